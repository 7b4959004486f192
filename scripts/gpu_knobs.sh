#!/bin/bash
# knob sweep on one box: scripts/gpu_knobs.sh "<bench args>" "ENV=1 ENV2=2" "ENV=3" ...   ("X=1" = defaults)
R=$GRAFT_REPO_ROOT
ARGS=$1; shift
for e in "$@"; do
  env $e python $R/bench.py --no-cpu-baseline --steps 100 --warmup 50 $ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print('%-50s' % '$e', c['kind'], 'T', c['lanes_per_row'], 'grid', c['grid_blocks'], 'win', c['x_window_elems'], 'seg', c['x_window_segments'], '%.2f us' % (r['kernel_ms']*1e3), 'min %.2f' % (r['kernel_ms_min']*1e3), 'frac %.3f' % r['frac'])"
done
