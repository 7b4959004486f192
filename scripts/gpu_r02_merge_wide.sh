#!/bin/bash
# merge on bands beyond a 256-thread window: the 512- / 1 024-thread run kernels (default) against MI355_MERGE_WIDE_WINDOW=0
for e in X=1 MI355_MERGE_WIDE_WINDOW=0; do for args in "--band-half-width 8192" "--band-half-width 16384" "--band-half-width 8192 --s32-values f64" "--s32-values f64"; do
  env $e python bench.py --no-cpu-baseline --steps 40 --warmup 40 --kind merge $args 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$e $args :', d['roofline']['kernel'], 'window', c['x_window_elems'], 'grid', c['grid_blocks'], round(d['roofline']['kernel_ms']*1e3,1), 'us', round(d['roofline']['achieved']), 'GB/s')"
done; done
