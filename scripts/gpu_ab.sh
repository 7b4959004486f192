#!/bin/bash
# A/B of library builds on one box, interleaved, 3 rounds: scripts/gpu_ab.sh "<bench args>" lib_a lib_b ...
# (lib = directory under spmv-samples_amd/; build variants with scripts/build_variant.sh).  Prints every round and the
# median per build: the boxes of the pool run the same kernel in 2-3 % wide modes from run to run.
R=$GRAFT_REPO_ROOT
ARGS=$1; shift
ROUNDS=${ROUNDS:-3}
T=$(mktemp)
for round in $(seq $ROUNDS); do
for L in "$@"; do
  MI355_SPMV_LIB=$R/spmv-samples_amd/$L/libmi355spmv.so python $R/bench.py --no-cpu-baseline --steps ${STEPS:-200} --warmup 50 $ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print('%-14s' % '$L', c['kind'], 'T', c['lanes_per_row'], 'grid', c['grid_blocks'], 'win', c['x_window_elems'], 'seg', c['x_window_segments'], '%.2f us' % (r['kernel_ms']*1e3), 'min %.2f' % (r['kernel_ms_min']*1e3), 'frac %.3f' % r['frac'])" | tee -a $T
done
done
python - $T "$ARGS" <<'PY'
import sys, statistics, collections
d = collections.OrderedDict()
for l in open(sys.argv[1]):
    p = l.split()
    d.setdefault(p[0], []).append(float(p[p.index('us') - 1]))
print('   median [%s]: ' % sys.argv[2] + '   '.join('%s %.2f' % (k, statistics.median(v)) for k, v in d.items()))
PY
rm -f $T
