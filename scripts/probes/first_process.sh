# the driver-style line (K = 20) against longer CONTINUOUS untimed warm-ups right before the timed region
R=$GRAFT_REPO_ROOT
show() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', 'step %.1f us' % (d['ms_per_step']*1e3), 'kernel %.1f' % (r['kernel_ms']*1e3), 'min %.1f' % (r['kernel_ms_min']*1e3), 'frac %.4f' % r['frac'], d['config']['kind'])"; }
for rep in 1 2; do
for w in 0 100 300 1000; do
MI355_BENCH_FINAL_WARM=$w python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show "K=20 final warm $w   "
done
python3 $R/bench.py --no-cpu-baseline 2>/dev/null | show "default K=200        "
done
