#!/bin/bash
# A/B of env-controlled variants on one workload/kind, interleaved rounds in separate processes
# usage: bash scripts/gpu_ab.sh "<bench args>" VAR=a VAR=b ...
ARGS="$1"; shift
for round in 1 2 3; do
  for v in "$@"; do
    r=$(env $v timeout -k 10 200 python bench.py $ARGS --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%9.1f us %7.0f GB/s' % (d['roofline']['kernel_ms']*1e3, d['achieved_hbm_gbps']))")
    echo "round $round  $v : $r"
  done
done
