#!/bin/bash
# time every kind on every workload (parity-test configs included), one JSON line each
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/sweep_${1:-x}.jsonl
: > $OUT
for w in s32-band s32-rand c2-cant c3-webgoogle c4-nlpkkt c5-rmat24; do
  timeout -k 10 280 python bench.py --workload $w --all-kinds --no-cpu-baseline --steps 50 --warmup 10 >> $OUT 2>> gpurun_out/sweep.err || echo "{\"workload\": \"$w\", \"failed\": true}" >> $OUT
done
python - <<PY
import json
for l in open("$OUT"):
    d = json.loads(l)
    if d.get("failed"): print(d); continue
    ks = d["all_kinds"]
    print("%-46s" % d["config"]["workload"][:46], " ".join("%s %8.1f us %6.0f GB/s %6.0f GF |" % (k, v["kernel_ms"]*1e3, v["gbps"], v["gflops"]) for k, v in ks.items()))
PY
