#!/bin/bash
# bench.py launch modes on one GPU: plain, self-spawned --gpus 1 is plain; torchrun N = 1 through the library's
# multi-GPU object with the exchange check rehearsed, and through the torch.distributed safety net.
set -o pipefail
O=gpurun_out/benchcheck; mkdir -p $O
python bench.py --steps 50 > $O/plain.json 2> $O/plain.err; echo "plain rc=$? lines=$(wc -l < $O/plain.json)"
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511"
MI355_BENCH_CHECK_EXCHANGE=1 $TR bench.py --gpus 1 --steps 50 --cpu-seconds 3 > $O/tr1.json 2> $O/tr1.err; echo "torchrun rc=$? lines=$(wc -l < $O/tr1.json)"
MI355_BENCH_CHECK_EXCHANGE=1 MI355_BENCH_FORCE_FALLBACK=1 $TR bench.py --gpus 1 --steps 50 --no-cpu-baseline > $O/tr1_fb.json 2> $O/tr1_fb.err; echo "fallback rc=$? lines=$(wc -l < $O/tr1_fb.json)"
MI355_BENCH_CHECK_EXCHANGE=1 $TR bench.py --gpus 1 --steps 20 --no-cpu-baseline --workload c5-rmat24 --sub-blocks 4 > $O/tr1_c5.json 2> $O/tr1_c5.err; echo "c5 rc=$? lines=$(wc -l < $O/tr1_c5.json)"
python - <<'P'
import json
for f in ("plain","tr1","tr1_fb","tr1_c5"):
    try:
        d=json.loads(open("gpurun_out/benchcheck/%s.json"%f).read())
        print(f, round(d["value"],1), round(d["ms_per_step"],4), d["roofline"]["frac"], d.get("exchange_check"), d.get("one_shot_ms"), "cpu" if "cpu_baseline" in d else "-", d["config"]["parallelism"][:70])
    except Exception as e:
        print(f, "ERR", e)
P
