#!/bin/bash
# full GPU check: parity tests, bench N=1 (plain) and the torch.distributed.run launch path (N=1 over RCCL)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -6 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit $?"; cut -c1-1200 gpurun_out/bench.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err
echo "torchrun bench exit $?"; cut -c1-700 gpurun_out/bench_torchrun1.json; tail -3 gpurun_out/bench_torchrun1.err
