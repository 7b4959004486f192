#!/bin/bash
# Round 2, call B: the new block / dist path — GPU suite, bench lines (plain, self-spawned, torchrun N=1).
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_b
rm -rf $O; mkdir -p $O
cd $R
echo "== dist tests"; timeout -k 10 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_host.py -x -q > $O/pytest_dist.log 2>&1; tail -15 $O/pytest_dist.log
echo "== config tests"; timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q > $O/pytest_configs.log 2>&1; tail -8 $O/pytest_configs.log
echo "== rest of the gpu suite"; timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_harness.py -x -q -m gpu > $O/pytest_rest.log 2>&1; tail -8 $O/pytest_rest.log
echo "== bench default"; timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
echo "== bench torchrun N=1"; timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 200 --warmup 20 > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; tail -c 600 $O/bench_torchrun1.json; tail -3 $O/bench_torchrun1.err
echo "== bench torchrun N=1 sub-blocks 4"; timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 200 --warmup 20 --sub-blocks 4 --no-cpu-baseline > $O/bench_torchrun1_s4.json 2> $O/bench_torchrun1_s4.err; tail -c 400 $O/bench_torchrun1_s4.json
echo "== bench c5 torchrun N=1 sub-blocks 4 (strong path)"; timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --workload c5-rmat24 --steps 20 --warmup 3 --sub-blocks 4 > $O/bench_c5_s4.json 2> $O/bench_c5_s4.err; tail -c 400 $O/bench_c5_s4.json; tail -3 $O/bench_c5_s4.err
echo "== two ranks on one GPU (exploratory)"; timeout -k 5 120 python scripts/gpu_rccl_two_ranks_one_gpu.py > $O/rccl_two_ranks.log 2>&1; tail -6 $O/rccl_two_ranks.log
echo done
