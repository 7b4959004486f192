#!/bin/bash
# Round 2, call D: new merge tile kernel (nnz-split, products in registers) + light whole rounds.
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_d
rm -rf $O; mkdir -p $O
cd $R
echo "== gpu suite"; timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
echo "== merge env matrix"; for env in "MI355_MERGE_TPS=1" "MI355_MERGE_TPS=5" "MI355_MERGE_BLOCK=512" "MI355_SPMV_WINDOW=0" "MI355_SPMV_WINDOW=1"; do echo $env; env $env timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merge" 2>&1 | tail -2; done > $O/pytest_merge_env.log 2>&1; cat $O/pytest_merge_env.log
echo "== sweep"; bash scripts/gpu_sweep.sh r02d > $O/sweep.txt 2>&1; cat $O/sweep.txt; cp gpurun_out/sweep_r02d.jsonl $O/
echo done
