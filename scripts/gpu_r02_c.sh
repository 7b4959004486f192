#!/bin/bash
# Round 2, call C: full GPU suite on the rel32 / off_t-free kernels + a sweep of every kind on every config.
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_c
rm -rf $O; mkdir -p $O
cd $R
echo "== gpu suite"; timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
echo "== rel32 limit forced low (wide path)"; MI355_SPMV_REL32_LIMIT=3000 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_ragged or integer_valued or empty_and or giant or weight" > $O/pytest_rel32.log 2>&1; tail -4 $O/pytest_rel32.log
echo "== sweep"; bash scripts/gpu_sweep.sh r02c > $O/sweep.txt 2>&1; cat $O/sweep.txt; cp gpurun_out/sweep_r02c.jsonl $O/
echo done
