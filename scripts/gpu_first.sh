#!/bin/bash
# first GPU contact: smoke, parity tests, first bench line
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { echo SMOKE FAILED; tail -30 gpurun_out/smoke.log; exit 1; }
cat gpurun_out/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -25 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --all-kinds > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit $?"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
