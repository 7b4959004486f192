#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -8 gpurun_out/pytest_gpu.log
bash scripts/gpu_sweep.sh ${1:-x}
