#!/bin/bash
# fixed row lengths just past a multiple of 4T: the plan's T against the next smaller one (two steps per row)
python scripts/gpu_r02_fixed_rows.py 18 20 22 24 2>&1 | grep -v amdgpu
for spec in "10 2" "12 2" "18 4" "20 4" "24 4" "34 8" "36 8" "40 8" "48 8" "72 16" "80 16" "100 16"; do set -- $spec
  echo "--- MI355_SPMV_LANES=$2"; MI355_SPMV_LANES=$2 python scripts/gpu_r02_fixed_rows.py $1 2>&1 | grep -v amdgpu
done
python scripts/gpu_r02_fixed_rows.py 10 12 34 36 40 48 72 80 100 2>&1 | grep -v amdgpu
