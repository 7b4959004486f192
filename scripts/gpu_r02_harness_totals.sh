#!/bin/bash
# the reference-shaped harness (spmv-samples_amd/host/main.cpp: the reference's main.cu loop, TEST_TIMES calls per kind) on a
# generated banded file: "total" per call with the plans kept between calls, and with MI355_SPMV_PLAN_CACHE=0 (a plan per call)
set -e
python - <<'P'
import numpy as np
n, k, hw = 400000, 32, 2000
rng = np.random.default_rng(1)
rows = np.repeat(np.arange(n), k)
cols = np.clip(rows + rng.integers(-hw, hw + 1, n * k), 0, n - 1)
key = np.unique(rows.astype(np.int64) * n + cols)            # distinct entries (a .mtx lists each once)
rows, cols = key // n, key % n
vals = rng.random(rows.size) * 2 - 1
with open("/tmp/band.mtx", "w") as f:
    f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (n, n, rows.size))
    np.savetxt(f, np.column_stack([rows + 1, cols + 1, vals]), fmt="%d %d %.9g")
print("wrote /tmp/band.mtx", rows.size, "entries")
P
make -s -C spmv-samples_amd/host
EXE=spmv-samples_amd/bin/spmv
echo "== plans kept between calls"; $EXE /tmp/band.mtx hip_vector hip_merge hip_light --iters 500 --unit-us | grep -E "total|kind" | tail -4
echo "== MI355_SPMV_PLAN_CACHE=0"; MI355_SPMV_PLAN_CACHE=0 $EXE /tmp/band.mtx hip_vector hip_merge hip_light --iters 500 --unit-us | grep -E "total|kind" | tail -4
