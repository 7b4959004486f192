#!/usr/bin/env python3
"""S32-band shape at 2^13 ... 2^21 rows: time, algorithmic GB/s and the plan of every kind (a scan of the
chunk-size / workgroup-size heuristics across matrix sizes)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
sp = g.load_package()
dev = torch.device("cuda:0")
def timeit(p, m, x, y, n=50):
    for _ in range(40): p.execute(m.Ax, x, y)   # (a big matrix runs its first ~35 executes 10-15 % slower)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): p.execute(m.Ax, x, y)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
PER_ROW = int(sys.argv[1]) if len(sys.argv) > 1 else 32
VAL = torch.float64 if "--f64" in sys.argv else torch.float32
for lg in (13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23):
    n = 1 << lg
    if n * PER_ROW > (1 << 28): break
    m = sp.synth.banded_fixed(n, PER_ROW, min(4096, n // 4), seed=2, device=dev, val_dtype=VAL)
    x = sp.synth.dense_vector(m.n_cols, VAL, 1, dev)
    y = torch.empty(m.n_rows, dtype=VAL, device=dev)
    out = []
    for kind in ("vector", "merge", "light"):
        p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, VAL)
        us = timeit(p, m, x, y); i = p.info()
        out.append("%s %6.1f us %5.0f GB/s (T%d b%d r%d w%d g%d %s)" % (kind[0], us, m.algorithmic_bytes() / us / 1e3, i["lanes_per_row"], i["block_threads"], i["rows_per_chunk"], i["window_elems"], i["grid_blocks"], i["main_kernel"].split("_")[1]))
        p.destroy()
    print("rows 2^%d: %s" % (lg, " | ".join(out)), flush=True)
