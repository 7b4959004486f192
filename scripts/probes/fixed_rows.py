#!/usr/bin/env python3
"""vector / light on fixed row lengths 4 .. 72 (2^27 nonzeros, band +-4096): a scan for cliffs of the plan rules."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
sp = g.load_package()
dev = torch.device("cuda:0")
def timeit(p, m, x, y, n=30):
    for _ in range(40): p.execute(m.Ax, x, y)   # (a big matrix runs its first ~35 executes 10-15 % slower)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): p.execute(m.Ax, x, y)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for k in [int(a) for a in sys.argv[1:]] or (4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 27, 28, 30, 32, 34, 36, 40, 44, 48, 56, 64, 72):
    m = sp.synth.banded_fixed((1 << 27) // k, k, 4096, 1, dev)
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, dev)
    y = torch.empty(m.n_rows, dtype=m.Ax.dtype, device=dev)
    out = []
    for kind in ("vector", "light"):
        p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
        us = timeit(p, m, x, y)
        i = p.info()
        out.append("%s %7.1f us %5.0f GB/s (T%d b%d rows/chunk %d w%d grid %d)" % (kind[0], us, m.algorithmic_bytes() / us / 1e3, i["lanes_per_row"], i["block_threads"], i["rows_per_chunk"], i["window_elems"], i["grid_blocks"]))
        p.destroy()
    print("fixed %3d per row : %s" % (k, " | ".join(out)), flush=True)
    del m, x, y
    torch.cuda.empty_cache()
