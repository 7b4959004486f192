#!/usr/bin/env python3
"""27-point stencils of growing edge length, fp64/i64 (the C4 stand-in's family) and fp32/i32: every kind with its plan —
a scan for cliffs where the three bands of the stencil stop fitting the LDS budget."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
sp = g.load_package()
dev = torch.device("cuda:0")
def timeit(p, m, x, y, n=20):
    for _ in range(40): p.execute(m.Ax, x, y)   # (a big matrix runs its first ~35 executes 10-15 % slower)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): p.execute(m.Ax, x, y)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for val, off in ((torch.float64, torch.int64), (torch.float32, torch.int32)):
    for d in (48, 64, 100, 128, 160, 203, 256):
        m = sp.synth.stencil27(d, d, d, 4, dev, val_dtype=val, off_dtype=off)
        x = sp.synth.dense_vector(m.n_cols, val, 1, dev)
        y = torch.empty(m.n_rows, dtype=val, device=dev)
        out = []
        for kind in ("vector", "merge", "light"):
            p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, val)
            us = timeit(p, m, x, y); i = p.info()
            out.append("%s %7.1f us %5.0f GB/s (T%d b%d r%d w%d seg%d %s)" % (kind[0], us, m.algorithmic_bytes() / us / 1e3, i["lanes_per_row"], i["block_threads"], i["rows_per_chunk"], i["window_elems"], i["window_segments"], i["main_kernel"].split("_")[1]))
            p.destroy()
        print("%s %d^3 (%.0f MB): %s" % ("f64/i64" if val == torch.float64 else "f32/i32", d, m.algorithmic_bytes() / 1e6, " | ".join(out)), flush=True)
        del m, x, y
        torch.cuda.empty_cache()
