#!/usr/bin/env python3
"""Banded matrices of many shapes (band half-width x nonzeros per row x value type): algorithmic GB/s of every
kind — a scan for cliffs in the plan heuristics (window fits / does not fit, lanes per row, chunk sizes)."""
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
vals = [torch.float32, torch.float64] if "--f64" in sys.argv else [torch.float32]
for val in vals:
    for per_row in (8, 32, 128):
        n = (1 << 26) // per_row                       # 64 M nonzeros each
        for hw in (128, 1024, 4096, 8192, 16384, 65536):
            if 2 * hw < per_row: continue
            m = sp.synth.banded_fixed(n, per_row, hw, seed=2, device=dev, val_dtype=val)
            x = sp.synth.dense_vector(m.n_cols, val, 1, dev)
            y = torch.empty(m.n_rows, dtype=val, device=dev)
            out = []
            for kind in ("vector", "merge", "light"):
                p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, val)
                us = timeit(p, m, x, y)
                i = p.info()
                out.append("%s %7.1f us %5.0f GB/s (T%d b%d w%d)" % (kind[0], us, m.algorithmic_bytes() / us / 1e3, i["lanes_per_row"], i["block_threads"], i["window_elems"]))
                p.destroy()
            print("%s nnz/row %3d half-width %6d : %s" % ("f32" if val == torch.float32 else "f64", per_row, hw, " | ".join(out)), flush=True)
            del m, x, y
            torch.cuda.empty_cache()
