#!/usr/bin/env python3
"""Host-side cost of the one-shot entry points (plan create + execute + sync + destroy per call, the
life cycle of the reference's kinds) next to a kept plan, per kind."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
sp = g.load_package()
dev = torch.device("cuda:0")
for w in (sys.argv[1:] or ["c2-cant", "s32-band"]):
    m = sp.synth.workload(w, dev)
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, dev)
    y = torch.empty(m.n_rows, dtype=m.Ax.dtype, device=dev)
    for kind in ("vector", "merge", "light"):
        for _ in range(3):
            sp.spmv(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            sp.spmv(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, y)
        one = (time.perf_counter() - t0) / n * 1e6
        p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
        for _ in range(3):
            p.execute(m.Ax, x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            p.execute(m.Ax, x, y)
            torch.cuda.synchronize()
        kept = (time.perf_counter() - t0) / n * 1e6
        t0 = time.perf_counter()
        for _ in range(n):
            q = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
            q.destroy()
        plan = (time.perf_counter() - t0) / n * 1e6
        p.destroy()
        print("%-10s %-7s one-shot %8.1f us   kept plan (execute+sync) %8.1f us   plan create+destroy %8.1f us" % (w, kind, one, kept, plan), flush=True)
