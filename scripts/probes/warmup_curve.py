#!/usr/bin/env python3
"""How long does a kernel take to reach its steady time?  Successive windows of executes from a fresh plan, per workload."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
sp = g.load_package()
dev = torch.device("cuda:0")
for w, kind, win in (("c2-cant", "vector", 100), ("c3-webgoogle", "merge", 50), ("s32-band", "vector", 10)):
    m = sp.synth.workload(w, dev)
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, dev)
    y = torch.empty(m.n_rows, dtype=m.Ax.dtype, device=dev)
    torch.cuda.synchronize()
    time.sleep(0.5)                      # (an idle gap, as between a harness's set-up and its timing loop)
    p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
    out = []
    for k in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(win): p.execute(m.Ax, x, y)
        b.record(); torch.cuda.synchronize()
        out.append(round(a.elapsed_time(b) / win * 1e3, 1))
    p.destroy()
    print("%s %s, windows of %d executes (us per execute): %s" % (w, kind, win, out), flush=True)
    del m, x, y
