#!/usr/bin/env python3
"""Small power-law matrices (R-MAT scale 12-17, 16 per row): which kind is fastest?  Run under rocprofv3 --kernel-trace
(the Python loop cannot issue kernels of a few microseconds back to back); MI355_SPMV_PLAIN=1 in the environment puts
the vector kind on its plain kernel."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
sp = g.load_package()
dev = "cuda:0"
for scale in [int(a) for a in sys.argv[1:]] or [14]:
    m = sp.synth.rmat(scale, 16, seed=5, device=dev)
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, dev)
    y = torch.empty(m.n_rows, device=dev)
    for kind in ("vector", "merge", "light"):
        p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
        for _ in range(300):
            p.execute(m.Ax, x, y)
        torch.cuda.synchronize()
        print(scale, kind, p.info()["main_kernel"], p.info()["n_kernels"], flush=True)
        p.destroy()
