#!/usr/bin/env python3
"""Small power-law matrices (R-MAT 16 .. 20): vector / light under a few knobs against merge — why is the row-based kind 2x behind?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
sp = g.load_package()
dev = torch.device("cuda:0")
def timeit(p, m, x, y, n=200):
    for _ in range(200): p.execute(m.Ax, x, y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): p.execute(m.Ax, x, y)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
def knobs(env):
    for k in list(os.environ):
        if k.startswith("MI355_") and k != "MI355_SPMV_LIB": os.environ.pop(k)
    os.environ.update(env)
    sp.capi.lib().mi355_spmv_knobs_reload()
for scale in (16, 18, 20):
    m = sp.synth.rmat(scale, 16, seed=5, device=dev)
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, dev)
    y = torch.empty(m.n_rows, dtype=m.Ax.dtype, device=dev)
    lens = (m.Ap[1:] - m.Ap[:-1])
    print("rmat%d: %d rows, %d nnz, longest row %d, rows > 1024: %d, empty %d" % (scale, m.n_rows, m.nnz, int(lens.max()), int((lens > 1024).sum()), int((lens == 0).sum())), flush=True)
    for kind, env in (("merge", {}), ("vector", {}), ("vector", {"MI355_SPMV_BALANCE": "0"}), ("vector", {"MI355_SPMV_LONG_STEPS": "2"}), ("vector", {"MI355_SPMV_LONG_STEPS": "8"}),
                      ("vector", {"MI355_SPMV_GIANT_ROW": "4096"}), ("vector", {"MI355_SPMV_ROWS_PER_CHUNK": "256"}), ("vector", {"MI355_SPMV_ROWS_PER_CHUNK": "1024"}),
                      ("vector", {"MI355_SPMV_WINDOW": "0"}), ("light", {})):
        knobs(env)
        p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
        i = p.info()
        print("   %-7s %-36s %7.1f us  (grid %d, %d kernels, bal %d, rows_cap %d, win %d)" % (kind, env, timeit(p, m, x, y), i["grid_blocks"], i["n_kernels"], i["balanced_chunks"], i["rows_cap"], i["window_elems"]), flush=True)
        p.destroy()
    knobs({})
    del m, x, y
