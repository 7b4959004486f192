#!/usr/bin/env python3
"""Variable-length rows: the vector width from the mean (default) against the next wider one (MI355_SPMV_LANES), which
takes the longest rows in one step too."""
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
def knob(name, v):
    if v is None: os.environ.pop(name, None)
    else: os.environ[name] = v
    sp.capi.lib().mi355_spmv_knobs_reload()
for (n, ml, j) in ((4000000, 64, 16), (2000000, 128, 32), (8000000, 20, 6), (6000000, 36, 10), (16000000, 10, 3), (3000000, 80, 30)):
    m = sp.synth.banded_variable(n, ml, j, 2048, 2, dev)
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, dev)
    y = torch.empty(m.n_rows, dtype=m.Ax.dtype, device=dev)
    out = []
    for lanes in (None, "2", "4", "8", "16", "32", "64"):
        knob("MI355_SPMV_LANES", lanes)
        p = sp.Plan("vector", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
        i = p.info()
        if lanes is not None and abs(int(lanes).bit_length() - default_t.bit_length()) > 1:
            p.destroy(); continue
        us = timeit(p, m, x, y)
        if lanes is None: default_t = i["lanes_per_row"]
        out.append("%s T%d %7.1f us %5.0f GB/s" % ("default" if lanes is None else "forced", i["lanes_per_row"], us, m.algorithmic_bytes() / us / 1e3))
        p.destroy()
    knob("MI355_SPMV_LANES", None)
    print("mean %3d +-%2d : %s" % (ml, j, " | ".join(out)), flush=True)
    del m, x, y
    torch.cuda.empty_cache()
