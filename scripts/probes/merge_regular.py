#!/usr/bin/env python3
"""merge kind on regular matrices of several row lengths: row-parallel runs (merge_rows_kernel) vs the item walk
(MI355_MERGE_ROWS=0), next to vector."""
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
mats = [("bandvar mean %d +-%d, %dk rows" % (ml, j, n // 1000), lambda n=n, ml=ml, j=j: sp.synth.banded_variable(n, ml, j, 2048, 2, dev))
        for (n, ml, j) in ((4000000, 64, 16), (1000000, 64, 16), (8000000, 24, 6), (6000000, 40, 8), (2000000, 128, 32), (1000000, 200, 50))]
mats += [("S32-band", lambda: sp.synth.workload("s32-band", dev))]
if "--fixed" in sys.argv:
    mats = [("fixed %d per row" % k, lambda k=k: sp.synth.banded_fixed((1 << 27) // k, k, 4096, 1, dev)) for k in (8, 12, 16, 24, 27, 30, 32, 40, 48, 64, 100, 128)]
for name, make in mats:
    m = make()
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, dev)
    y = torch.empty(m.n_rows, dtype=m.Ax.dtype, device=dev)
    out = []
    for kind, rows in (("vector", None), ("merge", "1"), ("merge", "0")):
        knob("MI355_MERGE_ROWS", rows)
        p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
        us = timeit(p, m, x, y)
        out.append("%s%s %7.1f us %5.0f GB/s (%s)" % (kind[0], "" if rows is None else "(rows=%s)" % rows, us, m.algorithmic_bytes() / us / 1e3, p.info()["main_kernel"]))
        p.destroy()
    knob("MI355_MERGE_ROWS", None)
    print("%-34s %7.1f MB : %s" % (name, m.algorithmic_bytes() / 1e6, " | ".join(out)), flush=True)
    del m, x, y
    torch.cuda.empty_cache()
