#!/usr/bin/env python3
"""Where do plain (cacheable) loads of the Aj / Ax stream beat nontemporal ones?  R-MAT (power-law) and banded-variable
matrices of growing size, every kind; run once per library (MI355_SPMV_LIB = lib / lib_plain built with
-DMI355_STREAM_PLAIN), merge also with the in-kernel search forced off / on."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
sp = g.load_package()
dev = torch.device("cuda:0")
tag = os.path.basename(os.path.dirname(os.environ.get("MI355_SPMV_LIB", "/lib/x")))
def timeit(p, m, x, y, n=40):
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
mats = [("rmat%d" % s, lambda s=s: sp.synth.rmat(s, 16, seed=5, device=dev)) for s in (18, 19, 20, 21, 22)]
mats += [("bandvar%dk" % (n // 1000), lambda n=n: sp.synth.banded_variable(n, 64, 16, 2048, 2, dev)) for n in (62451, 250000, 1000000, 4000000)]
for name, make in mats:
    m = make()
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, dev)
    y = torch.empty(m.n_rows, dtype=m.Ax.dtype, device=dev)
    mb = m.algorithmic_bytes() / 1e6
    out = []
    for kind, fused in (("vector", None), ("light", None), ("merge", None), ("merge", "0"), ("merge", "1")):
        knob("MI355_MERGE_FUSED", fused)
        p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
        out.append("%s%s %8.1f" % (kind[0], "" if fused is None else "(fused=%s)" % fused, timeit(p, m, x, y)))
        p.destroy()
    knob("MI355_MERGE_FUSED", None)
    print("%-9s %-12s %7.1f MB : %s" % (tag, name, mb, " | ".join(out)), flush=True)
    del m, x, y
    torch.cuda.empty_cache()
