#!/usr/bin/env python3
"""A banded matrix (2^21 rows x 32) with a few dense rows added: time of every kind and the plan it got —
the shapes that defeat equal-row chunks (hub rows) and single-workgroup rows (giant rows)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
sp = g.load_package()
dev = torch.device("cuda:0")
def timeit(p, m, x, y, n=20):
    for _ in range(3): p.execute(m.Ax, x, y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): p.execute(m.Ax, x, y)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
n = 1 << 21
for hubs, hub_len in ((0, 0), (10, 100000), (1000, 5000), (1, 4000000), (3, 1000000)):
    lens = torch.full((n,), 32, dtype=torch.int64, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    if hubs:
        lens[torch.randint(0, n, (hubs,), generator=gen, device=dev)] = hub_len
    Ap = torch.zeros(n + 1, dtype=torch.int64, device=dev); torch.cumsum(lens, 0, out=Ap[1:])
    nnz = int(Ap[-1])
    rows = torch.repeat_interleave(torch.arange(n, device=dev), lens)
    pos = torch.arange(nnz, device=dev) - Ap[rows]
    band_col = (rows - 4096 + (pos * 256) % 8192).clamp_(0, n - 1)
    rnd = torch.randint(0, n, (nnz,), generator=gen, device=dev)
    Aj = torch.where(lens[rows] != 32, rnd, band_col).to(torch.int32)
    Ax = torch.rand(nnz, generator=gen, device=dev) * 2 - 1
    m = sp.synth.Csr(n, n, nnz, Ap.to(torch.int32), Aj, Ax, "hubs", {})
    x = sp.synth.dense_vector(n, torch.float32, 1, dev)
    y = torch.empty(n, device=dev)
    out = []
    for kind in ("vector", "merge", "light"):
        p = sp.Plan(kind, n, n, nnz, m.Ap, m.Aj, torch.float32)
        us = timeit(p, m, x, y); i = p.info()
        out.append("%s %7.1f us (bal%d k%d b%d w%d)" % (kind[0], us, i["balanced_chunks"], i["n_kernels"], i["block_threads"], i["window_elems"]))
        p.destroy()
    print("dense rows %5d x %7d : %s" % (hubs, hub_len, " | ".join(out)), flush=True)
    del m, Aj, Ax, rows, pos, band_col, rnd
    torch.cuda.empty_cache()
