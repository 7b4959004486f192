#!/usr/bin/env python3
"""What holds the fp64 stencil (C4) at 4.5 TB/s when the fp64 band reaches 5.7?  One factor at a time:
size, nonzeros per row (27 vs 32: rows no longer start on 16-byte groups), one band vs three."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
sp = g.load_package()
dev = "cuda:0"

def timeit(kind, m, reps=30):
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, dev)
    y = torch.empty(m.n_rows, dtype=m.Ax.dtype, device=dev)
    p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
    for _ in range(40): p.execute(m.Ax, x, y)   # (a big matrix runs its first ~35 executes 10-15 % slower)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); p.execute(m.Ax, x, y); b.record()
    torch.cuda.synchronize()
    info = p.info(); p.destroy()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    return ms, info

def report(name, m):
    for kind in ("vector", "merge"):
        ms, info = timeit(kind, m)
        print("%-44s %-6s %8.1f us %6.0f GB/s  T=%d blk=%d win=%d seg=%d rows/chunk=%d" % (
            name, kind, ms * 1e3, m.algorithmic_bytes() / ms / 1e6, info["lanes_per_row"], info["block_threads"],
            info["window_elems"], info["window_segments"], info["rows_per_chunk"]), flush=True)

f64, i64, i32 = torch.float64, torch.int64, torch.int32
report("band32 f64 i32 2^22 rows w4096", sp.synth.banded_fixed(1 << 22, 32, 4096, 1, dev, f64, i32))
report("band32 f64 i64 2^22 rows w4096", sp.synth.banded_fixed(1 << 22, 32, 4096, 1, dev, f64, i64))
report("band32 f64 i64 2^23 rows w4096", sp.synth.banded_fixed(1 << 23, 32, 4096, 1, dev, f64, i64))
report("band32 f64 i64 2^23 rows w256", sp.synth.banded_fixed(1 << 23, 32, 256, 1, dev, f64, i64))
report("band28 f64 i64 2^23 rows w256", sp.synth.banded_fixed(1 << 23, 28, 256, 1, dev, f64, i64))
report("band27 f64 i64 2^23 rows w256", sp.synth.banded_fixed(1 << 23, 27, 256, 1, dev, f64, i64))
report("band27 f32 i32 2^23 rows w256", sp.synth.banded_fixed(1 << 23, 27, 256, 1, dev, torch.float32, i32))
report("stencil27 203^3 f64 i64 (C4)", sp.synth.workload("c4-nlpkkt", dev))
m = sp.synth.workload("c4-nlpkkt", dev)
m32 = sp.synth.Csr(m.n_rows, m.n_cols, m.nnz, m.Ap.to(i32), m.Aj, m.Ax.to(torch.float32), "c4-f32-i32")
report("stencil27 203^3 f32 i32", m32)
