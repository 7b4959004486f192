#!/usr/bin/env python3
"""Run-to-run bitwise reproducibility at BASELINE sizes: every kind is deterministic by construction
(no float atomics), so any difference between two runs on the same operands is a race.
usage: python scripts/gpu_repro_check.py [workload ...] [--runs N]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
sp = g.load_package()
dev = torch.device("cuda:0")
runs = 20
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--runs" in sys.argv:
    runs = int(sys.argv[sys.argv.index("--runs") + 1]); args = [a for a in args if a != str(runs)]
bad = 0
for w in (args or ["c4-nlpkkt", "s32-band", "c3-webgoogle"]):
    m = sp.synth.workload(w, dev)
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 5, dev)
    for kind in ("vector", "merge", "light"):
        ref = torch.full((m.n_rows,), float("nan"), dtype=m.Ax.dtype, device=dev)
        sp.spmv(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, ref)
        diffs = 0
        for i in range(runs):
            y = torch.full((m.n_rows,), float("nan"), dtype=m.Ax.dtype, device=dev)
            if i % 2:
                p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
                p.execute(m.Ax, x, y)
                torch.cuda.synchronize()
                p.destroy()
            else:
                sp.spmv(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, y)
            same = torch.equal(y, ref)
            if not same:
                d = (y != ref) | (y.isnan() != ref.isnan())
                idx = torch.nonzero(d).flatten()
                print("  %s %s run %d: %d rows differ, first %s  y=%s ref=%s" % (w, kind, i, idx.numel(), idx[:4].tolist(),
                      y[idx[:4]].tolist(), ref[idx[:4]].tolist()), flush=True)
                diffs += 1
        print("%-14s %-7s %d runs, %d differ" % (w, kind, runs, diffs), flush=True)
        bad += diffs
    del m, x
    torch.cuda.empty_cache()
sys.exit(1 if bad else 0)
