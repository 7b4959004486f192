#!/usr/bin/env python3
"""Wide bands (more columns than a CU's LDS holds): vector kind with the sweeping window (MI355_SPMV_SWEEP=1, forced)
against plain gathers (=0) and the library's own choice (unset), and light; every result is compared with the plain-gather
plan's (the parity tests proper, against the oracle, are tests/test_gpu_parity.py)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
sp = g.load_package()
dev = torch.device("cuda:0")
def timeit(p, m, x, y, n=20):
    for _ in range(40): p.execute(m.Ax, x, y)   # (a big matrix runs its first ~35 executes 10-15 % slower)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): p.execute(m.Ax, x, y)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
def plan_with(knob, kind, m, val):
    if knob is None: os.environ.pop("MI355_SPMV_SWEEP", None)
    else: os.environ["MI355_SPMV_SWEEP"] = knob
    sp.capi.lib().mi355_spmv_knobs_reload()
    return sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, val)
vals = [torch.float32, torch.float64]
quick = "--quick" in sys.argv
for val in vals:
    for per_row in (8, 32, 128):
        n = (1 << 26) // per_row
        for hw in ((32768,) if quick else (8192, 16384, 32768, 65536, 131072, 262144, 1 << 20)):
            m = sp.synth.banded_fixed(n, per_row, hw, seed=2, device=dev, val_dtype=val)
            x = sp.synth.dense_vector(m.n_cols, val, 1, dev)
            y = torch.empty(m.n_rows, dtype=val, device=dev)
            Ap, Aj, Ax = m.numpy()
            want, bound = None, None           # set by the first (plain-gather) plan below
            out = []
            for knob, kind in (("0", "vector"), (None, "vector"), (None, "light")):
                p = plan_with(knob, kind, m, val)
                y.fill_(float("nan"))
                us = timeit(p, m, x, y)
                i = p.info()
                got = y.cpu().numpy().astype(np.float64)
                if want is None:               # the reference of this script: the sweep-free plan of the same library
                    want = got
                    lens = np.diff(Ap.astype(np.int64))
                    bound = 2 * (lens + 2) * (2.0 ** -24 if val == torch.float32 else 2.0 ** -53) * (np.abs(want) + 1.0) * 8
                err = np.abs(got - want)
                ok = bool((err <= bound).all()) and not np.isnan(got).any()
                out.append("%s %7.1f us %5.0f GB/s (%s T%d b%d w%d) %s" % (kind[0] + " sweep=" + str(knob), us, m.algorithmic_bytes() / us / 1e3,
                           i["main_kernel"].split("_")[2], i["lanes_per_row"], i["block_threads"], i["window_elems"], "ok" if ok else "WRONG max %.3g" % err.max()))
                p.destroy()
            print("%s nnz/row %3d half-width %7d : %s" % ("f32" if val == torch.float32 else "f64", per_row, hw, " | ".join(out)), flush=True)
            del m, x, y
            torch.cuda.empty_cache()
os.environ.pop("MI355_SPMV_SWEEP", None)
