#!/usr/bin/env python3
"""Comparison column only (NOT part of the engine): the vendor library's CSR SpMV, reached through
torch.sparse (hipSPARSE/rocSPARSE underneath), timed on the same matrices as bench.py.
The reference's `cusparse` kind (include/spmv/cusparse.cuh) is the analogue on NVIDIA."""
import json
import sys
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

sp = g.load_package()
dev = torch.device("cuda:0")
out = {}

# rocSPARSE csrmv called directly (tools/cmp_rocsparse.cpp): with analysis (adaptive) and without (row split)
CMP = os.path.join(ROOT, "tools", "bin", "libcmp_rocsparse.so")
cmp_lib = None
if os.path.exists(CMP) and "--no-rocsparse" not in sys.argv:
    import ctypes
    cmp_lib = ctypes.CDLL(CMP)
    cmp_lib.cmp_rocsparse_csrmv.restype = ctypes.c_int
    cmp_lib.cmp_rocsparse_csrmv.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + \
        [ctypes.POINTER(ctypes.c_double)] * 2


# warm-up and loop lengths by matrix size: ~25 ms of warm-up (an HBM-bound kernel needs ~12 ms from a fresh start to reach its
# steady time, profiles/r02_warmup_curve.txt; 5 + 30 calls penalised whoever ran first), ~25 ms timed, same for both sides
def EST_US(m):
    return max(8.0, m.algorithmic_bytes() / 4.0e6)          # (at ~4 TB/s)
def WARM(m):
    return max(40, int(25e3 / EST_US(m)))
def ITERS(m):
    return max(30, int(25e3 / EST_US(m)))


def rocsparse_direct(m, x, y_ours):
    """{'adaptive': us, 'stream': us, 'analysis_ms': ms, 'rel_maxdiff': d} or None (64-bit offsets)."""
    if cmp_lib is None or m.nnz >= 2 ** 31:
        return None
    import ctypes
    res = {}
    Ap32 = m.Ap if m.Ap.dtype == torch.int32 else m.Ap.to(torch.int32)   # csrmv takes 32-bit row offsets
    if Ap32 is not m.Ap:
        res["note"] = "vendor run on a 32-bit-offset copy of Ap"
    y = torch.full_like(y_ours, float("nan"))
    for name, analyse in (("stream", 0), ("adaptive", 1)):
        us, ana = ctypes.c_double(0), ctypes.c_double(0)
        torch.cuda.synchronize()
        rc = cmp_lib.cmp_rocsparse_csrmv(0 if m.Ax.dtype == torch.float32 else 1, m.n_rows, m.n_cols, m.nnz,
                                         Ap32.data_ptr(), m.Aj.data_ptr(), m.Ax.data_ptr(), x.data_ptr(),
                                         y.data_ptr(), analyse, WARM(m), ITERS(m), ctypes.byref(us), ctypes.byref(ana))
        assert rc == 0
        torch.cuda.synchronize()
        res[name + "_us"] = us.value
        if analyse:
            res["analysis_ms"] = ana.value
        res[name + "_rel_maxdiff_vs_ours"] = float((y - y_ours).abs().max() / (y_ours.abs().max() + 1e-30))
    return res

for w in ([a for a in sys.argv[1:] if not a.startswith("--")] or ["s32-band", "c2-cant", "c3-webgoogle"]):
    if w.startswith("s32-band@"):                      # the target's shape at 2^k rows (mid-size studies)
        m = sp.synth.banded_fixed(1 << int(w.split("@")[1]), 32, 4096, 1, dev, name=w)
    else:
        m = sp.synth.workload(w, dev)
    with_torch = m.Ap.dtype == m.Aj.dtype and "--no-torch-sparse" not in sys.argv
    # (torch.sparse_csr_tensor does not validate index dtypes by default; int64 crow + int32 col indices faulted
    # inside the vendor path on MI355X, torch 2.10/ROCm 7.0: mixed dtypes are never handed to it)
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, dev)
    y = torch.zeros(m.n_rows, dtype=m.Ax.dtype, device=dev)
    if with_torch:
        A = torch.sparse_csr_tensor(m.Ap, m.Aj, m.Ax, size=(m.n_rows, m.n_cols))
        y = A @ x
    # our engine on the same operands, for the cross-check and the side-by-side time
    res = {}
    y2 = torch.empty_like(y)
    for kind in ("vector", "merge", "light"):
        p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
        for _ in range(WARM(m)):
            p.execute(m.Ax, x, y2)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(ITERS(m)):
            p.execute(m.Ax, x, y2)
        b.record()
        torch.cuda.synchronize()
        res[kind] = a.elapsed_time(b) / ITERS(m) * 1e3
        p.destroy()
    direct = rocsparse_direct(m, x, y2)
    t, err = None, None
    if with_torch:
        err = float((y2 - y).abs().max() / (y.abs().max() + 1e-30))
        for _ in range(5):
            y = A @ x
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            y = A @ x
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / 30 * 1e3
        del A
    out[w] = {"vendor_torch_sparse_us": t, "ours_us": res, "rel_maxdiff_vs_vendor": err, "rocsparse_csrmv": direct}
    print(w, json.dumps(out[w]), flush=True)
    del m, x, y, y2
    torch.cuda.empty_cache()
