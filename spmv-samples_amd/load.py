"""ctypes binding of libmi355load.so (include/mi355_load.h): Matrix Market file -> CSR, through the product loader
(host/load.hpp: the reference's LoadCoo + ToCsr, include/load.hpp:268-474, main.cu:32-39).  Host code only."""
import ctypes as C
import os

import numpy as np
import torch

from .synth import Csr

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355_LOAD_LIB") or os.path.join(_HERE, "lib", "libmi355load.so")
EXPORTS = ["mi355_load_mtx", "mi355_csr_host_dims", "mi355_csr_host_Ap", "mi355_csr_host_Aj", "mi355_csr_host_Ax",
           "mi355_csr_host_free", "mi355_load_last_error"]
STATUS = {1: "invalid argument", 2: "not a usable Matrix Market coordinate file", 3: "malformed entry",
          4: "does not fit the index / offset types"}
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: build it (python -c 'import __graft_entry__ as g; g.build()')" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.mi355_load_mtx.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.mi355_csr_host_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        for f in (L.mi355_csr_host_Ap, L.mi355_csr_host_Aj, L.mi355_csr_host_Ax):
            f.argtypes = [C.c_void_p]
            f.restype = C.c_void_p
        L.mi355_csr_host_free.argtypes = [C.c_void_p]
        L.mi355_csr_host_free.restype = None
        L.mi355_load_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def load_mtx(path, off_dtype=torch.int32, val_dtype=torch.float32, device="cpu"):
    """The file as the reference's harness would hold it after LoadCoo + ToCsr, as a synth.Csr on `device`."""
    L = lib()
    h = C.c_void_p()
    st = L.mi355_load_mtx(os.fsencode(path), 1 if off_dtype == torch.int64 else 0, 1 if val_dtype == torch.float64 else 0,
                          C.byref(h))
    if st != 0:
        raise RuntimeError("mi355_load_mtx(%s): %s (%s)" % (path, STATUS.get(st, st), L.mi355_load_last_error().decode()))
    try:
        nr, nc, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        L.mi355_csr_host_dims(h, C.byref(nr), C.byref(nc), C.byref(nnz))
        n_rows, n_cols, n = nr.value, nc.value, nnz.value
        np_off = np.int64 if off_dtype == torch.int64 else np.int32
        np_val = np.float64 if val_dtype == torch.float64 else np.float32

        def view(ptr, count, dt):
            if count == 0 or not ptr:
                return np.zeros(count, dtype=dt)
            buf = (C.c_char * (count * np.dtype(dt).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dt, count=count).copy()
        Ap = view(L.mi355_csr_host_Ap(h), n_rows + 1, np_off)
        Aj = view(L.mi355_csr_host_Aj(h), n, np.int32)
        Ax = view(L.mi355_csr_host_Ax(h), n, np_val)
    finally:
        L.mi355_csr_host_free(h)
    t = lambda a: torch.from_numpy(a).to(device)
    return Csr(n_rows, n_cols, n, t(Ap), t(Aj), t(Ax), os.path.basename(path), {"synthetic": False, "path": str(path)})
