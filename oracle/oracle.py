"""ctypes bindings for the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  The product path (spmv-samples_amd/) never does.

  Oracle()  -> oracle/liboracle.so   the CPU restatement (spmv_oracle.cpp)
  Ref()     -> oracle/_ref/libspmv_ref.so   the reference's own loader + serial
               SpMV headers compiled from /root/reference (ref_driver.cpp);
               present only where oracle/Makefile could build it (or where the
               prebuilt file travelled with the snapshot).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

_OFF = {"i32": (np.int32, C.c_int32), "i64": (np.int64, C.c_int64)}
_VAL = {"f32": (np.float32, C.c_float), "f64": (np.float64, C.c_double)}

MTX_ERRORS = {0: "ok", 1: "open", 2: "banner", 3: "array", 4: "size", 5: "overflow", 6: "entry", 7: "type"}


def suffix(Ap, Ax):
    o = {np.dtype(np.int32): "i32", np.dtype(np.int64): "i64"}[np.asarray(Ap).dtype]
    v = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64", np.dtype(np.int32): "i32"}[np.asarray(Ax).dtype]
    return o, v


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def build(ref=True):
    """(Re)build the oracle libraries with oracle/Makefile."""
    import subprocess
    subprocess.run(["make", "-s", "-C", _HERE, "liboracle.so"] + (["ref"] if ref else []), check=True)


class Oracle:
    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=os.path.isdir("/root/reference"))
        self.lib = C.CDLL(path)
        self.lib.oracle_mtx_load.restype = C.c_void_p
        self.lib.oracle_hardware_threads.restype = C.c_int

    def _check(self, Ap, Aj, Ax, x):
        n_rows = len(Ap) - 1
        assert Aj.dtype == np.int32 and Ax.dtype == x.dtype
        for a in (Ap, Aj, Ax, x):
            assert a.flags["C_CONTIGUOUS"]
        return n_rows

    def spmv_serial(self, Ap, Aj, Ax, x):
        """y = A x, serial CSR order (cpu_navie.hpp:5-17)."""
        n = self._check(Ap, Aj, Ax, x)
        o, v = suffix(Ap, Ax)
        y = np.empty(n, dtype=Ax.dtype)
        getattr(self.lib, f"oracle_spmv_serial_{o}_{v}")(C.c_int32(n), _p(Ap), _p(Aj), _p(Ax), _p(x), _p(y))
        return y

    def spmv_serial_mixed(self, Ap, Aj, Ax, x):
        """fp32 matrix, fp64 x -> fp64 y, serial CSR order (cpu_navie.hpp:5-17 as <float, double, double>)."""
        assert Ax.dtype == np.float32 and x.dtype == np.float64 and Aj.dtype == np.int32
        n = len(Ap) - 1
        o = "i32" if Ap.dtype == np.int32 else "i64"
        y = np.empty(n, dtype=np.float64)
        getattr(self.lib, f"oracle_spmv_serial_mixed_{o}")(C.c_int32(n), _p(Ap), _p(Aj), _p(Ax), _p(x), _p(y))
        return y

    def spmv_genl_serial(self, semiring, Ap, Aj, Ax, x):
        """Generalized serial SpMV (cpu_navie.hpp:20-34); semiring 0 = (+,*), 1 = (min,+), 2 = (max,*), 3 = (max,+),
        4 = (or,and); Ax / x float32, float64 or int32."""
        n = self._check(Ap, Aj, Ax, x)
        o, v = suffix(Ap, Ax)
        y = np.empty(n, dtype=Ax.dtype)
        getattr(self.lib, f"oracle_spmv_genl_serial_{o}_{v}")(
            C.c_int(semiring), C.c_int32(n), _p(Ap), _p(Aj), _p(Ax), _p(x), _p(y))
        return y

    def spmv_parallel(self, Ap, Aj, Ax, x, n_threads):
        n = self._check(Ap, Aj, Ax, x)
        o, v = suffix(Ap, Ax)
        y = np.empty(n, dtype=Ax.dtype)
        getattr(self.lib, f"oracle_spmv_parallel_{o}_{v}")(
            C.c_int32(n), _p(Ap), _p(Aj), _p(Ax), _p(x), _p(y), C.c_int(n_threads))
        return y

    def spmv_ref64(self, Ap, Aj, Ax, x, n_threads=1):
        """(fp64 serial sum, sum of |a*x|) per row: the two terms of the parity bound.
        n_threads > 1 splits the ROWS over threads; each row is the same serial sum."""
        n = self._check(Ap, Aj, Ax, x)
        o, v = suffix(Ap, Ax)
        y64 = np.empty(n, dtype=np.float64)
        ya = np.empty(n, dtype=np.float64)
        if n_threads > 1:
            getattr(self.lib, f"oracle_spmv_ref64_parallel_{o}_{v}")(
                C.c_int32(n), _p(Ap), _p(Aj), _p(Ax), _p(x), _p(y64), _p(ya), C.c_int(n_threads))
        else:
            getattr(self.lib, f"oracle_spmv_ref64_{o}_{v}")(C.c_int32(n), _p(Ap), _p(Aj), _p(Ax), _p(x), _p(y64), _p(ya))
        return y64, ya

    def spmv_vector_order(self, Ap, Aj, Ax, x, T, aligned_when_longer_than=-1):
        n = self._check(Ap, Aj, Ax, x)
        o, v = suffix(Ap, Ax)
        y = np.empty(n, dtype=Ax.dtype)
        getattr(self.lib, f"oracle_spmv_vector_order_{o}_{v}")(
            C.c_int32(n), _p(Ap), _p(Aj), _p(Ax), _p(x), _p(y), C.c_int(T), C.c_int(aligned_when_longer_than))
        return y

    def spmv_merge_order(self, Ap, Aj, Ax, x, block_threads, ipt):
        n = self._check(Ap, Aj, Ax, x)
        o, v = suffix(Ap, Ax)
        y = np.full(n, np.nan, dtype=Ax.dtype)
        getattr(self.lib, f"oracle_spmv_merge_order_{o}_{v}")(
            C.c_int32(n), _p(Ap), _p(Aj), _p(Ax), _p(x), _p(y), C.c_int(block_threads), C.c_int(ipt))
        return y

    def merge_tile_coords(self, Ap, tile_items):
        """Merge-path coordinates (row, nnz) of every tile boundary 0..n_tiles."""
        n = len(Ap) - 1
        nnz = int(Ap[-1])
        items = n + nnz
        n_tiles = (items + tile_items - 1) // tile_items
        xs = np.empty(n_tiles + 1, dtype=np.int64)
        ys = np.empty(n_tiles + 1, dtype=np.int64)
        o = "i32" if Ap.dtype == np.int32 else "i64"
        getattr(self.lib, f"oracle_merge_tile_coords_{o}")(
            C.c_int32(n), _p(Ap), C.c_int64(tile_items), C.c_int64(n_tiles), _p(xs), _p(ys))
        return xs, ys

    def load_mtx(self, path, off="i32", val="f32"):
        """Matrix Market -> CSR with the reference's ordering (load.hpp).  Returns
        (n_rows, n_cols, Ap, Aj, Ax) or raises ValueError(<error class>)."""
        st = C.c_int(0)
        h = self.lib.oracle_mtx_load(os.fsencode(path), C.c_int(32 if off == "i32" else 64),
                                     C.c_int(1 if val == "f64" else 0), C.byref(st))
        if not h:
            raise ValueError(MTX_ERRORS.get(st.value, str(st.value)))
        h = C.c_void_p(h)
        nr, nc, nz = C.c_int64(), C.c_int64(), C.c_int64()
        self.lib.oracle_mtx_dims(h, C.byref(nr), C.byref(nc), C.byref(nz))
        Ap = np.empty(nr.value + 1, dtype=np.int64)
        Aj = np.empty(nz.value, dtype=np.int32)
        Ax = np.empty(nz.value, dtype=np.float64)
        self.lib.oracle_mtx_copy(h, _p(Ap), _p(Aj), _p(Ax))
        self.lib.oracle_mtx_free(h)
        return (nr.value, nc.value, Ap.astype(_OFF[off][0]), Aj, Ax.astype(_VAL[val][0]))

    def hardware_threads(self):
        return int(self.lib.oracle_hardware_threads())


class Ref:
    """The reference's own headers, compiled from /root/reference (see ref_driver.cpp)."""

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libspmv_ref.so"))

    def __init__(self):
        self.lib = C.CDLL(os.path.join(_HERE, "_ref", "libspmv_ref.so"))
        for o in _OFF:
            for v in _VAL:
                getattr(self.lib, f"ref_load_{o}_{v}").restype = C.c_void_p

    def load_mtx(self, path, off="i32", val="f32"):
        s = f"{off}_{val}"
        h = getattr(self.lib, f"ref_load_{s}")(os.fsencode(path))
        if not h:
            raise ValueError("reference loader threw")
        h = C.c_void_p(h)
        nr, nc, nz = C.c_int64(), C.c_int64(), C.c_int64()
        getattr(self.lib, f"ref_dims_{s}")(h, C.byref(nr), C.byref(nc), C.byref(nz))
        Ap = np.empty(nr.value + 1, dtype=_OFF[off][0])
        Aj = np.empty(nz.value, dtype=np.int32)
        Ax = np.empty(nz.value, dtype=_VAL[val][0])
        getattr(self.lib, f"ref_copy_{s}")(h, _p(Ap), _p(Aj), _p(Ax))
        getattr(self.lib, f"ref_free_{s}")(h)
        return nr.value, nc.value, Ap, Aj, Ax

    def spmv_genl_cpu(self, semiring, n_cols, Ap, Aj, Ax, x):
        o, v = suffix(Ap, Ax)
        n = len(Ap) - 1
        y = np.empty(n, dtype=Ax.dtype)
        nnz = _OFF[o][1](int(Ap[-1]))
        getattr(self.lib, f"ref_spmv_genl_cpu_{o}_{v}")(
            C.c_int(semiring), C.c_int(n), C.c_int(n_cols), nnz, _p(Ap), _p(Aj), _p(Ax), _p(x), _p(y))
        return y

    def spmv_cpu_mixed(self, n_cols, Ap, Aj, Ax, x):
        """The reference's SpMV_cpu_navie<int, off, float, double, double>."""
        assert Ax.dtype == np.float32 and x.dtype == np.float64
        o = "i32" if Ap.dtype == np.int32 else "i64"
        n = len(Ap) - 1
        y = np.empty(n, dtype=np.float64)
        nnz = _OFF[o][1](int(Ap[-1]))
        getattr(self.lib, f"ref_spmv_cpu_mixed_{o}")(C.c_int(n), C.c_int(n_cols), nnz, _p(Ap), _p(Aj), _p(Ax), _p(x), _p(y))
        return y

    def spmv_cpu(self, n_cols, Ap, Aj, Ax, x):
        o, v = suffix(Ap, Ax)
        n = len(Ap) - 1
        y = np.empty(n, dtype=Ax.dtype)
        nnz = _OFF[o][1](int(Ap[-1]))
        getattr(self.lib, f"ref_spmv_cpu_{o}_{v}")(
            C.c_int(n), C.c_int(n_cols), nnz, _p(Ap), _p(Aj), _p(Ax), _p(x), _p(y))
        return y
