"""ctypes binding of libmi355spmv.so (the C ABI of include/mi355_spmv.h).

This is plumbing for tests and bench.py: torch provides device memory and
streams, the library does the work.  There is NO fallback: if the shared library
is missing or a call fails, a RuntimeError is raised.

The reference interface this mirrors (same argument order and meaning):
    SpMV(kind_str, n_rows, n_cols, nnz, Ap, Aj, Ax, x, y)   include/spmv.h:29-34
"""
import ctypes as C
import os

import torch  # imported before the library so both share one HIP runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355_SPMV_LIB") or os.path.join(_HERE, "lib", "libmi355spmv.so")

KINDS = {"vector": 0, "merge": 1, "light": 2, "auto": 100}   # auto: the library picks (MI355_KIND_AUTO)
KIND_NAMES = {0: "vector", 1: "merge", 2: "light"}
# labels the C++ host header registers in SPMV_KINDS (host/spmv.h)
LABELS = {"hip_vector": "vector", "hip_merge": "merge", "hip_light": "light", "hip_auto": "auto"}
OFF_TYPES = {torch.int32: (0, "i32"), torch.int64: (1, "i64")}
VAL_TYPES = {torch.float32: (0, "f32"), torch.float64: (1, "f64"), torch.int32: (2, "i32")}   # (int32 values: the merge kind only)
PLAN_REUSE_STRUCTURE = 1
SEMIRINGS = {"plus_times": 0, "min_plus": 1, "max_times": 2, "max_plus": 3, "or_and": 4}

EXPORTS = (
    ["mi355_spmv_%s_%s_%s" % (k, o, v) for k in KINDS for o in ("i32", "i64") for v in ("f32", "f64")]
    + ["mi355_spmv_merge_genl_%s_%s" % (o, v) for o in ("i32", "i64") for v in ("f32", "f64", "i32")]
    + ["mi355_spmv_plan_set_semiring", "mi355_spmv_plan_set_alpha_beta"]
    + ["mi355_spmv_plan_create", "mi355_spmv_plan_execute", "mi355_spmv_plan_destroy",
       "mi355_spmv_plan_get_info", "mi355_spmv_stream_synchronize", "mi355_spmv_plan_merge_coords", "mi355_spmv_version",
       "mi355_spmv_status_string", "mi355_spmv_last_error", "mi355_spmv_device_count"]
    + ["mi355_spmv_plan_get_shape", "mi355_spmv_plan_partition", "mi355_spmv_plan_create_block",
       "mi355_spmv_knobs_reload", "mi355_spmv_cache_release", "mi355_spmv_plan_acquire", "mi355_spmv_plan_release",
       "mi355_spmv_plan_create_typed", "mi355_spmv_merge_f32mat_f64vec_i32",
       "mi355_spmv_merge_f32mat_f64vec_i64"]
    + ["mi355_spmv_dist_" + n for n in ("create_local", "unique_id", "create_rank", "scatter_values", "replicate_x",
                                        "execute", "execute_ex", "set_exchange", "get_info", "structure_changed",
                                        "set_alpha_beta", "parts", "cuts", "part_info", "device_y", "device_x",
                                        "destroy")]
    + ["mi355_spmv_functor_" + n for n in ("compile", "compile_log", "spmv", "destroy")]
)


class PlanInfo(C.Structure):
    _fields_ = [("kind", C.c_int32), ("off_type", C.c_int32), ("val_type", C.c_int32),
                ("lanes_per_row", C.c_int32), ("elems_per_lane", C.c_int32), ("block_threads", C.c_int32),
                ("grid_blocks", C.c_int64), ("tile_items", C.c_int64), ("n_tiles", C.c_int64),
                ("rows_per_chunk", C.c_int64), ("scratch_bytes", C.c_int64), ("n_kernels", C.c_int32),
                ("window_elems", C.c_int32), ("window_segments", C.c_int32), ("main_kernel", C.c_char * 64),
                ("balanced_chunks", C.c_int32), ("rows_cap", C.c_int32), ("n_chunks", C.c_int64),
                ("knobs", C.c_char * 160)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["main_kernel"] = d["main_kernel"].decode()
        d["knobs"] = d["knobs"].decode()
        return d


class PlanShape(C.Structure):
    """mi355_spmv_plan_shape: the launch-shape decisions of a plan as plain data (picklable via bytes(shape))."""
    _fields_ = [("struct_bytes", C.c_int32), ("kind", C.c_int32), ("off_type", C.c_int32), ("val_type", C.c_int32),
                ("n_rows", C.c_int32), ("n_cols", C.c_int32), ("nnz", C.c_int64),
                ("lanes_per_row", C.c_int32), ("elems_per_lane", C.c_int32), ("block_threads", C.c_int32),
                ("balanced_chunks", C.c_int32), ("rows_cap", C.c_int32), ("giant_rows_enabled", C.c_int32),
                ("rows_per_chunk", C.c_int64), ("n_chunks", C.c_int64), ("bal_k", C.c_int64), ("bal_q", C.c_int64),
                ("giant_len", C.c_int64),
                ("window_elems", C.c_int32), ("window_bytes", C.c_int32), ("window_from_band", C.c_int32),
                ("window_segments", C.c_int32), ("probe_ok", C.c_int32), ("long_steps", C.c_int32),
                ("band_lo", C.c_int64), ("band_hi", C.c_int64), ("seg_lo", C.c_int64 * 4), ("seg_hi", C.c_int64 * 4),
                ("window_sweep", C.c_int32), ("small_plain", C.c_int32)]


class DistInfo(C.Structure):
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("sub_blocks", C.c_int32), ("local_mode", C.c_int32),
                ("exchange", C.c_int32), ("auto_picked", C.c_int32), ("allgather_in_place", C.c_int32),
                ("reserved0", C.c_int32), ("trial_us", C.c_float * 4), ("max_block_rows", C.c_int64),
                ("staging_bytes", C.c_int64), ("exchange_name", C.c_char * 16)]


EXCHANGES = {"auto": 0, "bcast": 1, "sendrecv": 2, "allgather": 3}
EXEC_DEFAULT, EXEC_SKIP_EXCHANGE, EXEC_EXCHANGE_ONLY = 0, 1, 2

_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.mi355_spmv_status_string.restype = C.c_char_p
        L.mi355_spmv_last_error.restype = C.c_char_p
        L.mi355_spmv_plan_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int32,
                                             C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        L.mi355_spmv_plan_create_typed.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        L.mi355_spmv_plan_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mi355_spmv_plan_destroy.argtypes = [C.c_void_p]
        L.mi355_spmv_plan_set_semiring.argtypes = [C.c_void_p, C.c_int]
        L.mi355_spmv_plan_set_alpha_beta.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.mi355_spmv_plan_get_info.argtypes = [C.c_void_p, C.POINTER(PlanInfo)]
        L.mi355_spmv_plan_merge_coords.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mi355_spmv_plan_get_shape.argtypes = [C.c_void_p, C.POINTER(PlanShape)]
        L.mi355_spmv_plan_partition.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mi355_spmv_plan_create_block.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                   C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                                   C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        L.mi355_spmv_dist_create_local.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int32,
                                                   C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                   C.c_int, C.c_int]
        L.mi355_spmv_dist_unique_id.argtypes = [C.c_void_p]
        L.mi355_spmv_dist_create_rank.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        L.mi355_spmv_dist_scatter_values.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mi355_spmv_dist_replicate_x.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mi355_spmv_dist_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mi355_spmv_dist_execute_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.mi355_spmv_dist_set_exchange.argtypes = [C.c_void_p, C.c_int]
        L.mi355_spmv_dist_get_info.argtypes = [C.c_void_p, C.POINTER(DistInfo)]
        L.mi355_spmv_dist_structure_changed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                        C.POINTER(C.c_int)]
        L.mi355_spmv_dist_set_alpha_beta.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.mi355_spmv_dist_parts.argtypes = [C.c_void_p]
        L.mi355_spmv_dist_cuts.argtypes = [C.c_void_p, C.c_void_p]
        L.mi355_spmv_dist_part_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(PlanInfo)]
        L.mi355_spmv_dist_device_y.argtypes = [C.c_void_p, C.c_int]
        L.mi355_spmv_dist_device_y.restype = C.c_void_p
        L.mi355_spmv_dist_device_x.argtypes = [C.c_void_p, C.c_int]
        L.mi355_spmv_dist_device_x.restype = C.c_void_p
        L.mi355_spmv_dist_destroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _check(status, what):
    if status != 0:
        L = lib()
        raise RuntimeError("%s failed: %s (%s)" % (
            what, L.mi355_spmv_status_string(status).decode(), L.mi355_spmv_last_error().decode()))


def _require_device(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("mi355 spmv takes device tensors only (no CPU path exists)")
        if not t.is_contiguous():
            raise RuntimeError("mi355 spmv takes contiguous tensors")


def _stream_ptr(stream):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def spmv(kind, n_rows, n_cols, nnz, Ap, Aj, Ax, x, y, stream=None):
    """One-shot call: the reference's `SpMV(kind_str, n_rows, n_cols, nnz, Ap, Aj, Ax, x, y)`
    (include/spmv.h:29-34) for kind in {"vector","merge","light"} (or their
    SPMV_KINDS labels "hip_vector", ...).  Synchronises the stream before returning."""
    kind = LABELS.get(kind, kind)
    if kind not in KINDS:
        # the reference prints 'SpMV kind "<k>" is NOT SUPPROT' and exits (spmv.h:46-47)
        raise ValueError('SpMV kind "%s" is NOT SUPPORTED' % kind)
    _require_device(Ap, Aj, Ax, x, y)
    if Aj.dtype != torch.int32 or Ax.dtype != x.dtype or Ax.dtype != y.dtype:
        raise TypeError("Aj must be int32 and Ax, x, y one value type")
    if Ax.dtype == torch.int32:            # integer values exist for the (generalized) merge kind only
        if LABELS.get(kind, kind) not in ("merge", "auto"):
            raise RuntimeError("mi355_spmv: integer values are not supported by the %s kind (merge only)" % kind)
        return spmv_genl("plus_times", n_rows, n_cols, nnz, Ap, Aj, Ax, x, y, stream)
    o = OFF_TYPES[Ap.dtype][1]
    v = VAL_TYPES[Ax.dtype][1]
    fn = getattr(lib(), "mi355_spmv_%s_%s_%s" % (kind, o, v))
    nnz_c = C.c_int32(nnz) if o == "i32" else C.c_int64(nnz)
    st = fn(C.c_int32(n_rows), C.c_int32(n_cols), nnz_c, C.c_void_p(Ap.data_ptr()), C.c_void_p(Aj.data_ptr()),
            C.c_void_p(Ax.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), _stream_ptr(stream))
    _check(st, "mi355_spmv_%s_%s_%s" % (kind, o, v))
    return y


def spmv_mixed(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y, stream=None):
    """One-shot merge-path SpMV with an fp32 matrix under fp64 vectors (mi355_spmv_merge_f32mat_f64vec_*)."""
    _require_device(Ap, Aj, Ax, x, y)
    if Aj.dtype != torch.int32 or Ax.dtype != torch.float32 or x.dtype != torch.float64 or y.dtype != torch.float64:
        raise TypeError("Aj int32, Ax float32, x and y float64")
    o = OFF_TYPES[Ap.dtype][1]
    fn = getattr(lib(), "mi355_spmv_merge_f32mat_f64vec_%s" % o)
    nnz_c = C.c_int32(nnz) if o == "i32" else C.c_int64(nnz)
    st = fn(C.c_int32(n_rows), C.c_int32(n_cols), nnz_c, C.c_void_p(Ap.data_ptr()), C.c_void_p(Aj.data_ptr()),
            C.c_void_p(Ax.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), _stream_ptr(stream))
    _check(st, "mi355_spmv_merge_f32mat_f64vec_%s" % o)
    return y


def cache_release():
    """Destroy the plans the one-shot entry points keep between calls (mi355_spmv_cache_release)."""
    _check(lib().mi355_spmv_cache_release(), "mi355_spmv_cache_release")


def spmv_genl(semiring, n_rows, n_cols, nnz, Ap, Aj, Ax, x, y, stream=None):
    """Generalized merge-path SpMV: the reference's SpMV_merge_based_generalized
    (include/spmv/merge_genl/merge_genl.cuh:41-79) with the semiring as an argument."""
    sr = SEMIRINGS[semiring] if isinstance(semiring, str) else int(semiring)
    _require_device(Ap, Aj, Ax, x, y)
    if Aj.dtype != torch.int32 or Ax.dtype != x.dtype or Ax.dtype != y.dtype:
        raise TypeError("Aj must be int32 and Ax, x, y one value type")
    o = OFF_TYPES[Ap.dtype][1]
    v = VAL_TYPES[Ax.dtype][1]
    fn = getattr(lib(), "mi355_spmv_merge_genl_%s_%s" % (o, v))
    nnz_c = C.c_int32(nnz) if o == "i32" else C.c_int64(nnz)
    st = fn(C.c_int(sr), C.c_int32(n_rows), C.c_int32(n_cols), nnz_c, C.c_void_p(Ap.data_ptr()),
            C.c_void_p(Aj.data_ptr()), C.c_void_p(Ax.data_ptr()), C.c_void_p(x.data_ptr()),
            C.c_void_p(y.data_ptr()), _stream_ptr(stream))
    _check(st, "mi355_spmv_merge_genl_%s_%s" % (o, v))
    return y


class Plan:
    """Scratch + launch shapes kept across calls (mi355_spmv_plan_*).  Holds
    references to Ap and Aj so they outlive the plan."""

    def __init__(self, kind, n_rows, n_cols, nnz, Ap, Aj, val_dtype, flags=0, mat_dtype=None):
        """val_dtype: the type of x and y (and of all arithmetic).  mat_dtype: the type the matrix values are stored
        in — None = val_dtype; torch.float32 under torch.float64 vectors is built for the merge kind
        (mi355_spmv_plan_create_typed; the reference's operator keeps the three value types apart, spmv.h:29-34)."""
        kind = LABELS.get(kind, kind)
        if kind not in KINDS:
            raise ValueError('SpMV kind "%s" is NOT SUPPORTED' % kind)
        _require_device(Ap, Aj)
        if Aj.dtype != torch.int32:
            raise TypeError("Aj must be int32")
        self.kind, self.n_rows, self.n_cols, self.nnz = kind, n_rows, n_cols, nnz
        self.Ap, self.Aj, self.val_dtype = Ap, Aj, val_dtype
        self.mat_dtype = mat_dtype if mat_dtype is not None else val_dtype
        self._h = C.c_void_p()
        if self.mat_dtype == val_dtype:
            st = lib().mi355_spmv_plan_create(C.byref(self._h), KINDS[kind], OFF_TYPES[Ap.dtype][0],
                                              VAL_TYPES[val_dtype][0], n_rows, n_cols, nnz,
                                              C.c_void_p(Ap.data_ptr()), C.c_void_p(Aj.data_ptr()), flags)
            _check(st, "mi355_spmv_plan_create")
        else:
            st = lib().mi355_spmv_plan_create_typed(C.byref(self._h), KINDS[kind], OFF_TYPES[Ap.dtype][0],
                                                    VAL_TYPES[self.mat_dtype][0], VAL_TYPES[val_dtype][0],
                                                    VAL_TYPES[val_dtype][0], n_rows, n_cols, nnz,
                                                    C.c_void_p(Ap.data_ptr()), C.c_void_p(Aj.data_ptr()), flags)
            _check(st, "mi355_spmv_plan_create_typed")

    def execute(self, Ax, x, y, stream=None):
        """Asynchronous on `stream` (default: torch's current stream)."""
        _require_device(Ax, x, y)
        if Ax.dtype != getattr(self, "mat_dtype", self.val_dtype) or x.dtype != self.val_dtype or y.dtype != self.val_dtype:
            raise TypeError("value type differs from the plan's")
        if Ax.numel() < self.nnz or x.numel() < self.n_cols or y.numel() < self.n_rows:
            raise ValueError("operand shorter than the plan's sizes")
        st = lib().mi355_spmv_plan_execute(self._h, C.c_void_p(Ax.data_ptr()), C.c_void_p(x.data_ptr()),
                                           C.c_void_p(y.data_ptr()), _stream_ptr(stream))
        _check(st, "mi355_spmv_plan_execute")
        return y

    def set_semiring(self, semiring):
        sr = SEMIRINGS[semiring] if isinstance(semiring, str) else int(semiring)
        _check(lib().mi355_spmv_plan_set_semiring(self._h, C.c_int(sr)), "mi355_spmv_plan_set_semiring")

    def set_alpha_beta(self, alpha, beta):
        """y = alpha * A x + beta * y for the following executes (default 1, 0)."""
        _check(lib().mi355_spmv_plan_set_alpha_beta(self._h, C.c_double(alpha), C.c_double(beta)),
               "mi355_spmv_plan_set_alpha_beta")

    def info(self):
        pi = PlanInfo()
        _check(lib().mi355_spmv_plan_get_info(self._h, C.byref(pi)), "mi355_spmv_plan_get_info")
        return pi.as_dict()

    def shape(self):
        """The plan's launch-shape decisions (mi355_spmv_plan_get_shape)."""
        sh = PlanShape()
        _check(lib().mi355_spmv_plan_get_shape(self._h, C.byref(sh)), "mi355_spmv_plan_get_shape")
        return sh

    def partition(self, parts):
        """nnz-balanced cuts on the plan's chunk boundaries: (row_cuts, chunk_cuts, nnz_cuts), parts + 1 each."""
        import numpy as np
        arrs = [np.zeros(parts + 1, dtype=np.int64) for _ in range(3)]
        _check(lib().mi355_spmv_plan_partition(self._h, parts, *[a.ctypes.data_as(C.c_void_p) for a in arrs]),
               "mi355_spmv_plan_partition")
        return tuple(a.tolist() for a in arrs)

    @classmethod
    def block(cls, kind, whole_shape, row_begin, chunk_begin, n_chunks, nnz_begin_whole, n_rows, n_cols, nnz_end,
              Ap, Aj, val_dtype, flags=0):
        """Plan for a row block given as a 16-byte-aligned VIEW of the whole CSR (Ap[0] in 0..3, nnz_end =
        Ap[n_rows]); inherits `whole_shape` (a PlanShape, or None) — mi355_spmv_plan_create_block."""
        kind = LABELS.get(kind, kind)
        _require_device(Ap, Aj)
        self = cls.__new__(cls)
        self.kind, self.n_rows, self.n_cols, self.nnz = kind, n_rows, n_cols, nnz_end
        self.Ap, self.Aj, self.val_dtype = Ap, Aj, val_dtype
        self._h = C.c_void_p()
        st = lib().mi355_spmv_plan_create_block(
            C.byref(self._h), KINDS[kind], OFF_TYPES[Ap.dtype][0], VAL_TYPES[val_dtype][0],
            C.cast(C.byref(whole_shape), C.c_void_p) if whole_shape is not None else None,
            row_begin, chunk_begin, n_chunks, nnz_begin_whole, n_rows, n_cols, nnz_end,
            C.c_void_p(Ap.data_ptr()), C.c_void_p(Aj.data_ptr()), flags)
        _check(st, "mi355_spmv_plan_create_block")
        return self

    def merge_coords(self):
        import numpy as np
        n = self.info()["n_tiles"] + 1
        rows = np.empty(n, dtype=np.int64)
        nz = np.empty(n, dtype=np.int64)
        _check(lib().mi355_spmv_plan_merge_coords(self._h, rows.ctypes.data_as(C.c_void_p),
                                                  nz.ctypes.data_as(C.c_void_p)), "mi355_spmv_plan_merge_coords")
        return rows, nz

    def destroy(self):
        if self._h:
            lib().mi355_spmv_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


C_TYPE_NAMES = {torch.float32: "float", torch.float64: "double", torch.int32: "int", torch.int64: "long long"}


class Functor:
    """mi355_spmv_functor_*: a generalized SpMV whose functor is C++ source text, compiled for gfx950 at run time.

    source        text defining the functor the way the reference writes one (merge_genl.cuh:19-38): a struct with
                  static initialize() / combine(nonzero, x) / reduce(lhs, rhs)
    functor_type  the type to use ("MyFunctor", "MergeFunctor<float, float, double>")
    off_dtype     torch.int32 / torch.int64;  mat / x / y: torch dtypes or C++ type names (a struct of the source)
    Compiling needs no device; .spmv() runs on the tensors' device, asynchronously on `stream`."""

    def __init__(self, source, functor_type, off_dtype=torch.int32, mat=torch.float32, x=torch.float32, y=torch.float32):
        self._h = C.c_void_p()
        self.off_dtype = off_dtype
        names = [t if isinstance(t, str) else C_TYPE_NAMES[t] for t in (mat, x, y)]
        L = lib()
        L.mi355_spmv_functor_compile.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_int, C.c_char_p,
                                                 C.c_char_p, C.c_char_p]
        L.mi355_spmv_functor_compile_log.restype = C.c_char_p
        L.mi355_spmv_functor_spmv.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64] + [C.c_void_p] * 6
        L.mi355_spmv_functor_destroy.argtypes = [C.c_void_p]
        st = L.mi355_spmv_functor_compile(C.byref(self._h), source.encode(), functor_type.encode(), OFF_TYPES[off_dtype][0],
                                          names[0].encode(), names[1].encode(), names[2].encode())
        self.log = (L.mi355_spmv_functor_compile_log() or b"").decode(errors="replace")
        _check(st, "mi355_spmv_functor_compile")

    def spmv(self, n_rows, n_cols, nnz, Ap, Aj, Ax, x, y, stream=None):
        _require_device(Ap, Aj, Ax, x, y)
        if Ap.dtype != self.off_dtype or Aj.dtype != torch.int32:
            raise TypeError("Functor.spmv: Ap must be %s and Aj int32" % self.off_dtype)
        _check(lib().mi355_spmv_functor_spmv(self._h, n_rows, n_cols, nnz, Ap.data_ptr(), Aj.data_ptr(), Ax.data_ptr(),
                                             x.data_ptr(), y.data_ptr(), _stream_ptr(stream)), "mi355_spmv_functor_spmv")

    def destroy(self):
        if self._h:
            lib().mi355_spmv_functor_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class DistPlan:
    """mi355_spmv_dist_*: row blocks over the GPUs of a node, x replicated, allgatherv(y) over RCCL.

    DistPlan.local(...)  one process drives `devices` (the whole structure lives on the current device)
    DistPlan.rank(...)   one process per GPU: this rank's slice + the global cut lists + a 128-byte id
    """

    def __init__(self):
        self._h = C.c_void_p()
        self._keep = ()

    @classmethod
    def local(cls, kind, n_rows, n_cols, nnz, Ap, Aj, val_dtype, parts=None, devices=None, sub_blocks=None, flags=0):
        kind = LABELS.get(kind, kind)
        _require_device(Ap, Aj)
        devices = list(devices) if devices is not None else [Ap.device.index or 0]
        if sub_blocks is None:
            sub_blocks = (parts // len(devices)) if parts else 1
        if parts is not None and parts != sub_blocks * len(devices):
            raise ValueError("parts must be len(devices) * sub_blocks")
        self = cls()
        self.kind, self.n_rows, self.n_cols, self.nnz, self.val_dtype = kind, n_rows, n_cols, nnz, val_dtype
        self._keep = (Ap, Aj)
        devs = (C.c_int * len(devices))(*devices)
        with torch.cuda.device(Ap.device):
            st = lib().mi355_spmv_dist_create_local(
                C.byref(self._h), KINDS[kind], OFF_TYPES[Ap.dtype][0], VAL_TYPES[val_dtype][0], n_rows, n_cols, nnz,
                C.c_void_p(Ap.data_ptr()), C.c_void_p(Aj.data_ptr()), len(devices), C.cast(devs, C.c_void_p),
                sub_blocks, flags)
        _check(st, "mi355_spmv_dist_create_local")
        return self

    @staticmethod
    def unique_id():
        """128 bytes from ONE rank; hand them to every rank by your own means."""
        buf = (C.c_char * 128)()
        _check(lib().mi355_spmv_dist_unique_id(C.cast(buf, C.c_void_p)), "mi355_spmv_dist_unique_id")
        return bytes(buf)

    @classmethod
    def rank(cls, kind, rank, world, unique_id, parts_per_rank, row_cuts, chunk_cuts, nnz_cuts, whole_shape,
             n_cols, n_rows_local, nnz_end_local, Ap_local, Aj_local, val_dtype, flags=0):
        import numpy as np
        kind = LABELS.get(kind, kind)
        _require_device(Ap_local, Aj_local)
        self = cls()
        self.kind, self.n_rows, self.n_cols, self.val_dtype = kind, int(row_cuts[-1]), n_cols, val_dtype
        self.nnz = int(nnz_cuts[-1]) - int(nnz_cuts[0])
        self._keep = (Ap_local, Aj_local)
        rc = np.ascontiguousarray(row_cuts, dtype=np.int64)
        cc = np.ascontiguousarray(chunk_cuts if chunk_cuts is not None else [0] * len(row_cuts), dtype=np.int64)
        nc = np.ascontiguousarray(nnz_cuts, dtype=np.int64)
        idbuf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        with torch.cuda.device(Ap_local.device):
            st = lib().mi355_spmv_dist_create_rank(
                C.byref(self._h), KINDS[kind], OFF_TYPES[Ap_local.dtype][0], VAL_TYPES[val_dtype][0], rank, world,
                C.cast(idbuf, C.c_void_p) if idbuf is not None else None, parts_per_rank,
                rc.ctypes.data_as(C.c_void_p), cc.ctypes.data_as(C.c_void_p), nc.ctypes.data_as(C.c_void_p),
                C.cast(C.byref(whole_shape), C.c_void_p) if whole_shape is not None else None,
                n_cols, n_rows_local, nnz_end_local, C.c_void_p(Ap_local.data_ptr()), C.c_void_p(Aj_local.data_ptr()),
                flags)
        _check(st, "mi355_spmv_dist_create_rank")
        return self

    def cuts(self):
        import numpy as np
        n = lib().mi355_spmv_dist_parts(self._h)
        a = np.zeros(n + 1, dtype=np.int64)
        _check(lib().mi355_spmv_dist_cuts(self._h, a.ctypes.data_as(C.c_void_p)), "mi355_spmv_dist_cuts")
        return a.tolist()

    def info(self, part=0):
        """Launch shape of this process's block `part`."""
        pi = PlanInfo()
        _check(lib().mi355_spmv_dist_part_info(self._h, part, C.byref(pi)), "mi355_spmv_dist_part_info")
        return pi.as_dict()

    def scatter_values(self, Ax, stream=None):
        _require_device(Ax)
        _check(lib().mi355_spmv_dist_scatter_values(self._h, C.c_void_p(Ax.data_ptr()), _stream_ptr(stream)),
               "mi355_spmv_dist_scatter_values")

    def replicate_x(self, x, stream=None):
        _require_device(x)
        _check(lib().mi355_spmv_dist_replicate_x(self._h, C.c_void_p(x.data_ptr()), _stream_ptr(stream)),
               "mi355_spmv_dist_replicate_x")

    def execute(self, Ax, x, y, stream=None, flags=EXEC_DEFAULT):
        """Asynchronous on `stream`.  LOCAL: home-device Ax / x (None = unchanged since scatter_values /
        replicate_x) and the full y; RANK: this rank's values view, its x, its full-length y.
        flags: EXEC_SKIP_EXCHANGE (kernels only) / EXEC_EXCHANGE_ONLY (the allgatherv of whatever y holds)."""
        for t in (Ax, x, y):
            if t is not None:
                _require_device(t)
                if t.dtype != self.val_dtype:
                    raise TypeError("value type differs from the plan's")
        if y.numel() < self.n_rows:
            raise ValueError("y is shorter than the matrix has rows")
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        with torch.cuda.device(y.device):
            st = lib().mi355_spmv_dist_execute_ex(self._h, p(Ax), p(x), p(y), _stream_ptr(stream), flags)
        _check(st, "mi355_spmv_dist_execute_ex")
        return y

    def set_exchange(self, name):
        """'bcast' | 'sendrecv' | 'allgather' (collective: every rank the same)."""
        _check(lib().mi355_spmv_dist_set_exchange(self._h, EXCHANGES[name]), "mi355_spmv_dist_set_exchange")

    def dist_info(self):
        di = DistInfo()
        _check(lib().mi355_spmv_dist_get_info(self._h, C.byref(di)), "mi355_spmv_dist_get_info")
        d = {n: getattr(di, n) for n, _ in di._fields_ if n not in ("trial_us", "exchange_name", "reserved0")}
        d["exchange_name"] = di.exchange_name.decode()
        d["trial_us"] = {k: float(di.trial_us[v]) for k, v in EXCHANGES.items() if v}
        return d

    def structure_changed(self, Ap, Aj, stream=None):
        """LOCAL mode: does the caller's Ap / Aj still match the fingerprint taken at create?"""
        _require_device(Ap, Aj)
        out = C.c_int(0)
        with torch.cuda.device(Ap.device):
            st = lib().mi355_spmv_dist_structure_changed(self._h, C.c_void_p(Ap.data_ptr()), C.c_void_p(Aj.data_ptr()),
                                                         _stream_ptr(stream), C.byref(out))
        _check(st, "mi355_spmv_dist_structure_changed")
        return bool(out.value)

    def set_alpha_beta(self, alpha, beta):
        _check(lib().mi355_spmv_dist_set_alpha_beta(self._h, C.c_double(alpha), C.c_double(beta)),
               "mi355_spmv_dist_set_alpha_beta")

    def destroy(self):
        if self._h:
            lib().mi355_spmv_dist_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
