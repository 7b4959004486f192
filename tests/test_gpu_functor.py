"""mi355_spmv_functor_*: the generalized SpMV whose functor is the caller's C++ TEXT, compiled for gfx950 at run time.

The reference's SpMV_merge_based_generalized is a template over a functor_t (merge_genl.cuh:19-38) and over five
independent types (spmv.h:29-34); its CPU twin is SpMV_genl_cpu_navie (cpu_navie.hpp:20-34).  The checker here is the
oracle's restatement of that twin for the five functors it knows, and a numpy restatement of the same left fold for
functors of this file's own.  CPU tests: the compile step alone (hiprtc needs no device)."""
import numpy as np
import pytest
import torch

from conftest import random_csr

DEV = "cuda:0"
NP = {"i32": np.int32, "i64": np.int64, "f32": np.float32, "f64": np.float64}
TORCH = {np.int32: torch.int32, np.int64: torch.int64, np.float32: torch.float32, np.float64: torch.float64}

# the five semirings the merge kind enumerates, written the way a user of the reference writes a functor
SEMIRING_TEXT = """
template <typename mat_value_t, typename vec_x_value_t, typename vec_y_value_t>
struct PlusTimesFn {
    __host__ __device__ __forceinline__ static vec_y_value_t initialize() { return vec_y_value_t(0); }
    __host__ __device__ __forceinline__ static vec_y_value_t combine(const mat_value_t& a, const vec_x_value_t& x) { return vec_y_value_t(a * x); }
    __host__ __device__ __forceinline__ static vec_y_value_t reduce(const vec_y_value_t& l, const vec_y_value_t& r) { return l + r; }
};
template <typename M, typename X, typename Y>
struct MinPlusFn {
    __device__ static Y initialize() { return Y(INFINITY); }
    __device__ static Y combine(const M& a, const X& x) { return Y(a + x); }
    __device__ static Y reduce(const Y& l, const Y& r) { return r < l ? r : l; }
};
template <typename M, typename X, typename Y>
struct MaxTimesFn {
    __device__ static Y initialize() { return Y(-INFINITY); }
    __device__ static Y combine(const M& a, const X& x) { return Y(a * x); }
    __device__ static Y reduce(const Y& l, const Y& r) { return l < r ? r : l; }
};
template <typename M, typename X, typename Y>
struct MaxPlusFn {
    __device__ static Y initialize() { return Y(-INFINITY); }
    __device__ static Y combine(const M& a, const X& x) { return Y(a + x); }
    __device__ static Y reduce(const Y& l, const Y& r) { return l < r ? r : l; }
};
template <typename M, typename X, typename Y>
struct OrAndFn {
    __device__ static Y initialize() { return Y(0); }
    __device__ static Y combine(const M& a, const X& x) { return (a != M(0) && x != X(0)) ? Y(1) : Y(0); }
    __device__ static Y reduce(const Y& l, const Y& r) { return (l != Y(0) || r != Y(0)) ? Y(1) : Y(0); }
};
"""
FN = {"plus_times": "PlusTimesFn", "min_plus": "MinPlusFn", "max_times": "MaxTimesFn", "max_plus": "MaxPlusFn",
      "or_and": "OrAndFn"}
CNAME = {np.float32: "float", np.float64: "double", np.int32: "int", np.int64: "long long"}


def d(a):
    return torch.from_numpy(a).to(DEV)


def fold(Ap, Aj, Ax, x, init, combine, reduce_):
    """cpu_navie.hpp:20-34 in numpy: per row a fold over combine(Ax[k], x[Aj[k]]) — here through ufunc.reduceat, valid
    for the exact / order-free reduce functions this file uses (min, max, integer-valued sums)."""
    n = len(Ap) - 1
    terms = combine(Ax, x[Aj])
    out = np.full(n, init, dtype=terms.dtype)
    lens = np.diff(Ap.astype(np.int64))
    rows = np.nonzero(lens > 0)[0]
    if rows.size:
        out[rows] = reduce_.reduceat(terms, Ap[:-1].astype(np.int64)[rows])
    return out


# ---- no GPU: the compile step ---------------------------------------------------------------------------------

def test_compiles_without_a_device_and_reports_errors(sp):
    f = sp.Functor(SEMIRING_TEXT, "MinPlusFn<float, float, double>", torch.int64, torch.float32, torch.float32, torch.float64)
    assert f.log == ""
    f.destroy()
    f.destroy()                                               # idempotent
    with pytest.raises(RuntimeError) as e:
        sp.Functor(SEMIRING_TEXT + "\nstruct Broken { int x }\n", "Broken")
    assert "invalid argument" in str(e.value) and "functor_source:" in str(e.value)     # the caller's line numbers
    with pytest.raises(RuntimeError) as e:                    # a functor without reduce(): the kernels' use of it fails
        sp.Functor("struct NoReduce { __device__ static float initialize() { return 0; } "
                   "__device__ static float combine(float a, float b) { return a * b; } };", "NoReduce")
    assert "reduce" in str(e.value)
    with pytest.raises(RuntimeError) as e:                    # type names are names, not code
        sp.Functor(SEMIRING_TEXT, "PlusTimesFn<float,float,float>", mat="float; int x")
    assert "type names" in str(e.value)
    with pytest.raises(RuntimeError):
        sp.Functor(SEMIRING_TEXT, "")


def test_functor_symbols_are_exported(sp):
    L = sp.capi.lib()
    for n in ("compile", "compile_log", "spmv", "destroy"):
        assert hasattr(L, "mi355_spmv_functor_" + n)


# ---- GPU ------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("semiring", ["plus_times", "min_plus", "max_times", "max_plus", "or_and"])
@pytest.mark.parametrize("off,val", [("i32", "f32"), ("i64", "f64"), ("i64", "f32")])
def test_text_functors_match_the_cpu_twin(sp, oracle, semiring, off, val):
    """The oracle's generalized serial loop (cpu_navie.hpp:20-34) knows five functors: the same five as TEXT must agree
    with it — bit for bit where nothing rounds twice (min / max / or), within the bound of SURVEY §8(c) for (+, *).
    Ragged rows, empty rows (initialize()), one row for the whole-wave kernel."""
    from test_gpu_parity import assert_parity
    rng = np.random.RandomState(5 + len(semiring))
    n_rows, n_cols = 20011, 1500
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 25, NP[off], NP[val], long_row=50000)
    x = (rng.rand(n_cols) * 2 - 1).astype(NP[val])
    if semiring == "or_and":
        Ax = (rng.rand(Ax.size) < 0.5).astype(NP[val])
        x = (rng.rand(n_cols) < 0.25).astype(NP[val])
    c = CNAME[NP[val]]
    f = sp.Functor(SEMIRING_TEXT, "%s<%s, %s, %s>" % (FN[semiring], c, c, c), TORCH[NP[off]], c, c, c)
    y = torch.full((n_rows,), float("nan"), dtype=TORCH[NP[val]], device=DEV)
    f.spmv(n_rows, n_cols, int(Ap[-1]), d(Ap), d(Aj), d(Ax), d(x), y)
    got = y.cpu().numpy()
    assert not np.any(np.isnan(got))
    if semiring == "plus_times":
        assert_parity(oracle, Ap, Aj, Ax, x, got)
    else:
        assert np.array_equal(got, oracle.spmv_genl_serial(sp.capi.SEMIRINGS[semiring], Ap, Aj, Ax, x))


@pytest.mark.gpu
@pytest.mark.parametrize("mean", [1, 3, 6, 12, 24, 48, 100, 300])
def test_every_lanes_per_row_choice(sp, mean):
    """T = 2 .. 64 follows the mean row length; integer-valued data make (+, *) exact in any order."""
    rng = np.random.RandomState(100 + mean)
    n_rows, n_cols = 6007, 4000
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 2 * mean, np.int32, np.float32, empty_frac=0.05, integer_values=True)
    x = rng.randint(-4, 5, size=n_cols).astype(np.float32)
    f = sp.Functor(SEMIRING_TEXT, "PlusTimesFn<float, float, float>")
    y = torch.full((n_rows,), float("nan"), device=DEV)
    f.spmv(n_rows, n_cols, int(Ap[-1]), d(Ap), d(Aj), d(Ax), d(x), y)
    want = fold(Ap, Aj, Ax, x, 0.0, lambda a, b: a * b, np.add)
    assert np.array_equal(y.cpu().numpy(), want)


@pytest.mark.gpu
def test_five_independent_types(sp):
    """spmv.h:29-34: index_t, offset_t, mat_value_t, vec_x_value_t, vec_y_value_t are five template parameters.  Here:
    int32 indices, int64 offsets, an fp64 matrix, an int32 x and an int64 y (a count), through a plain struct."""
    src = """
    struct CountAbove {
        __device__ static long long initialize() { return 0; }
        __device__ static long long combine(const double& a, const int& x) { return a * x > 2.5 ? 1 : 0; }
        __device__ static long long reduce(const long long& l, const long long& r) { return l + r; }
    };"""
    rng = np.random.RandomState(8)
    n_rows, n_cols = 9001, 700
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 60, np.int64, np.float64, long_row=30000)
    x = rng.randint(-5, 6, size=n_cols).astype(np.int32)
    f = sp.Functor(src, "CountAbove", torch.int64, torch.float64, torch.int32, torch.int64)
    y = torch.full((n_rows,), -1, dtype=torch.int64, device=DEV)
    f.spmv(n_rows, n_cols, int(Ap[-1]), d(Ap), d(Aj), d(Ax), d(x), y)
    want = fold(Ap, Aj, Ax, x, 0, lambda a, b: (a * b > 2.5).astype(np.int64), np.add)
    assert np.array_equal(y.cpu().numpy(), want)
    assert want.max() > 100 and (want == 0).any()


@pytest.mark.gpu
def test_a_struct_valued_y_the_source_defines(sp):
    """y_t need not be a number: an arg-max carries (value, column) pairs — x holds the pairs, y receives them.  The
    expected result is the plain loop of cpu_navie.hpp:20-34 over the same records."""
    src = """
    struct Best { float value; int column; };
    struct ArgMax {
        __device__ static Best initialize() { Best b; b.value = -INFINITY; b.column = -1; return b; }
        __device__ static Best combine(const float& a, const Best& x) { Best b; b.value = a * x.value; b.column = x.column; return b; }
        __device__ static Best reduce(const Best& l, const Best& r) {
            if (r.value > l.value) return r;
            if (r.value == l.value && r.column >= 0 && (l.column < 0 || r.column < l.column)) return r;   // ties: the lower column
            return l;
        }
    };"""
    rng = np.random.RandomState(21)
    n_rows, n_cols = 3001, 900
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 30, np.int32, np.float32, long_row=5000)
    rec = np.dtype([("value", np.float32), ("column", np.int32)])
    x = np.empty(n_cols, dtype=rec)
    x["value"] = (rng.rand(n_cols) * 2 - 1).astype(np.float32)
    x["column"] = np.arange(n_cols, dtype=np.int32)
    f = sp.Functor(src, "ArgMax", torch.int32, "float", "Best", "Best")
    dx = torch.from_numpy(x.view(np.int32).reshape(-1)).to(DEV)           # the records as raw 4-byte words
    y = torch.zeros(2 * n_rows, dtype=torch.int32, device=DEV)
    f.spmv(n_rows, n_cols, int(Ap[-1]), d(Ap), d(Aj), d(Ax), dx, y)
    got = y.cpu().numpy().view(rec)
    want = np.empty(n_rows, dtype=rec)
    for r in range(n_rows):
        bv, bc = np.float32(-np.inf), -1
        for k in range(Ap[r], Ap[r + 1]):
            v, c = np.float32(Ax[k] * x["value"][Aj[k]]), int(Aj[k])
            if v > bv or (v == bv and (bc < 0 or c < bc)):
                bv, bc = v, c
        want[r] = (bv, bc)
    assert np.array_equal(got["column"], want["column"])
    assert np.array_equal(got["value"], want["value"])
    assert (want["column"] < 0).any()                                     # empty rows keep initialize()


@pytest.mark.gpu
def test_edge_shapes_and_a_side_stream(sp):
    f = sp.Functor(SEMIRING_TEXT, "MaxPlusFn<float, float, float>")
    # no rows: nothing is touched, no launch
    z = torch.zeros(4, device=DEV)
    zi = torch.zeros(4, dtype=torch.int32, device=DEV)
    f.spmv(0, 5, 0, zi, zi, z, z, z)
    # rows but no nonzeros: every row is initialize()
    Ap = torch.zeros(1001, dtype=torch.int32, device=DEV)
    y = torch.zeros(1000, device=DEV)
    f.spmv(1000, 7, 0, Ap, zi, z, z, y)
    assert torch.all(torch.isinf(y) & (y < 0))
    # one column, on a side stream, twice (asynchronous: the stream orders the two calls)
    rng = np.random.RandomState(3)
    Ap, Aj, Ax = random_csr(rng, 5000, 1, 9, np.int32, np.float32)
    x = np.array([0.25], dtype=np.float32)
    s = torch.cuda.Stream()
    dAp, dAj, dAx, dx = d(Ap), d(Aj), d(Ax), d(x)
    y = torch.full((5000,), float("nan"), device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        f.spmv(5000, 1, int(Ap[-1]), dAp, dAj, dAx, dx, y, stream=s)
        f.spmv(5000, 1, int(Ap[-1]), dAp, dAj, dAx, dx, y, stream=s)
    s.synchronize()
    want = fold(Ap, Aj, Ax, x, np.float32(-np.inf), lambda a, b: a + b, np.maximum)
    assert np.array_equal(y.cpu().numpy(), want)
    # wrong offset dtype for the compiled handle: refused on the host
    with pytest.raises(TypeError):
        f.spmv(5000, 1, int(Ap[-1]), dAp.to(torch.int64), dAj, dAx, dx, y)


@pytest.mark.gpu
def test_target_shape_throughput_is_reported(sp):
    """Not a parity test: the general path on the S32-band shape at 2^20 rows, so that DESIGN.md's "general path, plain
    gathers" has a number behind it (printed; asserted only to be a sane bandwidth)."""
    m = sp.synth.workload("s32-band", DEV, scale_down=4)
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, DEV)
    f = sp.Functor(SEMIRING_TEXT, "PlusTimesFn<float, float, float>")
    y = torch.empty(m.n_rows, device=DEV)
    y2 = torch.empty(m.n_rows, device=DEV)
    for _ in range(20):
        f.spmv(m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, y)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        f.spmv(m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, y)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    sp.spmv("vector", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, y2)
    torch.cuda.synchronize()
    assert torch.allclose(y, y2, rtol=1e-4, atol=1e-4)
    gbps = m.algorithmic_bytes() / us / 1e3
    print("functor (+,*) on S32-band 2^20: %.1f us, %.0f GB/s algorithmic" % (us, gbps))
    assert 300 < gbps < 8000
