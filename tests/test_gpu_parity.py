"""GPU suite: the HIP path (through the C ABI) against the oracle.

Bar (north_star): Ap/Aj handling and every integer-valued case
bit-exact; floating point inside the per-row bound
    |y_gpu[r] - y64[r]| <= (len_r + 2) * eps * sum_k |Ax[k] x[Aj[k]]|,
eps = 2^-24 (fp32) / 2^-53 (fp64), y64 = fp64 serial sum — valid for any summation
order and FMA use.  y is NaN-poisoned before every call, so a
skipped row cannot pass.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, parity_bound, random_csr, seeded_x, unhex

pytestmark = pytest.mark.gpu

KINDS = ["vector", "merge", "light"]
COMBOS = [("i32", "f32"), ("i32", "f64"), ("i64", "f32"), ("i64", "f64")]
NP = {"i32": np.int32, "i64": np.int64, "f32": np.float32, "f64": np.float64}
FIXTURES = sorted(f for f in os.listdir(GOLD) if f.endswith(".mtx"))
DEV = "cuda:0"


def gpu_spmv(sp, kind, n_cols, Ap, Aj, Ax, x, plan=False, flags=0):
    """numpy in, numpy out, through the C ABI; y poisoned with NaN first."""
    n_rows = len(Ap) - 1
    nnz = int(Ap[-1]) if n_rows >= 0 and len(Ap) else 0
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    dAp, dAj, dAx, dx = d(Ap), d(Aj), d(Ax), d(x)
    y = torch.full((max(n_rows, 0),), float("nan"), dtype=dAx.dtype, device=DEV)
    if plan:
        p = sp.Plan(kind, n_rows, n_cols, nnz, dAp, dAj, dAx.dtype, flags)
        p.execute(dAx, dx, y)
        torch.cuda.synchronize()
        p.destroy()
    else:
        sp.spmv(kind, n_rows, n_cols, nnz, dAp, dAj, dAx, dx, y)
    return y.cpu().numpy()


def assert_parity(oracle, Ap, Aj, Ax, x, y, exact=False):
    assert not np.any(np.isnan(y)), "a row was skipped (NaN poison survived)"
    if exact:
        assert np.array_equal(y, oracle.spmv_serial(Ap, Aj, Ax, x))
        return
    y64, bound = parity_bound(oracle, Ap, Aj, Ax, x)
    err = np.abs(y.astype(np.float64) - y64)
    bad = np.nonzero(err > bound)[0]
    assert bad.size == 0, "rows outside the bound: %s (err %s, bound %s)" % (bad[:5], err[bad[:5]], bound[bad[:5]])


# ---- the reference's own inputs/outputs (golden vectors made from its build) -----------------

@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("off,val", COMBOS)
def test_golden_fixtures(sp, oracle, golden, kind, off, val):
    for name in FIXTURES:
        nr, nc, Ap, Aj, Ax = oracle.load_mtx(os.path.join(GOLD, name), off, val)
        g = golden[name]
        for xname, x in (("y_ones", np.ones(nc, dtype=NP[val])), ("y_seeded", seeded_x(nc, NP[val]))):
            y = gpu_spmv(sp, kind, nc, Ap, Aj, Ax, x)
            want = unhex(g[val][xname], NP[val])
            assert not np.any(np.isnan(y)), name
            integer_valued = name in ("pat3x4_dup_unsorted.mtx", "lattice9_cub_doc.mtx", "int5_general.mtx")
            if integer_valued and xname == "y_ones":
                assert np.array_equal(y, want), name                 # bit-exact
            else:
                y64, bound = parity_bound(oracle, Ap, Aj, Ax, x)
                assert np.all(np.abs(y.astype(np.float64) - y64) <= bound), name
                assert np.all(np.abs(want.astype(np.float64) - y64) <= bound), name


# ---- ragged random matrices: empty rows, duplicates, unsorted columns, one long row ----------

@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("off,val", COMBOS)
@pytest.mark.parametrize("shape", [(1, 1, 1, None), (5, 3, 2, None), (257, 100, 9, None), (4099, 700, 40, 30000),
                                   (20011, 5000, 6, 9000), (3000, 1, 2, None), (1500, 2000, 300, None)])
def test_random_ragged(sp, oracle, kind, off, val, shape):
    n_rows, n_cols, max_len, long_row = shape
    rng = np.random.RandomState(n_rows * 7 + max_len)
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, max_len, NP[off], NP[val], long_row=long_row)
    x = (rng.rand(n_cols) * 2 - 1).astype(NP[val])
    assert_parity(oracle, Ap, Aj, Ax, x, gpu_spmv(sp, kind, n_cols, Ap, Aj, Ax, x))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("off,val", COMBOS)
def test_integer_valued_is_bit_exact(sp, oracle, kind, off, val):
    """Small integers sum exactly in any order -> the GPU result must equal the serial
    CPU result bit for bit (pattern matrices with x = 1)."""
    rng = np.random.RandomState(99)
    Ap, Aj, Ax = random_csr(rng, 6007, 900, 50, NP[off], NP[val], long_row=20000, integer_values=True)
    x = rng.randint(-2, 3, size=900).astype(NP[val])
    assert_parity(oracle, Ap, Aj, Ax, x, gpu_spmv(sp, kind, 900, Ap, Aj, Ax, x), exact=True)


# ---- edge cases ---------------------------------------------------------------------------------

@pytest.mark.parametrize("kind", KINDS)
def test_empty_and_degenerate(sp, oracle, kind):
    f32 = np.float32
    # no rows at all: a no-op that must not fail
    y = gpu_spmv(sp, kind, 5, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, f32), np.ones(5, f32))
    assert y.shape == (0,)
    # rows but no nonzeros: y = 0 everywhere (cpu_navie.hpp:10-15)
    y = gpu_spmv(sp, kind, 5, np.zeros(1001, np.int32), np.zeros(0, np.int32), np.zeros(0, f32), np.ones(5, f32))
    assert np.array_equal(y, np.zeros(1000, f32))
    # every nonzero in the last row
    Ap = np.zeros(301, np.int32); Ap[-1] = 7777
    rng = np.random.RandomState(4)
    Aj = rng.randint(0, 50, 7777).astype(np.int32)
    Ax = rng.randint(-3, 4, 7777).astype(f32)
    x = rng.randint(-2, 3, 50).astype(f32)
    assert_parity(oracle, Ap, Aj, Ax, x, gpu_spmv(sp, kind, 50, Ap, Aj, Ax, x), exact=True)
    # every nonzero in the first row, then empty rows only
    Ap = np.full(301, 7777, np.int32); Ap[0] = 0
    assert_parity(oracle, Ap, Aj, Ax, x, gpu_spmv(sp, kind, 50, Ap, Aj, Ax, x), exact=True)
    # one row, one column
    y = gpu_spmv(sp, kind, 1, np.array([0, 3], np.int32), np.zeros(3, np.int32), np.array([1, 2, 4], f32),
                 np.array([3], f32))
    assert y.tolist() == [21.0]


@pytest.mark.parametrize("kind", KINDS)
def test_nan_and_inf_propagate_only_to_their_rows(sp, kind):
    """A NaN/Inf in x must reach exactly the rows that reference it (masked lanes of the
    16-byte loads must not leak a neighbour's value)."""
    rng = np.random.RandomState(8)
    Ap, Aj, Ax = random_csr(rng, 2000, 64, 11)
    Ax[:] = 1.0
    x = np.ones(64, np.float32)
    x[13] = np.nan
    y = gpu_spmv(sp, kind, 64, Ap, Aj, Ax, x)
    touched = np.zeros(2000, bool)
    for r in range(2000):
        touched[r] = np.any(Aj[Ap[r]:Ap[r + 1]] == 13)
    assert np.array_equal(np.isnan(y), touched)


@pytest.mark.parametrize("kind", KINDS)
def test_unaligned_views_take_the_4_byte_path(sp, oracle, kind):
    """Aj/Ax that are not 16-byte aligned (offset views of a larger buffer) must still work."""
    rng = np.random.RandomState(21)
    Ap, Aj, Ax = random_csr(rng, 3000, 500, 30)
    x = (rng.rand(500) * 2 - 1).astype(np.float32)
    nnz = int(Ap[-1])
    big_j = torch.zeros(nnz + 3, dtype=torch.int32, device=DEV)
    big_x = torch.zeros(nnz + 3, dtype=torch.float32, device=DEV)
    big_j[1:nnz + 1] = torch.from_numpy(Aj).to(DEV)
    big_x[3:nnz + 3] = torch.from_numpy(Ax).to(DEV)
    dAj, dAx = big_j[1:nnz + 1], big_x[3:nnz + 3]
    assert dAj.data_ptr() % 16 != 0 and dAx.data_ptr() % 16 != 0
    y = torch.full((3000,), float("nan"), device=DEV)
    sp.spmv(kind, 3000, 500, nnz, torch.from_numpy(Ap).to(DEV), dAj, dAx, torch.from_numpy(x).to(DEV), y)
    assert_parity(oracle, Ap, Aj, Ax, x, y.cpu().numpy())


# ---- merge-path integers are bit-exact against the restatement of thread_search.cuh ---------

@pytest.mark.parametrize("off", ["i32", "i64"])
def test_merge_tile_coordinates_bit_exact(sp, oracle, off):
    rng = np.random.RandomState(17)
    Ap, Aj, Ax = random_csr(rng, 50021, 1000, 12, NP[off], np.float32, long_row=100000)
    x = np.ones(1000, np.float32)
    d = lambda a: torch.from_numpy(a).to(DEV)
    dAp, dAj, dAx, dx = d(Ap), d(Aj), d(Ax), d(x)
    y = torch.empty(50021, device=DEV)
    p = sp.Plan("merge", 50021, 1000, int(Ap[-1]), dAp, dAj, torch.float32)
    info = p.info()
    p.execute(dAx, dx, y)
    rows, nz = p.merge_coords()
    want_rows, want_nz = oracle.merge_tile_coords(Ap, info["tile_items"])
    assert info["n_tiles"] + 1 == len(want_rows)
    assert np.array_equal(rows, want_rows) and np.array_equal(nz, want_nz)
    p.destroy()


# ---- determinism, plans, row-local equality -----------------------------------------------------

@pytest.mark.parametrize("kind", KINDS)
def test_run_to_run_bitwise_reproducible(sp, kind):
    """Also for merge-path: the fix-up is ordered, not float atomics."""
    rng = np.random.RandomState(33)
    Ap, Aj, Ax = random_csr(rng, 30000, 4000, 25, long_row=60000)
    x = (rng.rand(4000) * 2 - 1).astype(np.float32)
    a = gpu_spmv(sp, kind, 4000, Ap, Aj, Ax, x)
    for _ in range(3):
        assert np.array_equal(a, gpu_spmv(sp, kind, 4000, Ap, Aj, Ax, x))


@pytest.mark.parametrize("tail", [1, 2, 3])
def test_array_tail_is_race_free(sp, oracle, tail):
    """nnz % 4 != 0: the last 1-3 nonzeros of the arrays cannot be read with a 16-byte load and are redone
    by a scalar path.  In the merge kernel that path rewrites LDS slots another wave has just written
    (regression: without a barrier in between, the last row came out wrong in about one process out of 25 on
    the C4 stand-in).  Integer-valued data: every repetition must equal the serial CPU result bit for bit."""
    rng = np.random.RandomState(50 + tail)
    n_rows, n_cols = 4000, 900
    lens = rng.randint(20, 40, size=n_rows)
    lens[-1] = 37
    while int(lens.sum()) % 4 != tail:
        lens[rng.randint(0, n_rows - 1)] += 1
    Ap = np.zeros(n_rows + 1, dtype=np.int32)
    np.cumsum(lens, out=Ap[1:])
    nnz = int(Ap[-1])
    assert nnz % 4 == tail
    Aj = rng.randint(0, n_cols, size=nnz).astype(np.int32)
    Ax = rng.randint(-3, 4, size=nnz).astype(np.float32)
    x = rng.randint(-2, 3, size=n_cols).astype(np.float32)
    want = torch.from_numpy(oracle.spmv_serial(Ap, Aj, Ax, x)).to(DEV)
    d = lambda a: torch.from_numpy(a).to(DEV)
    dAp, dAj, dAx, dx = d(Ap), d(Aj), d(Ax), d(x)
    for kind in KINDS:
        p = sp.Plan(kind, n_rows, n_cols, nnz, dAp, dAj, torch.float32)
        y = torch.empty(n_rows, device=DEV)
        for rep in range(150):
            y.fill_(float("nan"))
            p.execute(dAx, dx, y)
            assert torch.equal(y, want), (kind, rep, torch.nonzero(y != want).flatten()[:4].tolist())
        p.destroy()


@pytest.mark.skipif(bool(os.environ.get("MI355_SPMV_PLAIN")), reason="the forced 4-byte kernel sums in another order")
def test_vector_and_light_agree_bitwise(sp):
    """Both use the same per-row arithmetic; only the row -> wave assignment differs
    (results are assignment-independent)."""
    rng = np.random.RandomState(34)
    Ap, Aj, Ax = random_csr(rng, 25000, 3000, 70, long_row=5000)
    x = (rng.rand(3000) * 2 - 1).astype(np.float32)
    assert np.array_equal(gpu_spmv(sp, "vector", 3000, Ap, Aj, Ax, x), gpu_spmv(sp, "light", 3000, Ap, Aj, Ax, x))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("flags", [0, 1])
def test_plan_reuse_across_executes(sp, oracle, kind, flags):
    rng = np.random.RandomState(35)
    Ap, Aj, Ax = random_csr(rng, 9000, 800, 33, long_row=12000)
    d = lambda a: torch.from_numpy(a).to(DEV)
    dAp, dAj, dAx = d(Ap), d(Aj), d(Ax)
    p = sp.Plan(kind, 9000, 800, int(Ap[-1]), dAp, dAj, torch.float32, flags)
    for i in range(4):
        x = (rng.rand(800) * 2 - 1).astype(np.float32)
        y = torch.full((9000,), float("nan"), device=DEV)
        p.execute(dAx, d(x), y)
        torch.cuda.synchronize()
        assert_parity(oracle, Ap, Aj, Ax, x, y.cpu().numpy())
    p.destroy()


@pytest.mark.parametrize("kind", KINDS)
def test_execute_on_a_side_stream(sp, oracle, kind):
    rng = np.random.RandomState(36)
    Ap, Aj, Ax = random_csr(rng, 5000, 600, 20)
    x = (rng.rand(600) * 2 - 1).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to(DEV)
    dAp, dAj, dAx, dx = d(Ap), d(Aj), d(Ax), d(x)
    y = torch.full((5000,), float("nan"), device=DEV)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    p = sp.Plan(kind, 5000, 600, int(Ap[-1]), dAp, dAj, torch.float32)
    with torch.cuda.stream(s):
        p.execute(dAx, dx, y)
    s.synchronize()
    assert_parity(oracle, Ap, Aj, Ax, x, y.cpu().numpy())
    p.destroy()


@pytest.mark.parametrize("kind", KINDS)
def test_execute_is_capturable_in_a_hip_graph(sp, oracle, kind):
    """plan_execute only enqueues (kernels + one memset for `light`): it can be captured once and
    replayed, the way the 2000-iteration loop of main.cu:102-113 would be run launch-free.  The
    replay reads the operands where they were at capture time, so new x values are copied in place."""
    rng = np.random.RandomState(37)
    Ap, Aj, Ax = random_csr(rng, 6000, 800, 24, long_row=9000)
    d = lambda a: torch.from_numpy(a).to(DEV)
    dAp, dAj, dAx = d(Ap), d(Aj), d(Ax)
    dx = torch.zeros(800, device=DEV)
    y = torch.full((6000,), float("nan"), device=DEV)
    p = sp.Plan(kind, 6000, 800, int(Ap[-1]), dAp, dAj, torch.float32)
    p.execute(dAx, dx, y)                      # warm-up outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        p.execute(dAx, dx, y)
    for seed in (1, 2, 3):
        x = (np.random.RandomState(seed).rand(800) * 2 - 1).astype(np.float32)
        dx.copy_(torch.from_numpy(x))
        y.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        got = y.cpu().numpy()
        assert_parity(oracle, Ap, Aj, Ax, x, got)
        y2 = torch.full_like(y, float("nan"))
        p.execute(dAx, dx, y2)                 # the replay and a direct launch give the same bits
        torch.cuda.synchronize()
        assert np.array_equal(got, y2.cpu().numpy())
    del g
    p.destroy()


def test_chunks_are_cut_by_weight_only_when_rows_are_uneven(sp, oracle):
    """decide_balance (analyze.hip): equal-row chunks for uniform matrices (no table), weight-cut
    chunks when the heaviest equal-row chunk is more than twice the mean.  Either way every row is
    computed exactly once (integer-valued data: bit-exact), hub rows (summed by a whole workgroup) included."""
    rng = np.random.RandomState(41)
    n, n_cols = 60000, 5000
    lens = rng.randint(0, 6, size=n).astype(np.int64)
    lens[:300] = rng.randint(200, 1200, size=300)        # the heavy head of a power-law matrix
    lens[7] = 9000                                       # hub rows: whole-workgroup pass
    lens[40000] = 20000
    Ap = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=Ap[1:])
    nnz = int(Ap[-1])
    Aj = rng.randint(0, n_cols, size=nnz).astype(np.int32)
    Ax = rng.randint(-3, 4, size=nnz).astype(np.float32)
    x = rng.randint(-2, 3, size=n_cols).astype(np.float32)
    want = oracle.spmv_serial(Ap.astype(np.int32), Aj, Ax, x)
    d = lambda a: torch.from_numpy(a).to(DEV)
    for off in (np.int32, np.int64):
        dAp, dAj, dAx, dx = d(Ap.astype(off)), d(Aj), d(Ax), d(x)
        for kind in ("vector", "light"):
            p = sp.Plan(kind, n, n_cols, nnz, dAp, dAj, torch.float32)
            info = p.info()
            if os.environ.get("MI355_SPMV_BALANCE") != "0":
                assert info["balanced_chunks"] == 1 and info["n_chunks"] >= 1
            y = torch.full((n,), float("nan"), device=DEV)
            p.execute(dAx, dx, y)
            torch.cuda.synchronize()
            assert np.array_equal(y.cpu().numpy(), want), (kind, off)
            p.destroy()
    # a uniform matrix keeps equal-row chunks
    Ap2, Aj2, Ax2 = random_csr(rng, 60000, 5000, 12)
    p = sp.Plan("vector", 60000, 5000, int(Ap2[-1]), d(Ap2), d(Aj2), torch.float32)
    if "MI355_SPMV_BALANCE" not in os.environ:      # (scripts/gpu_env_matrix.sh forces either plan)
        assert p.info()["balanced_chunks"] == 0
    p.destroy()


@pytest.mark.parametrize("off", [np.int32, np.int64])
def test_giant_rows_are_split_across_workgroups(sp, oracle, off):
    """Rows of more than 65 536 nonzeros in a plan with weight-cut chunks: the chunk kernel leaves them to the
    slice kernels (giant_rows.hpp), which add the slices' partial sums in slice order.  Integer-valued data:
    bit-exact, every repetition; alpha/beta included."""
    rng = np.random.RandomState(61)
    n, n_cols = 30000, 20000
    lens = rng.randint(0, 9, size=n).astype(np.int64)
    lens[5] = 300001                                     # 10 slices, the last one short
    lens[29999] = 70000                                  # the last row of the matrix
    lens[12345] = 65536                                  # exactly at the threshold: NOT giant
    Ap = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=Ap[1:])
    nnz = int(Ap[-1])
    Aj = rng.randint(0, n_cols, size=nnz).astype(np.int32)
    Ax = rng.randint(-2, 3, size=nnz).astype(np.float32)
    x = rng.randint(-2, 3, size=n_cols).astype(np.float32)
    want = oracle.spmv_serial(Ap.astype(np.int32), Aj, Ax, x)
    d = lambda a: torch.from_numpy(a).to(DEV)
    dAp, dAj, dAx, dx = d(Ap.astype(off)), d(Aj), d(Ax), d(x)
    for kind in ("vector", "light"):
        p = sp.Plan(kind, n, n_cols, nnz, dAp, dAj, torch.float32)
        info = p.info()
        if not any(k.startswith("MI355_SPMV_") for k in os.environ if k not in ("MI355_SPMV_LIB", "MI355_SPMV_SMALL")):
            assert info["balanced_chunks"] == 1 and info["n_kernels"] == 3, info
        y = torch.empty(n, device=DEV)
        for rep in range(20):
            y.fill_(float("nan"))
            p.execute(dAx, dx, y)
            torch.cuda.synchronize()
            assert np.array_equal(y.cpu().numpy(), want), (kind, rep)
        y0 = rng.randint(-3, 4, size=n).astype(np.float32)
        y.copy_(torch.from_numpy(y0))
        p.set_alpha_beta(2.0, -1.0)
        p.execute(dAx, dx, y)
        torch.cuda.synchronize()
        assert np.array_equal(y.cpu().numpy(), 2.0 * want - y0), kind
        p.destroy()


@pytest.mark.parametrize("kind", ["vector", "light"])
def test_weight_cut_chunks_with_a_band_window_in_fp64_i64(sp, oracle, kind):
    """A banded fp64 matrix with 64-bit offsets and a few dense rows: the plan cuts chunks by weight AND stages
    a band-placed window of x; the workgroup's LDS layout (2 052 rows of 8-byte bounds and results + the
    window) passes the default 64 KB, so the launch has to raise the kernel's limit first (a launch that
    does not returns an error: this case used to be untested)."""
    rng = np.random.RandomState(77)
    n = 1_200_000
    lens = np.full(n, 20, dtype=np.int64)
    hubs = rng.choice(n, size=40, replace=False)
    lens[hubs] = 30_000
    Ap = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=Ap[1:])
    nnz = int(Ap[-1])
    rows = np.repeat(np.arange(n, dtype=np.int64), lens)
    Aj = np.clip(rows + rng.randint(-1500, 1501, size=nnz), 0, n - 1).astype(np.int32)
    del rows
    Ax = (rng.rand(nnz) * 2 - 1)
    x = (rng.rand(n) * 2 - 1)
    d = lambda a: torch.from_numpy(a).to(DEV)
    dAp, dAj, dAx, dx = d(Ap), d(Aj), d(Ax), d(x)
    p = sp.Plan(kind, n, n, nnz, dAp, dAj, torch.float64)
    info = p.info()
    if not any(k.startswith("MI355_SPMV_") for k in os.environ if k not in ("MI355_SPMV_LIB", "MI355_SPMV_SMALL")):   # (no forced code path)
        assert info["balanced_chunks"] == 1 and info["window_elems"] > 0, info
    y = torch.full((n,), float("nan"), dtype=torch.float64, device=DEV)
    p.execute(dAx, dx, y)
    torch.cuda.synchronize()
    p.destroy()
    assert_parity(oracle, Ap, Aj, Ax, x, y.cpu().numpy())


@pytest.mark.parametrize("kind", ["vector", "light"])
@pytest.mark.parametrize("val", ["f32", "f64"])
def test_band_too_wide_for_two_workgroups_takes_one_of_1024_threads(sp, oracle, val, kind):
    """A band of 32 769 columns in fp32 (16 385 in fp64) is more than two workgroups per CU can hold in LDS; the row-based
    kinds then run ONE 1 024-thread workgroup per CU with ~150 KB (round 1: plain gathers at 1.7 TB/s).  Every row
    against the oracle bound; a matrix too small to fill the chip that way keeps the 256-thread plan."""
    hw = 16384 if val == "f32" else 8192
    n = 1_500_000
    dt = torch.float32 if val == "f32" else torch.float64
    m = sp.synth.banded_fixed(n, 32, hw, seed=4, device=DEV, val_dtype=dt)
    x = sp.synth.dense_vector(m.n_cols, dt, 10, DEV)
    p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, dt)
    info = p.info()
    if not any(k.startswith("MI355_SPMV_") for k in os.environ if k not in ("MI355_SPMV_LIB", "MI355_SPMV_SMALL")):   # (no forced code path)
        assert info["block_threads"] == 1024 and info["window_elems"] * m.Ax.element_size() > 100 * 1024, info
    y = torch.full((n,), float("nan"), dtype=dt, device=DEV)
    p.execute(m.Ax, x, y)
    torch.cuda.synchronize()
    p.destroy()
    Ap, Aj, Ax = m.numpy()
    assert_parity(oracle, Ap, Aj, Ax, x.cpu().numpy(), y.cpu().numpy())
    small = sp.synth.banded_fixed(100_000, 32, hw, seed=4, device=DEV, val_dtype=dt)
    p = sp.Plan(kind, small.n_rows, small.n_cols, small.nnz, small.Ap, small.Aj, dt)
    assert p.info()["block_threads"] != 1024 or any(k.startswith("MI355_SPMV_") for k in os.environ if k not in ("MI355_SPMV_LIB", "MI355_SPMV_SMALL"))
    p.destroy()


@pytest.mark.parametrize("kind", ["vector", "light"])
def test_wide_band_fp64_takes_more_than_64_kb_of_lds(sp, oracle, kind):
    """The S32-band shape in fp64 (band of 8 193 columns = 64 KB of doubles): the window only fits when the
    workgroup takes more than the default 64 KB of LDS (two 512-thread workgroups of ~78 KB per CU).  Every
    row against the fp64 oracle bound."""
    n = 700000
    m = sp.synth.banded_fixed(n, 32, 4096, seed=3, device=DEV, val_dtype=torch.float64)
    x = sp.synth.dense_vector(m.n_cols, torch.float64, 9, DEV)
    p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, torch.float64)
    info = p.info()
    if not any(k.startswith("MI355_SPMV_") for k in os.environ if k not in ("MI355_SPMV_LIB", "MI355_SPMV_SMALL")):   # (no forced code path)
        assert info["block_threads"] == 512 and info["window_elems"] * 8 > 64 * 1024, info
    y = torch.full((n,), float("nan"), dtype=torch.float64, device=DEV)
    p.execute(m.Ax, x, y)
    torch.cuda.synchronize()
    p.destroy()
    Ap, Aj, Ax = m.numpy()
    assert_parity(oracle, Ap, Aj, Ax, x.cpu().numpy(), y.cpu().numpy())


@pytest.mark.parametrize("kind", ["vector", "light"])
@pytest.mark.parametrize("off,val", [("i32", "f32"), ("i64", "f64")])
def test_band_wider_than_any_window_is_swept(sp, oracle, off, val, kind):
    """A band of ~80 K columns (fp64: ~40 K) is more than ONE CU's LDS holds: the vector kind then makes a chunk one group
    of rows held in registers and lets the window sweep the band (csr_vector_sweep_kernel / light_rows_sweep_kernel; plain gathers ran at
    1.6-2.4 TB/s).  The plan's band and longest row come from a SAMPLE of 256 rows, so the matrix also holds what the
    sample cannot see: rows of 200 nonzeros (more than one step of their vector: the plain-gather tail), columns far
    outside the band (gathered from global memory), an empty row, a short last chunk.  Every row against the oracle
    bound; row blocks of the same plan must reproduce the whole result bit for bit."""
    rng = np.random.default_rng(31)
    n, per_row = 300_003, 32
    hw = 40_000 if val == "f32" else 20_000
    lens = np.full(n, per_row, dtype=np.int64)
    probed = set(((n - 1) * np.arange(256)) // 255)
    special = [r for r in (5, 1001, 150_001, n - 2) if r not in probed]
    lens[special] = 200
    lens[7] = 0
    Ap = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=Ap[1:])
    nnz = int(Ap[-1])
    rows = np.repeat(np.arange(n, dtype=np.int64), lens)
    lo = np.clip(rows - hw, 0, n - 1)
    hi = np.clip(rows + hw, 0, n - 1)
    cols = lo + (rng.random(nnz) * (hi - lo + 1)).astype(np.int64)
    far = [r for r in (9, 2000, 250_000) if r not in probed]
    for r in far:                                       # first / last element of the row far outside the band
        cols[Ap[r]] = 0 if r > n // 2 else n - 1
        cols[Ap[r + 1] - 1] = n - 1 if r > n // 2 else 0
    order = np.lexsort((cols, rows))                    # sorted columns inside each row (duplicates allowed: they add up)
    Aj = cols[order].astype(np.int32)
    Ax = (rng.random(nnz) - 0.5).astype(NP[val])
    x = seeded_x(n, NP[val])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    dAp, dAj, dAx, dx = d(Ap.astype(NP[off])), d(Aj), d(Ax), d(x)
    sweep_kernel = "csr_vector_sweep_kernel" if kind == "vector" else "light_rows_sweep_kernel"
    p = sp.Plan(kind, n, n, nnz, dAp, dAj, dAx.dtype)
    info = p.info()
    forced = any(k.startswith("MI355_") for k in os.environ if k not in ("MI355_SPMV_LIB", "MI355_SPMV_SMALL"))
    if not forced:
        assert info["main_kernel"] == sweep_kernel and info["block_threads"] == 1024, info
    y = torch.full((n,), float("nan"), dtype=dAx.dtype, device=DEV)
    p.execute(dAx, dx, y)
    torch.cuda.synchronize()
    assert_parity(oracle, Ap.astype(NP[off]), Aj, Ax, x, y.cpu().numpy())
    # y = alpha * A x + beta * y_old through the same kernel: against the plain result, two roundings per row
    y_old = torch.from_numpy(seeded_x(n, NP[val])).to(DEV) * 0.5
    y_ab = y_old.clone()
    p.set_alpha_beta(-0.75, 3.0)
    p.execute(dAx, dx, y_ab)
    torch.cuda.synchronize()
    p.set_alpha_beta(1.0, 0.0)
    want_ab = -0.75 * y.double() + 3.0 * y_old.double()
    eps = 2.0 ** -23 if val == "f32" else 2.0 ** -52
    assert bool(((y_ab.double() - want_ab).abs() <= 2 * eps * (0.75 * y.double().abs() + 3.0 * y_old.double().abs()) + 1e-300).all())
    # the same plan cut into three row blocks: bit-identical rows
    shape = p.shape()
    row_cuts, chunk_cuts, nnz_cuts = p.partition(3)
    y_blocks = torch.full((n,), float("nan"), dtype=dAx.dtype, device=DEV)
    for b in range(3):
        r0, r1 = row_cuts[b], row_cuts[b + 1]
        if r1 == r0:
            continue
        a, j, v, _ = sp.dist.block_view(dAp, dAj, dAx, r0, r1)
        pb = sp.Plan.block(kind, shape, r0, chunk_cuts[b], chunk_cuts[b + 1] - chunk_cuts[b], nnz_cuts[b], r1 - r0, n,
                           int(a[-1].item()), a, j, dAx.dtype)
        if not forced:
            assert pb.info()["main_kernel"] == sweep_kernel
        pb.execute(v, dx, y_blocks[r0:r1])
        torch.cuda.synchronize()
        pb.destroy()
    p.destroy()
    if os.environ.get("MI355_SPMV_PLAIN"):      # (the forced 4-byte kernel sums the WHOLE plan in another order; blocks keep the plan's)
        assert_parity(oracle, Ap.astype(NP[off]), Aj, Ax, x, y_blocks.cpu().numpy())
    else:
        assert torch.equal(y.view(torch.int32 if val == "f32" else torch.int64), y_blocks.view(torch.int32 if val == "f32" else torch.int64))


@pytest.mark.parametrize("kind", ["vector", "light"])
def test_swept_window_that_ends_inside_a_16_byte_group(sp, oracle, kind):
    """Regression (round 3): the sweep's windows are staged by LDS-DMA in whole 16-byte groups; a window whose last column
    is not the last of a group (c_hi + 1 - w0 not a multiple of 4) must still hold that group.  The 1 024-thread-plan
    matrix, forced onto the sweep kernel by forbidding band-placed windows, has such windows in most chunks (found by
    scripts/gpu_env_matrix.sh, not by the default suite: 26 wrong rows of 1.3 M)."""
    old = os.environ.get("MI355_SPMV_WINDOW_FROM_BAND")
    os.environ["MI355_SPMV_WINDOW_FROM_BAND"] = "0"
    sp.capi.lib().mi355_spmv_knobs_reload()
    try:
        for val in ("f32", "f64"):
            test_band_too_wide_for_two_workgroups_takes_one_of_1024_threads(sp, oracle, val, kind)
    finally:
        if old is None:
            os.environ.pop("MI355_SPMV_WINDOW_FROM_BAND", None)
        else:
            os.environ["MI355_SPMV_WINDOW_FROM_BAND"] = old
        sp.capi.lib().mi355_spmv_knobs_reload()


@pytest.mark.parametrize("off,val", [("i32", "f32"), ("i64", "f64")])
def test_merge_runs_sweep_a_band_wider_than_any_window(sp, oracle, off, val):
    """The merge kind on the same shape: its row-parallel runs take the sweeping body too (merge_rows_kernel with 1 024
    threads, a piece = one group of rows held in registers; plain gathers ran such runs at 1.6 TB/s).  Regular rows only
    (the probe must see rows that all but fill one step), plus what the probe's sample cannot see: a few rows of 200
    nonzeros, columns far outside the band, an empty row; 31 per row so that rows straddle the 16-byte groups and the run
    boundaries fall inside rows (carries through the fix-up kernel)."""
    rng = np.random.default_rng(37)
    n, per_row = 600_003, 31
    hw = 40_000 if val == "f32" else 20_000
    lens = np.full(n, per_row, dtype=np.int64)
    probed = set(((n - 1) * np.arange(256)) // 255)
    special = [r for r in (5, 1001, 150_001, n - 2) if r not in probed]
    lens[special] = 200
    lens[[r for r in (7, 333_333) if r not in probed]] = 0
    Ap = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=Ap[1:])
    nnz = int(Ap[-1])
    rows = np.repeat(np.arange(n, dtype=np.int64), lens)
    lo = np.clip(rows - hw, 0, n - 1)
    hi = np.clip(rows + hw, 0, n - 1)
    cols = lo + (rng.random(nnz) * (hi - lo + 1)).astype(np.int64)
    for r in [r for r in (9, 2000, 250_000) if r not in probed]:
        cols[Ap[r]] = 0 if r > n // 2 else n - 1
        cols[Ap[r + 1] - 1] = n - 1 if r > n // 2 else 0
    order = np.lexsort((cols, rows))
    Aj = cols[order].astype(np.int32)
    Ax = (rng.random(nnz) - 0.5).astype(NP[val])
    x = seeded_x(n, NP[val])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    dAp, dAj, dAx, dx = d(Ap.astype(NP[off])), d(Aj), d(Ax), d(x)
    p = sp.Plan("merge", n, n, nnz, dAp, dAj, dAx.dtype)
    info = p.info()
    forced = any(k.startswith("MI355_") for k in os.environ if k not in ("MI355_SPMV_LIB", "MI355_SPMV_SMALL"))
    if not forced:
        assert info["main_kernel"] == "merge_rows_kernel" and info["block_threads"] == 1024 and info["window_elems"] > 30_000 // (2 if val == "f64" else 1), info
    for _ in range(2):                                   # twice: the run boundaries / carries are rewritten by every execute
        y = torch.full((n,), float("nan"), dtype=dAx.dtype, device=DEV)
        p.execute(dAx, dx, y)
        torch.cuda.synchronize()
        assert_parity(oracle, Ap.astype(NP[off]), Aj, Ax, x, y.cpu().numpy())
    y_old = torch.from_numpy(seeded_x(n, NP[val])).to(DEV) * 0.5
    y_ab = y_old.clone()
    p.set_alpha_beta(-0.75, 3.0)
    p.execute(dAx, dx, y_ab)
    torch.cuda.synchronize()
    p.destroy()
    # bound of test_alpha_beta: (len + 3) eps (|alpha| sum|a x| + |beta y0|) around alpha * y64 + beta * y0 (a row that
    # straddles runs gets its carry added by the fix-up kernel, so "two roundings of the plain result" does not hold here)
    y64, yabs = oracle.spmv_ref64(Ap.astype(NP[off]), Aj, Ax, x)
    y0 = y_old.cpu().numpy().astype(np.float64)
    eps = 2.0 ** -24 if val == "f32" else 2.0 ** -53
    bound = (lens + 3) * eps * (0.75 * yabs + 3.0 * np.abs(y0)) + 1e-300
    assert np.all(np.abs(y_ab.cpu().numpy().astype(np.float64) - (-0.75 * y64 + 3.0 * y0)) <= bound)


@pytest.mark.parametrize("kind", KINDS)
def test_one_shot_calls_keep_their_plan_and_survive_a_rewritten_matrix(sp, oracle, kind):
    """The one-shot entry points find the plan of their previous call again by the pointers and sizes of Ap / Aj (the
    reference's harness calls a kind 2 000 times in a row, main.cu:102-113).  A kept plan holds launch-shape decisions
    only, so results must stay right when the caller REWRITES the arrays in place with another structure of the same
    sizes — here a banded matrix turned into a scattered, ragged one with a hub row, same n_rows / nnz, same device
    buffers — and again after mi355_spmv_cache_release()."""
    rng = np.random.RandomState(123)
    n, k = 40_000, 24
    nnz = n * k
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    # structure 1: exactly k per row, narrow band
    Ap1 = (np.arange(n + 1, dtype=np.int64) * k).astype(np.int32)
    rows = np.repeat(np.arange(n), k)
    Aj1 = np.clip(rows + rng.randint(-300, 301, size=nnz), 0, n - 1).astype(np.int32)
    # structure 2: same sizes — ragged rows (many empty), scattered columns, one row of 60 000 nonzeros
    lens = rng.multinomial(nnz - 60_000, np.ones(n - 1) / (n - 1))
    lens = np.insert(lens, 777, 60_000)
    Ap2 = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=Ap2[1:])
    assert int(Ap2[-1]) == nnz
    Ap2 = Ap2.astype(np.int32)
    Aj2 = rng.randint(0, n, size=nnz).astype(np.int32)
    Ax = (rng.rand(nnz) * 2 - 1).astype(np.float32)
    x = (rng.rand(n) * 2 - 1).astype(np.float32)
    dAp, dAj, dAx, dx = d(Ap1), d(Aj1), d(Ax), d(x)
    sp.capi.cache_release()
    y = torch.full((n,), float("nan"), device=DEV)
    for _ in range(3):                                  # first call makes the plan, the next two find it again
        y.fill_(float("nan"))
        sp.spmv(kind, n, n, nnz, dAp, dAj, dAx, dx, y)
        assert_parity(oracle, Ap1, Aj1, Ax, x, y.cpu().numpy())
    dAp.copy_(d(Ap2))                                   # the same device buffers, another matrix
    dAj.copy_(d(Aj2))
    for _ in range(2):
        y.fill_(float("nan"))
        sp.spmv(kind, n, n, nnz, dAp, dAj, dAx, dx, y)
        assert_parity(oracle, Ap2, Aj2, Ax, x, y.cpu().numpy())
    sp.capi.cache_release()
    y.fill_(float("nan"))
    sp.spmv(kind, n, n, nnz, dAp, dAj, dAx, dx, y)      # a fresh plan for structure 2 (weight-cut chunks, long-row passes)
    assert_parity(oracle, Ap2, Aj2, Ax, x, y.cpu().numpy())
    dAp.copy_(d(Ap1))                                   # ... and back, under the plan made for structure 2
    dAj.copy_(d(Aj1))
    y.fill_(float("nan"))
    sp.spmv(kind, n, n, nnz, dAp, dAj, dAx, dx, y)
    assert_parity(oracle, Ap1, Aj1, Ax, x, y.cpu().numpy())
    sp.capi.cache_release()


@pytest.mark.parametrize("kind", KINDS)
def test_kept_one_shot_plan_meets_a_row_beyond_every_giant_threshold(sp, oracle, kind):
    """The hole the plan cache leaves open by design: a plan kept for a matrix WITHOUT giant rows meets, at the same
    addresses and sizes, a matrix with a row of 1.5 M nonzeros — far beyond the 64 K at which any fresh plan would cut
    it into slices for several workgroups.  The kept plan has no slice list: the row is summed by the one workgroup
    that owns it (long-row / whole-workgroup passes of the chunk body, the merge kind's tiles as ever).  Slow, and
    right: that is the documented trade ("What the one-shot entry points retain")."""
    rng = np.random.RandomState(321)
    n, k = 100_000, 32
    nnz = n * k
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    Ap1 = (np.arange(n + 1, dtype=np.int64) * k).astype(np.int32)
    Aj1 = np.clip(np.repeat(np.arange(n), k) + rng.randint(-500, 501, size=nnz), 0, n - 1).astype(np.int32)
    hub = 1_500_000
    lens = rng.multinomial(nnz - hub, np.ones(n - 1) / (n - 1))
    lens = np.insert(lens, 31_337, hub)
    Ap2 = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=Ap2[1:])
    Ap2 = Ap2.astype(np.int32)
    Aj2 = rng.randint(0, n, size=nnz).astype(np.int32)
    Ax = (rng.rand(nnz) * 2 - 1).astype(np.float32)
    x = (rng.rand(n) * 2 - 1).astype(np.float32)
    dAp, dAj, dAx, dx = d(Ap1), d(Aj1), d(Ax), d(x)
    sp.capi.cache_release()
    y = torch.full((n,), float("nan"), device=DEV)
    sp.spmv(kind, n, n, nnz, dAp, dAj, dAx, dx, y)       # the plan of the banded matrix is kept
    assert_parity(oracle, Ap1, Aj1, Ax, x, y.cpu().numpy())
    dAp.copy_(d(Ap2))
    dAj.copy_(d(Aj2))
    y.fill_(float("nan"))
    sp.spmv(kind, n, n, nnz, dAp, dAj, dAx, dx, y)       # ... and meets the hub row
    assert_parity(oracle, Ap2, Aj2, Ax, x, y.cpu().numpy())
    # a fresh plan for the same arrays does cut the row into slices (more kernels per execute) and agrees
    sp.capi.cache_release()
    p = sp.Plan(kind, n, n, nnz, dAp, dAj, torch.float32)
    if kind != "merge" and not any(k.startswith("MI355_") for k in os.environ if k not in ("MI355_SPMV_LIB", "MI355_SPMV_SMALL")):   # (no forced code path)
        assert p.info()["n_kernels"] == 3
    y2 = torch.full((n,), float("nan"), device=DEV)
    p.execute(dAx, dx, y2)
    torch.cuda.synchronize()
    p.destroy()
    assert_parity(oracle, Ap2, Aj2, Ax, x, y2.cpu().numpy())


def test_one_shot_calls_from_several_threads_share_nothing(sp, oracle):
    """Four host threads call the one-shot entry point on the SAME matrix at once, each on its own stream with its own y:
    a kept plan is taken OUT of the cache by its user (its scratch serves one execute at a time), so a second thread
    makes its own; every result must be right and every thread's results identical from call to call."""
    import threading
    rng = np.random.RandomState(5)
    Ap, Aj, Ax = random_csr(rng, 50_000, 9000, 60, long_row=30_000)
    x = (rng.rand(9000) * 2 - 1).astype(np.float32)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    dAp, dAj, dAx, dx = d(Ap), d(Aj), d(Ax), d(x)
    nnz = int(Ap[-1])
    sp.capi.cache_release()
    results, errors = {}, []

    def work(tid):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            kind = ("vector", "merge", "light", "merge")[tid]
            outs = []
            for _ in range(12):
                y = torch.full((50_000,), float("nan"), device=DEV)
                torch.cuda.current_stream().synchronize()     # (the poison is written on another stream than the SpMV's)
                sp.spmv(kind, 50_000, 9000, nnz, dAp, dAj, dAx, dx, y, stream=stream)   # (synchronises its stream)
                outs.append(y.cpu().numpy())
            results[tid] = outs
        except Exception as e:            # noqa: BLE001 (reported below)
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    sp.capi.cache_release()
    assert not errors, errors
    for tid, outs in results.items():
        assert_parity(oracle, Ap, Aj, Ax, x, outs[0])
        for o in outs[1:]:
            assert np.array_equal(o, outs[0])


@pytest.mark.parametrize("off,val,hw,block", [("i32", "f64", 4096, 512), ("i64", "f64", 4096, 512), ("i32", "f32", 16384, 1024),
                                              ("i64", "f64", 8192, 1024)])
def test_merge_runs_take_a_wide_workgroup_when_the_band_needs_it(sp, oracle, off, val, hw, block):
    """The S32-band shape in fp64: 8 193 columns of doubles + the rows of a run are more than a 256-thread workgroup's
    window holds (plain gathers: 718 us; a 56 KB window covering 78 %: 583), so the run kernel takes two workgroups of 512
    threads per CU with ~78 KB each and runs as long as the band leaves room for (306 us); a wider band still gets ONE
    workgroup of 1 024 threads with ~155 KB.  Every row against the bound, bitwise reproducible, alpha / beta."""
    n = 1_300_000
    dt = torch.float64 if val == "f64" else torch.float32
    m = sp.synth.banded_fixed(n, 32, hw, seed=8, device=DEV, val_dtype=dt,
                              off_dtype=torch.int32 if off == "i32" else torch.int64)
    x = sp.synth.dense_vector(m.n_cols, dt, 8, DEV)
    p = sp.Plan("merge", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, dt)
    info = p.info()
    if not any(k.startswith("MI355_") for k in os.environ if k not in ("MI355_SPMV_LIB", "MI355_SPMV_SMALL")):
        assert info["main_kernel"] == "merge_rows_kernel" and info["block_threads"] == block and \
            info["window_elems"] * m.Ax.element_size() > 64 * 1024, info
    y = torch.full((n,), float("nan"), dtype=dt, device=DEV)
    p.execute(m.Ax, x, y)
    torch.cuda.synchronize()
    Ap, Aj, Ax = m.numpy()
    assert_parity(oracle, Ap, Aj, Ax, x.cpu().numpy(), y.cpu().numpy())
    y2 = torch.full((n,), float("nan"), dtype=dt, device=DEV)
    p.execute(m.Ax, x, y2)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    p.set_alpha_beta(-1.5, 0.5)
    y0 = sp.synth.dense_vector(n, dt, 9, DEV)
    y3 = y0.clone()
    p.execute(m.Ax, x, y3)
    torch.cuda.synchronize()
    p.destroy()
    y64, bound = parity_bound(oracle, Ap, Aj, Ax, x.cpu().numpy(), 8)
    want = -1.5 * y64 + 0.5 * y0.cpu().numpy().astype(np.float64)
    eps = 2.0 ** -52 if val == "f64" else 2.0 ** -23
    assert np.all(np.abs(y3.cpu().numpy().astype(np.float64) - want) <= 1.5 * bound + 4 * eps * (np.abs(want) + np.abs(y64) + 1.0) + 1e-300)


# ---- BASELINE-sized inputs ---------------------------------------------------------------------

@pytest.mark.parametrize("kind", KINDS)
def test_full_size_s32_band_against_oracle(sp, oracle, kind):
    """North-star target shape (2^22 rows x 32 nnz/row, fp32): every row against the
    fp64 oracle bound (the oracle takes about a second at this size)."""
    m = sp.synth.workload("s32-band", DEV)
    x = sp.synth.dense_vector(m.n_cols, torch.float32, 1, DEV)
    y = torch.full((m.n_rows,), float("nan"), device=DEV)
    sp.spmv(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, y)
    Ap, Aj, Ax = m.numpy()
    assert_parity(oracle, Ap, Aj, Ax, x.cpu().numpy(), y.cpu().numpy())


@pytest.mark.parametrize("kind", KINDS)
def test_full_size_powerlaw_pattern_is_exact(sp, kind):
    """Config C3 stand-in (916 428 rows, 5 105 039 nnz, values 1.0, power-law rows) with
    x = 1: y must equal the row lengths exactly (bit-exact integer case)."""
    m = sp.synth.workload("c3-webgoogle", DEV)
    x = torch.ones(m.n_cols, device=DEV)
    y = torch.full((m.n_rows,), float("nan"), device=DEV)
    sp.spmv(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, y)
    lens = (m.Ap[1:] - m.Ap[:-1]).to(torch.float32)
    assert lens.max().item() > 1000                                  # it is a load-balance stress
    assert torch.equal(y, lens)


@pytest.mark.parametrize("kind", KINDS)
def test_full_size_fp64_i64_linearity_and_rowsums(sp, oracle, kind):
    """Config C4 stand-in (8.4 M rows, ~224 M nnz, fp64 values, 64-bit Ap): size-independent
    properties — row sums for x = 1 against a torch segment sum, linearity in x — and the
    first 200 000 rows against the oracle."""
    m = sp.synth.workload("c4-nlpkkt", DEV)
    assert m.Ap.dtype == torch.int64 and m.Ax.dtype == torch.float64
    n = m.n_rows
    ones = torch.ones(m.n_cols, dtype=torch.float64, device=DEV)
    y1 = torch.full((n,), float("nan"), dtype=torch.float64, device=DEV)
    sp.spmv(kind, n, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, ones, y1)
    csum = torch.zeros(m.nnz + 1, dtype=torch.float64, device=DEV)
    torch.cumsum(m.Ax, 0, out=csum[1:])
    rowsum = csum[m.Ap[1:]] - csum[m.Ap[:-1]]
    assert torch.allclose(y1, rowsum, rtol=0, atol=1e-9)             # prefix-sum differences: loose but global
    del csum, rowsum
    xa = sp.synth.dense_vector(m.n_cols, torch.float64, 5, DEV)
    xb = sp.synth.dense_vector(m.n_cols, torch.float64, 6, DEV)
    ya, yb, yc = (torch.empty(n, dtype=torch.float64, device=DEV) for _ in range(3))
    sp.spmv(kind, n, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, xa, ya)
    sp.spmv(kind, n, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, xb, yb)
    sp.spmv(kind, n, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, 2.0 * xa - 0.5 * xb, yc)
    lin_err = (yc - (2.0 * ya - 0.5 * yb)).abs()
    bad = torch.nonzero(~(lin_err <= 1e-12 * 28 * 4)).flatten()
    assert bad.numel() == 0, "linearity: %d rows off, first %s, errors %s, yc %s" % (
        bad.numel(), bad[:6].tolist(), lin_err[bad[:6]].tolist(), yc[bad[:6]].tolist())
    rows = 200000
    hi = int(m.Ap[rows].item())
    Ap = m.Ap[:rows + 1].cpu().numpy(); Aj = m.Aj[:hi].cpu().numpy(); Ax = m.Ax[:hi].cpu().numpy()
    y64, bound = parity_bound(oracle, Ap, Aj, Ax, xa.cpu().numpy())
    assert np.all(np.abs(ya[:rows].cpu().numpy() - y64) <= bound)


@pytest.mark.parametrize("kind", KINDS)
def test_more_than_2_to_31_nonzeros_with_64bit_offsets(sp, kind):
    """nnz = 2^31 + 2^21 > INT32_MAX: only expressible with offset_t = int64 (the reference's
    harness cannot; its loader overflows there, quirk 2).  Values are 1.0 and
    x = 1, so y must equal the row lengths exactly; x = column parity checks the gather too.
    ~17 GB of HBM."""
    n_rows, n_cols = 1 << 20, 1 << 16
    per_row = 2048 + 2                      # nnz = 2^20 * 2050 = 2^31 + 2^21
    nnz = n_rows * per_row
    assert nnz > 2 ** 31
    Ap = torch.arange(n_rows + 1, dtype=torch.int64, device=DEV) * per_row
    # columns: (row * 7 + position * 3) mod n_cols — cheap, deterministic, spread out
    Aj = torch.empty(nnz, dtype=torch.int32, device=DEV)
    step = 1 << 14
    pos = torch.arange(per_row, dtype=torch.int64, device=DEV)[None, :] * 3
    for r0 in range(0, n_rows, step):
        rows = torch.arange(r0, r0 + step, dtype=torch.int64, device=DEV)[:, None] * 7
        Aj[r0 * per_row:(r0 + step) * per_row] = ((rows + pos) % n_cols).reshape(-1).to(torch.int32)
    Ax = torch.ones(nnz, dtype=torch.float32, device=DEV)
    y = torch.full((n_rows,), float("nan"), device=DEV)
    sp.spmv(kind, n_rows, n_cols, nnz, Ap, Aj, Ax, torch.ones(n_cols, device=DEV), y)
    assert torch.equal(y, torch.full((n_rows,), float(per_row), device=DEV))
    # x = 1 on even columns: row r sees position p at column (7r + 3p) mod 2^16, even iff (r + p) even
    xe = (torch.arange(n_cols, device=DEV) % 2 == 0).to(torch.float32)
    sp.spmv(kind, n_rows, n_cols, nnz, Ap, Aj, Ax, xe, y)
    assert torch.equal(y, torch.full((n_rows,), float(per_row // 2), device=DEV))


# ---- generalized merge-path SpMV ---------------------------------------------

@pytest.mark.parametrize("semiring", ["plus_times", "min_plus", "max_times", "max_plus", "or_and"])
@pytest.mark.parametrize("off,val", COMBOS)
def test_generalized_merge_semirings(sp, oracle, semiring, off, val):
    """min and max never round and a+x / a*x round once, so (min,+) and (max,*) must equal the
    reference's generalized CPU loop bit for bit whatever the reduction order; (+,*) within the bound.
    Empty rows yield the semiring's identity; one row spans many tiles."""
    rng = np.random.RandomState(77)
    Ap, Aj, Ax = random_csr(rng, 30011, 2000, 25, NP[off], NP[val], long_row=70000)
    x = (rng.rand(2000) * 2 - 1).astype(NP[val])
    if semiring == "or_and":                                  # booleans as 0.0 / 1.0, about one in eight set
        Ax = (rng.rand(Ax.size) < 0.5).astype(NP[val])
        x = (rng.rand(2000) < 0.25).astype(NP[val])
    d = lambda a: torch.from_numpy(a).to(DEV)
    y = torch.full((30011,), float("nan"), dtype=d(Ax).dtype, device=DEV)
    sp.spmv_genl(semiring, 30011, 2000, int(Ap[-1]), d(Ap), d(Aj), d(Ax), d(x), y)
    got = y.cpu().numpy()
    want = oracle.spmv_genl_serial(sp.capi.SEMIRINGS[semiring], Ap, Aj, Ax, x)
    assert not np.any(np.isnan(got))
    if semiring == "plus_times":
        assert_parity(oracle, Ap, Aj, Ax, x, got)
    else:
        assert np.array_equal(got, want)


@pytest.mark.parametrize("semiring", ["plus_times", "min_plus", "max_times", "max_plus", "or_and"])
@pytest.mark.parametrize("off", ["i32", "i64"])
def test_generalized_merge_on_integer_values(sp, oracle, semiring, off):
    """32-bit integer values (MI355_VAL_I32; the reference's generalized kind is a template over the value types):
    every semiring bit-exact against the oracle — ragged rows, empty rows (identity 0 / INT32_MAX / INT32_MIN), a
    long row across many tiles, a matrix big enough for several runs; one-shot symbol and a kept plan."""
    rng = np.random.RandomState(90 + len(semiring))
    Ap, Aj, _ = random_csr(rng, 60011, 3000, 40, NP[off], np.float32, long_row=70000)
    nnz = int(Ap[-1])
    Ax = rng.randint(-9, 10, size=nnz).astype(np.int32)
    x = rng.randint(-7, 8, size=3000).astype(np.int32)
    if semiring == "or_and":
        Ax = (rng.rand(nnz) < 0.5).astype(np.int32)
        x = (rng.rand(3000) < 0.1).astype(np.int32)
    d = lambda a: torch.from_numpy(a).to(DEV)
    dAp, dAj, dAx, dx = d(Ap), d(Aj), d(Ax), d(x)
    y = torch.full((60011,), 12345, dtype=torch.int32, device=DEV)
    sp.spmv_genl(semiring, 60011, 3000, nnz, dAp, dAj, dAx, dx, y)
    want = oracle.spmv_genl_serial(sp.capi.SEMIRINGS[semiring], Ap, Aj, Ax, x)
    assert np.array_equal(y.cpu().numpy(), want)
    assert (np.diff(Ap.astype(np.int64)) == 0).any()
    p = sp.Plan("merge", 60011, 3000, nnz, dAp, dAj, torch.int32)
    p.set_semiring(semiring)
    y2 = torch.full((60011,), -777, dtype=torch.int32, device=DEV)
    p.execute(dAx, dx, y2)
    p.execute(dAx, dx, y2)
    torch.cuda.synchronize()
    assert np.array_equal(y2.cpu().numpy(), want)
    with pytest.raises(RuntimeError, match="not supported"):
        p.set_alpha_beta(2.0, 0.0)
    p.destroy()


def test_integer_values_are_a_merge_feature_and_wrap_around(sp, oracle):
    rng = np.random.RandomState(7)
    Ap, Aj, _ = random_csr(rng, 3000, 500, 20, np.int32, np.float32)
    nnz = int(Ap[-1])
    d = lambda a: torch.from_numpy(a).to(DEV)
    for kind in ("vector", "light"):
        with pytest.raises(RuntimeError, match="not supported"):
            sp.Plan(kind, 3000, 500, nnz, d(Ap), d(Aj), torch.int32)
    Ax = np.full(nnz, 2 ** 30, dtype=np.int32)                 # sums far beyond 2^31: two's-complement wrap-around
    x = np.full(500, 3, dtype=np.int32)
    y = torch.zeros(3000, dtype=torch.int32, device=DEV)
    sp.spmv("merge", 3000, 500, nnz, d(Ap), d(Aj), d(Ax), d(x), y)
    assert np.array_equal(y.cpu().numpy(), oracle.spmv_genl_serial(0, Ap, Aj, Ax, x))


def test_semiring_is_a_merge_feature_and_plans_keep_it(sp, oracle):
    rng = np.random.RandomState(78)
    Ap, Aj, Ax = random_csr(rng, 5000, 300, 12)
    x = (rng.rand(300) * 2 - 1).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to(DEV)
    dAp, dAj, dAx, dx = d(Ap), d(Aj), d(Ax), d(x)
    p = sp.Plan("vector", 5000, 300, int(Ap[-1]), dAp, dAj, torch.float32)
    with pytest.raises(RuntimeError, match="not supported"):
        p.set_semiring("min_plus")                           # only the merge kind is generalized, as in the reference
    p.destroy()
    p = sp.Plan("merge", 5000, 300, int(Ap[-1]), dAp, dAj, torch.float32)
    p.set_semiring("min_plus")
    y = torch.empty(5000, device=DEV)
    for _ in range(2):
        p.execute(dAx, dx, y)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), oracle.spmv_genl_serial(1, Ap, Aj, Ax, x))
    p.set_semiring("plus_times")
    p.execute(dAx, dx, y)
    torch.cuda.synchronize()
    assert_parity(oracle, Ap, Aj, Ax, x, y.cpu().numpy())
    p.destroy()


# ---- y = alpha * A x + beta * y -----------------------------------------------

@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("off,val", [("i32", "f32"), ("i64", "f64")])
@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (2.5, 0.0), (1.0, 1.0), (-0.75, 3.0), (0.0, 2.0)])
def test_alpha_beta(sp, oracle, kind, off, val, alpha, beta):
    """Bound: (len+3) eps (|alpha| sum|a x| + |beta y0|) around alpha*y64 + beta*y0 in fp64.
    beta = 0 must ignore y's old content (NaN poison), as the reference's beta = 0 does
    (cusparse.cuh:42-43; cpu_navie.hpp:15 overwrites)."""
    rng = np.random.RandomState(91)
    Ap, Aj, Ax = random_csr(rng, 9001, 700, 28, NP[off], NP[val], long_row=15000)
    x = (rng.rand(700) * 2 - 1).astype(NP[val])
    y0 = (rng.rand(9001) * 2 - 1).astype(NP[val]) if beta != 0.0 else np.full(9001, np.nan, NP[val])
    d = lambda a: torch.from_numpy(a).to(DEV)
    dAp, dAj, dAx, dx, y = d(Ap), d(Aj), d(Ax), d(x), d(y0.copy())
    p = sp.Plan(kind, 9001, 700, int(Ap[-1]), dAp, dAj, dAx.dtype)
    p.set_alpha_beta(alpha, beta)
    p.execute(dAx, dx, y)
    torch.cuda.synchronize()
    p.destroy()
    got = y.cpu().numpy().astype(np.float64)
    y64, yabs = oracle.spmv_ref64(Ap, Aj, Ax, x)
    eps = 2.0 ** -24 if val == "f32" else 2.0 ** -53
    y0z = np.where(np.isnan(y0), 0.0, y0).astype(np.float64)
    want = alpha * y64 + beta * y0z
    bound = (np.diff(Ap.astype(np.int64)) + 3) * eps * (abs(alpha) * yabs + np.abs(beta * y0z)) + 1e-300
    assert not np.any(np.isnan(got))
    assert np.all(np.abs(got - want) <= bound)


def test_alpha_beta_is_for_the_ordinary_semiring_only(sp):
    Ap = torch.tensor([0, 1], dtype=torch.int32, device=DEV)
    Aj = torch.tensor([0], dtype=torch.int32, device=DEV)
    p = sp.Plan("merge", 1, 1, 1, Ap, Aj, torch.float32)
    p.set_semiring("min_plus")
    with pytest.raises(RuntimeError, match="not supported"):
        p.set_alpha_beta(2.0, 0.0)
    p.set_alpha_beta(1.0, 0.0)
    p.destroy()


# ---- randomized structures: bands, several bands, hubs, tails ----------------------------------

def _structured_csr(rng, kind_of):
    """Small matrices built to hit the kernels' special paths: windows (one band), segments
    (several bands), no window (scattered), long rows (hubs), empty rows, nnz not a multiple of 4."""
    n_rows = int(rng.randint(1, 6000))
    n_cols = int(rng.randint(1, 9000))
    mean = [1, 3, 9, 33, 70, 200][rng.randint(6)]
    lens = rng.poisson(mean, size=n_rows)
    lens[rng.rand(n_rows) < 0.1] = 0
    if rng.rand() < 0.5:
        lens[rng.randint(n_rows)] = int(rng.randint(500, 20000))           # a hub row
    Ap = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(lens, out=Ap[1:])
    nnz = int(Ap[-1])
    rows = np.repeat(np.arange(n_rows), lens)
    if kind_of == "band":
        w = int(rng.randint(1, 400))
        cols = rows * n_cols // max(n_rows, 1) + rng.randint(-w, w + 1, size=nnz)
    elif kind_of == "bands3":
        off = np.array([-n_cols // 3, 0, n_cols // 3])[rng.randint(3, size=nnz)]
        cols = rows * n_cols // max(n_rows, 1) + off + rng.randint(-20, 21, size=nnz)
    else:
        cols = rng.randint(0, n_cols, size=nnz)
    cols = np.clip(cols, 0, n_cols - 1).astype(np.int32)
    if rng.rand() < 0.5:                                                   # sorted inside rows, as most files are
        order = np.lexsort((cols, rows))
        cols = cols[order]
    return n_rows, n_cols, Ap, cols, nnz


@pytest.mark.parametrize("structure", ["band", "bands3", "scatter"])
@pytest.mark.parametrize("off,val", [("i32", "f32"), ("i64", "f64"), ("i64", "f32")])
def test_randomized_structures(sp, oracle, structure, off, val):
    rng = np.random.RandomState(hash((structure, off, val)) % 2 ** 31)
    for trial in range(12):
        n_rows, n_cols, Ap, Aj, nnz = _structured_csr(rng, structure)
        Ap = Ap.astype(NP[off])
        integer = trial % 3 == 0
        if integer:
            Ax = rng.randint(-3, 4, size=nnz).astype(NP[val])
            x = rng.randint(-2, 3, size=n_cols).astype(NP[val])
        else:
            Ax = (rng.rand(nnz) * 2 - 1).astype(NP[val])
            x = (rng.rand(n_cols) * 2 - 1).astype(NP[val])
        for kind in KINDS:
            y = gpu_spmv(sp, kind, n_cols, Ap, Aj, Ax, x, plan=(trial % 2 == 0))
            assert_parity(oracle, Ap, Aj, Ax, x, y, exact=integer)


# ---- separate matrix / vector value types (reference include/spmv.h:29-34) -------------------------------

@pytest.mark.parametrize("off", ["i32", "i64"])
def test_fp32_matrix_under_fp64_vectors_merge(sp, oracle, off):
    """mi355_spmv_plan_create_typed: matrix stored in fp32, x / y and all arithmetic fp64 (merge kind).  The
    result must sit inside the fp64 bound around the serial fp64 sum of the WIDENED matrix — i.e. no fp32
    rounding anywhere — for ragged rows, a window-friendly band and a hub row; other combinations are refused."""
    rng = np.random.RandomState(71)
    cases = [random_csr(rng, 20011, 5000, 12, NP[off], np.float32, long_row=30000)]
    m = sp.synth.banded_fixed(200_000, 32, 700, seed=8, device="cpu", off_dtype=torch.int64 if off == "i64" else torch.int32)
    cases.append(m.numpy())
    for Ap, Aj, Ax in cases:
        n_rows, n_cols = len(Ap) - 1, int(Aj.max()) + 1
        x = (rng.rand(n_cols) * 2 - 1)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
        dAp, dAj, dAx, dx = d(Ap), d(Aj), d(Ax), d(x)
        y = torch.full((n_rows,), float("nan"), dtype=torch.float64, device=DEV)
        sp.spmv_mixed(n_rows, n_cols, int(Ap[-1]), dAp, dAj, dAx, dx, y)
        y64, bound = parity_bound(oracle, Ap, Aj, Ax.astype(np.float64), x)
        err = np.abs(y.cpu().numpy() - y64)
        assert not np.isnan(err).any() and np.all(err <= bound), int((err > bound).sum())
        # the same through a plan, with alpha / beta
        p = sp.Plan("merge", n_rows, n_cols, int(Ap[-1]), dAp, dAj, torch.float64, mat_dtype=torch.float32)
        p.set_alpha_beta(-0.5, 0.25)
        y0 = rng.rand(n_rows)
        y2 = d(y0)
        p.execute(dAx, dx, y2)
        torch.cuda.synchronize()
        p.destroy()
        assert np.all(np.abs(y2.cpu().numpy() - (-0.5 * y64 + 0.25 * y0)) <= 0.5 * bound + 1e-300)
    dAp, dAj = d(cases[0][0]), d(cases[0][1])
    for kind in ("vector", "light"):
        with pytest.raises(RuntimeError, match="not supported"):
            sp.Plan(kind, 20011, 5000, int(cases[0][0][-1]), dAp, dAj, torch.float64, mat_dtype=torch.float32)
    with pytest.raises(RuntimeError, match="not supported"):
        sp.Plan("merge", 20011, 5000, int(cases[0][0][-1]), dAp, dAj, torch.float32, mat_dtype=torch.float64)


def test_merge_on_a_regular_matrix_takes_row_parallel_runs(sp, oracle):
    """A big matrix whose rows are alike (the target's shape): the merge kind keeps its merge-path runs but sums each
    run row-parallel (merge_rows_kernel).  Every row inside the bound, alpha / beta, bitwise reproducible, the
    tile coordinates still available (computed on demand) and still the oracle's; a ragged matrix keeps the item walk."""
    m = sp.synth.banded_fixed(1_200_000, 32, 900, seed=6, device=DEV)    # (big enough for runs of 16 tiles)
    x = sp.synth.dense_vector(m.n_cols, torch.float32, 6, DEV)
    p = sp.Plan("merge", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, torch.float32)
    info = p.info()
    if not any(k.startswith("MI355_") for k in os.environ if k not in ("MI355_SPMV_LIB", "MI355_SPMV_SMALL")):
        assert info["main_kernel"] == "merge_rows_kernel", info
    y = torch.full((m.n_rows,), float("nan"), device=DEV)
    p.execute(m.Ax, x, y)
    torch.cuda.synchronize()
    Ap, Aj, Ax = m.numpy()
    assert_parity(oracle, Ap, Aj, Ax, x.cpu().numpy(), y.cpu().numpy())
    y2 = torch.full((m.n_rows,), float("nan"), device=DEV)
    p.execute(m.Ax, x, y2)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    rows, nz = p.merge_coords()
    xs, ys = oracle.merge_tile_coords(Ap, info["tile_items"])
    assert np.array_equal(rows, xs) and np.array_equal(nz, ys)
    p.set_alpha_beta(3.0, -0.5)
    y0 = sp.synth.dense_vector(m.n_rows, torch.float32, 7, DEV)
    y3 = y0.clone()
    p.execute(m.Ax, x, y3)
    torch.cuda.synchronize()
    p.destroy()
    y64, bound = parity_bound(oracle, Ap, Aj, Ax, x.cpu().numpy())
    want = 3.0 * y64 - 0.5 * y0.cpu().numpy().astype(np.float64)
    assert np.all(np.abs(y3.cpu().numpy() - want) <= 3.0 * bound + 1e-6 * np.abs(want) + 1e-30)
    rng = np.random.RandomState(3)
    Ap2, Aj2, Ax2 = random_csr(rng, 200_000, 5000, 40)
    d = lambda a: torch.from_numpy(a).to(DEV)
    p = sp.Plan("merge", 200_000, 5000, int(Ap2[-1]), d(Ap2), d(Aj2), torch.float32)
    if "MI355_MERGE_ROWS" not in os.environ:
        assert p.info()["main_kernel"] == "merge_tile_kernel"
    p.destroy()


# ---- MI355_KIND_AUTO: the library picks the kind ---------------------------------------------------------------

def test_auto_kind_picks_merge_on_skewed_rows_and_vector_otherwise(sp, oracle):
    """The reference leaves the kind to the command line (main.cu:26-30).  MI355_KIND_AUTO: merge-path when the row
    lengths are skewed (the VECTOR plan's own analysis had to cut its chunks by weight), VECTOR otherwise; integer
    values are merge's.  Whatever is picked, the result is that kind's result, bit for bit."""
    band = sp.synth.banded_fixed(1 << 16, 32, 2048, 1, DEV)
    skew = sp.synth.rmat(16, 16, seed=5, device=DEV)
    for m, want in ((band, "vector"), (skew, "merge")):
        x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, DEV)
        p = sp.Plan("auto", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
        if any(k.startswith("MI355_") for k in os.environ if k not in ("MI355_SPMV_LIB", "MI355_SPMV_SMALL")):   # (a forced code path may force the pick:
            want = sp.capi.KIND_NAMES[p.info()["kind"]]                                  #  MI355_SPMV_BALANCE=1 makes every matrix "skewed")
        assert sp.capi.KIND_NAMES[p.info()["kind"]] == want
        q = sp.Plan(want, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
        y = torch.full((m.n_rows,), float("nan"), device=DEV)
        y2 = torch.full((m.n_rows,), float("nan"), device=DEV)
        p.execute(m.Ax, x, y)
        q.execute(m.Ax, x, y2)
        torch.cuda.synchronize()
        assert torch.equal(y, y2) and not torch.isnan(y).any()
        # the one-shot symbol (mi355_spmv_auto_*), twice: the second call finds the kept plan under the AUTO key
        y3 = torch.full((m.n_rows,), float("nan"), device=DEV)
        sp.spmv("auto", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, y3)
        sp.spmv("hip_auto", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, y3)
        assert torch.equal(y3, y2)
        assert_parity(oracle, m.Ap.cpu().numpy(), m.Aj.cpu().numpy(), m.Ax.cpu().numpy(), x.cpu().numpy(), y.cpu().numpy())
        p.destroy(); q.destroy()
    # integer values: merge, the only kind that has them
    rng = np.random.RandomState(4)
    Ap, Aj, _ = random_csr(rng, 3001, 500, 12, np.int32, np.float32)
    Ax = rng.randint(-5, 6, size=int(Ap[-1])).astype(np.int32)
    xi = rng.randint(-5, 6, size=500).astype(np.int32)
    d = lambda a: torch.from_numpy(a).to(DEV)
    p = sp.Plan("auto", 3001, 500, int(Ap[-1]), d(Ap), d(Aj), torch.int32)
    assert sp.capi.KIND_NAMES[p.info()["kind"]] == "merge"
    yi = torch.zeros(3001, dtype=torch.int32, device=DEV)
    p.execute(d(Ax), d(xi), yi)
    torch.cuda.synchronize()
    assert np.array_equal(yi.cpu().numpy(), oracle.spmv_genl_serial(0, Ap, Aj, Ax, xi))
    # a row block cannot be AUTO (its kind is the whole matrix's)
    with pytest.raises(RuntimeError):
        sp.Plan.block("auto", None, 0, 0, 1, 0, 3001, 500, int(Ap[-1]), d(Ap), d(Aj), torch.float32)


@pytest.mark.parametrize("off,val", [("i64", "f64"), ("i32", "f32")])
def test_merge_runs_on_a_stencil_stage_a_window_segment_per_band(sp, oracle, off, val):
    """A 27-point stencil on a 90^3 box: three far-apart bands of columns and boundary rows of 18 / 12 / 8 nonzeros among
    the 27s.  The merge kind takes row-parallel runs with one segment of the window per band (shape_merge; the probe's
    count of short rows lets the few boundary rows through) — every row within the bound of SURVEY §8(c), the rows that
    straddle runs included (their partial sums meet in the fix-up)."""
    tv = {"f32": torch.float32, "f64": torch.float64}[val]
    to = {"i32": torch.int32, "i64": torch.int64}[off]
    m = sp.synth.stencil27(90, 90, 90, 4, DEV, val_dtype=tv, off_dtype=to)
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, DEV)
    p = sp.Plan("merge", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
    info = p.info()
    if not any(k.startswith("MI355_") for k in os.environ if k not in ("MI355_SPMV_LIB", "MI355_SPMV_SMALL")):   # (no forced code path)
        assert info["main_kernel"] == "merge_rows_kernel" and info["window_segments"] == 3, info
    y = torch.full((m.n_rows,), float("nan"), dtype=m.Ax.dtype, device=DEV)
    p.execute(m.Ax, x, y)
    p.execute(m.Ax, x, y)
    torch.cuda.synchronize()
    Ap, Aj, Ax, xh = m.Ap.cpu().numpy(), m.Aj.cpu().numpy(), m.Ax.cpu().numpy(), x.cpu().numpy()
    assert_parity(oracle, Ap, Aj, Ax, xh, y.cpu().numpy())
    lens = np.diff(Ap.astype(np.int64))
    assert lens.max() == 27 and lens.min() == 8
    # alpha / beta ride along the same kernels
    p.set_alpha_beta(0.5, 2.0)
    y2 = torch.ones(m.n_rows, dtype=m.Ax.dtype, device=DEV)
    p.execute(m.Ax, x, y2)
    torch.cuda.synchronize()
    want = 0.5 * y.cpu().numpy().astype(np.float64) + 2.0
    tol = 1e-5 if val == "f32" else 1e-13
    assert np.allclose(y2.cpu().numpy(), want, rtol=tol, atol=tol * 32)
    p.destroy()


def test_mid_size_plans_are_one_round_of_two_workgroups_per_cu(sp, oracle):
    """Plan rules measured in round 3 (profiles/r03_mid_size_*.txt), pinned: a regular banded matrix whose kernel is a
    single round of the chip takes 512-thread workgroups, two per CU (or one with twice the rows), and its merge plan
    runs of 8 tiles summed row-parallel.  Results: the oracle's bound, as everywhere."""
    if any(k.startswith("MI355_") for k in os.environ if k not in ("MI355_SPMV_LIB", "MI355_SPMV_SMALL")):
        pytest.skip("a forced code path decides the plan")
    for lg, grid in ((17, 256), (18, 512)):
        m = sp.synth.banded_fixed(1 << lg, 32, 4096, 1, DEV)
        x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, DEV)
        for kind in ("vector", "light", "merge"):
            p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
            info = p.info()
            if kind == "merge":
                assert info["main_kernel"] == "merge_rows_kernel" and info["window_elems"] > 0, info
                assert info["grid_blocks"] == -(-info["n_tiles"] // 8), info
            else:
                assert info["block_threads"] == 512 and info["window_elems"] > 0, info
                if kind == "vector":
                    assert info["grid_blocks"] == grid, info
            y = torch.full((m.n_rows,), float("nan"), device=DEV)
            p.execute(m.Ax, x, y)
            torch.cuda.synchronize()
            assert_parity(oracle, m.Ap.cpu().numpy(), m.Aj.cpu().numpy(), m.Ax.cpu().numpy(), x.cpu().numpy(), y.cpu().numpy())
            p.destroy()
