"""CPU suite, part 1: the oracle is pinned before anything trusts it.

  * against the golden vectors (tests/golden/golden.json) that oracle/make_golden.py
    produced from the REFERENCE'S OWN loader and serial SpMV (load.hpp, cpu_navie.hpp)
  * against the reference library itself where it is built (oracle/_ref)
  * against the one known-answer vector the reference holds: the 9x9 lattice of
    include/spmv/merge_based/device_spmv.cuh:95-128
"""
import os

import numpy as np
import pytest

from conftest import GOLD, parity_bound, random_csr, seeded_x, unhex

FIXTURES = sorted(f for f in os.listdir(GOLD) if f.endswith(".mtx"))
COMBOS = [("i32", "f32"), ("i32", "f64"), ("i64", "f32"), ("i64", "f64")]
NP = {"i32": np.int32, "i64": np.int64, "f32": np.float32, "f64": np.float64}


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("off,val", COMBOS)
def test_loader_and_serial_spmv_match_golden(oracle, golden, name, off, val):
    g = golden[name]
    nr, nc, Ap, Aj, Ax = oracle.load_mtx(os.path.join(GOLD, name), off, val)
    s = g["struct"]
    assert (nr, nc, int(Ap[-1])) == (s["n_rows"], s["n_cols"], s["nnz"])
    assert Ap.dtype == NP[off] and Ap.tolist() == s["Ap"]          # bit-exact Ap
    assert Aj.tolist() == s["Aj"]                                  # bit-exact Aj (file order kept)
    assert np.array_equal(Ax, unhex(g[val]["Ax"], NP[val]))        # bit-exact values
    y1 = oracle.spmv_serial(Ap, Aj, Ax, np.ones(nc, dtype=NP[val]))
    ys = oracle.spmv_serial(Ap, Aj, Ax, seeded_x(nc, NP[val]))
    assert np.array_equal(y1, unhex(g[val]["y_ones"], NP[val]))    # bit-exact y
    assert np.array_equal(ys, unhex(g[val]["y_seeded"], NP[val]))


def test_known_answer_lattice(oracle):
    """device_spmv.cuh:95-128: row_offsets, column_indices and y for x = 1."""
    nr, nc, Ap, Aj, Ax = oracle.load_mtx(os.path.join(GOLD, "lattice9_cub_doc.mtx"))
    assert Ap.tolist() == [0, 2, 5, 7, 10, 14, 17, 19, 22, 24]
    assert Aj.tolist() == [1, 3, 0, 2, 4, 1, 5, 0, 4, 6, 1, 3, 5, 7, 2, 4, 8, 3, 7, 4, 6, 8, 5, 7]
    want = [2, 3, 2, 3, 4, 3, 2, 3, 2]
    x = np.ones(9, dtype=np.float32)
    assert oracle.spmv_serial(Ap, Aj, Ax, x).tolist() == want
    for bt, ipt in [(2, 2), (4, 3), (128, 7), (64, 5), (256, 8)]:
        assert oracle.spmv_merge_order(Ap, Aj, Ax, x, bt, ipt).tolist() == want
    for T in (2, 4, 8, 16, 32, 64):
        assert oracle.spmv_vector_order(Ap, Aj, Ax, x, T).tolist() == want


def test_loader_quirks(oracle):
    """SURVEY.md §2c quirk 1, observed on the reference build in the survey container."""
    _, _, Ap, Aj, Ax = oracle.load_mtx(os.path.join(GOLD, "pat3x4_dup_unsorted.mtx"))
    assert Ap.tolist() == [0, 2, 2, 4] and Aj.tolist() == [3, 1, 0, 0] and Ax.tolist() == [1, 1, 1, 1]
    _, _, Ap, Aj, Ax = oracle.load_mtx(os.path.join(GOLD, "skew3_not_expanded.mtx"))
    assert int(Ap[-1]) == 2                                        # skew-symmetric is NOT expanded
    _, _, Ap, Aj, Ax = oracle.load_mtx(os.path.join(GOLD, "sym4_real.mtx"))
    assert int(Ap[-1]) == 8                                        # 5 stored, 3 off-diagonal mirrored
    nr, nc, Ap, Aj, Ax = oracle.load_mtx(os.path.join(GOLD, "c1_1138_bus_standin.mtx"))
    assert (nr, nc, int(Ap[-1])) == (1138, 1138, 4054)             # BASELINE config C1 shape


@pytest.mark.parametrize("text,err", [
    ("", "banner"),
    ("%%MatrixMarket matrix coordinate real\n1 1 1\n1 1 1\n", "banner"),          # 4 banner tokens
    ("%MatrixMarket matrix coordinate real general\n1 1 1\n1 1 1\n", "banner"),    # wrong banner word
    ("%%MatrixMarket vector coordinate real general\n1 1 1\n1 1 1\n", "banner"),   # not "matrix"
    ("%%MatrixMarket matrix array real general\n1 1\n1\n", "array"),               # dense file
    ("%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1 0\n", "type"),
    ("%%MatrixMarket matrix coordinate real general\n% only comments\n", "size"),
    ("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n", "entry"),  # short file
    ("%%MatrixMarket matrix coordinate real general\n2 2 1\n0 1 1.0\n", "entry"),  # zero-based index
    ("%%MatrixMarket matrix coordinate real general\n2147483647 1 0\n", "overflow"),
])
def test_loader_error_classes(oracle, tmp_path, text, err):
    """Where the reference exits or throws (load.hpp:278-306, :324-329, :357-360) the
    restatement reports the same failure class."""
    p = tmp_path / "bad.mtx"
    p.write_text(text)
    with pytest.raises(ValueError, match=err):
        oracle.load_mtx(str(p))


def test_loader_missing_file(oracle, tmp_path):
    with pytest.raises(ValueError, match="open"):
        oracle.load_mtx(str(tmp_path / "nope.mtx"))


# ---- against the reference library itself (build container; prebuilt on the GPU box) ----

@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("off,val", COMBOS)
def test_loader_matches_reference_build(oracle, ref, name, off, val):
    path = os.path.join(GOLD, name)
    a = oracle.load_mtx(path, off, val)
    b = ref.load_mtx(path, off, val)
    assert a[0] == b[0] and a[1] == b[1]
    for u, v in zip(a[2:], b[2:]):
        assert u.dtype == v.dtype and np.array_equal(u, v)


@pytest.mark.parametrize("off,val", COMBOS)
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_serial_spmv_matches_reference_build(oracle, ref, off, val, seed):
    rng = np.random.RandomState(seed)
    Ap, Aj, Ax = random_csr(rng, 700 + 13 * seed, 300, 40, NP[off], NP[val], long_row=2000)
    x = (rng.rand(300) * 2 - 1).astype(NP[val])
    y_ref = ref.spmv_cpu(300, Ap, Aj, Ax, x)
    assert np.array_equal(oracle.spmv_serial(Ap, Aj, Ax, x), y_ref)            # bit-exact
    assert np.array_equal(oracle.spmv_parallel(Ap, Aj, Ax, x, 4), y_ref)       # row-parallel: same bits


def test_random_mtx_roundtrip_against_reference_build(oracle, ref, tmp_path):
    """A generated symmetric file with duplicates: both loaders must agree entry for entry."""
    rng = np.random.RandomState(7)
    n, nz = 60, 400
    rows = rng.randint(1, n + 1, size=nz)
    cols = rng.randint(1, n + 1, size=nz)
    lo, hi = np.minimum(rows, cols), np.maximum(rows, cols)
    vals = rng.randn(nz)
    p = tmp_path / "gen.mtx"
    with open(p, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n%c\n")
        f.write("%d %d %d\n" % (n, n, nz))
        for r, c, v in zip(hi, lo, vals):
            f.write("%d %d %.17g\n" % (r, c, v))
    a = oracle.load_mtx(str(p), "i32", "f64")
    b = ref.load_mtx(str(p), "i32", "f64")
    for u, v in zip(a[2:], b[2:]):
        assert np.array_equal(u, v)


# ---- the algorithm restatements ----------------------------------

@pytest.mark.parametrize("bt,ipt", [(4, 3), (8, 2), (128, 7), (64, 5)])
def test_merge_order_is_exact_on_integer_data(oracle, bt, ipt):
    """Integer-valued data sums exactly in any order: the merge-path decomposition
    (tiles, per-thread walks, scan, carry fix-up) must reproduce the serial result
    bit for bit, including empty rows and rows that straddle several tiles."""
    rng = np.random.RandomState(bt * 31 + ipt)
    for trial in range(40):
        n_rows = rng.randint(1, 60)
        Ap, Aj, Ax = random_csr(rng, n_rows, 17, 12, integer_values=True,
                                long_row=(bt * ipt * 3 if trial % 3 == 0 else None))
        x = rng.randint(-2, 3, size=17).astype(np.float32)
        want = oracle.spmv_serial(Ap, Aj, Ax, x)
        got = oracle.spmv_merge_order(Ap, Aj, Ax, x, bt, ipt)
        assert np.array_equal(got, want), (trial, n_rows)


@pytest.mark.parametrize("T", [2, 4, 8, 16, 32, 64])
def test_vector_order_within_bound_and_exact_on_integers(oracle, T):
    rng = np.random.RandomState(T)
    Ap, Aj, Ax = random_csr(rng, 300, 50, 90, integer_values=True, long_row=700)
    x = rng.randint(-2, 3, size=50).astype(np.float32)
    assert np.array_equal(oracle.spmv_vector_order(Ap, Aj, Ax, x, T), oracle.spmv_serial(Ap, Aj, Ax, x))
    assert np.array_equal(oracle.spmv_vector_order(Ap, Aj, Ax, x, T, 32), oracle.spmv_serial(Ap, Aj, Ax, x))
    Ap, Aj, Ax = random_csr(rng, 300, 50, 90, long_row=700)
    x = (rng.rand(50) * 2 - 1).astype(np.float32)
    y64, bound = parity_bound(oracle, Ap, Aj, Ax, x)
    for aligned in (-1, 32):
        y = oracle.spmv_vector_order(Ap, Aj, Ax, x, T, aligned)
        assert np.all(np.abs(y.astype(np.float64) - y64) <= bound)


def test_merge_tile_coords_properties(oracle):
    rng = np.random.RandomState(3)
    Ap, _, _ = random_csr(rng, 500, 10, 30, long_row=5000)
    n, nnz = 500, int(Ap[-1])
    xs, ys = oracle.merge_tile_coords(Ap, 2048)
    assert xs[0] == 0 and ys[0] == 0 and xs[-1] == n and ys[-1] == nnz
    assert np.all(np.diff(xs) >= 0) and np.all(np.diff(ys) >= 0)
    d = xs + ys
    assert np.all(d[:-1] == np.arange(len(d) - 1) * 2048) and d[-1] == n + nnz
    # a row that has ended before a boundary has all its nonzeros before it
    for t in range(len(xs)):
        assert Ap[xs[t]] <= ys[t]
        if xs[t] < n:
            assert ys[t] <= Ap[xs[t] + 1]


# ---- generalized (semiring) SpMV: cpu_navie.hpp:20-34 -------------------------------------------

@pytest.mark.parametrize("semiring", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("off,val", COMBOS)
def test_generalized_serial_matches_reference_build(oracle, ref, semiring, off, val):
    """The reference's own SpMV_genl_cpu_navie<functor_t>, instantiated in oracle/ref_driver.cpp with
    (+,*), (min,+), (max,*), (max,+) and (or,and), against the restatement — bit for bit, empty rows included."""
    rng = np.random.RandomState(40 + semiring)
    Ap, Aj, Ax = random_csr(rng, 500, 200, 30, NP[off], NP[val], long_row=900)
    x = (rng.rand(200) * 2 - 1).astype(NP[val])
    want = ref.spmv_genl_cpu(semiring, 200, Ap, Aj, Ax, x)
    got = oracle.spmv_genl_serial(semiring, Ap, Aj, Ax, x)
    assert np.array_equal(got, want)
    empty = np.diff(Ap.astype(np.int64)) == 0
    assert empty.any()
    ident = {0: 0.0, 1: np.inf, 2: -np.inf, 3: -np.inf, 4: 0.0}[semiring]
    assert np.all(got[empty] == ident)                       # an empty row yields initialize()
    if semiring == 0:
        assert np.array_equal(got, oracle.spmv_serial(Ap, Aj, Ax, x))


@pytest.mark.parametrize("semiring", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("off", ["i32", "i64"])
def test_generalized_serial_on_integer_values(oracle, ref, semiring, off):
    """32-bit integer values (MI355_VAL_I32): the restatement against the reference's own SpMV_genl_cpu_navie
    instantiated on int (oracle/ref_driver.cpp) and against a plain numpy loop — exact; empty rows yield the identity
    (0 / INT32_MAX / INT32_MIN); the (+,*) sum wraps around like the GPU's (checked against numpy's modular int32)."""
    rng = np.random.RandomState(60 + semiring)
    Ap, Aj, _ = random_csr(rng, 400, 150, 25, NP[off], np.float32, long_row=700)
    nnz = int(Ap[-1])
    Ax = rng.randint(-9, 10, size=nnz).astype(np.int32)
    x = rng.randint(-7, 8, size=150).astype(np.int32)
    if semiring == 4:
        Ax = (rng.rand(nnz) < 0.5).astype(np.int32)
        x = (rng.rand(150) < 0.3).astype(np.int32)
    got = oracle.spmv_genl_serial(semiring, Ap, Aj, Ax, x)
    assert got.dtype == np.int32
    assert np.array_equal(got, ref.spmv_genl_cpu(semiring, 150, Ap, Aj, Ax, x))
    ident = {0: 0, 1: 2 ** 31 - 1, 2: -2 ** 31, 3: -2 ** 31, 4: 0}[semiring]
    for r in range(400):
        a, b = Ax[Ap[r]:Ap[r + 1]].astype(np.int64), x[Aj[Ap[r]:Ap[r + 1]]].astype(np.int64)
        if a.size == 0:
            assert got[r] == ident
            continue
        want = {0: (a * b).sum(), 1: (a + b).min(), 2: (a * b).max(), 3: (a + b).max(), 4: int(((a != 0) & (b != 0)).any())}[semiring]
        assert got[r] == want
    if semiring == 0:                                   # wrap-around: defined here (uint32 arithmetic), as on the GPU
        big = np.full(nnz, 2 ** 30, dtype=np.int32)
        xs = np.full(150, 3, dtype=np.int32)
        wrapped = oracle.spmv_genl_serial(0, Ap, Aj, big, xs)
        lens = np.diff(Ap.astype(np.int64))
        assert np.array_equal(wrapped, ((lens * (3 * 2 ** 30) + 2 ** 31) % 2 ** 32 - 2 ** 31).astype(np.int32))


def test_ref64_threads_do_not_change_values(oracle):
    """The threaded pass used by the BASELINE-sized GPU tests splits ROWS only."""
    rng = np.random.RandomState(5)
    Ap, Aj, Ax = random_csr(rng, 5003, 700, 40, np.int64, np.float64, long_row=9000)
    x = (rng.rand(700) * 2 - 1).astype(np.float64)
    a = oracle.spmv_ref64(Ap, Aj, Ax, x)
    for t in (2, 7, 16):
        b = oracle.spmv_ref64(Ap, Aj, Ax, x, t)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("off", [np.int32, np.int64])
def test_mixed_types_restatement_equals_the_reference_template(oracle, ref, off):
    """fp32 matrix under fp64 vectors: the oracle's loop against the reference's own SpMV_cpu_navie
    instantiated <int, off, float, double, double> (include/spmv.h:29-34 keeps the three value types apart),
    bit for bit; and against the fp64 loop on the widened matrix (a float widens exactly)."""
    rng = np.random.RandomState(31)
    Ap, Aj, Ax = random_csr(rng, 3001, 800, 30, off, np.float32, long_row=5000)
    x = (rng.rand(800) * 2 - 1).astype(np.float64)
    y = oracle.spmv_serial_mixed(Ap, Aj, Ax, x)
    assert np.array_equal(y, ref.spmv_cpu_mixed(800, Ap, Aj, Ax, x))
    assert np.array_equal(y, oracle.spmv_serial(Ap, Aj, Ax.astype(np.float64), x))
