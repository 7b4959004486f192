"""CPU suite, part 5: the PRODUCT Matrix Market loader (spmv-samples_amd/host/load.hpp,
SURVEY §8(f)-1) against the golden vectors, the oracle and the reference build — entry for
entry, for every type combination, with the parallel parser forced onto tiny chunks so that
entries straddle chunk boundaries."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLD, ROOT, unhex

FIXTURES = sorted(f for f in os.listdir(GOLD) if f.endswith(".mtx"))
COMBOS = [("i32", "f32"), ("i32", "f64"), ("i64", "f32"), ("i64", "f64")]
NP = {"i32": np.int32, "i64": np.int64, "f32": np.float32, "f64": np.float64}


@pytest.fixture(scope="module")
def hostlib():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "libhostload.so"], check=True)
    lib = C.CDLL(os.path.join(ROOT, "tests", "cpp", "libhostload.so"))
    for o, v in COMBOS:
        getattr(lib, "host_load_%s_%s" % (o, v)).restype = C.c_void_p
    return lib


def host_load(lib, path, off, val, roundtrip=None):
    s = "%s_%s" % (off, val)
    st = C.c_int(0)
    h = getattr(lib, "host_load_" + s)(os.fsencode(path), C.byref(st))
    if not h:
        raise ValueError({5: "overflow", 6: "entry"}.get(st.value, str(st.value)))
    h = C.c_void_p(h)
    nr, nc, nz = C.c_int64(), C.c_int64(), C.c_int64()
    getattr(lib, "host_dims_" + s)(h, C.byref(nr), C.byref(nc), C.byref(nz))
    Ap = np.empty(nr.value + 1, dtype=NP[off])
    Aj = np.empty(nz.value, dtype=np.int32)
    Ax = np.empty(nz.value, dtype=NP[val])
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    getattr(lib, "host_copy_" + s)(h, p(Ap), p(Aj), p(Ax))
    if roundtrip:
        assert getattr(lib, "host_roundtrip_" + s)(h, os.fsencode(roundtrip)) == 0
    getattr(lib, "host_free_" + s)(h)
    return nr.value, nc.value, Ap, Aj, Ax


@pytest.mark.parametrize("chunk", ["1048576", "7", "1"])
@pytest.mark.parametrize("name", FIXTURES)
def test_product_loader_matches_golden(hostlib, golden, name, chunk, monkeypatch, tmp_path):
    monkeypatch.setenv("MI355_LOAD_CHUNK", chunk)       # tiny chunks: entries straddle chunk boundaries
    monkeypatch.setenv("MI355_LOAD_THREADS", "4")
    g = golden[name]
    for off, val in COMBOS:
        nr, nc, Ap, Aj, Ax = host_load(hostlib, os.path.join(GOLD, name), off, val,
                                       roundtrip=str(tmp_path / "cache.bin"))
        s = g["struct"]
        assert (nr, nc, int(Ap[-1])) == (s["n_rows"], s["n_cols"], s["nnz"])
        assert Ap.tolist() == s["Ap"] and Aj.tolist() == s["Aj"]
        assert np.array_equal(Ax, unhex(g[val]["Ax"], NP[val]))


def _write_random_mtx(path, rng, n, nz, field, symmetry, messy):
    rows = rng.randint(1, n + 1, size=nz)
    cols = rng.randint(1, n + 1, size=nz)
    if symmetry != "general":
        rows, cols = np.maximum(rows, cols), np.minimum(rows, cols)
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate %s %s\n%% generated\n%d %d %d\n" % (field, symmetry, n, n, nz))
        for r, c in zip(rows, cols):
            sep = ["\n", " \n ", "\t", "   "][rng.randint(4)] if messy else " "
            if field == "pattern":
                f.write("%d%s%d\n" % (r, sep, c))
            elif field == "integer":
                f.write("%d%s%d %d\n" % (r, sep, c, rng.randint(-9, 10)))
            else:
                v = rng.randn() * 10.0 ** rng.randint(-30, 30)
                fmt = ["%.17g", "%.6e", "%.3f", "%g"][rng.randint(4)]
                f.write(("%d%s%d " + fmt + "\n") % (r, sep, c, v))


@pytest.mark.parametrize("field", ["real", "integer", "pattern"])
@pytest.mark.parametrize("symmetry", ["general", "symmetric", "skew-symmetric"])
@pytest.mark.parametrize("messy", [False, True])
def test_product_loader_matches_oracle_and_reference_on_generated_files(hostlib, oracle, tmp_path, monkeypatch,
                                                                         field, symmetry, messy):
    """Values in many notations and magnitudes (the fast double path must equal strtod bit
    for bit), entries split over lines, symmetric expansion order."""
    if field == "pattern" and symmetry == "skew-symmetric":
        pytest.skip("not a Matrix Market combination")
    rng = np.random.RandomState(hash((field, symmetry, messy)) % 2 ** 31)
    path = str(tmp_path / "gen.mtx")
    _write_random_mtx(path, rng, 97, 1500, field, symmetry, messy)
    from oracle.oracle import Ref
    for chunk in ("1048576", "13"):
        monkeypatch.setenv("MI355_LOAD_CHUNK", chunk)
        for off, val in COMBOS:
            got = host_load(hostlib, path, off, val)
            want = oracle.load_mtx(path, off, val)
            assert got[0] == want[0] and got[1] == want[1]
            for u, v in zip(got[2:], want[2:]):
                assert u.dtype == v.dtype and np.array_equal(u, v)
            if Ref.available():
                ref = Ref().load_mtx(path, off, val)
                for u, v in zip(got[2:], ref[2:]):
                    assert np.array_equal(u, v)


@pytest.mark.parametrize("text,err", [
    ("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n", "entry"),   # short file
    ("%%MatrixMarket matrix coordinate real general\n2 2 1\n0 1 1.0\n", "entry"),   # zero-based index
    ("%%MatrixMarket matrix coordinate pattern general\n2 2 2\n1 1\n2 x\n", "entry"),
    ("%%MatrixMarket matrix coordinate real general\n2147483647 1 0\n", "overflow"),
    # an index beyond the header's size: the reference does not look and then writes past row_offsets in ToCsr
    # (and a column beyond n_cols becomes x[col] on the device); the product loader refuses the file
    ("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n3 1 2.0\n", "entry"),
    ("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n2 5 2.0\n", "entry"),
])
def test_product_loader_exceptions(hostlib, tmp_path, text, err):
    """Malformed entries / overflow throw exception_t like the reference (load.hpp:302-306, :324-351)."""
    p = tmp_path / "bad.mtx"
    p.write_text(text)
    with pytest.raises(ValueError, match=err):
        host_load(hostlib, str(p), "i32", "f32")
