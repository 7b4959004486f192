import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")

# The product runs a SMALL regular matrix (VECTOR kind, up to 4.1 M nonzeros) on its plain one-pass kernel (capi.hip,
# small_plain).  Most matrices of this suite are small on purpose — empty rows, one long row, ragged lengths — and are
# there to put the CHUNKED kernels through those cases, so the suite switches that choice off for itself;
# tests/test_gpu_small.py runs the same kinds of matrices through the product's default, and the forced-code-path
# matrix (scripts/gpu_env_matrix.sh) has a MI355_SPMV_SMALL=1 pass over the whole parity file.
os.environ.setdefault("MI355_SPMV_SMALL", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sp():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The reference's own loader + serial SpMV, compiled from /root/reference by
    oracle/Makefile (present in the build container; the prebuilt file also travels
    to the GPU box).  Tests that need it skip when it is absent."""
    from oracle.oracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libspmv_ref.so not built (no /root/reference here)")
    return Ref()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLD, "golden.json")) as f:
        return json.load(f)


def unhex(lst, dtype):
    return np.array([float.fromhex(v) for v in lst], dtype=dtype)


def seeded_x(n, dtype):
    """The non-trivial x of oracle/make_golden.py."""
    i = np.arange(n, dtype=np.int64)
    return (((7 * i + 3) % 11 - 5) / 4.0).astype(dtype)


def random_csr(rng, n_rows, n_cols, max_len, off_dtype=np.int32, val_dtype=np.float32,
               empty_frac=0.2, long_row=None, integer_values=False):
    """Ragged CSR with empty rows, duplicates and unsorted columns."""
    lens = rng.randint(0, max_len + 1, size=n_rows)
    lens[rng.rand(n_rows) < empty_frac] = 0
    if long_row is not None and n_rows > 0:
        lens[rng.randint(0, n_rows)] = long_row
    Ap = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(lens, out=Ap[1:])
    nnz = int(Ap[-1])
    Aj = rng.randint(0, max(n_cols, 1), size=nnz).astype(np.int32)
    if integer_values:
        Ax = rng.randint(-3, 4, size=nnz).astype(val_dtype)
    else:
        Ax = (rng.rand(nnz) * 2 - 1).astype(val_dtype)
    return Ap.astype(off_dtype), Aj, Ax


def parity_bound(oracle, Ap, Aj, Ax, x, n_threads=1):
    """Per-row bound of SURVEY.md §8(c): (len+2) * eps * sum|a*x| around the fp64 serial sum.
    n_threads splits the rows of the oracle pass over host threads (same values)."""
    y64, yabs = oracle.spmv_ref64(Ap, Aj, Ax, x, n_threads)
    eps = 2.0 ** -24 if Ax.dtype == np.float32 else 2.0 ** -53
    lens = np.diff(Ap.astype(np.int64))
    return y64, (lens + 2) * eps * yabs
