"""GPU suite: the product's DEFAULT on small matrices — the VECTOR kind's plain one-pass kernel (capi.hip, small_plain;
common.hpp, kSmallPlainNnz).  tests/conftest.py switches that choice off for the rest of the suite (whose small matrices
are there to exercise the chunked kernels); here it is switched back on, the knobs re-read, and the same kinds of
matrices — ragged, empty rows, one long row, one column, every type combination, row blocks — go through it."""
import os

import numpy as np
import pytest
import torch

from conftest import random_csr
from test_gpu_parity import assert_parity

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NP = {"i32": np.int32, "i64": np.int64, "f32": np.float32, "f64": np.float64}


@pytest.fixture()
def small_on(sp):
    saved = os.environ.pop("MI355_SPMV_SMALL", None)
    sp.capi.lib().mi355_spmv_knobs_reload()
    yield
    if saved is not None:
        os.environ["MI355_SPMV_SMALL"] = saved
    sp.capi.lib().mi355_spmv_knobs_reload()


def d(a):
    return torch.from_numpy(a).to(DEV)


@pytest.mark.parametrize("off,val", [("i32", "f32"), ("i32", "f64"), ("i64", "f32"), ("i64", "f64")])
def test_small_matrices_take_the_plain_kernel_and_keep_parity(sp, oracle, small_on, off, val):
    rng = np.random.RandomState(11)
    for n_rows, n_cols, max_len, long_row in ((3001, 2500, 24, 900), (777, 1, 3, None), (20011, 4000, 70, None), (5, 5, 2, None)):
        Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, max_len, NP[off], NP[val], long_row=long_row)
        x = (rng.rand(n_cols) * 2 - 1).astype(NP[val])
        nnz = int(Ap[-1])
        pl = sp.Plan("light", n_rows, n_cols, nnz, d(Ap), d(Aj), d(Ax).dtype)
        yl = torch.full((n_rows,), float("nan"), dtype=d(Ax).dtype, device=DEV)
        pl.execute(d(Ax), d(x), yl)
        p = sp.Plan("vector", n_rows, n_cols, nnz, d(Ap), d(Aj), d(Ax).dtype)
        info = p.info()
        if nnz > 0:
            assert info["main_kernel"] == "csr_vector_kernel" and info["window_elems"] == 0, info
        y = torch.full((n_rows,), float("nan"), dtype=d(Ax).dtype, device=DEV)
        p.execute(d(Ax), d(x), y)
        torch.cuda.synchronize()
        assert_parity(oracle, Ap, Aj, Ax, x, y.cpu().numpy())
        assert torch.equal(yl, y)                                          # light takes the same kernel here: the same sums
        pl.destroy()
        # alpha / beta
        p.set_alpha_beta(2.0, -1.0)
        y2 = torch.ones(n_rows, dtype=d(Ax).dtype, device=DEV)
        p.execute(d(Ax), d(x), y2)
        torch.cuda.synchronize()
        tol = 1e-5 if val == "f32" else 1e-13
        assert np.allclose(y2.cpu().numpy(), 2.0 * y.cpu().numpy().astype(np.float64) - 1.0, rtol=tol, atol=tol * 64)
        # the one-shot symbol and the library's own pick go the same way, bit for bit
        y3 = torch.full((n_rows,), float("nan"), dtype=d(Ax).dtype, device=DEV)
        sp.spmv("vector", n_rows, n_cols, nnz, d(Ap), d(Aj), d(Ax), d(x), y3)
        assert torch.equal(y3, y)
        p.destroy()


def test_a_skewed_or_big_matrix_keeps_the_chunked_kernels(sp, small_on):
    skew = sp.synth.rmat(14, 16, seed=5, device=DEV)                       # weight-cut chunks: not the plain kernel
    p = sp.Plan("vector", skew.n_rows, skew.n_cols, skew.nnz, skew.Ap, skew.Aj, skew.Ax.dtype)
    assert p.info()["main_kernel"] != "csr_vector_kernel"
    big = sp.synth.banded_fixed(1 << 17, 32, 4096, 1, DEV)                 # 4.19 M nonzeros: above the threshold
    q = sp.Plan("vector", big.n_rows, big.n_cols, big.nnz, big.Ap, big.Aj, big.Ax.dtype)
    assert q.info()["main_kernel"] == "csr_vector_window_kernel"
    small = sp.synth.banded_fixed(1 << 14, 32, 4096, 1, DEV)
    r = sp.Plan("vector", small.n_rows, small.n_cols, small.nnz, small.Ap, small.Aj, small.Ax.dtype)
    assert r.info()["main_kernel"] == "csr_vector_kernel" and r.info()["lanes_per_row"] == 16
    t = sp.Plan("light", small.n_rows, small.n_cols, small.nnz, small.Ap, small.Aj, small.Ax.dtype)
    assert t.info()["main_kernel"] == "csr_vector_kernel"                  # handing rows out would cost more than summing them
    t = sp.Plan("merge", small.n_rows, small.n_cols, small.nnz, small.Ap, small.Aj, small.Ax.dtype)
    assert t.info()["main_kernel"] != "csr_vector_kernel"                  # merge-path is what it was


def test_row_blocks_of_a_small_matrix_equal_the_one_gpu_result(sp, small_on):
    """A block inherits the whole plan's choice (mi355_spmv_plan_shape.small_plain) and its lanes per row: the same
    sums bit for bit — what the multi-GPU path promises for the VECTOR kind."""
    m = sp.synth.banded_fixed(30000, 24, 500, 3, DEV)
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, DEV)
    whole = sp.Plan("vector", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
    assert whole.shape().small_plain == 1
    y1 = torch.full((m.n_rows,), float("nan"), device=DEV)
    whole.execute(m.Ax, x, y1)
    for kind in ("vector", "light"):
        dp = sp.DistPlan.local(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype, sub_blocks=3)
        y2 = torch.full((m.n_rows,), float("nan"), device=DEV)
        dp.execute(m.Ax, x, y2)
        torch.cuda.synchronize()
        assert torch.equal(y1, y2) and not torch.isnan(y1).any()
        dp.destroy()
