"""GPU suite, BASELINE.json configs at FULL size (SURVEY.md §8 sizes; synthetic stand-ins where the
SuiteSparse file is not available offline, same shape statistics, seeded):

  C2  CSR-vector config: cant stand-in (62 451 rows, ~4.0 M nnz, fp32)           every row, every kind
  C4  fp64 + 64-bit offsets: nlpkkt160 stand-in (8.4 M rows, 224 M nnz)          EVERY row against the oracle
  C5  R-MAT scale 24 (16.8 M rows, 268 M nnz, fp32): the multi-GPU config        every row on one GPU, and the
      8-way row-partitioned path executed block by block on the one GPU here

Bar as everywhere (tests/test_gpu_parity.py): per-row bound (len+2)·eps·Σ|a·x| around the fp64 serial sum,
bit-exact where the arithmetic is exact.  The oracle pass over 2-3·10^8 gathers is split over host
threads by ROWS (each row the same serial sum: tests/test_oracle.py).
"""
import os

import numpy as np
import pytest
import torch

from conftest import parity_bound

pytestmark = pytest.mark.gpu

KINDS = ["vector", "merge", "light"]
DEV = "cuda:0"
THREADS = max(1, min(16, os.cpu_count() or 1))


def check_rows(oracle, m, x, y, what):
    yh = y.cpu().numpy()
    assert not np.any(np.isnan(yh)), "%s: a row was skipped (NaN poison survived)" % what
    Ap, Aj, Ax = m.numpy()
    y64, bound = parity_bound(oracle, Ap, Aj, Ax, x.cpu().numpy(), THREADS)
    err = np.abs(yh.astype(np.float64) - y64)
    bad = np.nonzero(err > bound)[0]
    assert bad.size == 0, "%s: %d rows outside the bound, first %s (err %s, bound %s)" % (
        what, bad.size, bad[:5], err[bad[:5]], bound[bad[:5]])
    return y64, bound


@pytest.fixture(scope="module")
def c2(sp):
    m = sp.synth.workload("c2-cant", DEV)
    return m, sp.synth.dense_vector(m.n_cols, torch.float32, 2, DEV)


@pytest.mark.parametrize("kind", KINDS)
def test_c2_cant_every_row(sp, oracle, c2, kind):
    """Config 2 (CSR-vector wave-reduce fp32 on cant): every row against the oracle bound, one-shot
    and through a plan; x = 1 must also agree (the reference's harness input, main.cu:41)."""
    m, x = c2
    assert m.n_rows == 62451 and abs(m.nnz / m.n_rows - 64) < 1
    y = torch.full((m.n_rows,), float("nan"), device=DEV)
    sp.spmv(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, y)
    check_rows(oracle, m, x, y, "c2 %s one-shot" % kind)
    p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, torch.float32)
    ones = torch.ones(m.n_cols, device=DEV)
    y2 = torch.full((m.n_rows,), float("nan"), device=DEV)
    p.execute(m.Ax, ones, y2)
    torch.cuda.synchronize()
    check_rows(oracle, m, ones, y2, "c2 %s plan, x = 1" % kind)
    # a plan is reusable and deterministic: the same execute twice gives the same bits
    y3 = torch.full((m.n_rows,), float("nan"), device=DEV)
    p.execute(m.Ax, ones, y3)
    torch.cuda.synchronize()
    p.destroy()
    assert torch.equal(y2, y3)


@pytest.fixture(scope="module")
def c4(sp):
    m = sp.synth.workload("c4-nlpkkt", DEV)
    return m, sp.synth.dense_vector(m.n_cols, torch.float64, 5, DEV)


@pytest.mark.parametrize("kind", KINDS)
def test_c4_fp64_i64_every_row(sp, oracle, c4, kind):
    """Config 4 (fp64 values, 64-bit row offsets, 8.4 M rows / 224 M nnz): EVERY row against the oracle."""
    m, x = c4
    assert m.Ap.dtype == torch.int64 and m.Ax.dtype == torch.float64 and m.n_rows > 8_000_000
    y = torch.full((m.n_rows,), float("nan"), dtype=torch.float64, device=DEV)
    sp.spmv(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, y)
    check_rows(oracle, m, x, y, "c4 %s" % kind)


@pytest.fixture(scope="module")
def c5(sp):
    m = sp.synth.workload("c5-rmat24", DEV)
    return m, sp.synth.dense_vector(m.n_cols, torch.float32, 5, DEV)


@pytest.fixture(scope="module")
def c5_truth(oracle, c5):
    """(y64, bound) of the whole C5 matrix, computed once for the tests below."""
    m, x = c5
    Ap, Aj, Ax = m.numpy()
    return parity_bound(oracle, Ap, Aj, Ax, x.cpu().numpy(), THREADS)


def _within(y, truth):
    yh = y.cpu().numpy()
    assert not np.any(np.isnan(yh))
    err = np.abs(yh.astype(np.float64) - truth[0])
    return np.nonzero(err > truth[1])[0]


@pytest.mark.parametrize("kind", KINDS)
def test_c5_rmat24_every_row_one_gpu(sp, c5, c5_truth, kind):
    """Config 5's matrix (R-MAT scale 24, edge factor 16, duplicates kept) on ONE GPU: every row."""
    m, x = c5
    assert m.n_rows == 1 << 24 and m.nnz == 1 << 28
    y = torch.full((m.n_rows,), float("nan"), device=DEV)
    sp.spmv(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, y)
    bad = _within(y, c5_truth)
    assert bad.size == 0, "%d rows outside the bound, first %s" % (bad.size, bad[:5])


@pytest.mark.parametrize("kind", KINDS)
def test_c5_rmat24_eight_row_blocks_on_one_gpu(sp, c5, c5_truth, kind):
    """Config 5's PATH without 8 GPUs (SURVEY §8(e) parity): the matrix is cut into 8 nnz-balanced
    row blocks exactly as the 8-GPU run cuts it, every block is executed in turn on the one GPU here
    through the library's own distributed object (all 8 blocks resident on device 0, no exchange to
    make), and the concatenated y must equal the 1-GPU y BIT FOR BIT for the row-local kinds
    (vector, light) and stay inside the bound for merge, whose tile boundaries move with the cut."""
    m, x = c5
    whole = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, torch.float32)
    y1 = torch.full((m.n_rows,), float("nan"), device=DEV)
    whole.execute(m.Ax, x, y1)
    torch.cuda.synchronize()
    d = sp.DistPlan.local(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, torch.float32, parts=8, devices=[0])
    cuts = d.cuts()
    assert len(cuts) == 9 and cuts[0] == 0 and cuts[-1] == m.n_rows
    nnz_of = [int(m.Ap[cuts[p + 1]].item()) - int(m.Ap[cuts[p]].item()) for p in range(8)]
    assert max(nnz_of) < 1.05 * m.nnz / 8 + 70000, "blocks are nnz-balanced: %s" % nnz_of
    y8 = torch.full((m.n_rows,), float("nan"), device=DEV)
    d.execute(m.Ax, x, y8)
    torch.cuda.synchronize()
    bad = _within(y8, c5_truth)
    assert bad.size == 0, "%d rows outside the bound, first %s" % (bad.size, bad[:5])
    if kind in ("vector", "light"):
        diff = torch.nonzero(y8 != y1).flatten()
        assert diff.numel() == 0, "row-local kind differs from the 1-GPU result in %d rows, first %s" % (
            diff.numel(), diff[:5].tolist())
    d.destroy()
    whole.destroy()
