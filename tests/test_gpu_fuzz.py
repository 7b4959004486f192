"""GPU suite, seeded fuzz: matrices of random size, row-length law, column pattern and type through every kind,
against the oracle bound (tests/test_gpu_parity.py states it).  The shapes are drawn to land on as many plan
decisions as possible — window / no window / several bands / swept band, 256- / 512- / 1 024-thread plans, equal-row
and weight-cut chunks, giant rows, merge's item walk and row-parallel runs, in-kernel and separate search — and the
plan each case got is printed on failure.  Deterministic: the seeds are the test ids."""
import numpy as np
import pytest
import torch

from conftest import parity_bound

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KINDS = ["vector", "merge", "light"]


def draw_matrix(seed):
    rng = np.random.default_rng(1000 + seed)
    if seed < 8:                                                # tiny and odd sizes
        n_rows = int(np.exp(rng.uniform(np.log(1), np.log(3000))))
    elif seed < 28:                                             # medium
        n_rows = int(np.exp(rng.uniform(np.log(20_000), np.log(1_000_000))))
    else:                                                       # big enough for the wide-workgroup / sweep / run plans
        n_rows = int(np.exp(rng.uniform(np.log(1_000_000), np.log(2_200_000))))
    n_cols = max(1, int(n_rows * np.exp(rng.uniform(np.log(0.3), np.log(3.0)))))
    if seed >= 8 and rng.random() < 0.7:
        n_cols = n_rows                                         # (square: the band of column - row is then narrow)
    law = rng.choice(["fixed", "uniform", "geometric", "powerlaw", "mostly_empty"])
    k = int(np.exp(rng.uniform(np.log(1), np.log(150 if seed < 8 else 90)))) if seed < 28 else int(rng.integers(8, 41))
    if law == "fixed":
        lens = np.full(n_rows, min(k, n_cols))
    elif law == "uniform":
        lens = rng.integers(0, k + 1, n_rows)
    elif law == "geometric":
        lens = rng.geometric(1.0 / (k + 1), n_rows) - 1
    elif law == "powerlaw":
        lens = np.minimum((rng.pareto(1.3, n_rows) * max(1, k // 8)).astype(np.int64), 200_000)
    else:
        lens = np.where(rng.random(n_rows) < 0.9, 0, rng.integers(1, k + 2, n_rows))
    if rng.random() < 0.3 and n_rows > 10:                      # a hub row or three (one possibly giant)
        for r in rng.integers(0, n_rows, 3):
            lens[r] = int(np.exp(rng.uniform(np.log(500), np.log(120_000))))
    lens = np.minimum(lens.astype(np.int64), 4 * n_cols + 7)      # (duplicates allowed, but keep it sane)
    while lens.sum() > (48_000_000 if seed >= 28 else 16_000_000):
        lens = lens // 2
    Ap = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(lens, out=Ap[1:])
    nnz = int(Ap[-1])
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), lens)
    pattern = rng.choice(["band", "wideband", "scatter", "bands3"])
    centre = rows * n_cols // max(n_rows, 1)
    if pattern == "band":
        hw = int(np.exp(rng.uniform(np.log(1), np.log(6000))))
        cols = centre + rng.integers(-hw, hw + 1, nnz)
    elif pattern == "wideband":
        hw = int(np.exp(rng.uniform(np.log(6000), np.log(70_000))))
        cols = centre + rng.integers(-hw, hw + 1, nnz)
    elif pattern == "bands3":
        d = max(2, int(round(n_cols ** (1 / 3))))
        cols = centre + rng.choice([-d * d, 0, d * d], nnz) + rng.choice([-d, 0, d], nnz) + rng.integers(-1, 2, nnz)
    else:
        cols = rng.integers(0, n_cols, nnz)
    cols = np.clip(cols, 0, n_cols - 1)
    if rng.random() < 0.7:                                       # sorted columns inside a row (most loaders), else as drawn
        order = np.lexsort((cols, rows))
        cols = cols[order]
    off = rng.choice([np.int32, np.int64])
    val = rng.choice([np.float32, np.float64])
    Ax = (rng.random(nnz) * 2 - 1).astype(val)
    x = (rng.random(n_cols) * 2 - 1).astype(val)
    desc = "seed %d: %d x %d, nnz %d, %s rows (k=%d), %s columns, %s/%s" % (
        seed, n_rows, n_cols, nnz, law, k, pattern, np.dtype(off).name, np.dtype(val).name)
    return Ap.astype(off), cols.astype(np.int32), Ax, x, n_cols, desc


@pytest.mark.parametrize("seed", range(36))
def test_fuzz_all_kinds_against_the_oracle(sp, oracle, seed):
    Ap, Aj, Ax, x, n_cols, desc = draw_matrix(seed)
    n_rows = len(Ap) - 1
    nnz = int(Ap[-1])
    y64, bound = parity_bound(oracle, Ap, Aj, Ax, x, 8)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    dAp, dAj, dAx, dx = d(Ap), d(Aj), d(Ax), d(x)
    rng = np.random.default_rng(seed)
    for kind in KINDS:
        p = sp.Plan(kind, n_rows, n_cols, nnz, dAp, dAj, dAx.dtype)
        info = p.info()
        y = torch.full((n_rows,), float("nan"), dtype=dAx.dtype, device=DEV)
        p.execute(dAx, dx, y)
        torch.cuda.synchronize()
        got = y.cpu().numpy().astype(np.float64)
        where = "%s | %s: %s" % (desc, kind, {k: info[k] for k in ("main_kernel", "lanes_per_row", "block_threads",
                                                                   "rows_per_chunk", "window_elems", "window_segments",
                                                                   "balanced_chunks", "grid_blocks", "n_kernels")})
        assert not np.isnan(got).any(), "a row was skipped: " + where
        bad = np.nonzero(np.abs(got - y64) > bound)[0]
        assert bad.size == 0, "rows outside the bound %s: %s" % (bad[:5], where)
        if seed % 3 == 0:                                        # y = alpha A x + beta y_old through the same plan
            alpha, beta = float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2))
            y_old = torch.from_numpy((rng.random(n_rows) - 0.5).astype(Ax.dtype)).to(DEV)
            y2 = y_old.clone()
            p.set_alpha_beta(alpha, beta)
            p.execute(dAx, dx, y2)
            torch.cuda.synchronize()
            want = alpha * y64 + beta * y_old.cpu().numpy().astype(np.float64)
            eps = 2.0 ** -23 if Ax.dtype == np.float32 else 2.0 ** -52
            tol = abs(alpha) * bound + 2 * eps * (np.abs(alpha * y64) + np.abs(beta * y_old.cpu().numpy().astype(np.float64))) + 1e-300
            bad = np.nonzero(np.abs(y2.cpu().numpy().astype(np.float64) - want) > tol)[0]
            assert bad.size == 0, "alpha/beta rows outside the bound %s: %s" % (bad[:5], where)
        p.destroy()
