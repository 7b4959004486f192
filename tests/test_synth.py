"""CPU suite, part 3: the seeded stand-ins have the shapes SURVEY.md §8(d) asks for."""
import numpy as np
import torch


def _check_csr(m):
    Ap, Aj, Ax = m.numpy()
    assert Ap[0] == 0 and Ap[-1] == m.nnz == len(Aj) == len(Ax)
    assert np.all(np.diff(Ap) >= 0)
    assert Aj.min() >= 0 and Aj.max() < m.n_cols
    return Ap, Aj, Ax


def test_banded_fixed(sp):
    m = sp.synth.banded_fixed(5000, 32, 256, seed=1)
    Ap, Aj, _ = _check_csr(m)
    assert np.all(np.diff(Ap) == 32)
    cols = Aj.reshape(5000, 32)
    assert np.all(np.diff(cols, axis=1) > 0)                        # sorted, distinct
    r = np.arange(5000)[:, None]
    assert np.all(np.abs(cols - r) <= 256 + 32)                     # inside the band
    m2 = sp.synth.banded_fixed(5000, 32, 256, seed=1)
    assert torch.equal(m.Aj, m2.Aj) and torch.equal(m.Ax, m2.Ax)    # seeded
    full = sp.synth.banded_fixed(2000, 32, None, seed=1)
    _check_csr(full)
    # one rank's row block of a taller matrix keeps global column ids
    blk = sp.synth.banded_fixed(1000, 32, 256, seed=3, row_offset=4000, n_cols=8000)
    _, Ajb, _ = _check_csr(blk)
    assert Ajb.min() >= 4000 - 256 - 32 and Ajb.max() < 8000


def test_banded_variable_and_rmat_and_stencil(sp):
    m = sp.synth.banded_variable(3000, 64, 16, 512, seed=2)
    Ap, Aj, _ = _check_csr(m)
    lens = np.diff(Ap)
    assert lens.min() >= 48 and lens.max() <= 80
    for r in (0, 17, 2999):
        c = Aj[Ap[r]:Ap[r + 1]]
        assert np.all(np.diff(c) > 0)
    g = sp.synth.rmat(12, 8, seed=5)
    Ap, Aj, _ = _check_csr(g)
    assert g.n_rows == 4096 and g.nnz == 8 * 4096
    lens = np.diff(Ap)
    assert lens.max() > 20 * lens.mean()                            # power-law skew
    w = sp.synth.rmat(12, seed=3, n=3000, nnz=20000, ones=True)
    _check_csr(w)
    assert w.n_rows == 3000 and torch.all(w.Ax == 1)
    s = sp.synth.stencil27(6, 5, 4, off_dtype=torch.int64, val_dtype=torch.float64)
    Ap, Aj, _ = _check_csr(s)
    assert s.Ap.dtype == torch.int64 and s.Ax.dtype == torch.float64
    assert np.diff(Ap).max() == 27 and np.diff(Ap).min() == 8


def test_algorithmic_bytes_formula(sp):
    """the S32 target is 1 124.1 MB at 2^22 rows."""
    n = 1 << 22
    b = n * 32 * 8 + (n + 1) * 4 + n * 4 + n * 4
    assert abs(b / 1e6 - 1124.1) < 0.1
    m = sp.synth.banded_fixed(1024, 32, 64)
    assert m.algorithmic_bytes() == 1024 * 32 * 8 + 1025 * 4 + 1024 * 4 + 1024 * 4
