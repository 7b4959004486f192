"""Seeded synthetic CSR matrices (torch tensors, any device).

SuiteSparse files are not available offline, so the BASELINE configs are replaced
by stand-ins with the same shape statistics; every matrix
built here is flagged synthetic.  CSR conventions are the reference loader's
(include/load.hpp:420-474): Ap has n_rows+1 entries, duplicates are kept, and
inside a row entries stay in generation order (R-MAT) — columns are sorted only
where the generator itself emits them sorted (banded matrices).

Generators are chunked so that peak temporary memory stays a small multiple of
the result.
"""
from dataclasses import dataclass, field

import torch


@dataclass
class Csr:
    n_rows: int
    n_cols: int
    nnz: int
    Ap: torch.Tensor  # int32 or int64, n_rows + 1
    Aj: torch.Tensor  # int32, nnz
    Ax: torch.Tensor  # float32 or float64, nnz
    name: str = ""
    meta: dict = field(default_factory=dict)

    def to(self, device):
        return Csr(self.n_rows, self.n_cols, self.nnz, self.Ap.to(device), self.Aj.to(device),
                   self.Ax.to(device), self.name, dict(self.meta))

    def numpy(self):
        return self.Ap.cpu().numpy(), self.Aj.cpu().numpy(), self.Ax.cpu().numpy()

    def algorithmic_bytes(self):
        """Compulsory HBM traffic of one SpMV:
        nnz*(|val|+4) + (n_rows+1)*|off| + n_rows*|val| (y) + n_cols*|val| (x)."""
        v = self.Ax.element_size()
        o = self.Ap.element_size()
        return self.nnz * (v + 4) + (self.n_rows + 1) * o + self.n_rows * v + self.n_cols * v


def _gen(seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def _values(nnz, dtype, g, device, ones=False):
    if ones:
        return torch.ones(nnz, dtype=dtype, device=device)
    return (torch.rand(nnz, generator=g, device=device, dtype=torch.float32) * 2 - 1).to(dtype)


def dense_vector(n, dtype, seed, device, ones=False):
    """x: ones (main.cu:41, parity runs) or U(-1,1) (bandwidth runs)."""
    if ones:
        return torch.ones(n, dtype=dtype, device=device)
    return (torch.rand(n, generator=_gen(seed + 7919, device), device=device, dtype=torch.float32) * 2 - 1).to(dtype)


def banded_fixed(n, per_row=32, half_width=4096, seed=1, device="cpu", val_dtype=torch.float32,
                 off_dtype=torch.int32, row_offset=0, n_cols=None, name="S32-band"):
    """Exactly `per_row` nonzeros in every row; columns sorted and distinct inside the
    window [r-w, r+w] clipped to [0, n_cols): column_i = lo + i + floor(u_(i) * (W - per_row + 1))
    with u_(0) <= ... sorted uniforms (a sorted draw of `per_row` distinct columns).
    half_width=None -> window = all columns (S32-rand, the gather-stress companion).
    row_offset / n_cols: build rows [row_offset, row_offset+n) of a taller matrix whose
    column space is n_cols (used to make one rank's row block of a sharded matrix)."""
    n_cols = n_cols if n_cols is not None else n
    g = _gen(seed, device)
    Aj = torch.empty(n * per_row, dtype=torch.int32, device=device)
    chunk = 1 << 20
    ar = torch.arange(per_row, device=device, dtype=torch.int64)
    for r0 in range(0, n, chunk):
        r1 = min(n, r0 + chunk)
        rows = torch.arange(r0, r1, device=device, dtype=torch.int64) + row_offset
        if half_width is None:
            lo = torch.zeros_like(rows)
            hi = torch.full_like(rows, n_cols - 1)
        else:
            lo = (rows - half_width).clamp_(min=0)
            hi = (rows + half_width).clamp_(max=n_cols - 1)
        span = (hi - lo + 1 - per_row + 1).clamp_(min=1)  # slack positions
        u = torch.rand((r1 - r0, per_row), generator=g, device=device, dtype=torch.float64)
        u, _ = torch.sort(u, dim=1)
        cols = lo[:, None] + ar[None, :] + torch.floor(u * span[:, None].to(torch.float64)).to(torch.int64)
        cols.clamp_(max=n_cols - 1)
        Aj[r0 * per_row:r1 * per_row] = cols.reshape(-1).to(torch.int32)
        del u, cols
    nnz = n * per_row
    Ap = (torch.arange(n + 1, device=device, dtype=torch.int64) * per_row).to(off_dtype)
    Ax = _values(nnz, val_dtype, g, device)
    return Csr(n, n_cols, nnz, Ap, Aj, Ax, name,
               {"synthetic": True, "per_row": per_row, "half_width": half_width, "seed": seed})


def _csr_from_lengths(lens, off_dtype):
    Ap = torch.zeros(lens.numel() + 1, dtype=torch.int64, device=lens.device)
    torch.cumsum(lens, 0, out=Ap[1:])
    return Ap.to(off_dtype)


def banded_variable(n, mean_len=64, jitter=16, half_width=2048, seed=2, device="cpu",
                    val_dtype=torch.float32, off_dtype=torch.int32, name="C2-cant-standin"):
    """FEM-like stand-in for cant.mtx (62 451 rows, ~64 nnz/row): row lengths uniform in
    [mean-jitter, mean+jitter], columns sorted and distinct inside a band."""
    g = _gen(seed, device)
    lens = torch.randint(mean_len - jitter, mean_len + jitter + 1, (n,), generator=g, device=device,
                         dtype=torch.int64)
    Ap64 = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=Ap64[1:])
    nnz = int(Ap64[-1].item())
    rows = torch.repeat_interleave(torch.arange(n, device=device, dtype=torch.int64), lens)
    pos = torch.arange(nnz, device=device, dtype=torch.int64) - Ap64[rows]  # rank inside the row
    lo = (rows - half_width).clamp_(min=0)
    hi = (rows + half_width).clamp_(max=n - 1)
    span = (hi - lo + 1 - lens[rows] + 1).clamp_(min=1)
    u = torch.rand(nnz, generator=g, device=device, dtype=torch.float64)
    # sort u inside each row: sort by value, then stable sort by row
    o1 = torch.argsort(u)
    o2 = torch.argsort(rows[o1], stable=True)
    u_sorted = u[o1][o2]
    cols = lo + pos + torch.floor(u_sorted * span.to(torch.float64)).to(torch.int64)
    cols.clamp_(max=n - 1)
    Ax = _values(nnz, val_dtype, g, device)
    return Csr(n, n, nnz, Ap64.to(off_dtype), cols.to(torch.int32), Ax, name,
               {"synthetic": True, "mean_len": mean_len, "seed": seed})


def rmat(scale, edge_factor=16, abcd=(0.57, 0.19, 0.19, 0.05), seed=5, device="cpu",
         val_dtype=torch.float32, off_dtype=torch.int32, n=None, nnz=None, ones=False, name="rmat"):
    """R-MAT (Chakrabarti et al.) edge list -> CSR with the reference loader's semantics:
    duplicates kept, counting sort on row only, generation order kept inside a row
    (load.hpp:443-471).  n: fold the 2^scale ids onto [0, n) (stand-in for web-Google's
    916 428 rows); nnz: exact edge count (default edge_factor * 2^scale)."""
    g = _gen(seed, device)
    N = 1 << scale
    E = nnz if nnz is not None else edge_factor * N
    a, b, c, _ = abcd
    rows = torch.zeros(E, dtype=torch.int64, device=device)
    cols = torch.zeros(E, dtype=torch.int64, device=device)
    for _bit in range(scale):
        r = torch.rand(E, generator=g, device=device, dtype=torch.float32)
        row_bit = (r >= a + b)
        col_bit = ((r >= a) & (r < a + b)) | (r >= a + b + c)
        rows.mul_(2).add_(row_bit)
        cols.mul_(2).add_(col_bit)
        del r, row_bit, col_bit
    n_out = n if n is not None else N
    if n is not None:
        rows.remainder_(n)
        cols.remainder_(n)
    order = torch.argsort(rows, stable=True)
    lens = torch.bincount(rows, minlength=n_out)
    Ap = _csr_from_lengths(lens, off_dtype)
    Aj = cols[order].to(torch.int32)
    del rows, cols, order
    Ax = _values(E, val_dtype, g, device, ones=ones)
    return Csr(n_out, n_out, E, Ap, Aj, Ax, name, {"synthetic": True, "scale": scale, "seed": seed})


def stencil27(nx, ny, nz, seed=4, device="cpu", val_dtype=torch.float64, off_dtype=torch.int64,
              name="C4-nlpkkt-standin"):
    """3-D 27-point-stencil bands (stand-in for nlpkkt160: 8 345 600 rows, ~27.5 nnz/row,
    fp64, 64-bit Ap): row r=(i,j,k) couples to the neighbours inside the nx*ny*nz box."""
    n = nx * ny * nz
    g = _gen(seed, device)
    r = torch.arange(n, device=device, dtype=torch.int64)
    i = r % nx
    j = (r // nx) % ny
    k = r // (nx * ny)
    lens = torch.zeros(n, dtype=torch.int64, device=device)
    offs = [(di, dj, dk) for dk in (-1, 0, 1) for dj in (-1, 0, 1) for di in (-1, 0, 1)]
    valids = []
    for (di, dj, dk) in offs:
        v = (i + di >= 0) & (i + di < nx) & (j + dj >= 0) & (j + dj < ny) & (k + dk >= 0) & (k + dk < nz)
        valids.append(v)
        lens += v
    Ap64 = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=Ap64[1:])
    nnz = int(Ap64[-1].item())
    Aj = torch.empty(nnz, dtype=torch.int32, device=device)
    slot = Ap64[:-1].clone()
    for (di, dj, dk), v in zip(offs, valids):
        col = r + di + dj * nx + dk * nx * ny
        Aj[slot[v]] = col[v].to(torch.int32)
        slot += v
    Ax = _values(nnz, val_dtype, g, device)
    return Csr(n, n, nnz, Ap64.to(off_dtype), Aj, Ax, name, {"synthetic": True, "box": (nx, ny, nz), "seed": seed})


# ---- named workloads (BASELINE.json configs and the north-star target) ------------------

def workload(name, device="cpu", scale_down=1):
    """scale_down > 1 shrinks the row count (tests); 1 = BASELINE size."""
    s = scale_down
    if name == "s32-band":      # north-star target: 2^22 rows, 32 nnz/row, w = 4096, fp32
        return banded_fixed((1 << 22) // s, 32, 4096, 1, device, name="S32-band")
    if name == "s32-rand":      # gather-stress companion
        return banded_fixed((1 << 22) // s, 32, None, 1, device, name="S32-rand")
    if name == "c2-cant":       # 62 451 rows, ~4.0 M nnz
        return banded_variable(62451 // s, 64, 16, 2048, 2, device)
    if name == "c3-webgoogle":  # 916 428 rows, 5 105 039 nnz, pattern (values 1.0)
        return rmat(20 if s == 1 else 14, seed=3, device=device, n=916428 // s, nnz=5105039 // s,
                    ones=True, name="C3-webGoogle-standin")
    if name == "c4-nlpkkt":     # 8 345 600 rows (203^3 = 8 365 427 here), fp64, i64 offsets
        d = max(4, round(203 / s ** (1 / 3)))
        return stencil27(d, d, d, 4, device)
    if name == "c5-rmat24":     # scale 24, edge factor 16
        sc = 24 if s == 1 else max(10, 24 - (s.bit_length() - 1))
        return rmat(sc, 16, seed=5, device=device, name="C5-rmat%d" % sc)
    raise ValueError("unknown workload %r" % name)
