"""GPU suite: row-block plans and the multi-GPU object (mi355_spmv_dist_*) on the ONE GPU of the test box.

SURVEY §8(e) parity: "concatenated y must equal the 1-GPU y bit-for-bit when the per-row algorithm is
row-local (CSR-vector, LightSpMV); merge-path tile boundaries move with the partition ⇒ same bound as (c)".
What makes that hold here (include/mi355_spmv.h, "row-block plans"): blocks are cut on chunk boundaries of
the whole matrix's plan, inherit its launch shape, and are 16-byte-aligned views that keep every row's
position modulo 4.  Everything except the RCCL calls themselves runs with one device: the partition, the
block plans, the in-place y displacements, the sub-block loop.
"""
import numpy as np
import pytest
import torch

from conftest import parity_bound

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cases(sp):
    """(name, csr) — uniform chunks, weight-cut chunks, a giant row, fp64 / 64-bit offsets."""
    out = []
    out.append(("band-f32", sp.synth.banded_fixed(150_001, 32, 900, seed=3, device=DEV)))
    out.append(("rmat17-f32", sp.synth.rmat(17, 16, seed=9, device=DEV)))
    m = sp.synth.rmat(16, 8, seed=10, device=DEV, val_dtype=torch.float64, off_dtype=torch.int64)
    out.append(("rmat16-f64-i64", m))
    out.append(("stencil-f64-i64", sp.synth.stencil27(40, 40, 40, device=DEV)))
    # a banded matrix with one row of 300 000 nonzeros (giant-row slices) and a few of 5 000 (whole-workgroup rows)
    rng = np.random.RandomState(12)
    lens = np.full(60_000, 24, dtype=np.int64)
    lens[31_111] = 300_000
    lens[[5, 20_000, 59_999]] = 5_000
    Ap = np.zeros(lens.size + 1, dtype=np.int64)
    np.cumsum(lens, out=Ap[1:])
    Aj = rng.randint(0, 60_000, size=int(Ap[-1])).astype(np.int32)
    Ax = (rng.rand(int(Ap[-1])) * 2 - 1).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    out.append(("hub-f32", sp.synth.Csr(60_000, 60_000, int(Ap[-1]), t(Ap.astype(np.int32)), t(Aj), t(Ax), "hub")))
    return out


@pytest.fixture(scope="module")
def cases(sp):
    return _cases(sp)


def _whole(sp, kind, m, x):
    p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
    y = torch.full((m.n_rows,), float("nan"), dtype=m.Ax.dtype, device=DEV)
    p.execute(m.Ax, x, y)
    torch.cuda.synchronize()
    return p, y


@pytest.mark.parametrize("kind", ["vector", "light"])
@pytest.mark.parametrize("parts", [3, 8])
def test_block_plans_reproduce_the_whole_plan_bit_for_bit(sp, cases, kind, parts):
    for name, m in cases:
        x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 21, DEV)
        whole, y1 = _whole(sp, kind, m, x)
        shape = whole.shape()
        rows, chunks, nnzs = whole.partition(parts)
        assert rows[0] == 0 and rows[-1] == m.n_rows and all(a <= b for a, b in zip(rows, rows[1:])), name
        assert all(r % 4 == 0 or r == m.n_rows for r in rows), name
        assert nnzs == [int(m.Ap[r].item()) for r in rows], name
        y = torch.full((m.n_rows,), float("nan"), dtype=m.Ax.dtype, device=DEV)
        for b in range(parts):
            a, j, v, lo = sp.dist.block_view(m.Ap, m.Aj, m.Ax, rows[b], rows[b + 1])
            if rows[b + 1] == rows[b]:
                continue
            blk = sp.Plan.block(kind, shape, rows[b], chunks[b], chunks[b + 1] - chunks[b], nnzs[b],
                                rows[b + 1] - rows[b], m.n_cols, int(a[-1].item()), a, j, m.Ax.dtype)
            info = blk.info()
            assert info["lanes_per_row"] == shape.lanes_per_row and info["block_threads"] == shape.block_threads, name
            assert info["balanced_chunks"] == shape.balanced_chunks, name
            blk.execute(v, x, y[rows[b]:rows[b + 1]])
            torch.cuda.synchronize()
            blk.destroy()
        whole.destroy()
        diff = torch.nonzero(~((y == y1) | (torch.isnan(y) & torch.isnan(y1)))).flatten()
        assert not torch.isnan(y).any(), name
        assert diff.numel() == 0, "%s: %d rows differ from the whole plan's, first %s" % (name, diff.numel(), diff[:5].tolist())


@pytest.mark.parametrize("kind", ["vector", "merge", "light"])
def test_partition_is_nnz_balanced_and_matches_the_restated_rule(sp, cases, kind):
    """mi355_spmv_plan_partition against dist.partition_rows (the same rule in torch, which the CPU suite
    runs under gloo): identical cuts for plans with equal-row chunks and for merge (units of 4 rows)."""
    for name, m in cases:
        p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
        info = p.info()
        rows, chunks, nnzs = p.partition(4)
        p.destroy()
        heaviest_row = int((m.Ap[1:] - m.Ap[:-1]).max().item())
        per = [nnzs[i + 1] - nnzs[i] for i in range(4)]
        unit_rows = 4 if kind == "merge" else max(info["rows_per_chunk"], info["rows_cap"])
        slack = heaviest_row + unit_rows * max(1, m.nnz // max(m.n_rows, 1)) * 4
        assert max(per) <= m.nnz / 4 + slack, (name, per)
        if kind == "merge":
            assert rows == sp.dist.partition_rows(m.Ap, 4, unit=4), name
        elif not info["balanced_chunks"]:
            assert rows == sp.dist.partition_rows(m.Ap, 4, unit=info["rows_per_chunk"]), name


@pytest.mark.parametrize("kind", ["vector", "merge", "light"])
def test_dist_local_eight_blocks_one_device(sp, oracle, cases, kind):
    """The multi-GPU object with every block on device 0: y must be the 1-GPU y bit for bit (vector, light) /
    inside the bound (merge); also through alpha / beta."""
    for name, m in cases:
        x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 22, DEV)
        whole, y1 = _whole(sp, kind, m, x)
        whole.destroy()
        d = sp.DistPlan.local(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype, parts=8, devices=[0])
        cuts = d.cuts()
        assert len(cuts) == 9 and cuts[0] == 0 and cuts[-1] == m.n_rows
        y = torch.full((m.n_rows,), float("nan"), dtype=m.Ax.dtype, device=DEV)
        d.execute(m.Ax, x, y)
        torch.cuda.synchronize()
        assert not torch.isnan(y).any(), name
        if kind == "merge":
            Ap, Aj, Ax = m.numpy()
            y64, bound = parity_bound(oracle, Ap, Aj, Ax, x.cpu().numpy())
            assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - y64) <= bound), name
        else:
            assert torch.equal(y, y1), name
            d.set_alpha_beta(0.5, -2.0)
            y0 = sp.synth.dense_vector(m.n_rows, m.Ax.dtype, 23, DEV)
            ya = y0.clone()
            d.execute(m.Ax, x, ya)
            p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
            p.set_alpha_beta(0.5, -2.0)
            yb = y0.clone()
            p.execute(m.Ax, x, yb)
            torch.cuda.synchronize()
            p.destroy()
            assert torch.equal(ya, yb), name
        d.destroy()


@pytest.mark.parametrize("kind", ["vector", "light"])
def test_dist_rank_mode_world_one_with_sub_blocks(sp, cases, kind):
    """RANK mode as bench.py drives it, world = 1: the rank's slice is the whole matrix, cut into 4
    sub-blocks on the whole plan's chunk boundaries; no communicator is made."""
    name, m = cases[1]
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 24, DEV)
    whole, y1 = _whole(sp, kind, m, x)
    shape = whole.shape()
    rows, chunks, nnzs = whole.partition(4)
    whole.destroy()
    d = sp.DistPlan.rank(kind, 0, 1, None, 4, rows, chunks, nnzs, shape, m.n_cols, m.n_rows, m.nnz, m.Ap, m.Aj,
                         m.Ax.dtype)
    y = torch.full((m.n_rows,), float("nan"), dtype=m.Ax.dtype, device=DEV)
    d.execute(m.Ax, x, y)
    torch.cuda.synchronize()
    d.destroy()
    assert torch.equal(y, y1)


def test_dist_execute_is_capturable_and_cheap_with_one_block(sp):
    """N = 1, one block: execute is the plain plan execute on the caller's stream (no events, no RCCL)."""
    m = sp.synth.banded_fixed(1 << 16, 32, 512, seed=5, device=DEV)
    x = sp.synth.dense_vector(m.n_cols, torch.float32, 5, DEV)
    d = sp.DistPlan.local("vector", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, torch.float32, parts=1, devices=[0])
    y = torch.full((m.n_rows,), float("nan"), device=DEV)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d.execute(m.Ax, x, y, stream=s)                # warm-up outside the capture
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        y.fill_(float("nan"))
        with torch.cuda.graph(g, stream=s):
            d.execute(m.Ax, x, y, stream=s)
        g.replay()
    torch.cuda.synchronize()
    p, y1 = _whole(sp, "vector", m, x)
    p.destroy()
    d.destroy()
    assert torch.equal(y, y1)


@pytest.mark.parametrize("kind", ["vector", "light"])
def test_block_that_ends_inside_the_arrays_last_partial_group(sp, kind):
    """a block that is not the last one reads the tail of its last row in whole 16-byte groups — but never
    past the END of the whole arrays.  Here the matrix ends with rows that hold 2 nonzeros in all and nnz % 4 == 1 + 2:
    the middle block's rounded-up view would reach 1-3 elements past Aj / Ax.  Aj / Ax are allocated EXACTLY (views of a
    larger buffer poisoned behind the end would show a stray read as NaN / a wild column); results bit for bit the
    whole plan's."""
    m0 = sp.synth.banded_fixed(8192, 32, 300, seed=77, device=DEV)
    probe = sp.Plan(kind, m0.n_rows, m0.n_cols, m0.nnz, m0.Ap, m0.Aj, m0.Ax.dtype)
    rpc = probe.info()["rows_per_chunk"]
    probe.destroy()
    assert rpc > 0 and 8192 % rpc == 0
    # rows [0, 8192): 32 per row, minus 3 nonzeros of the last one -> A = 8192 * 32 - 3 (A % 4 == 1); then rpc rows with
    # two 1-nonzero rows among them -> nnz = A + 2
    n = 8192 + rpc
    lens = np.full(n, 0, dtype=np.int64)
    lens[:8192] = 32
    lens[8191] = 29
    lens[8192 + 1] = 1
    lens[n - 1] = 1
    Ap = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=Ap[1:])
    nnz = int(Ap[-1])
    assert (nnz - 2) % 4 == 1
    rng = np.random.RandomState(5)
    rows = np.repeat(np.arange(n), lens)
    Aj = np.clip(rows + rng.randint(-200, 201, size=nnz), 0, n - 1).astype(np.int32)
    Ax = (rng.rand(nnz) * 2 - 1).astype(np.float32)
    # exact-size device arrays carved out of poisoned buffers: element nnz .. nnz + 7 hold a wild column / NaN
    bufj = torch.full((nnz + 8,), 2 ** 30, dtype=torch.int32, device=DEV)
    bufx = torch.full((nnz + 8,), float("nan"), dtype=torch.float32, device=DEV)
    bufj[:nnz] = torch.from_numpy(Aj).to(DEV)
    bufx[:nnz] = torch.from_numpy(Ax).to(DEV)
    dAj, dAx = bufj[:nnz], bufx[:nnz]
    dAp = torch.from_numpy(Ap.astype(np.int32)).to(DEV)
    x = sp.synth.dense_vector(n, torch.float32, 6, DEV)
    whole = sp.Plan(kind, n, n, nnz, dAp, dAj, torch.float32)
    y1 = torch.full((n,), float("nan"), device=DEV)
    whole.execute(dAx, x, y1)
    torch.cuda.synchronize()
    shape = whole.shape()
    assert whole.info()["rows_per_chunk"] == rpc and not whole.info()["balanced_chunks"]
    whole.destroy()
    cuts = [0, 4096, 8192, n]
    y = torch.full((n,), float("nan"), device=DEV)
    for b in range(3):
        r0, r1 = cuts[b], cuts[b + 1]
        a, j, v, lo = sp.dist.block_view(dAp, dAj, dAx, r0, r1)
        blk = sp.Plan.block(kind, shape, r0, r0 // rpc, (r1 - r0 + rpc - 1) // rpc, int(Ap[r0]), r1 - r0, n, int(a[-1].item()),
                            a, j, torch.float32)
        blk.execute(v, x, y[r0:r1])
        torch.cuda.synchronize()
        blk.destroy()
    assert not torch.isnan(y).any()
    assert torch.equal(y, y1)
