"""GPU suite: the N > 1 schedule of mi355_spmv_dist_* on the ONE GPU of a test box.

The real RCCL refuses two ranks on one device, so until round 3 nothing of the multi-GPU exchange had ever run.
Here the library's RCCL table is bound to tests/cpp/libfakerccl.so (MI355_SPMV_RCCL_LIB) — an emulation that moves
the data with hipMemcpyAsync on the communication streams and orders them with events the way a collective orders
ranks — and a device may be listed several times (MI355_DIST_SHARED_DEVICE=1).  Every "GPU" is then its own set of
buffers, streams, events and communicator on device 0: what runs is the product's whole N > 1 path — remote copies
of Ap / Aj / Ax / x, the sub-block loop, the three exchanges (grouped in-place broadcasts, send / recv pairs,
all-gather in place or through the padded staging buffer with its pack / unpack kernels), the timed pick between
them — in LOCAL mode (one thread drives all ranks) and in RANK mode (one thread per rank here; one process per GPU
in bench.py).  Parity: SURVEY §8(e) — every GPU's y equals the one-GPU y bit for bit for vector / light, inside the
bound for merge.  What this cannot show is RCCL's own behaviour and the xGMI timings.
"""
import ctypes as C
import os
import threading

import numpy as np
import pytest
import torch

from conftest import ROOT, parity_bound

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FAKE = os.path.join(ROOT, "tests", "cpp", "libfakerccl.so")
EXCHANGES = ("bcast", "sendrecv", "allgather")


@pytest.fixture(scope="module", autouse=True)
def emulated_rccl(sp):
    if not os.path.exists(FAKE):
        pytest.fail("tests/cpp/libfakerccl.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    old = {k: os.environ.get(k) for k in ("MI355_SPMV_RCCL_LIB", "MI355_DIST_SHARED_DEVICE", "MI355_DIST_TRIALS")}
    os.environ["MI355_SPMV_RCCL_LIB"] = FAKE
    os.environ["MI355_DIST_SHARED_DEVICE"] = "1"
    os.environ["MI355_DIST_TRIALS"] = "2"
    sp.capi.lib().mi355_spmv_knobs_reload()
    yield C.CDLL(FAKE)
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    sp.capi.lib().mi355_spmv_knobs_reload()


def _matrices(sp):
    out = [("band-f32", sp.synth.banded_fixed(120_000, 32, 700, seed=31, device=DEV)),        # equal blocks
           ("rmat16-f32", sp.synth.rmat(16, 16, seed=32, device=DEV)),                        # uneven, weight-cut chunks
           ("stencil-f64-i64", sp.synth.stencil27(30, 30, 30, device=DEV)),
           ("tiny-f32", sp.synth.banded_fixed(40, 8, 6, seed=33, device=DEV))]                # most blocks empty
    return out


@pytest.fixture(scope="module")
def matrices(sp):
    return _matrices(sp)


def _one_gpu(sp, kind, m, x):
    p = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
    y = torch.full((m.n_rows,), float("nan"), dtype=m.Ax.dtype, device=DEV)
    p.execute(m.Ax, x, y)
    torch.cuda.synchronize()
    p.destroy()
    return y


def _device_y(sp, d, i, n, dtype):
    """GPU i's own full-length y (LOCAL mode) as a tensor."""
    ptr = sp.capi.lib().mi355_spmv_dist_device_y(d._h, i)
    if not ptr:
        return None
    t = torch.empty(n, dtype=dtype, device=DEV)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), C.c_size_t(n * t.element_size()), 3) == 0
    return t


def _check(sp, oracle, kind, m, x, y, y1, what):
    assert not torch.isnan(y).any(), what
    if kind == "merge":
        Ap, Aj, Ax = m.numpy()
        y64, bound = parity_bound(oracle, Ap, Aj, Ax, x.cpu().numpy())
        assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - y64) <= bound), what
    else:
        assert torch.equal(y, y1), what


@pytest.mark.parametrize("kind", ["vector", "merge", "light"])
@pytest.mark.parametrize("gpus,sub", [(2, 1), (4, 1), (3, 2), (8, 4)])
def test_local_mode_every_exchange_gives_every_gpu_the_one_gpu_y(sp, oracle, matrices, kind, gpus, sub):
    for name, m in matrices:
        x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 41, DEV)
        y1 = _one_gpu(sp, kind, m, x)
        d = sp.DistPlan.local(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype, devices=[0] * gpus, sub_blocks=sub)
        info = d.dist_info()
        assert info["world"] == gpus and info["sub_blocks"] == sub and info["auto_picked"] == 1, name
        assert info["exchange_name"] in EXCHANGES and all(info["trial_us"][e] > 0 for e in EXCHANGES), (name, info)
        for exch in ("picked",) + EXCHANGES:
            if exch != "picked":
                d.set_exchange(exch)
                assert d.dist_info()["exchange_name"] == exch
            y = torch.full((m.n_rows,), float("nan"), dtype=m.Ax.dtype, device=DEV)
            d.execute(m.Ax, x, y)
            torch.cuda.synchronize()
            what = "%s %s x%d/%d %s" % (name, kind, gpus, sub, exch)
            _check(sp, oracle, kind, m, x, y, y1, what + " home")
            for i in range(1, gpus):
                yi = _device_y(sp, d, i, m.n_rows, m.Ax.dtype)
                assert torch.equal(yi, y), what + " gpu %d" % i     # the very bits of the home GPU's y
        d.destroy()


def test_values_and_x_are_handed_over_on_every_call(sp, matrices):
    """The drop-in semantics of SpMV(kind, ...): Ax and x passed to execute are scattered / replicated first, so
    values rewritten in place (an iterative solver) reach the other GPUs; NULL = unchanged."""
    name, m = matrices[0]
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 42, DEV)
    d = sp.DistPlan.local("vector", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype, devices=[0, 0, 0], sub_blocks=2)
    y = torch.empty(m.n_rows, dtype=m.Ax.dtype, device=DEV)
    d.execute(m.Ax, x, y)
    Ax2 = m.Ax.clone()
    Ax2.mul_(-3.0)
    m2 = sp.synth.Csr(m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, Ax2, "scaled")
    x2 = x * 0.5
    y_ref = _one_gpu(sp, "vector", m2, x2)
    m.Ax.copy_(Ax2)                           # in place, same pointer
    x.copy_(x2)
    d.execute(m.Ax, x, y)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    y.fill_(float("nan"))
    d.execute(None, None, y)                  # unchanged since the last hand-over
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    d.destroy()


@pytest.mark.parametrize("which", [1, 0])
def test_exchange_calls_match_the_restated_schedule(sp, emulated_rccl, matrices, which):
    """What RCCL is called with, read back from the emulation's log, against dist.exchange_schedule (the same count /
    displacement logic in Python, which the CPU suite runs over gloo): roots, peers, counts, pointers, order — for the
    grouped in-place broadcasts, the send / recv pairs, and the all-gather (padded through staging on the R-MAT's
    uneven blocks, in place on the band's equal ones)."""
    name, m = matrices[which]
    gpus, sub = (3, 2) if which == 1 else (4, 1)
    vb = m.Ax.element_size()
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 43, DEV)
    d = sp.DistPlan.local("vector", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype, devices=[0] * gpus, sub_blocks=sub)
    cuts = d.cuts()
    y = torch.empty(m.n_rows, dtype=m.Ax.dtype, device=DEV)
    lib = sp.capi.lib()
    base = {0: y.data_ptr()}
    for i in range(1, gpus):
        base[i] = lib.mi355_spmv_dist_device_y(d._h, i)
    for mode in EXCHANGES:
        d.set_exchange(mode)
        torch.cuda.synchronize()
        emulated_rccl.fake_rccl_log_clear()
        d.execute(m.Ax, x, y)
        torch.cuda.synchronize()
        rec = (C.c_longlong * 6)()
        calls = []
        for i in range(emulated_rccl.fake_rccl_log_size()):
            assert emulated_rccl.fake_rccl_log_get(i, rec) == 0
            calls.append(tuple(rec))
        expect, padded = [], []
        for s in range(sub):
            for i in range(gpus):
                for op in sp.dist.exchange_schedule(cuts, gpus, sub, s, mode, i):
                    if op[0] == "bcast":
                        at = base[i] + op[2] * vb
                        expect.append((1, i, op[1], op[3], at, at))
                    elif op[0] == "send":
                        expect.append((2, i, op[1], op[3], base[i] + op[2] * vb, 0))
                    elif op[0] == "recv":
                        expect.append((3, i, op[1], op[3], 0, base[i] + op[2] * vb))
                    elif op[0] == "allgather_in_place":
                        at = base[i] + op[1] * vb
                        expect.append((4, i, -1, op[2], at + i * op[2] * vb, at))
                    elif op[0] == "allgather_padded":
                        expect.append((4, i, -1, op[1], None, None))
                        padded.append(len(expect) - 1)
        assert len(calls) == len(expect), (mode, len(calls), len(expect))
        for k, (got, want) in enumerate(zip(calls, expect)):
            if k in padded:                      # staging buffer: its address is the library's, its layout is not
                assert got[:4] == want[:4] and got[4] == got[5] + got[1] * got[3] * vb, (mode, k, got)
            else:
                assert got == want, (mode, k, got, want)
        if mode == "allgather":
            assert (d.dist_info()["allgather_in_place"] == 1) == (not padded)
            assert which == 0 or len(padded) == sub * gpus          # R-MAT: nnz-balanced blocks differ in rows
    d.destroy()


def test_skip_exchange_and_exchange_only_add_up_to_a_step(sp, matrices):
    """bench.py's compute-only and exchange-only legs: SKIP_EXCHANGE leaves every GPU with ITS rows only,
    EXCHANGE_ONLY then completes y everywhere."""
    name, m = matrices[0]
    gpus, sub = 4, 2
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 44, DEV)
    y1 = _one_gpu(sp, "light", m, x)
    d = sp.DistPlan.local("light", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype, devices=[0] * gpus, sub_blocks=sub)
    cuts = d.cuts()
    for exch in EXCHANGES:
        d.set_exchange(exch)
        y = torch.full((m.n_rows,), float("nan"), dtype=m.Ax.dtype, device=DEV)
        d.execute(m.Ax, x, y, flags=sp.capi.EXEC_SKIP_EXCHANGE)
        torch.cuda.synchronize()
        own = torch.zeros(m.n_rows, dtype=torch.bool, device=DEV)
        for s in range(sub):
            own[cuts[s]:cuts[s + 1]] = True                               # GPU 0 owns blocks 0 .. sub-1
        assert torch.equal(y[own], y1[own]) and torch.isnan(y[~own]).all(), exch
        d.execute(None, None, y, flags=sp.capi.EXEC_EXCHANGE_ONLY)
        torch.cuda.synchronize()
        assert torch.equal(y, y1), exch
        for i in range(1, gpus):
            assert torch.equal(_device_y(sp, d, i, m.n_rows, m.Ax.dtype), y1), exch
    d.destroy()


def test_structure_fingerprint_sees_a_matrix_rewritten_in_place(sp, matrices):
    """A dist handle holds copies of the structure (advisor r2): the C++ kinds ask structure_changed on every call."""
    name, m = matrices[0]
    Ap, Aj = m.Ap.clone(), m.Aj.clone()
    d = sp.DistPlan.local("vector", m.n_rows, m.n_cols, m.nnz, Ap, Aj, m.Ax.dtype, devices=[0, 0], sub_blocks=2)
    assert not d.structure_changed(Ap, Aj)
    stride = max(1, m.nnz // 65536)
    Aj[7 * stride] = (int(Aj[7 * stride].item()) + 1) % m.n_cols         # a sampled column
    assert d.structure_changed(Ap, Aj)
    Aj.copy_(m.Aj)
    assert not d.structure_changed(Ap, Aj)
    # move one nonzero from row 10 to row 11 (same nnz, same sizes, same pointers)
    Ap[11] += -1
    assert d.structure_changed(Ap, Aj)
    d.destroy()


@pytest.mark.parametrize("kind", ["vector", "merge", "light"])
@pytest.mark.parametrize("world,sub", [(2, 1), (3, 4)])
def test_rank_mode_one_thread_per_rank(sp, oracle, matrices, kind, world, sub):
    """RANK mode as bench.py drives it (one process per GPU there, one thread per rank here): ONE matrix cut on the
    chunk boundaries of its one-GPU plan, every rank holds its view only, unique id from rank 0, each rank its own
    stream and its own full-length y; the timed pick agrees on one exchange over all ranks (ncclAllReduce)."""
    for name, m in matrices[:3]:
        x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 45, DEV)
        y1 = _one_gpu(sp, kind, m, x)
        whole = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
        shape = whole.shape()
        rows, chunks, nnzs = whole.partition(world * sub)
        whole.destroy()
        uid = sp.DistPlan.unique_id()
        ys = [torch.full((m.n_rows,), float("nan"), dtype=m.Ax.dtype, device=DEV) for _ in range(world)]
        picked, errors = [None] * world, []

        def run(rank):
            try:
                r0, r1 = rows[rank * sub], rows[(rank + 1) * sub]
                Ap_l, Aj_l, Ax_l, _ = sp.dist.block_view(m.Ap, m.Aj, m.Ax, r0, r1)
                d = sp.DistPlan.rank(kind, rank, world, uid, sub, rows, chunks, nnzs, shape, m.n_cols, r1 - r0,
                                     int(Ap_l[-1].item()) if r1 > r0 else 0, Ap_l, Aj_l, m.Ax.dtype)
                picked[rank] = d.dist_info()
                st = torch.cuda.Stream()
                for exch in (None,) + EXCHANGES:
                    if exch:
                        d.set_exchange(exch)
                    ys[rank].fill_(float("nan"))
                    torch.cuda.synchronize()
                    d.execute(Ax_l, x, ys[rank], stream=st)
                    st.synchronize()
                    what = "%s %s rank %d/%d sub %d %s" % (name, kind, rank, world, sub, exch)
                    _check(sp, oracle, kind, m, x, ys[rank], y1, what)
                d.destroy()
            except Exception as e:            # noqa: BLE001 - reported by the main thread
                errors.append("rank %d: %r" % (rank, e))

        ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in ts), "a rank is stuck"
        assert not errors, errors
        assert len({p["exchange_name"] for p in picked}) == 1, picked         # all ranks agreed
        assert all(p["trial_us"] == picked[0]["trial_us"] for p in picked), picked
        if kind != "merge":
            for r in range(1, world):
                assert torch.equal(ys[r], ys[0])


def test_allgather_in_place_only_when_blocks_are_equal_and_adjacent(sp, matrices):
    name, m = matrices[0]
    d = sp.DistPlan.local("vector", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype, devices=[0, 0], sub_blocks=1)
    cuts = d.cuts()
    equal = cuts[1] - cuts[0] == cuts[2] - cuts[1]
    assert d.dist_info()["allgather_in_place"] == (1 if equal else 0)
    d.destroy()
    name, m = matrices[1]                      # R-MAT: nnz-balanced blocks have different row counts
    d = sp.DistPlan.local("vector", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype, devices=[0, 0, 0], sub_blocks=1)
    info = d.dist_info()
    assert info["allgather_in_place"] == 0 and info["staging_bytes"] >= 3 * info["max_block_rows"] * 4
    d.destroy()
