"""CPU suite, part 4: the N>1 path (row partition, shard, allgatherv) with gloo, world_size 2.
The per-rank SpMV here is the ORACLE (test infrastructure standing in for the GPU
kernel, which cannot run in this container); tests/test_gpu_host.py and
tests/test_gpu_configs.py run the same flow with the HIP path (mi355_spmv_dist_*)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, random_csr


def test_partition_rows_balances_nnz(sp):
    rng = np.random.RandomState(0)
    Ap, _, _ = random_csr(rng, 1000, 50, 20, long_row=30000)
    Ap_t = torch.from_numpy(Ap)
    for parts in (1, 2, 3, 8):
        cuts = sp.dist.partition_rows(Ap_t, parts)
        assert cuts[0] == 0 and cuts[-1] == 1000 and len(cuts) == parts + 1
        assert all(a <= b for a, b in zip(cuts, cuts[1:]))
    cuts = sp.dist.partition_rows(Ap_t, 2)
    left = int(Ap[cuts[1]])
    assert abs(left - Ap[-1] / 2) <= 30000                          # within one (long) row of the target
    assert sp.dist.partition_rows(Ap_t, 4, balance="rows") == [0, 250, 500, 750, 1000]


def test_partition_on_unit_boundaries(sp):
    """The library's cut rule (mi355_spmv_plan_partition), restated in dist.partition_rows: cuts fall on
    chunk boundaries only, each is the first boundary at or after its nonzero target."""
    rng = np.random.RandomState(3)
    Ap, _, _ = random_csr(rng, 1003, 50, 20, long_row=9000)
    Ap_t = torch.from_numpy(Ap)
    for unit in (4, 64, 1000):
        for parts in (2, 3, 8):
            cuts = sp.dist.partition_rows(Ap_t, parts, unit=unit)
            assert cuts[0] == 0 and cuts[-1] == 1003 and len(cuts) == parts + 1
            assert all(a <= b for a, b in zip(cuts, cuts[1:]))
            for p in range(1, parts):
                c = cuts[p]
                assert c % unit == 0 or c == 1003
                target = int(Ap[-1]) * p // parts
                assert Ap[c] >= target or c == 1003
                assert c == 0 or Ap[max(c - unit, 0)] < target or cuts[p - 1] == c
    table = [0, 8, 8, 200, 640, 1003]                               # a weight-cut plan's table (an empty chunk too)
    cuts = sp.dist.partition_rows(Ap_t, 3, table=table)
    assert all(c in table for c in cuts) and cuts[0] == 0 and cuts[-1] == 1003


def test_block_view_keeps_the_16_byte_phase(sp, oracle):
    """block_view: no copy of Aj / Ax, Ap_l[0] = the first row's position modulo 4, and the rows of the
    view are the rows of the parent."""
    rng = np.random.RandomState(4)
    Ap, Aj, Ax = random_csr(rng, 300, 90, 17)
    Ap_t, Aj_t, Ax_t = map(torch.from_numpy, (Ap, Aj, Ax))
    x = (rng.rand(90) * 2 - 1).astype(np.float32)
    y = oracle.spmv_serial(Ap, Aj, Ax, x)
    for r0, r1 in ((0, 300), (4, 120), (120, 300), (296, 300), (300, 300)):
        a, j, v, lo = sp.dist.block_view(Ap_t, Aj_t, Ax_t, r0, r1)
        assert lo % 4 == 0 and int(a[0]) == int(Ap[r0]) - lo and 0 <= int(a[0]) <= 3
        if j.numel():
            assert j.data_ptr() == Aj_t.data_ptr() + 4 * lo and v.data_ptr() == Ax_t.data_ptr() + 4 * lo
        assert int(a[-1]) == j.numel() == v.numel()
        for i in range(r1 - r0):
            s, e = int(a[i]), int(a[i + 1])
            assert np.array_equal(j[s:e].numpy(), Aj[Ap[r0 + i]:Ap[r0 + i + 1]])
        # the oracle's serial loop takes any Ap[0]
        assert np.array_equal(oracle.spmv_serial(a.numpy(), j.numpy(), v.numpy(), x), y[r0:r1])


def test_shard_csr_is_a_standalone_matrix(sp, oracle):
    rng = np.random.RandomState(1)
    Ap, Aj, Ax = random_csr(rng, 400, 70, 25)
    x = (rng.rand(70) * 2 - 1).astype(np.float32)
    y = oracle.spmv_serial(Ap, Aj, Ax, x)
    cuts = sp.dist.partition_rows(torch.from_numpy(Ap), 3)
    parts = []
    for p in range(3):
        a, j, v = sp.dist.shard_csr(torch.from_numpy(Ap), torch.from_numpy(Aj), torch.from_numpy(Ax),
                                    cuts[p], cuts[p + 1])
        assert int(a[0]) == 0 and int(a[-1]) == j.numel() == v.numel()
        parts.append(oracle.spmv_serial(a.numpy(), j.numpy(), v.numpy(), x))
    assert np.array_equal(np.concatenate(parts), y)                 # row-local: bit-exact


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, balance, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as g
    from oracle.oracle import Oracle
    sp = g.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(5)                                   # same matrix on every rank
    Ap, Aj, Ax = random_csr(rng, 1001, 333, 30, long_row=4000)
    x = (rng.rand(333) * 2 - 1).astype(np.float32)
    Ap_t, Aj_t, Ax_t = map(torch.from_numpy, (Ap, Aj, Ax))
    cuts = sp.dist.partition_rows(Ap_t, world, balance=balance)
    a, j, v = sp.dist.shard_csr(Ap_t, Aj_t, Ax_t, cuts[rank], cuts[rank + 1])
    y_local = torch.from_numpy(Oracle().spmv_serial(a.numpy(), j.numpy(), v.numpy(), x))
    y_full = torch.full((1001,), float("nan"))
    sp.dist.allgatherv(y_local, y_full, cuts)
    want = Oracle().spmv_serial(Ap, Aj, Ax, x)
    ok = np.array_equal(y_full.numpy(), want)
    open(os.path.join(out_dir, "rank%d.%s" % (rank, "ok" if ok else "bad")), "w").close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("balance", ["nnz", "rows"])
def test_allgatherv_world2_gloo(tmp_path, balance):
    """nnz balance gives unequal counts (broadcast-per-root path); 'rows' on 1001 rows too;
    both must reproduce the single-process y bit for bit on every rank."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, balance, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["rank0.ok", "rank1.ok"]


def _worker_views(rank, world, port, out_dir):
    """The flow of bench.py / mi355_spmv_dist_create_rank on the CPU: cuts on chunk boundaries (the library's
    partition rule), zero-copy block VIEWS that keep the 16-byte phase (Ap_l[0] != 0), two sub-blocks per rank,
    in-place exchange into the full-length y."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as g
    from oracle.oracle import Oracle
    sp = g.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(9)                                   # same matrix on every rank
    Ap, Aj, Ax = random_csr(rng, 4003, 500, 21, long_row=7000)
    x = (rng.rand(500) * 2 - 1).astype(np.float32)
    Ap_t, Aj_t, Ax_t = map(torch.from_numpy, (Ap, Aj, Ax))
    sub = 2
    rows = sp.dist.partition_rows(Ap_t, world * sub, unit=64)        # 64-row chunks of a uniform plan
    y_full = torch.full((4003,), float("nan"))
    phases = []
    for b in range(rank * sub, (rank + 1) * sub):
        a, j, v, lo = sp.dist.block_view(Ap_t, Aj_t, Ax_t, rows[b], rows[b + 1])
        phases.append(int(a[0]))
        y_full[rows[b]:rows[b + 1]] = torch.from_numpy(Oracle().spmv_serial(a.numpy(), j.numpy(), v.numpy(), x))
    rank_cuts = [rows[r * sub] for r in range(world)] + [rows[-1]]
    lo, hi = rank_cuts[rank], rank_cuts[rank + 1]
    sp.dist.allgatherv(y_full[lo:hi].clone(), y_full, rank_cuts)
    want = Oracle().spmv_serial(Ap, Aj, Ax, x)
    ok = np.array_equal(y_full.numpy(), want) and all(0 <= ph <= 3 for ph in phases)
    ok = ok and all(r % 64 == 0 or r == 4003 for r in rows)
    open(os.path.join(out_dir, "rank%d.%s" % (rank, "ok" if ok else "bad")), "w").close()
    dist.barrier()
    dist.destroy_process_group()


def test_rank_flow_with_block_views_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker_views, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["rank0.ok", "rank1.ok"]


def _worker_equal(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    sp = g.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cuts = [0, 8, 16]
    y_local = torch.arange(8, dtype=torch.float32) + 100 * rank
    y_full = torch.zeros(16)
    sp.dist.allgatherv(y_local, y_full, cuts)                        # equal counts: all_gather path
    want = torch.cat([torch.arange(8.0), torch.arange(8.0) + 100])
    ok = torch.equal(y_full, want)
    open(os.path.join(out_dir, "rank%d.%s" % (rank, "ok" if ok else "bad")), "w").close()
    dist.destroy_process_group()


def test_allgatherv_equal_counts_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker_equal, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["rank0.ok", "rank1.ok"]


def _worker_exchanges(rank, world, port, out_dir):
    """The count / displacement logic of the library's three exchanges (csrc/dist.hip exchange_sub_block, restated in
    dist.exchange_schedule) over gloo: every GPU's rows in 3 sub-blocks, uneven blocks, an EMPTY block, then equal
    adjacent blocks (the in-place all-gather)."""
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    sp = g.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for case in ("uneven", "equal"):
        sub = 3 if case == "uneven" else 1
        if case == "uneven":
            sizes = [((7 * g_ + 3) % 11) * 4 for g_ in range(world * sub)]      # multiples of 4 rows, one of them 0
            sizes[1] = 0
        else:
            sizes = [24] * world
        cuts = [0]
        for n in sizes:
            cuts.append(cuts[-1] + n)
        want = torch.arange(cuts[-1], dtype=torch.float32) * 0.5 + 1
        for mode in ("bcast", "sendrecv", "allgather"):
            y = torch.full((cuts[-1],), float("nan"))
            for s in range(sub):                                               # "compute" this rank's blocks
                b = rank * sub + s
                y[cuts[b]:cuts[b + 1]] = want[cuts[b]:cuts[b + 1]]
            for s in range(sub):
                sp.dist.exchange_sub_block(y, cuts, world, sub, s, mode)
            ok = ok and torch.equal(y, want)
            sched = sp.dist.exchange_schedule(cuts, world, sub, 0, mode, rank)
            if case == "equal" and mode == "allgather":
                ok = ok and sched == [("allgather_in_place", 0, 24)]
            if case == "uneven" and mode == "allgather":
                ok = ok and sched[0][0] in ("pack", "allgather_padded") and any(o[0] == "allgather_padded" and o[1] % 4 == 0 and o[1] >= max(sizes) for o in sched)
            if mode == "bcast":
                ok = ok and [o[1] for o in sched] == [r for r in range(world) if sizes[r * sub] > 0]
    open(os.path.join(out_dir, "rank%d.%s" % (rank, "ok" if ok else "bad")), "w").close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_three_exchanges_counts_and_displacements_gloo(tmp_path, world):
    port = _free_port()
    mp.spawn(_worker_exchanges, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["rank%d.ok" % r for r in range(world)]
