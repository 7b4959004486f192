#!/usr/bin/env python3
"""Exploratory (not part of the test suite): can two RANK-mode processes share the ONE GPU of a test box, so
that the RCCL leg of mi355_spmv_dist_* can be rehearsed without a multi-GPU node?  RCCL normally refuses
("Duplicate GPU detected"); this prints what happens.  Run under `timeout`."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)


def worker(rank, world, port):
    import __graft_entry__ as g
    sp = g.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    m = sp.synth.banded_fixed(1 << 16, 32, 512, seed=5, device=dev)
    x = sp.synth.dense_vector(m.n_cols, torch.float32, 5, dev)
    whole = sp.Plan("vector", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, torch.float32)
    y1 = torch.empty(m.n_rows, device=dev)
    whole.execute(m.Ax, x, y1)
    shape = whole.shape()
    rows, chunks, nnzs = whole.partition(world * 2)
    whole.destroy()
    box = [sp.DistPlan.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    r0, r1 = rows[rank * 2], rows[rank * 2 + 2]
    a, j, v, _ = sp.dist.block_view(m.Ap, m.Aj, m.Ax, r0, r1)
    try:
        d = sp.DistPlan.rank("vector", rank, world, box[0], 2, rows, chunks, nnzs, shape, m.n_cols, r1 - r0,
                             int(a[-1].item()), a, j, torch.float32)
    except RuntimeError as e:
        print("rank %d: create_rank failed: %s" % (rank, e), flush=True)
        return
    y = torch.full((m.n_rows,), float("nan"), device=dev)
    for _ in range(3):
        d.execute(v, x, y)
    torch.cuda.synchronize()
    print("rank %d: two ranks on one GPU worked, y == one-GPU y: %s" % (rank, bool(torch.equal(y, y1))), flush=True)
    d.destroy()
    dist.barrier()


if __name__ == "__main__":
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
