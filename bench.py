#!/usr/bin/env python3
"""bench.py — CSR SpMV throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload s32-band] [--kind auto]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one SpMV  y = A x  through the C ABI (mi355_spmv_plan_execute: every kernel of
the kind runs every step; the plan only holds scratch and launch shapes), inputs
resident in HBM.  N > 1: one process per GPU, the matrix is row-sharded (each rank
holds 2^22 rows of a banded matrix with N*2^22 rows: weak scaling), x is replicated
and the step ends with the allgatherv of the y slices over RCCL.

Default workload = the north-star target named in BASELINE.json / SURVEY.md §8(d):
S32-band, 2^22 rows, exactly 32 nnz/row inside a +-4096 band, fp32 values, 32-bit
offsets, seed 1 (1 124.1 MB of compulsory traffic, far beyond the 256 MiB Infinity
Cache).  The other configs are parity-test cases (tests/test_gpu_parity.py); they can
be timed with --workload.

One JSON line on stdout (rank 0).  Extra objects:
  roofline      achieved = ALGORITHMIC bytes of one SpMV / mean device time of one
                execute, from HIP events recorded on the launch stream around every
                timed step; peak = 8000 GB/s (HBM3E spec, MI355X_MICROARCH.md);
                traffic = HBM bytes per launch from rocprofv3 PMC passes (read from
                profiles/, null if that file is absent)
  cpu_baseline  the reference's serial CPU SpMV (cpu_navie.hpp:5-17) timed on this
                box's host cores, rank 0 / N = 1 only: kind "reference" = the
                reference's own header compiled by oracle/Makefile (oracle/_ref),
                else "port" = oracle/spmv_oracle.cpp.  Reported, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBPS = 8000.0
try:
    with open(os.path.join(ROOT, "BASELINE.json")) as _f:
        BASELINE_METRIC = json.load(_f)["metric"]
except Exception:
    BASELINE_METRIC = "GFLOP/s (2*nnz/t) and achieved HBM GB/s, CSR SpMV fp32, 1/2/4/8 MI355X"
KINDS = ("vector", "merge", "light")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="s32-band",
                    choices=["s32-band", "s32-rand", "c2-cant", "c3-webgoogle", "c4-nlpkkt", "c5-rmat24"])
    ap.add_argument("--kind", default="auto", choices=("auto",) + KINDS)
    ap.add_argument("--rows-log2", type=int, default=22, help="rows per GPU of the s32 workloads (2^k)")
    ap.add_argument("--s32-offsets", choices=("i32", "i64"), default="i32", help="offset type of the s32 workloads")
    ap.add_argument("--s32-values", choices=("f32", "f64"), default="f32", help="value type of the s32 workloads")
    ap.add_argument("--reuse-structure", action="store_true",
                    help="plan flag MI355_PLAN_REUSE_STRUCTURE (merge: keep tile coordinates)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cold", action="store_true",
                    help="write 512 MiB between steps so that nothing of the matrix is left in the 256 MiB Infinity "
                         "Cache (SURVEY 8d: C2/C3 fit it); the HIP-event kernel time is then the figure to read")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--all-kinds", action="store_true", help="time every kind for K steps (extra field)")
    return ap.parse_args()


def build_local(sp, args, rank, world, dev):
    """This rank's row block (global column ids) + the replicated x."""
    if args.workload in ("s32-band", "s32-rand"):
        n = 1 << args.rows_log2
        hw = 4096 if args.workload == "s32-band" else None
        m = sp.synth.banded_fixed(n, 32, hw, seed=1 + rank, device=dev, row_offset=rank * n, n_cols=world * n,
                                  val_dtype=torch.float64 if args.s32_values == "f64" else torch.float32,
                                  off_dtype=torch.int64 if args.s32_offsets == "i64" else torch.int32,
                                  name="S32-band" if hw else "S32-rand")
        cuts = [p * n for p in range(world + 1)]
        return m, cuts
    full = sp.synth.workload(args.workload, dev)
    if world == 1:
        return full, [0, full.n_rows]
    cuts = sp.dist.partition_rows(full.Ap, world)
    a, j, v = sp.dist.shard_csr(full.Ap, full.Aj, full.Ax, cuts[rank], cuts[rank + 1])
    m = sp.synth.Csr(cuts[rank + 1] - cuts[rank], full.n_cols, int(j.numel()), a, j, v, full.name, full.meta)
    del full
    return m, cuts


_FLUSH = {"buf": None}


def time_steps(plan, m, x, y_local, y_full, cuts, use_dist, steps, sp):
    """K steps; returns (wall seconds between the two syncs, mean device ms of one execute).
    use_dist: a process group exists (launched by torch.distributed.run) -> every step ends
    with the allgatherv of the y slices, and the timed region is bracketed by barriers.
    The exchange of step k runs beside the SpMV of step k+1 (y_local / y_full are pairs of
    buffers; a buffer is reused only after the exchange that read it has finished); every
    exchange completes inside the timed region."""
    # HIP events bracket EVERY timed execute.  A pair costs the stream ~6 us of wall per step (184.1 vs 178.8 us
    # with a pair on every 6th step), but sampled pairs read the bracketed execute 2-5 % too long (the marker
    # before it is then not back to back with the one after the previous execute), and the kernel time is
    # what the roofline fraction is computed from: the rocprofv3 trace agrees with the per-step pairs.
    stride = 1
    evs = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for i in range(0, steps, stride)}
    pending = [None, None]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        k = i & 1
        if pending[k] is not None and not pending[k].done():   # (a finished exchange needs no stream-level wait)
            pending[k].wait()
        if _FLUSH["buf"] is not None:
            _FLUSH["buf"].fill_(float(i))      # cold mode: evict the matrix from L2 / Infinity Cache
        ev = evs.get(i)
        if ev:
            ev[0].record()
        plan.execute(m.Ax, x, y_local[k])
        if ev:
            ev[1].record()
        if use_dist:
            pending[k] = sp.dist.allgatherv(y_local[k], y_full[k], cuts, async_op=True)
    for w in pending:
        if w is not None:
            w.wait()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    wall = time.perf_counter() - t0
    dev_ms = float(np.mean([a.elapsed_time(b) for a, b in evs.values()]))
    return wall, dev_ms


def cpu_baseline(m, x, budget_s):
    """Serial CPU SpMV (the reference's correctness path) on this box's host cores."""
    from oracle.oracle import Oracle, Ref
    Ap, Aj, Ax = m.numpy()
    xh = x.cpu().numpy()
    orc = Oracle()
    use_ref = Ref.available()
    ref = Ref() if use_ref else None
    run = (lambda: ref.spmv_cpu(m.n_cols, Ap, Aj, Ax, xh)) if use_ref else (lambda: orc.spmv_serial(Ap, Aj, Ax, xh))
    run()  # warm-up (page-in)
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 400):
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    out = {"value": 2.0 * m.nnz / med / 1e9, "unit": "GFLOP/s", "cores": 1,
           "kind": "reference" if use_ref else "port",
           "sample": "full %s matrix (%d rows, %d nnz), median of %d serial passes, %.3f s each"
                     % (m.name, m.n_rows, m.nnz, len(times), med),
           "gbps": m.algorithmic_bytes() / med / 1e9}
    # all-core row-parallel variant of the port (BASELINE.md §2b ii), 3 passes
    nthr = orc.hardware_threads()
    orc.spmv_parallel(Ap, Aj, Ax, xh, nthr)
    tp = []
    for _ in range(3):
        t0 = time.perf_counter()
        orc.spmv_parallel(Ap, Aj, Ax, xh, nthr)
        tp.append(time.perf_counter() - t0)
    out["all_cores"] = {"value": 2.0 * m.nnz / float(np.median(tp)) / 1e9, "unit": "GFLOP/s", "cores": nthr,
                        "kind": "port"}
    return out


def read_traffic(kernel, algorithmic_bytes):
    """HBM bytes per launch of `kernel` from the committed PMC summaries (profiles/traffic_latest.json),
    only when a summary was taken on THIS configuration (same kernel, same algorithmic bytes)."""
    p = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(p) as f:
            entries = json.load(f)["entries"]
        for e in entries:
            if e["kernel"] == kernel and int(e["algorithmic_bytes"]) == int(algorithmic_bytes):
                return e["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one process per GPU)" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # launched by torch.distributed.run (RANK set): one process per GPU over RCCL, also for N = 1
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout when the communicator comes up; stdout is
        # reserved for the one JSON line, so route fd 1 to stderr until the first collective is done
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    sp = graft.load_package()

    if args.cold:
        _FLUSH["buf"] = torch.empty(512 << 18, dtype=torch.float32, device=dev)   # 512 MiB
    m, cuts = build_local(sp, args, rank, world, dev)
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, dev)  # same seed on every rank: replicated x
    y_local = [torch.empty(m.n_rows, dtype=m.Ax.dtype, device=dev) for _ in range(2)]
    y_full = [torch.empty(cuts[-1], dtype=m.Ax.dtype, device=dev) for _ in range(2)] if use_dist else y_local
    flags = sp.capi.PLAN_REUSE_STRUCTURE if args.reuse_structure else 0
    plans = {k: sp.Plan(k, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype, flags)
             for k in (KINDS if args.kind == "auto" or args.all_kinds else (args.kind,))}

    # warm-up; with --kind auto the warm-up also picks the kind (outside the timed region)
    probe = {}
    for k, p in plans.items():
        time_steps(p, m, x, y_local, y_full, cuts, use_dist, max(args.warmup, 1), sp)   # warm-up proper
    for k, p in plans.items():
        _, ms = time_steps(p, m, x, y_local, y_full, cuts, use_dist, max(args.warmup, 30), sp)
        probe[k] = ms
    kind = args.kind
    if kind == "auto":
        best = torch.tensor([probe[k] for k in KINDS], device=dev)
        if use_dist:
            dist.all_reduce(best, op=dist.ReduceOp.MAX)
        kind = KINDS[int(torch.argmin(best).item())]
    plan = plans[kind]

    wall, dev_ms = time_steps(plan, m, x, y_local, y_full, cuts, use_dist, args.steps, sp)
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
    nnz_all = torch.tensor([m.nnz], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(nnz_all, op=dist.ReduceOp.SUM)
    wall = float(tmax.item())
    total_nnz = float(nnz_all.item())

    others = {}
    if args.all_kinds:
        for k, p in plans.items():
            _, ms = time_steps(p, m, x, y_local, y_full, cuts, use_dist, args.steps, sp)
            others[k] = {"kernel_ms": ms, "gflops": 2.0 * m.nnz / ms / 1e6,
                         "gbps": m.algorithmic_bytes() / ms / 1e6}

    if rank == 0:
        info = plan.info()
        bytes_alg = m.algorithmic_bytes()
        achieved = bytes_alg / (dev_ms * 1e-3) / 1e9
        out = {
            # BASELINE.json's metric, verbatim; `value` is its GFLOP/s part, the achieved HBM GB/s part
            # is `achieved_hbm_gbps` / `roofline.achieved`
            "metric": BASELINE_METRIC if m.Ax.dtype == torch.float32 else BASELINE_METRIC.replace("fp32", "fp64"),
            "value": 2.0 * total_nnz * args.steps / wall / 1e9,
            "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if m.Ax.dtype == torch.float32 else "f64",
            "data": "synthetic",
            "config": {"workload": "%s: %d rows/GPU x %d GPU, %d nnz/GPU, %s offsets, seeded"
                                   % (m.name, m.n_rows, world, m.nnz, "i32" if m.Ap.dtype == torch.int32 else "i64"),
                       "kind": kind, "lanes_per_row": info["lanes_per_row"], "grid_blocks": info["grid_blocks"],
                       "kernels_per_step": info["n_kernels"], "x_window_elems": info["window_elems"], "x_window_segments": info["window_segments"], "reuse_structure": bool(args.reuse_structure),
                       "parallelism": ("row-block x%d, x replicated, allgatherv(y) over RCCL overlapped with the next "
                                       "step's SpMV" % world) if use_dist else "single GPU"},
            "achieved_hbm_gbps": achieved,
            # per-GPU SpMV alone (device time of the execute) vs the whole step incl. the exchange
            "compute_only": {"ms": dev_ms, "gflops_per_gpu": 2.0 * m.nnz / dev_ms / 1e6,
                             "step_ms_incl_exchange": wall / args.steps * 1e3},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": read_traffic(info["main_kernel"], bytes_alg),
                         "kernel": info["main_kernel"], "kernel_ms": dev_ms, "algorithmic_bytes": bytes_alg},
            "warmup_probe_ms": probe,
        }
        if others:
            out["all_kinds"] = others
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(m, x, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    for p in plans.values():
        p.destroy()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
