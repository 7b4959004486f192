#!/usr/bin/env python3
"""bench.py — CSR SpMV throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload s32-band] [--kind auto]
    python bench.py --mtx FILE.mtx [--dtype f32|f64] [--offset 32|64]        # a Matrix Market file (main.cu:32-39)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one SpMV  y = A x  through the C ABI (mi355_spmv_plan_execute: every kernel of
the kind runs every step; the plan only holds scratch and launch shapes), inputs
resident in HBM.  N > 1: one process per GPU — `python bench.py --gpus N` starts its N
rank processes itself (fresh children, before anything here touches a GPU), or the
driver starts them with torch.distributed.run — the matrix is row-sharded (each rank
holds 2^22 rows of a banded matrix with N*2^22 rows: weak scaling; any other workload,
e.g. --workload c5-rmat24 or --mtx, cuts ONE matrix into N nnz-balanced row blocks:
strong scaling), x is replicated and every step ends with the allgatherv of the y slices
over RCCL, made by the library itself (mi355_spmv_dist_*: grouped broadcasts, send/recv
pairs or one all-gather per sub-block on a communication stream — the library times the
three when the communicator comes up and keeps the fastest — each GPU's rows in
sub-blocks so that a slice travels while the next is computed).
torch.distributed (gloo) only carries the control plane: the 128-byte RCCL id, the
barriers and the max-over-ranks of the timings.  For N > 1 the line separates the step
into its parts: `compute_only` (the blocks' kernels, no exchange), `exchange_only` (the
allgatherv alone) and the step itself (both, overlapped).

Default workload = the north-star target named in BASELINE.json / SURVEY.md §8(d):
S32-band, 2^22 rows, exactly 32 nnz/row inside a +-4096 band, fp32 values, 32-bit
offsets, seed 1 (1 124.1 MB of compulsory traffic, far beyond the 256 MiB Infinity
Cache).  The other configs are parity-test cases (tests/test_gpu_parity.py); they can
be timed with --workload.

One JSON line on stdout (rank 0) — also when the run fails: a rank that dies, an RCCL
or gloo call that does not return and a wall-clock limit all end in ONE line with an
"error" field and a non-zero exit code, never in a hang.  Extra objects:
  roofline      achieved = ALGORITHMIC bytes of one SpMV / mean device time of one
                execute = (HIP events recorded on the launch stream around the K timed
                executes) / K — for N > 1: this GPU's bytes / its compute-only time;
                min / median from a second pass with events around every execute;
                peak = 8000 GB/s (HBM3E spec, MI355X_MICROARCH.md);
                traffic = HBM bytes per launch from rocprofv3 PMC passes (read from
                profiles/, null if that file is absent)
  cpu_baseline  the reference's serial CPU SpMV (cpu_navie.hpp:5-17) timed on this
                box's host cores, rank 0 / N = 1 only: kind "reference" = the
                reference's own header compiled by oracle/Makefile (oracle/_ref),
                else "port" = oracle/spmv_oracle.cpp.  Reported, not the target.
"""
import argparse
import datetime
import json
import os
import sys
import threading
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
WORKLOADS = ["s32-band", "s32-rand", "c2-cant", "c3-webgoogle", "c4-nlpkkt", "c5-rmat24"]
# wall-clock limit of a whole run and of one control-plane collective (a dead peer must not park the others)
LIMIT_S = float(os.environ.get("MI355_BENCH_LIMIT_S", "900"))
COLLECTIVE_S = float(os.environ.get("MI355_BENCH_COLLECTIVE_S", "180"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="s32-band", choices=WORKLOADS)
    ap.add_argument("--mtx", default=None, metavar="PATH",
                    help="a Matrix Market coordinate file instead of a synthetic workload, read by the product loader "
                         "(host/load.hpp through include/mi355_load.h: the reference's LoadCoo + ToCsr, main.cu:32-39)")
    ap.add_argument("--dtype", choices=("f32", "f64"), default="f32", help="value type of --mtx")
    ap.add_argument("--offset", choices=("32", "64"), default="32", help="offset width of --mtx")
    ap.add_argument("--kind", default="auto", choices=("auto",) + KINDS)
    ap.add_argument("--rows-log2", type=int, default=22, help="rows per GPU of the s32 workloads (2^k)")
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU of the s32 workloads when not a power of two (overrides --rows-log2)")
    ap.add_argument("--band-half-width", type=int, default=4096,
                    help="s32-band: columns within +-this of the diagonal (4096 = the north-star target; wider bands "
                         "exercise the 1 024-thread and the sweeping-window plans)")
    ap.add_argument("--nnz-per-row", type=int, default=32,
                    help="nonzeros per row of the s32 workloads (32 = the north-star target; other lengths for plan studies)")
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
    ap.add_argument("--sub-blocks", type=int, default=0,
                    help="row blocks per GPU in the multi-GPU path (0 = 4 when N > 1, else 1)")
    ap.add_argument("--exchange", default="auto", choices=("auto", "bcast", "sendrecv", "allgather"),
                    help="how the y slices travel when N > 1 (auto = the library's timed pick at create)")
    return ap.parse_args()


def error_line(msg, world, args=None):
    """The JSON line of a failed run: same keys as a good one where they are known, value null, an "error"."""
    return json.dumps({"metric": BASELINE_METRIC, "value": None, "unit": "GFLOP/s", "n_gpus": world,
                       "steps": getattr(args, "steps", None), "warmup": getattr(args, "warmup", None),
                       "higher_is_better": True, "vs_baseline": None, "data": "synthetic", "error": msg})


def spawn_ranks(n, args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N rank processes here (fresh
    children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; this parent never initialises a GPU), pass rank 0's
    JSON line through and exit with the worst exit code.  The children are WATCHED: the first one that exits
    non-zero, or the wall-clock limit, ends the others (SIGTERM, then SIGKILL), and the parent prints one JSON line
    with an "error" field — a rank that died in hipMalloc or ncclCommInitRank must not leave rank 0 waiting in a
    barrier for the driver's time limit."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    t_end = time.monotonic() + LIMIT_S + 30          # (the ranks' own watchdogs fire first)
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = "rank %d exited with code %d" % bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > t_end:
            failed = "wall-clock limit of %.0f s reached" % (LIMIT_S + 30)
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 5
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
    out = procs[0].stdout.read().decode() if procs[0].stdout else ""
    lines = [l for l in out.splitlines() if l.strip().startswith("{")]
    if failed and not lines:
        lines = [error_line("bench.py --gpus %d: %s; the other ranks were stopped" % (n, failed), n, args)]
    sys.stdout.write("\n".join(lines) + ("\n" if lines else ""))
    sys.stdout.flush()
    codes = [p.returncode for p in procs]
    sys.exit(1 if failed else max(abs(c or 0) for c in codes))


class Watchdog(threading.Thread):
    """Every rank watches its own clock: past the limit (a collective that never returns, a peer that died without
    the launcher noticing) rank 0 prints the error line and the process ends — os._exit works from a thread even while
    the main thread sits in a C call.  `phase` names where the run was."""

    def __init__(self, rank, world, args, json_fd):
        super().__init__(daemon=True)
        self.rank, self.world, self.args, self.json_fd = rank, world, args, json_fd
        self.phase, self.done = "start", False
        self.deadline = time.monotonic() + LIMIT_S

    def run(self):
        while not self.done:
            if time.monotonic() > self.deadline:
                msg = "bench.py rank %d/%d: no progress within %.0f s (last phase: %s)" % (self.rank, self.world, LIMIT_S, self.phase)
                sys.stderr.write(msg + "\n")
                if self.rank == 0:
                    os.write(self.json_fd, (error_line(msg, self.world, self.args) + "\n").encode())
                os._exit(3)
            time.sleep(0.5)


def load_matrix(sp, args, dev):
    if args.mtx:
        m = sp.load.load_mtx(args.mtx, torch.int64 if args.offset == "64" else torch.int32,
                             torch.float64 if args.dtype == "f64" else torch.float32, dev)
        if m.n_rows <= 0 or m.nnz <= 0:
            raise SystemExit("bench.py: %s holds no rows / no nonzeros" % args.mtx)
        return m
    return sp.synth.workload(args.workload, dev)


def build_local(sp, args, rank, world, dev, sub_blocks):
    """This rank's row block as a 16-byte-aligned view (spmv-samples_amd/dist.py block_view) + what
    mi355_spmv_dist_create_rank needs: the global cut lists of all world * sub_blocks blocks and, when ONE
    matrix is cut (every workload but the s32 ones), the whole matrix's plan shape so that the blocks sum every row
    as one GPU would.  Returns (local Csr whose Ap[0] is the block's phase, cuts dict)."""
    parts = world * sub_blocks
    if not args.mtx and args.workload in ("s32-band", "s32-rand"):
        # weak scaling: every rank generates its own 2^k rows of a banded matrix with world * 2^k rows; the
        # blocks are statistically alike, every rank can write down everybody's cuts (32 nonzeros per row)
        n = args.rows if args.rows > 0 else 1 << args.rows_log2
        hw = args.band_half_width if args.workload == "s32-band" else None
        m = sp.synth.banded_fixed(n, args.nnz_per_row, hw, seed=1 + rank, device=dev, row_offset=rank * n, n_cols=world * n,
                                  val_dtype=torch.float64 if args.s32_values == "f64" else torch.float32,
                                  off_dtype=torch.int64 if args.s32_offsets == "i64" else torch.int32,
                                  name=(("S32-band" if hw == 4096 else "S32-band(+-%d)" % hw) if hw else "S32-rand") + ("" if args.nnz_per_row == 32 else "[%d per row]" % args.nnz_per_row))
        sub = [(n * s // sub_blocks) & ~3 for s in range(sub_blocks)]
        rows = [r * n + o for r in range(world) for o in sub] + [world * n]
        return m, {"rows": rows, "chunks": None, "nnz": [args.nnz_per_row * r for r in rows], "shape": None, "kind_shape": {}}
    full = load_matrix(sp, args, dev)
    if world == 1 and sub_blocks == 1:
        return full, {"rows": [0, full.n_rows], "chunks": None, "nnz": [0, full.nnz], "shape": None, "kind_shape": {}}
    # strong scaling: ONE matrix, cut on the chunk boundaries of its own one-GPU plan (per kind)
    kind_shape = {}
    for k in (KINDS if args.kind == "auto" or args.all_kinds else (args.kind,)):
        whole = sp.Plan(k, full.n_rows, full.n_cols, full.nnz, full.Ap, full.Aj, full.Ax.dtype)
        kind_shape[k] = (whole.shape(),) + whole.partition(parts)
        whole.destroy()
    return full, {"rows": None, "chunks": None, "nnz": None, "shape": None, "kind_shape": kind_shape}


class _NoPlan:
    """A rank that owns no row (strong scaling of a tiny matrix): nothing to launch, still part of every collective."""

    def execute(self, *a, **k):
        pass

    def info(self):
        return {"lanes_per_row": 0, "grid_blocks": 0, "n_kernels": 0, "window_elems": 0, "window_segments": 0,
                "knobs": "", "main_kernel": "none (this rank owns no rows)"}

    def destroy(self):
        pass


class Runner:
    """One kind on this rank: a plain plan (single GPU, one block) or the library's multi-GPU object."""

    def __init__(self, sp, kind, m, cuts, rank, world, sub_blocks, unique_id, flags, use_dist, exchange="auto"):
        self.kind, self.m, self.use_dist, self.sp = kind, m, use_dist, sp
        self.blocks, self.exchange, self.dist_info = None, "none (single GPU)", None
        if not use_dist:
            self.plan = sp.Plan(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype, flags)
            self.Ax, self.n_rows_global, self.n_rows_local, self.nnz_local = m.Ax, m.n_rows, m.n_rows, m.nnz
            return
        if kind in cuts["kind_shape"]:       # one matrix cut into blocks: this rank keeps its view only
            shape, rows, chunks, nnzs = cuts["kind_shape"][kind]
            first = rank * sub_blocks
            r0, r1 = rows[first], rows[first + sub_blocks]
            Ap_l, Aj_l, Ax_l, _ = sp.dist.block_view(m.Ap, m.Aj, m.Ax, r0, r1)
        else:                                # every rank generated its own block (s32): Ap[0] == 0 already
            shape, rows, chunks, nnzs = None, cuts["rows"], cuts["chunks"], cuts["nnz"]
            r0, r1 = rows[rank * sub_blocks], rows[(rank + 1) * sub_blocks]
            Ap_l, Aj_l, Ax_l = m.Ap, m.Aj, m.Ax
        self.rows = rows
        self.Ax = Ax_l
        self.n_rows_global, self.n_rows_local = rows[-1], r1 - r0
        self.nnz_local = nnzs[(rank + 1) * sub_blocks] - nnzs[rank * sub_blocks]
        err = ""
        try:
            self.plan = sp.DistPlan.rank(kind, rank, world, unique_id, sub_blocks, rows, chunks, nnzs, shape, m.n_cols,
                                         r1 - r0, int(Ap_l[-1].item()) if r1 > r0 else 0, Ap_l, Aj_l, m.Ax.dtype, flags)
            if exchange != "auto" and world > 1:
                self.plan.set_exchange(exchange)
            ok = 1
        except RuntimeError as e:
            self.plan, ok, err = None, 0, str(e)
        if os.environ.get("MI355_BENCH_FORCE_FALLBACK"):     # (rehearsal of the safety net on one GPU)
            ok, err = 0, "forced by MI355_BENCH_FORCE_FALLBACK"
        if world > 1:
            flag = torch.tensor([ok])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        self._fb = (sp, rank, world, sub_blocks, rows, chunks, nnzs, shape, r0, Ap_l, Aj_l, Ax_l, flags)
        if not ok:
            self.fall_back(err)
        else:
            self.dist_info = self.plan.dist_info()
            t = self.dist_info["trial_us"]
            self.exchange = ("mi355_spmv_dist_* %s per sub-block on a communication stream" % self.dist_info["exchange_name"]) + (
                " (the library's timed pick: bcast %.0f / sendrecv %.0f / allgather %.0f us per exchange of all sub-blocks)"
                % (t["bcast"], t["sendrecv"], t["allgather"]) if self.dist_info["auto_picked"] else
                (" (forced)" if world > 1 else " (one GPU: no communicator)"))

    def fall_back(self, err):
        """Safety net so that a scaling run still measures something: the same row blocks through the C ABI's
        block plans, the exchange through torch.distributed's RCCL group instead of the library's own."""
        sp, rank, world, sub_blocks, rows, chunks, nnzs, shape, r0, Ap_l, Aj_l, Ax_l, flags = self._fb
        m, kind = self.m, self.kind
        if self.plan is not None:
            self.plan.destroy()
        sys.stderr.write("bench.py: the library's multi-GPU object failed (%s); exchanging y through torch.distributed\n" % err)
        self.exchange = "FALLBACK torch.distributed over RCCL (the library's own communicator failed: %s)" % (err or "on another rank")
        self.dist_info = None
        self.pg = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=COLLECTIVE_S))
        self.rank_cuts = [rows[r * sub_blocks] for r in range(world)] + [rows[-1]]
        self.blocks = []
        for b in range(rank * sub_blocks, (rank + 1) * sub_blocks):
            if rows[b + 1] == rows[b]:
                continue
            a, j, v, _ = sp.dist.block_view(Ap_l, Aj_l, Ax_l, rows[b] - r0, rows[b + 1] - r0)
            cb = chunks[b] if chunks is not None else 0
            nc = (chunks[b + 1] - chunks[b]) if chunks is not None else 0
            pl = sp.Plan.block(kind, shape, rows[b], cb, nc, nnzs[b], rows[b + 1] - rows[b], m.n_cols,
                               int(a[-1].item()), a, j, m.Ax.dtype, flags)
            self.blocks.append((pl, v, rows[b], rows[b + 1]))
        # (a rank whose blocks are all empty keeps an empty list: it still enters every collective below)
        self.plan = self.blocks[0][0] if self.blocks else _NoPlan()
        self.rank = rank

    def check_exchange(self, x, y):
        """N > 1 (and the N = 1 rehearsal): one step on a y poisoned with NaN; afterwards every rank must hold, for
        every rank's rows, the very bits their owner holds (a checksum of the raw words per slice, compared over the
        gloo control plane), and no NaN.  A failed step or a mismatch anywhere sends every rank to the
        torch.distributed fallback, once."""
        sp, rank, world, sub_blocks, rows = self._fb[:5]
        cuts = [rows[r * sub_blocks] for r in range(world)] + [rows[-1]]
        for attempt in range(2):
            ok, err = 1, ""
            try:
                y.fill_(float("nan"))
                self.execute(x, y)
                torch.cuda.synchronize()
                words = y.view(torch.int32 if y.element_size() == 4 else torch.int64)
                sums = torch.stack([words[cuts[r]:cuts[r + 1]].to(torch.int64).sum() for r in range(world)]).cpu()
                if bool(torch.isnan(y).any().item()):
                    ok, err = 0, "rows left unwritten after the exchange"
            except RuntimeError as e:
                ok, err, sums = 0, str(e), torch.zeros(world, dtype=torch.int64)
            box = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(box, sums)
            if any(not torch.equal(b, box[0]) for b in box):
                ok, err = 0, err or "a rank holds other bits than the owner of the rows"
            flag = torch.tensor([ok])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()):
                return "bitwise equal on all ranks" + (" (after falling back)" if attempt else "")
            if attempt or self.blocks is not None:
                return "FAILED: " + (err or "on another rank")
            self.fall_back(err)
        return "FAILED"

    def execute(self, x, y, leg="step"):
        """leg: "step" (kernels + exchange), "compute" (kernels only), "exchange" (the allgatherv only)."""
        if self.blocks is None:
            if not self.use_dist:
                self.plan.execute(self.Ax, x, y)
            else:
                self.plan.execute(self.Ax, x, y, flags={"step": 0, "compute": 1, "exchange": 2}[leg])
            return
        if leg != "exchange":
            for pl, v, b0, b1 in self.blocks:
                pl.execute(v, x, y[b0:b1])
        if leg != "compute":
            lo, hi = self.rank_cuts[self.rank], self.rank_cuts[self.rank + 1]
            self.sp.dist.allgatherv(y[lo:hi], y, self.rank_cuts, group=self.pg)

    def info(self):
        return self.plan.info()        # (multi-GPU object: its first block's launch shape)

    def destroy(self):
        if self.blocks is None:
            self.plan.destroy()
        else:
            for pl, _, _, _ in self.blocks:
                pl.destroy()


_FLUSH = {"buf": None}
_STATE = {}       # json_fd / rank / world / args once stdout has been redirected (for the error line of a crashed run)


def time_steps(run, x, y, use_dist, steps, per_step=False, leg="step"):
    """K steps; returns (wall seconds between the two syncs, list of device ms).
    per_step=False (the timed region): ONE pair of HIP events around the K executes, on the stream they are
    launched on — the list holds their mean, K times.  per_step=True (a separate, instrumented pass; always in
    --cold mode): a pair around EVERY execute, for the minimum / median — each pair costs the stream ~6 us of wall
    per step (184.1 vs 178.8 us), which is why the timed region does not carry them.
    A step of the multi-GPU path is complete when this GPU holds the WHOLE y (the library makes the caller's stream
    wait for its communication stream), so the events bracket SpMV + exchange there; leg = "compute" / "exchange"
    runs one half only (mi355_spmv_dist_execute_ex)."""
    per_step = per_step or _FLUSH["buf"] is not None
    n_ev = steps if per_step else 1
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if not per_step:
        evs[0][0].record()
    for i in range(steps):
        if _FLUSH["buf"] is not None:
            _FLUSH["buf"].fill_(float(i))      # cold mode: evict the matrix from L2 / Infinity Cache
        if per_step:
            evs[i][0].record()
        run.execute(x, y, leg)
        if per_step:
            evs[i][1].record()
    if not per_step:
        evs[0][1].record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    wall = time.perf_counter() - t0
    if per_step:
        return wall, [a.elapsed_time(b) for a, b in evs]
    return wall, [evs[0][0].elapsed_time(evs[0][1]) / steps] * steps


def pick_kind(runs, x, y, use_dist, warmup):
    """--kind auto, outside the timed region.  Every kind is first warmed for max(W, 70) executes (an HBM-bound kernel
    needs ~60-70 from a fresh plan to reach its steady time, profiles/r02_warmup_curve.txt); then the kinds are probed
    INTERLEAVED — A B C A B C ..., 7 rounds of 10 executes each — so that no kind owns the clock / cache state the
    previous one left, and the MEDIAN of a kind's rounds decides (max over ranks when N > 1)."""
    for r in runs.values():
        time_steps(r, x, y, use_dist, max(warmup, 70))
    samples = {k: [] for k in runs}
    for _ in range(7):
        for k, r in runs.items():
            _, ms = time_steps(r, x, y, use_dist, 10)
            samples[k].append(float(ms[0]))
    med = {k: float(np.median(v)) for k, v in samples.items()}
    return med


def one_shot_ms(sp, kind, m, x, y, reps=5, first=False):
    """The reference's per-call life cycle (SURVEY §8(d) "also reported as one-shot time") through the one-shot entry
    point, host clock, median of `reps` after one warm-up.  first=False: calls in a row on the same matrix, as the
    reference's harness makes them (main.cu:102-113) — the entry point finds its plan of the previous call again;
    first=True: the kept plans are released before every call, i.e. plan create (structure probe, scratch) + execute +
    synchronise every time."""
    ts = []
    for i in range(reps + 1):
        if first:
            sp.capi.cache_release()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sp.spmv(kind, m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax, x, y)   # (synchronises the stream itself)
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts[1:]))


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
    """(HBM bytes per launch, source) of `kernel` from the committed PMC summaries (profiles/traffic_latest.json):
    a rocprofv3 --pmc pass cannot run inside this process, so the figure is a RECORDED one — quoted only when
    a summary was taken on THIS configuration (same kernel, same algorithmic bytes), with its source."""
    p = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(p) as f:
            entries = json.load(f)["entries"]
        for e in entries:
            if e["kernel"] == kernel and int(e["algorithmic_bytes"]) == int(algorithmic_bytes):
                return e["hbm_bytes_per_launch"], "recorded: %s (%s)" % (e.get("source", "profiles/traffic_latest.json"),
                                                                        e.get("commit", "commit not recorded"))
    except Exception:
        pass
    return None, None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ     # by torch.distributed.run or by spawn_ranks
    if not launched and args.gpus > 1:
        spawn_ranks(args.gpus, args)                                    # (does not return)
    if launched:
        args.gpus = world
    # stdout carries exactly ONE line, the JSON: gloo ("[Gloo] Rank 0 is connected ...") and RCCL (its version banner)
    # print to fd 1 when a process group / communicator comes up, so fd 1 points at stderr until that line is due
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    _STATE.update(json_fd=json_fd, rank=rank, world=world, args=args)
    dog = Watchdog(rank, world, args, json_fd)
    dog.start()
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one process per GPU (RANK set), also for N = 1: the library's multi-GPU object, control plane over gloo
    use_dist = launched
    sub_blocks = args.sub_blocks if args.sub_blocks > 0 else (4 if world > 1 else 1)
    sp = graft.load_package()
    unique_id = None
    if use_dist:
        dog.phase = "gloo rendezvous"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (every control-plane collective inherits this limit: a rank whose peer died gets an exception, not a wait)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=COLLECTIVE_S))
        if world > 1:
            box = [sp.DistPlan.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            unique_id = box[0]

    if args.cold:
        _FLUSH["buf"] = torch.empty(512 << 18, dtype=torch.float32, device=dev)   # 512 MiB
    dog.phase = "building the matrix"
    m, cuts = build_local(sp, args, rank, world, dev, sub_blocks) if use_dist else build_local(sp, args, 0, 1, dev, 1)
    x = sp.synth.dense_vector(m.n_cols, m.Ax.dtype, 1, dev)  # same seed on every rank: replicated x
    flags = sp.capi.PLAN_REUSE_STRUCTURE if args.reuse_structure else 0
    kinds = KINDS if args.kind == "auto" or args.all_kinds else (args.kind,)
    # (a fresh 128-byte id per communicator: one per kind when several kinds are timed)
    runs = {}
    for k in kinds:
        dog.phase = "creating the %s plan / communicator (exchange trial inside)" % k
        uid = unique_id
        if use_dist and world > 1 and runs:
            box = [sp.DistPlan.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
        runs[k] = Runner(sp, k, m, cuts, rank, world, sub_blocks, uid, flags, use_dist, args.exchange)
    n_rows_global = next(iter(runs.values())).n_rows_global
    y = torch.empty(n_rows_global, dtype=m.Ax.dtype, device=dev)
    dog.phase = "exchange self-check"
    checks = {k: r.check_exchange(x, y) for k, r in runs.items()} if use_dist and (world > 1 or os.environ.get("MI355_BENCH_CHECK_EXCHANGE")) else {}   # (the knob: N = 1 rehearsal)

    # warm-up; with --kind auto the warm-up also picks the kind (outside the timed region)
    dog.phase = "warm-up / kind probe"
    kind = args.kind
    if kind == "auto":
        probe = pick_kind(runs, x, y, use_dist, args.warmup)
        best = torch.tensor([probe[k] for k in KINDS], dtype=torch.float64)
        if use_dist:
            dist.all_reduce(best, op=dist.ReduceOp.MAX)
        kind = KINDS[int(torch.argmin(best).item())]
    else:
        probe = {}
        for k, r in runs.items():
            time_steps(r, x, y, use_dist, max(args.warmup, 70))
    run = runs[kind]
    # the W warm-up steps proper, on the kind that is timed — at least 100 of them in ONE uninterrupted stream: the probe
    # above runs in bursts of 10 with a synchronisation between them, and a short timed region (the driver's K = 20 is
    # 3.5 ms) right behind such bursts measured ~1 % slower than behind 17 ms of continuous executes
    # (profiles/r03_final_warm.txt: 175.0 / 175.7 us without, 173.9 / 174.5 with 100, 173.1 for K = 200)
    time_steps(run, x, y, use_dist, max(args.warmup, 100))

    dog.phase = "timed region"
    wall, dev_list = time_steps(run, x, y, use_dist, args.steps)            # THE timed region
    step_dev_ms = float(np.mean(dev_list))
    multi = use_dist and world > 1
    legs = {}
    if multi or (use_dist and os.environ.get("MI355_BENCH_LEGS")):          # (the knob: N = 1 rehearsal of the two legs)
        dog.phase = "compute-only / exchange-only legs"
        for leg in ("compute", "exchange"):
            time_steps(run, x, y, use_dist, 5, leg=leg)
            w, ms = time_steps(run, x, y, use_dist, args.steps, leg=leg)
            t = torch.tensor([float(np.mean(ms)), w / args.steps * 1e3], dtype=torch.float64)
            tmax, tmin = t.clone(), t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
            legs[leg] = {"ms": float(t[0].item()), "ms_max_over_gpus": float(tmax[0].item()),
                         "ms_min_over_gpus": float(tmin[0].item()), "wall_ms_per_step": float(tmax[1].item())}
        run.execute(x, y)                                                    # (leave a whole y behind)
    dev_ms = legs["compute"]["ms"] if "compute" in legs else step_dev_ms     # this GPU's kernels
    dog.phase = "instrumented pass"
    _, step_list = time_steps(run, x, y, use_dist, args.steps, per_step=True, leg="compute" if legs else "step")   # spread of single executes
    tmax = torch.tensor([wall], dtype=torch.float64)
    nnz_all = torch.tensor([float(run.nnz_local)], dtype=torch.float64)
    v, o = m.Ax.element_size(), m.Ap.element_size()
    bytes_alg = run.nnz_local * (v + 4) + (run.n_rows_local + 1) * o + run.n_rows_local * v + m.n_cols * v
    frac = torch.tensor([bytes_alg / (max(dev_ms, 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBPS], dtype=torch.float64)
    frac_min, frac_max = frac.clone(), frac.clone()
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(nnz_all, op=dist.ReduceOp.SUM)
        dist.all_reduce(frac_min, op=dist.ReduceOp.MIN)
        dist.all_reduce(frac_max, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
    total_nnz = float(nnz_all.item())

    others = {}
    if args.all_kinds:
        dog.phase = "--all-kinds"
        for k, r in runs.items():
            _, ms = time_steps(r, x, y, use_dist, args.steps)
            ms = float(np.mean(ms))
            others[k] = {"kernel_ms": ms, "gflops": 2.0 * r.nnz_local / ms / 1e6, "gbps": bytes_alg / ms / 1e6}

    if rank == 0:
        dog.phase = "report"
        strong = bool(cuts["kind_shape"])
        info = run.info()
        achieved = bytes_alg / (dev_ms * 1e-3) / 1e9
        traffic, traffic_source = read_traffic(info["main_kernel"], bytes_alg)
        step_ms = wall / args.steps * 1e3
        out = {
            # BASELINE.json's metric, verbatim; `value` is its GFLOP/s part, the achieved HBM GB/s part
            # is `achieved_hbm_gbps` / `roofline.achieved`
            "metric": BASELINE_METRIC if m.Ax.dtype == torch.float32 else BASELINE_METRIC.replace("fp32", "fp64"),
            "value": 2.0 * total_nnz * args.steps / wall / 1e9,
            "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "warmup_executed": max(args.warmup, 100),       # untimed executes of the timed kind right before the region (>= W)
            "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32" if m.Ax.dtype == torch.float32 else "f64",
            "data": "file" if args.mtx else "synthetic",
            "config": {"workload": ("%s: %d rows, %d nnz cut into %d nnz-balanced row blocks (%d per GPU), %s offsets%s"
                                    % (m.name, m.n_rows, m.nnz, world * sub_blocks, sub_blocks,
                                       "i32" if m.Ap.dtype == torch.int32 else "i64", "" if args.mtx else ", seeded")) if strong else
                                   ("%s: %d rows/GPU x %d GPU, %d nnz/GPU, %s offsets%s"
                                    % (m.name, run.n_rows_local, world, run.nnz_local,
                                       "i32" if m.Ap.dtype == torch.int32 else "i64",
                                       ", Matrix Market file through the product loader" if args.mtx else ", seeded")),
                       "kind": kind, "lanes_per_row": info["lanes_per_row"], "grid_blocks": info["grid_blocks"],
                       "kernels_per_step": info["n_kernels"], "x_window_elems": info["window_elems"],
                       "x_window_segments": info["window_segments"], "reuse_structure": bool(args.reuse_structure),
                       "knobs": info["knobs"],
                       "parallelism": ("%d GPU x %d row blocks, x replicated, allgatherv(y): %s" % (world, sub_blocks, run.exchange))
                                      if use_dist else "single GPU"},
            "achieved_hbm_gbps": achieved,
            # this GPU's kernels by HIP events (N > 1: the compute-only leg), the step with its exchange beside it
            "compute_only": {"ms": dev_ms, "gflops_per_gpu": 2.0 * run.nnz_local / dev_ms / 1e6,
                             "step_ms_incl_exchange": step_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": info["main_kernel"], "kernel_ms": dev_ms,
                         "kernel_ms_min": float(np.min(step_list)), "kernel_ms_median": float(np.median(step_list)),
                         "kernel_ms_mean_instrumented": float(np.mean(step_list)),
                         "algorithmic_bytes": bytes_alg,
                         "timed": ("kernel_ms = one pair of HIP events around the K executes of the timed region / K, on the "
                                   "launch stream; min / median / mean_instrumented = a second pass of K executes with a "
                                   "pair around each" if _FLUSH["buf"] is None else
                                   "HIP events around every execute of the timed region (cold mode)") +
                                  (" (N > 1: the compute-only leg of this GPU, its own bytes; the step with the exchange is "
                                   "ms_per_step)" if legs else "")},
            "warmup_probe_ms": probe,
        }
        if not use_dist:
            # what the library itself would run for a caller with no opinion (MI355_KIND_AUTO: a rule on the structure,
            # no timing) — reported next to the measured pick
            try:
                pa = sp.Plan("auto", m.n_rows, m.n_cols, m.nnz, m.Ap, m.Aj, m.Ax.dtype)
                out["library_auto_kind"] = sp.capi.KIND_NAMES[pa.info()["kind"]]
                pa.destroy()
            except Exception as e:            # (never fails the bench line)
                out["library_auto_kind"] = "error: %s" % e
        if probe:
            out["kind_pick"] = "every kind warmed for %d executes, then 7 interleaved rounds of 10; median per kind%s" % (
                max(args.warmup, 70), ", max over ranks" if multi else "")
        if legs:
            ex = legs["exchange"]
            out["compute_only"].update({"ms_max_over_gpus": legs["compute"]["ms_max_over_gpus"],
                                        "ms_min_over_gpus": legs["compute"]["ms_min_over_gpus"],
                                        "wall_ms_per_step": legs["compute"]["wall_ms_per_step"]})
            out["exchange_only"] = {"ms": ex["ms"], "ms_max_over_gpus": ex["ms_max_over_gpus"],
                                    "wall_ms_per_step": ex["wall_ms_per_step"],
                                    "y_bytes_received_per_gpu": int((n_rows_global - run.n_rows_local) * v),
                                    "gbps_in_per_gpu": (n_rows_global - run.n_rows_local) * v / max(ex["ms_max_over_gpus"], 1e-9) / 1e6}
            out["exchange_ms"] = ex["ms_max_over_gpus"]
            out["overlap"] = {"step_ms": step_ms,
                              "compute_plus_exchange_ms": legs["compute"]["ms_max_over_gpus"] + ex["ms_max_over_gpus"],
                              "hidden_ms": legs["compute"]["ms_max_over_gpus"] + ex["ms_max_over_gpus"] - step_ms}
            out["roofline"]["frac_min_over_gpus"] = float(frac_min.item())
            out["roofline"]["frac_max_over_gpus"] = float(frac_max.item())
            out["gflops_compute_only_all_gpus"] = 2.0 * total_nnz / max(legs["compute"]["ms_max_over_gpus"], 1e-9) / 1e6
        if run.dist_info:
            out["exchange"] = run.dist_info
        if checks:
            out["exchange_check"] = checks[kind]
        if others:
            out["all_kinds"] = others
        if world == 1:
            out["one_shot_ms"] = one_shot_ms(sp, kind, m, x, y)                     # repeated calls (plan found again)
            out["one_shot_first_ms"] = one_shot_ms(sp, kind, m, x, y, first=True)   # a first call: plan created
            sp.capi.cache_release()
        if world == 1 and not args.no_cpu_baseline:
            dog.phase = "cpu baseline"
            dog.deadline += args.cpu_seconds + 120
            out["cpu_baseline"] = cpu_baseline(m, x, args.cpu_seconds)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    dog.phase = "teardown"
    for r in runs.values():
        r.destroy()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    dog.done = True


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:      # noqa: BLE001 - a run that dies still ends in ONE JSON line (rank 0) and a non-zero code
        import traceback
        traceback.print_exc()
        if _STATE.get("rank", 0) == 0:
            line = error_line("bench.py: %s: %s" % (type(e).__name__, str(e)[:400]), _STATE.get("world", 1), _STATE.get("args"))
            os.write(_STATE.get("json_fd", 1), (line + "\n").encode())
        os._exit(1)                 # (not sys.exit: a rank stuck in a collective's destructor must not outlive the error)
