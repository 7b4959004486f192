"""Row-block sharding of one CSR SpMV across the GPUs of a node.

The reference is single-GPU (main.cu:53, common.cuh:8; no collective anywhere).
Rows of y = A x are independent (cpu_navie.hpp:9-16), so the matrix is cut into P
contiguous row blocks, one per rank (one process per GPU); x is replicated and the
y slices are concatenated with an allgatherv over RCCL/xGMI — SURVEY.md §8(e).

  partition_rows   nnz-balanced cut points from Ap (binary search), so R-MAT-like
                   skew does not leave one GPU with most of the nonzeros
  shard_csr        rank p's slice: Ap rebased to 0, Aj/Ax slice (global column
                   ids kept), so the slice is an ordinary CSR matrix for the C ABI
  allgatherv       RCCL has no allgatherv: equal counts -> one all_gather into the
                   full vector; unequal -> one broadcast per root into its
                   displacement (grouped, asynchronous), the pattern SURVEY §5 names

Backend-agnostic (nccl = RCCL on ROCm for GPUs, gloo for the CPU tests).
"""
import torch
import torch.distributed as dist


def partition_rows(Ap, parts, balance="nnz", unit=None, table=None):
    """Cut points r_0=0 <= r_1 <= ... <= r_P=n_rows.
    balance="nnz": Ap[r_p] ~ p*nnz/P;  "rows": equal row counts.
    unit / table: the rule of the library's mi355_spmv_plan_partition (csrc/analyze.hip, partition_kernel),
    restated with torch so that it also runs on the CPU: a cut may only fall on a UNIT boundary — row
    u * unit (uniform chunks of `unit` rows; 4 for the merge kind), or table[u] (the chunk table of a
    weight-cut plan) — and cut p is the first unit whose first row starts at or after nonzero
    Ap[0] + (Ap[n] - Ap[0]) * p // P.  Blocks then own whole chunks of the whole matrix's plan."""
    n_rows = Ap.numel() - 1
    if balance == "rows":
        return [n_rows * p // parts for p in range(parts + 1)]
    if unit is not None or table is not None:
        if table is not None:
            rows = torch.as_tensor(table, dtype=torch.int64, device=Ap.device)
        else:
            n_units = (n_rows + unit - 1) // unit
            rows = torch.clamp(torch.arange(n_units + 1, dtype=torch.int64, device=Ap.device) * unit, max=n_rows)
        starts = Ap.to(torch.int64)[rows]                     # Ap at every unit boundary (ascending)
        first, last = int(Ap[0].item()), int(Ap[-1].item())
        targets = torch.tensor([first + (last - first) * p // parts for p in range(1, parts)], dtype=torch.int64,
                               device=Ap.device)
        u = torch.searchsorted(starts.contiguous(), targets, right=False).clamp_(max=rows.numel() - 1)
        cuts = [0] + rows[u].tolist() + [n_rows] if parts > 1 else [0, n_rows]
        for i in range(1, len(cuts)):
            cuts[i] = max(cuts[i], cuts[i - 1])
        return cuts
    nnz = int(Ap[-1].item())
    targets = torch.tensor([nnz * p // parts for p in range(1, parts)], dtype=Ap.dtype, device=Ap.device)
    mid = torch.searchsorted(Ap.contiguous(), targets, right=False).tolist() if parts > 1 else []
    cuts = [0] + [min(int(m), n_rows) for m in mid] + [n_rows]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts


def shard_csr(Ap, Aj, Ax, r0, r1):
    """Rows [r0, r1) as a stand-alone CSR (local Ap starts at 0, global column ids)."""
    base = Ap[r0]
    lo, hi = int(base.item()), int(Ap[r1].item())
    Ap_l = (Ap[r0:r1 + 1] - base).contiguous()
    # .clone(): a fresh allocation is 256-byte aligned, which the 16-byte-per-lane
    # loads of the kernels rely on (an offset view of the parent array is not)
    return Ap_l, Aj[lo:hi].clone(), Ax[lo:hi].clone()


def block_view(Ap, Aj, Ax, r0, r1):
    """Rows [r0, r1) as a 16-byte-aligned VIEW of the parent arrays — the form
    mi355_spmv_plan_create_block / mi355_spmv_dist_create_rank take: Aj / Ax start at element
    lo = Ap[r0] & ~3 (no copy), Ap_l[i] = Ap[r0 + i] - lo, so Ap_l[0] is the block's phase (0..3) and
    Ap_l[-1] the END offset.  Every row keeps the position it has in the parent modulo 4, which is what
    the 16-byte sweeps of the row-local kernels sum by.  Returns (Ap_l, Aj_view, Ax_view, lo)."""
    lo = int(Ap[r0].item()) & ~3
    hi = int(Ap[r1].item())
    Ap_l = (Ap[r0:r1 + 1] - lo).contiguous()
    return Ap_l, Aj[lo:hi], Ax[lo:hi], lo


class _Works:
    """Handle for an in-flight allgatherv: wait() blocks the current stream until it is done."""

    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()

    def done(self):
        """Host-side query: True once every piece has finished on the device.  A finished exchange
        needs no stream-level wait (on ROCm a cross-stream event wait costs the waiting stream tens of
        microseconds even when the event has long fired)."""
        try:
            return all(w.is_completed() for w in self.works)
        except Exception:
            return False


def allgatherv(y_local, y_full, cuts, group=None, async_op=False):
    """y_full[cuts[p]:cuts[p+1]] <- rank p's y_local, on every rank.
    async_op=True returns a handle (wait() before y_local is overwritten or y_full is read):
    the exchange then runs beside whatever the caller launches next (bench.py overlaps it
    with the following SpMV, double-buffering y)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [cuts[p + 1] - cuts[p] for p in range(world)]
    assert y_local.numel() == counts[rank]
    if len(set(counts)) == 1 and y_full.numel() == counts[0] * world:
        w = dist.all_gather_into_tensor(y_full, y_local, group=group, async_op=True)
        if async_op:
            return _Works([w])
        w.wait()
        return y_full
    y_full[cuts[rank]:cuts[rank + 1]].copy_(y_local)
    works = []
    for p in range(world):
        if counts[p] == 0:
            continue
        src = dist.get_global_rank(group, p) if group is not None else p
        works.append(dist.broadcast(y_full[cuts[p]:cuts[p + 1]], src=src, group=group, async_op=True))
    if async_op:
        return _Works(works)
    for w in works:
        w.wait()
    return y_full


# ---- the three allgatherv exchanges of csrc/dist.hip, restated ------------------------------------------------
# mi355_spmv_dist_* makes the allgatherv of sub-block s in one of three ways (include/mi355_spmv.h,
# MI355_DIST_EXCHANGE_*).  exchange_schedule is the same count / displacement logic as plain data — what each rank
# calls, in order — so that it can be checked without a GPU (tests/test_dist_cpu.py runs it over gloo with 2 and 3
# ranks) and against the calls the library really makes (tests/test_gpu_dist_multi.py reads them back from the
# emulated RCCL).  Block g = rank * sub_blocks + s holds rows [row_cuts[g], row_cuts[g + 1]).
def exchange_schedule(row_cuts, world, sub_blocks, s, mode, rank):
    """The calls of `rank` for sub-block s: a list of tuples
         ("bcast", root, first_row, count)            in-place broadcast of block (root, s)
         ("send", peer, first_row, count) / ("recv", peer, first_row, count)
         ("allgather_in_place", first_row_of_rank_0s_block, count)
         ("pack", first_row, count, slot) ("allgather_padded", pad) ("unpack", first_row, count, slot) ...
    """
    blk = lambda r: r * sub_blocks + s
    rows = lambda r: row_cuts[blk(r) + 1] - row_cuts[blk(r)]
    if mode == "bcast":
        return [("bcast", r, row_cuts[blk(r)], rows(r)) for r in range(world) if rows(r) > 0]
    if mode == "sendrecv":
        out = []
        for r in range(world):
            if r == rank:
                continue
            if rows(rank) > 0:
                out.append(("send", r, row_cuts[blk(rank)], rows(rank)))
            if rows(r) > 0:
                out.append(("recv", r, row_cuts[blk(r)], rows(r)))
        return out
    if mode == "allgather":
        cnt = rows(0)
        if cnt > 0 and all(rows(r) == cnt and row_cuts[blk(r)] == row_cuts[blk(0)] + r * cnt for r in range(world)):
            return [("allgather_in_place", row_cuts[blk(0)], cnt)]
        widest = max(row_cuts[g + 1] - row_cuts[g] for g in range(world * sub_blocks))
        pad = (widest + 3) & ~3
        out = []
        if rows(rank) > 0:
            out.append(("pack", row_cuts[blk(rank)], rows(rank), rank))
        out.append(("allgather_padded", pad))
        out += [("unpack", row_cuts[blk(r)], rows(r), r) for r in range(world) if r != rank and rows(r) > 0]
        return out
    raise ValueError(mode)


def exchange_sub_block(y_full, row_cuts, world, sub_blocks, s, mode, group=None):
    """Run exchange_schedule with torch.distributed on this rank's full-length y (blocking)."""
    rank = dist.get_rank(group)
    sched = exchange_schedule(row_cuts, world, sub_blocks, s, mode, rank)
    works, stage = [], None
    for op in sched:
        if op[0] == "bcast":
            _, root, first, cnt = op
            works.append(dist.broadcast(y_full[first:first + cnt], src=root, group=group, async_op=True))
        elif op[0] == "send":
            _, peer, first, cnt = op
            works.append(dist.isend(y_full[first:first + cnt].contiguous(), dst=peer, group=group))
        elif op[0] == "recv":
            _, peer, first, cnt = op
            works.append(dist.irecv(y_full[first:first + cnt], src=peer, group=group))
        elif op[0] == "allgather_in_place":
            _, first, cnt = op
            out = y_full[first:first + world * cnt]
            dist.all_gather_into_tensor(out, out[rank * cnt:(rank + 1) * cnt].clone(), group=group)
        elif op[0] == "pack":
            _, first, cnt, slot = op
            pad = next(o[1] for o in sched if o[0] == "allgather_padded")
            stage = torch.zeros(world * pad, dtype=y_full.dtype, device=y_full.device)
            stage[slot * pad:slot * pad + cnt] = y_full[first:first + cnt]
        elif op[0] == "allgather_padded":
            pad = op[1]
            if stage is None:
                stage = torch.zeros(world * pad, dtype=y_full.dtype, device=y_full.device)
            dist.all_gather_into_tensor(stage, stage[rank * pad:(rank + 1) * pad].clone(), group=group)
        elif op[0] == "unpack":
            _, first, cnt, slot = op
            pad = stage.numel() // world
            y_full[first:first + cnt] = stage[slot * pad:slot * pad + cnt]
    for w in works:
        w.wait()
    return y_full
