// merge_path.hip — kind MERGE: merge-path load-balanced CSR SpMV for gfx950.
//
// Replaces the reference's vendored CUB pipeline:
//   search   include/spmv/merge_based/thread_search.cuh:15-49,
//            dispatch_spmv_orig.cuh:104-148            -> merge_search_kernel
//   tile     include/spmv/merge_based/agent_spmv_orig.cuh:454-757,
//            dispatch_spmv_orig.cuh:154-191            -> merge_tile_kernel
//   fix-up   include/spmv/merge_based/agent_segment_fixup.cuh:97-384,
//            dispatch_spmv_orig.cuh:197-229            -> merge_fixup_kernel
// and the offset_t-typed twin include/spmv/merge_genl/ (the only reference merge
// variant that accepts 64-bit offsets): both offset widths are instantiated here.
//
// The decomposition is the reference's: the merge of the row-end offsets
// Ap[1..n_rows] with the counting sequence 0..nnz-1 is cut into tiles of
// TILE = 256 * IPT items; a tile owns the rows that END inside it and the
// nonzeros inside it, whatever the row lengths.  What is rebuilt for MI355X:
//  * nonzeros are streamed with 16-byte-per-lane loads from the tile start
//    rounded down to a multiple of 4 (the reference reads 4 bytes per lane,
//    strided by the block, agent_spmv_orig.cuh:474-506);
//  * row ends are staged tile-relative as 32-bit integers whatever offset_t is,
//    so a 64-bit-offset matrix costs no extra LDS;
//  * the walk bounds each thread by the tile's real item count instead of
//    padding the row-end list and clamping the carry (agent_spmv_orig.cuh:543-548,
//    :744-753);
//  * the reference's cub::BlockScan of (key, value) pairs with ReduceByKeyOp
//    (agent_spmv_orig.cuh:616-629) becomes a flag-segmented wave64 scan
//    (six __shfl_up steps) plus a 4-entry cross-wave pass, written here;
//  * rows finished by a thread go straight to an LDS partial array during the
//    walk (no per-item key/value register arrays), then to y with coalesced stores;
//  * the fix-up is deterministic: one thread per distinct carried row sums that
//    row's tile carries in tile order and adds once, instead of float atomics
//    (agent_segment_fixup.cuh:228-271), so results do not change
//    from run to run;
//  * tile ids are remapped so each XCD walks a contiguous range of the matrix.
//
// n_cols == 1 (the reference's special kernel, dispatch_spmv_orig.cuh:68-96,
// :572-597) needs no special case here.

#include <climits>

#include "common.hpp"
#include "row_dot.hpp"

namespace mi355 {

// ---- K6: tile start coordinates ------------------------------------------------
template <typename off_t>
__global__ __launch_bounds__(kBlock) void merge_search_kernel(
    int32_t n_rows, int64_t nnz, const off_t* __restrict__ Ap, int64_t tile_items, int64_t n_tiles,
    int32_t* __restrict__ tile_row, int64_t* __restrict__ tile_nnz) {
    const int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (t > n_tiles) return;
    const int64_t items = int64_t(n_rows) + nnz;
    int64_t diag = t * tile_items;
    if (diag > items) diag = items;
    int64_t lo = diag - nnz > 0 ? diag - nnz : 0;
    int64_t hi = diag < n_rows ? diag : n_rows;
    while (lo < hi) {
        const int64_t p = (lo + hi) >> 1;
        if (int64_t(Ap[p + 1]) <= diag - p - 1) lo = p + 1;
        else hi = p;
    }
    tile_row[t] = int32_t(lo);
    tile_nnz[t] = diag - lo;
}

// ---- K7: one tile per workgroup ---------------------------------------------------
template <int IPT, bool VEC, typename off_t, typename val_t>
__global__ __launch_bounds__(kBlock) void merge_tile_kernel(
    int32_t n_rows, int64_t nnz, const off_t* __restrict__ Ap, const int32_t* __restrict__ Aj,
    const val_t* __restrict__ Ax, const val_t* __restrict__ x, val_t* __restrict__ y,
    const int32_t* __restrict__ tile_row, const int64_t* __restrict__ tile_nnz,
    int32_t* __restrict__ carry_row, val_t* __restrict__ carry_val) {
    constexpr int TILE = kBlock * IPT;
    using v4 = typename Vec4<val_t>::type;
    __shared__ __attribute__((aligned(32))) val_t s_nz[TILE + 4];  // products, index = nnz - (y0 & ~3)
    __shared__ int s_re[TILE + 1];                                 // tile-relative row ends
    __shared__ val_t s_part[TILE];                                 // sums of the rows ending here
    __shared__ val_t s_wave_sum[kBlock / kWave];
    __shared__ int s_wave_flag[kBlock / kWave];

    const unsigned t = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int tid = threadIdx.x;
    const int lane64 = tid & (kWave - 1);
    const int wave = tid / kWave;

    const int x0 = tile_row[t], x1 = tile_row[t + 1];
    const int64_t y0 = tile_nnz[t], y1 = tile_nnz[t + 1];
    const int tr = x1 - x0;            // rows that end in this tile
    const int tn = int(y1 - y0);       // nonzeros in this tile
    const int shift = VEC ? int(y0 & 3) : 0;
    const int64_t yb = y0 - shift;     // 16-byte aligned start of the stream

    // (1) products a*x for the tile's nonzeros, 4 per lane per load
    if constexpr (!VEC) {
        // Aj / Ax not 16-byte aligned (an offset view): 4-byte-per-lane form
        for (int i = tid; i < tn; i += kBlock) s_nz[i] = Ax[y0 + i] * x[Aj[y0 + i]];
    } else
    for (int g = tid; 4 * g < tn + shift; g += kBlock) {
        const int64_t j = yb + 4 * int64_t(g);
        int4v c;
        v4 a;
        if (j + 4 <= nnz) {
            c = stream_load(reinterpret_cast<const int4v*>(Aj + j));
            a = stream_load(reinterpret_cast<const v4*>(Ax + j));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = (j + e) < nnz;
                c[e] = in ? Aj[j + e] : 0;
                a[e] = in ? Ax[j + e] : val_t(0);
            }
        }
        v4 p;
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = a[e] * x[c[e]];
        *reinterpret_cast<v4*>(&s_nz[4 * g]) = p;
    }
    // (2) row ends, relative to y0; the row still open at the tile end never ends here
    for (int i = tid; i <= tr; i += kBlock) {
        s_re[i] = (i < tr) ? int(int64_t(Ap[int64_t(x0) + i + 1]) - y0) : INT_MAX;
    }
    __syncthreads();

    // (3) this thread's piece of the merge path: items [d0, d1) of the tile
    const int items = tr + tn;
    const int d0 = min(tid * IPT, items);
    const int d1 = min(d0 + IPT, items);
    int lo = max(d0 - tn, 0), hi = min(d0, tr);
    while (lo < hi) {
        const int p = (lo + hi) >> 1;
        if (s_re[p] <= d0 - p - 1) lo = p + 1;
        else hi = p;
    }
    int cx = lo, cy = d0 - lo;

    // (4) walk: a nonzero extends the running row sum, a row end closes it
    val_t run = val_t(0), first_val = val_t(0);
    int first_end = -1;
    int re = s_re[cx];
    const int cnt = d1 - d0;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        if (k < cnt) {
            if (cy < re) {
                run += s_nz[cy + shift];
                ++cy;
            } else {
                if (first_end < 0) {
                    first_end = cx;      // may continue a row opened by earlier threads
                    first_val = run;
                } else {
                    s_part[cx] = run;    // opened and closed inside this thread
                }
                run = val_t(0);
                ++cx;
                re = s_re[cx];
            }
        }
    }

    // (5) carry-in = sum of the open-row tails of the preceding threads back to
    //     the last thread that closed a row: a flag-segmented inclusive scan
    val_t sv = run;
    int sf = first_end >= 0 ? 1 : 0;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const val_t ov = __shfl_up(sv, d, kWave);
        const int of = __shfl_up(sf, d, kWave);
        if (lane64 >= d) {
            if (!sf) sv = ov + sv;
            sf |= of;
        }
    }
    if (lane64 == kWave - 1) {
        s_wave_sum[wave] = sv;
        s_wave_flag[wave] = sf;
    }
    __syncthreads();
    val_t prefix = val_t(0);  // block-inclusive value at the end of the previous wave
    for (int w = 0; w < wave; ++w) prefix = s_wave_flag[w] ? s_wave_sum[w] : prefix + s_wave_sum[w];
    const val_t incl = sf ? sv : prefix + sv;
    val_t carry_in = __shfl_up(incl, 1, kWave);
    if (lane64 == 0) carry_in = prefix;

    if (first_end >= 0) s_part[first_end] = carry_in + first_val;
    if (tid == kBlock - 1) {
        // the row still open at the end of the tile: x1 (== n_rows means none)
        carry_row[t] = x1;
        carry_val[t] = incl;
    }
    __syncthreads();

    // (6) rows that ended in this tile, coalesced
    for (int i = tid; i < tr; i += kBlock) y[int64_t(x0) + i] = s_part[i];
}

// ---- K8: add the tile carries of rows that straddle tiles ---------------------------
template <typename val_t>
__global__ __launch_bounds__(kBlock) void merge_fixup_kernel(
    int64_t n_tiles, int32_t n_rows, const int32_t* __restrict__ carry_row,
    const val_t* __restrict__ carry_val, val_t* __restrict__ y) {
    const int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (t >= n_tiles) return;
    const int32_t r = carry_row[t];
    if (r >= n_rows) return;
    if (t > 0 && carry_row[t - 1] == r) return;  // not the first tile carrying row r
    val_t s = carry_val[t];
    for (int64_t u = t + 1; u < n_tiles && carry_row[u] == r; ++u) s += carry_val[u];
    y[r] += s;
}

// ---- host side -----------------------------------------------------------------------
constexpr int kMergeIpt = 8;

void shape_merge(Plan& p) {
    p.lanes_per_row = 0;
    p.tile_items = int64_t(kBlock) * kMergeIpt;
    const int64_t items = int64_t(p.n_rows) + p.nnz;
    p.n_tiles = (items + p.tile_items - 1) / p.tile_items;
    p.grid_blocks = p.n_tiles;
    p.n_kernels = p.n_tiles > 1 ? 3 : 2;
    p.coords_valid = false;
    snprintf(p.main_kernel, sizeof(p.main_kernel), "merge_tile_kernel");
}

template <typename off_t, typename val_t>
int launch_merge(Plan& p, const off_t* Ap, const val_t* Ax, const val_t* x, val_t* y, hipStream_t s) {
    if (p.n_rows == 0 || p.n_tiles == 0) return MI355_SPMV_OK;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.Aj) | reinterpret_cast<uintptr_t>(Ax)) & 15u) == 0;
    const bool reuse = (p.flags & MI355_PLAN_REUSE_STRUCTURE) && p.coords_valid;
    if (!reuse) {
        const unsigned g = unsigned((p.n_tiles + 1 + kBlock - 1) / kBlock);
        hipLaunchKernelGGL((merge_search_kernel<off_t>), dim3(g), dim3(kBlock), 0, s, p.n_rows, p.nnz, Ap,
                           p.tile_items, p.n_tiles, p.tile_row, p.tile_nnz);
        MI355_HIP_TRY(hipGetLastError());
        p.coords_valid = true;
    }
    if (aligned) {
        hipLaunchKernelGGL((merge_tile_kernel<kMergeIpt, true, off_t, val_t>), dim3((unsigned)p.n_tiles),
                           dim3(kBlock), 0, s, p.n_rows, p.nnz, Ap, p.Aj, Ax, x, y, p.tile_row, p.tile_nnz,
                           p.carry_row, static_cast<val_t*>(p.carry_val));
    } else {
        hipLaunchKernelGGL((merge_tile_kernel<kMergeIpt, false, off_t, val_t>), dim3((unsigned)p.n_tiles),
                           dim3(kBlock), 0, s, p.n_rows, p.nnz, Ap, p.Aj, Ax, x, y, p.tile_row, p.tile_nnz,
                           p.carry_row, static_cast<val_t*>(p.carry_val));
    }
    MI355_HIP_TRY(hipGetLastError());
    if (p.n_tiles > 1) {
        const unsigned g = unsigned((p.n_tiles + kBlock - 1) / kBlock);
        hipLaunchKernelGGL((merge_fixup_kernel<val_t>), dim3(g), dim3(kBlock), 0, s, p.n_tiles, p.n_rows,
                           p.carry_row, static_cast<const val_t*>(p.carry_val), y);
        MI355_HIP_TRY(hipGetLastError());
    }
    return MI355_SPMV_OK;
}

template int launch_merge<int32_t, float>(Plan&, const int32_t*, const float*, const float*, float*, hipStream_t);
template int launch_merge<int32_t, double>(Plan&, const int32_t*, const double*, const double*, double*, hipStream_t);
template int launch_merge<int64_t, float>(Plan&, const int64_t*, const float*, const float*, float*, hipStream_t);
template int launch_merge<int64_t, double>(Plan&, const int64_t*, const double*, const double*, double*, hipStream_t);

}  // namespace mi355
