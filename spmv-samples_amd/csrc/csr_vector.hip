// csr_vector.hip — kind VECTOR: CSR-vector SpMV with per-row sub-wave reduction.
//
// Replaces the reference's CUSP CSR-vector family on the hot path
// (include/spmv/cusp/cusp_warp_reduce.cuh:11-59 kernel, :93-133 width selection,
// include/spmv/cusp/utils.cuh:38-47 shuffle tree).  Written for gfx950:
//   * 64-lane waves, 256-thread workgroups, T lanes per row (T = 2..64);
//   * 16-byte-per-lane nontemporal loads of Aj/Ax, R rows per vector in flight;
//   * a workgroup owns a CHUNK of consecutive rows (~32 K nonzeros) and stages the
//     window of x the chunk touches through LDS (xwindow.hpp) — the plain global
//     gather is what bounds this kernel on MI355X, not the Aj/Ax stream;
//   * chunk ids are remapped so each XCD walks a contiguous range of rows and
//     neighbouring windows of x hit that XCD's L2.
// The 4-byte-per-lane kernel at the bottom is the form of the reference
// (one row per vector, grid = ceil(rows / vectors per block), cusp_warp_reduce.cuh:70-87);
// it is used only when Aj/Ax/x are not 16-byte aligned.

#include <cstdlib>

#include "common.hpp"
#include "row_dot.hpp"
#include "giant_rows.hpp"
#include "xwindow.hpp"

namespace mi355 {

// (Registers: 512-thread workgroups and the R = 2 bodies — fp64, and fp32 rows of 33+ nonzeros — are held to 128
// VGPRs, i.e. two / four workgroups per CU; the 256-thread fp32 R = 4 body needs ~135 and gets 168: three per CU,
// which is what its 36 KB window of x allows anyway.  shape_chunks sizes a small matrix's single round of chunks
// by the same numbers: a kernel that silently needs a few more registers than its plan assumed loses 30-40 %
// there — cant stand-in: 14.7 -> 19-21 us when its body went from 127 to 139 VGPRs.)  The kernel does not depend on the width
// of the row offsets: a chunk is walked with 32-bit offsets relative to its own first nonzero (xwindow.hpp).
template <int BLOCK, int T, int R, int NSEG, bool ADAPT, typename val_t>
__global__ __launch_bounds__(BLOCK, (BLOCK >= kWideBlock || R == 2 ? 4 : 3)) void csr_vector_window_kernel(
    int32_t n_rows, int32_t n_cols, int64_t nnz, const ApView Ap, const int32_t* __restrict__ Aj,
    const val_t* __restrict__ Ax, const val_t* __restrict__ x, val_t* __restrict__ y, ChunkMap cmap,
    int32_t window_cap, BandHint hint, SegmentPlan segs, val_t alpha, val_t beta) {
    // NSEG: 0 = no window (plain gathers), 1 = one window of x in LDS, kMaxSegments = several bands
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];   // window | bounds | y | flags
    __shared__ int s_red[2];
    ChunkScratch<val_t> scr(s_dyn, window_cap, cmap.rows_cap);
    scr.alpha = alpha;
    scr.beta = beta;
    scr.long_steps = cmap.long_steps;
    scr.giant_len = cmap.giant_len;
    // Equal-row chunks: each XCD walks a contiguous range (neighbouring windows of x hit that L2).  Weight-cut
    // chunks (power-law matrices, no window to share): chunk = block index, i.e. consecutive chunks go to
    // different XCDs — the chunks of the dense head of the matrix and those of its near-empty tail take very
    // different times at equal weight, and a contiguous eighth per XCD leaves the XCDs unevenly loaded.
    const unsigned chunk = cmap.table ? blockIdx.x : xcd_contiguous_id(blockIdx.x, gridDim.x);
    int64_t rb, re;
    cmap.range(chunk, n_rows, rb, re);
    if (rb >= re) return;   // (balanced plans: a hub row heavier than a chunk leaves empty chunks behind it)
    bool fits;
    const int64_t base = stage_chunk_bounds<val_t>(scr, rb, re, Ap, cmap.rel_limit, fits);
    if (!fits) {            // (uniform) more nonzeros than 32-bit chunk-relative offsets reach
        chunk_rows_wide<BLOCK, val_t>(rb, re, Ap, Aj, Ax, x, y, alpha, beta, cmap.giant_len);
        return;
    }
    __syncthreads();
    const int32_t* const Aj_c = Aj + base;       // the chunk's view: element 0 = its first 16-byte group
    const val_t* const Ax_c = Ax + base;
    const int64_t left = nnz - base;
    const int32_t nnz_c = int32_t(left < kRel32Limit + 32768 ? left : kRel32Limit + 32768);
    // the window is staged inside chunk_rows, behind the first group's stream loads
    if constexpr (NSEG > 1) {
        auto stage = [&] { return stage_x_segments<val_t>(rb, re, n_cols, x, scr.s_x, window_cap, segs); };
        chunk_rows_any<BLOCK, T, R, true, ADAPT, val_t>(rb, re, nnz_c, Aj_c, Ax_c, x, y, stage, scr);
    } else {
        auto first_last = [&](int64_t r, int& first, int& last) {
            const int32_t s = scr.s_b[r - rb], e = scr.s_b[r - rb + 1];
            if (e <= s) return false;
            first = Aj_c[s];
            last = Aj_c[e - 1];
            return true;
        };
        auto stage = [&] {
            return stage_x_window<val_t>(rb, re, n_cols, first_last, x, scr.s_x, window_cap, s_red, hint);
        };
        chunk_rows_any<BLOCK, T, R, NSEG == 1, ADAPT, val_t>(rb, re, nnz_c, Aj_c, Ax_c, x, y, stage, scr);
    }
}

// The band is wider than any window of x: one 1 024-thread workgroup per CU, a chunk = one group of rows held in
// registers, the window sweeps the band (xwindow.hpp, chunk_rows_sweep).
template <int T, int R, typename val_t>
__global__ __launch_bounds__(kHugeBlock, 4) void csr_vector_sweep_kernel(
    int32_t n_rows, int32_t n_cols, int64_t nnz, const ApView Ap, const int32_t* __restrict__ Aj,
    const val_t* __restrict__ Ax, const val_t* __restrict__ x, val_t* __restrict__ y, ChunkMap cmap,
    int32_t window_cap, BandHint hint, val_t alpha, val_t beta) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];   // window | bounds | y | flags
    ChunkScratch<val_t> scr(s_dyn, window_cap, cmap.rows_cap);
    scr.alpha = alpha;
    scr.beta = beta;
    const unsigned chunk = xcd_contiguous_id(blockIdx.x, gridDim.x);   // (an XCD's chunks sweep neighbouring columns: its L2 holds them)
    int64_t rb, re;
    cmap.range(chunk, n_rows, rb, re);
    if (rb >= re) return;
    bool fits;
    const int64_t base = stage_chunk_bounds<val_t>(scr, rb, re, Ap, cmap.rel_limit, fits);
    if (!fits) {
        chunk_rows_wide<kHugeBlock, val_t>(rb, re, Ap, Aj, Ax, x, y, alpha, beta, 0);
        return;
    }
    __syncthreads();
    const int64_t left = nnz - base;
    const int32_t nnz_c = int32_t(left < kRel32Limit + 32768 ? left : kRel32Limit + 32768);
    // (a persistent workgroup per CU walking its share of the chunks measured WORSE, 194 vs 187 us at two passes,
    // 438 vs 358 at seven: the hardware dispatcher's refill costs less than the registers the loop does)
    chunk_rows_sweep<kHugeBlock, T, R, val_t>(rb, re, nnz_c, Aj + base, Ax + base, x, y, n_cols, window_cap, hint, scr);
}

template <int T, typename off_t, typename val_t>
__global__ __launch_bounds__(kBlock) void csr_vector_kernel(
    int32_t n_rows, off_t nnz, const off_t* __restrict__ Ap, const int32_t* __restrict__ Aj,
    const val_t* __restrict__ Ax, const val_t* __restrict__ x, val_t* __restrict__ y, val_t alpha, val_t beta) {
    constexpr int ROWS_PER_BLOCK = kBlock / T;
    const unsigned blk = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & (T - 1);
    const int64_t row = int64_t(blk) * ROWS_PER_BLOCK + (threadIdx.x / T);
    const bool live = row < n_rows;
    // a vector past the last row runs as an empty row so that every lane of the
    // wave reaches the shuffles below
    off_t start = 0, end = 0;
    if (live) {
        start = Ap[row];
        end = Ap[row + 1];
    }
    val_t sum = row_partial<T, off_t, val_t>(start, end, lane, Aj, Ax, x);
    sum = vector_reduce<T, val_t>(sum);
    if (live && lane == 0) y[row] = (beta != val_t(0)) ? alpha * sum + beta * y[row] : alpha * sum;
}

template <typename val_t> constexpr int rows_in_flight() { return sizeof(val_t) == 4 ? 4 : 2; }

#ifndef MI355_TU_F64   // the host-side shape functions live in the fp32 translation unit only
void shape_vector(Plan& p) {
    p.lanes_per_row = pick_lanes_per_row(p.nnz - p.nnz_begin, p.n_rows, p.elems_per_lane);
    const int R = p.val_type == MI355_VAL_F64 ? rows_in_flight<double>() : rows_in_flight<float>();
    {                                                          // tuning knob
        const int t = p.knob.lanes;
        if (t == 2 || t == 4 || t == 8 || t == 16 || t == 32 || t == 64) p.lanes_per_row = t;
    }
    shape_chunks(p, R, 1, true, true);   // (analyze.hip)
    p.grid_blocks = (int64_t(p.n_rows) + p.rows_per_chunk - 1) / p.rows_per_chunk;
    if (p.grid_blocks < 1) p.grid_blocks = 1;
    p.n_tiles = p.grid_blocks;
    p.n_kernels = 1;
    snprintf(p.main_kernel, sizeof(p.main_kernel), "csr_vector_window_kernel");
}

// after decide_balance: one workgroup per chunk, window sized for the rows a chunk may hold
void reshape_vector_balanced(Plan& p) {
    if (!p.balanced) return;
    p.block_threads = kBlock;          // (weight-cut chunks are sized for 256 threads)
    p.window_bytes = kWindowBytes;
    p.grid_blocks = p.n_chunks;
    p.n_tiles = p.n_chunks;
    p.window_elems = pick_window_elems(p, p.rows_cap);
    if (p.n_seg >= 2) { p.window_elems = 0; p.n_seg = 0; }   // (the multi-band plan is sized for uniform chunks)
}

void block_grid_vector(Plan& p) {
    p.grid_blocks = p.n_chunks;
    p.n_tiles = p.n_chunks;
    snprintf(p.main_kernel, sizeof(p.main_kernel), p.sweep ? "csr_vector_sweep_kernel" : "csr_vector_window_kernel");
}

#endif  // MI355_TU_F64

template <int BLOCK, typename val_t>
static int launch_vector_window(const Plan& p, const ApView Ap, const val_t* Ax, const val_t* x, val_t* y,
                                hipStream_t s) {
    if constexpr (BLOCK >= kWideBlock) {
        // a 512- / 1 024-thread plan is only ever shaped around ONE window of x; without it (a forced knob) the
        // 256-thread kernel walks the same chunks (any workgroup size walks any chunk)
        if (p.window_elems <= 0 || p.n_seg >= 2) return launch_vector_window<kBlock, val_t>(p, Ap, Ax, x, y, s);
    }
    constexpr int R = rows_in_flight<val_t>();
    const BandHint hint{p.band_lo, p.band_hi, p.window_from_band};
    const dim3 grid((unsigned)p.grid_blocks), block(BLOCK);
    const int64_t nnz = p.nnz_read;
    const size_t lds = chunk_lds_bytes(p.window_elems, p.rows_cap, sizeof(val_t));
    const ChunkMap cmap{p.balanced ? p.chunk_row : nullptr, (int32_t)p.rows_per_chunk, (int32_t)p.rows_cap, p.n_chunks,
                        long_steps_for(p), p.n_giant > 0 ? p.giant_len : int64_t(0),
                        p.knob.rel32_limit > 0 ? p.knob.rel32_limit : kRel32Limit, 0};
    SegmentPlan segs;
    segs.n = p.n_seg;
    for (int i = 0; i < kMaxSegments; ++i) { segs.lo[i] = p.seg_lo[i]; segs.hi[i] = p.seg_hi[i]; }
#define MI355_VEC_ARGS s, p.n_rows, p.n_cols, nnz, Ap, p.Aj, Ax, x, y, cmap, (int32_t)p.window_elems, hint, segs, (val_t)p.alpha, (val_t)p.beta
    // rows a vector keeps in flight: fp32 with 16 or more lanes per row (rows of 33+ nonzeros) runs with 2 instead of
    // 4 - a long row keeps its lanes' loads busy by itself, and the body then needs ~95 VGPRs instead of ~135 (four
    // 256-thread workgroups per CU instead of three: what a small matrix's single round of chunks is sized for)
    constexpr auto wide_r = [](int tt) constexpr { return (sizeof(val_t) == 4 && tt >= 16) ? 2 : R; };
#define MI355_VEC_LAUNCH(TT, NSEG_, ADAPT_)                                                                   \
    do {                                                                                                      \
        if (const int st = allow_dynamic_lds((const void*)csr_vector_window_kernel<BLOCK, TT, (wide_r(TT)), NSEG_, ADAPT_, val_t>, lds)) return st; \
        hipLaunchKernelGGL((csr_vector_window_kernel<BLOCK, TT, (wide_r(TT)), NSEG_, ADAPT_, val_t>), grid, block, lds, MI355_VEC_ARGS); \
    } while (0)
#define MI355_VEC_CASE(TT)                                                                                    \
    case TT:                                                                                                  \
        if (p.window_elems > 0 && p.n_seg >= 2) {                                                             \
            /* (several bands: shape_chunks keeps those plans on 256 threads) */                              \
            if constexpr (BLOCK == kBlock) MI355_VEC_LAUNCH(TT, kMaxSegments, false);                         \
            else { set_error("csr_vector: no 512-thread kernel for a multi-band window"); return MI355_SPMV_EINVAL; } \
        }                                                                                                     \
        else if (p.window_elems > 0) MI355_VEC_LAUNCH(TT, 1, false);                                          \
        else if constexpr (BLOCK == kBlock) MI355_VEC_LAUNCH(TT, 0, false);                                   \
        break;
    if constexpr (BLOCK == kBlock) if (p.balanced) {   // vector width per chunk (chunk_rows_any); the T of the template is not used
        // (the weight-cut layout holds up to 2 K rows of bounds and results next to the window: may pass 64 KB)
        if (p.window_elems > 0) MI355_VEC_LAUNCH(2, 1, true);
        else MI355_VEC_LAUNCH(2, 0, true);
        MI355_HIP_TRY(hipGetLastError());
        return p.off_type == MI355_OFF_I64
                   ? launch_giant_rows<int64_t, val_t>(p, static_cast<const int64_t*>(Ap.p), Ax, x, y, s)
                   : launch_giant_rows<int32_t, val_t>(p, static_cast<const int32_t*>(Ap.p), Ax, x, y, s);   // (rows too long for one workgroup, if any)
    }
    switch (p.lanes_per_row) {
        MI355_VEC_CASE(2)
        MI355_VEC_CASE(4)
        MI355_VEC_CASE(8)
        MI355_VEC_CASE(16)
        MI355_VEC_CASE(32)
        MI355_VEC_CASE(64)
        default:
            set_error("csr_vector: bad lanes_per_row %d", p.lanes_per_row);
            return MI355_SPMV_EINVAL;
    }
#undef MI355_VEC_CASE
#undef MI355_VEC_LAUNCH
#undef MI355_VEC_ARGS
    MI355_HIP_TRY(hipGetLastError());
    return MI355_SPMV_OK;
}

template <typename val_t>
static int launch_vector_sweep(const Plan& p, const ApView Ap, const val_t* Ax, const val_t* x, val_t* y, hipStream_t s) {
    const BandHint hint{p.band_lo, p.band_hi, true};
    const dim3 grid((unsigned)p.grid_blocks), block(kHugeBlock);
    const size_t lds = chunk_lds_bytes(p.window_elems, p.rows_cap, sizeof(val_t));
    const ChunkMap cmap{nullptr, (int32_t)p.rows_per_chunk, (int32_t)p.rows_cap, p.n_chunks, 0, int64_t(0),
                        p.knob.rel32_limit > 0 ? p.knob.rel32_limit : kRel32Limit, 0};
    // rows a vector holds: 4, or 8 (fp32; sweep_rows_for) — the plan's rows per chunk say which
    const int64_t vectors = kHugeBlock / p.lanes_per_row;
    const int held = int(p.rows_per_chunk / vectors);
    constexpr bool kHasEight = sizeof(val_t) == 4;
    if (p.rows_per_chunk != vectors * held || !(held == kSweepRows || (kHasEight && held == 8 && p.lanes_per_row >= 4)) ||
        p.rows_cap < p.rows_per_chunk || p.window_elems < int(kHugeBlock * 16 / sizeof(val_t))) {
        set_error("csr_vector: sweep plan with %lld rows per chunk at %d lanes per row", (long long)p.rows_per_chunk, p.lanes_per_row);
        return MI355_SPMV_EINVAL;
    }
#define MI355_VEC_SWEEP(TT, RR)                                                                               \
    do {                                                                                                      \
        if (const int st = allow_dynamic_lds((const void*)csr_vector_sweep_kernel<TT, RR, val_t>, lds)) return st; \
        hipLaunchKernelGGL((csr_vector_sweep_kernel<TT, RR, val_t>), grid, block, lds, s, p.n_rows, p.n_cols, p.nnz_read, Ap, \
                           p.Aj, Ax, x, y, cmap, (int32_t)p.window_elems, hint, (val_t)p.alpha, (val_t)p.beta);   \
    } while (0)
#define MI355_VEC_CASE(TT)                                                                                    \
    case TT:                                                                                                  \
        if constexpr (kHasEight && TT >= 4) { if (held == 8) { MI355_VEC_SWEEP(TT, 8); break; } }              \
        MI355_VEC_SWEEP(TT, kSweepRows);                                                                      \
        break;
    switch (p.lanes_per_row) {
        MI355_VEC_CASE(2)
        MI355_VEC_CASE(4)
        MI355_VEC_CASE(8)
        MI355_VEC_CASE(16)
        MI355_VEC_CASE(32)
        MI355_VEC_CASE(64)
        default:
            set_error("csr_vector: bad lanes_per_row %d", p.lanes_per_row);
            return MI355_SPMV_EINVAL;
    }
#undef MI355_VEC_CASE
#undef MI355_VEC_SWEEP
    MI355_HIP_TRY(hipGetLastError());
    return MI355_SPMV_OK;
}

template <typename off_t, typename val_t>
static int launch_vector_plain(const Plan& p, const off_t* Ap, const val_t* Ax, const val_t* x, val_t* y,
                               hipStream_t s) {
    const int rows_per_block = kBlock / p.lanes_per_row;
    const dim3 grid((unsigned)((int64_t(p.n_rows) + rows_per_block - 1) / rows_per_block)), block(kBlock);
    const off_t nnz = (off_t)p.nnz;
#define MI355_VEC_CASE(TT)                                                                                \
    case TT:                                                                                              \
        hipLaunchKernelGGL((csr_vector_kernel<TT, off_t, val_t>), grid, block, 0, s, p.n_rows, nnz, Ap, p.Aj, \
                           Ax, x, y, (val_t)p.alpha, (val_t)p.beta);                                      \
        break;
    switch (p.lanes_per_row) {
        MI355_VEC_CASE(2)
        MI355_VEC_CASE(4)
        MI355_VEC_CASE(8)
        MI355_VEC_CASE(16)
        MI355_VEC_CASE(32)
        MI355_VEC_CASE(64)
        default:
            set_error("csr_vector: bad lanes_per_row %d", p.lanes_per_row);
            return MI355_SPMV_EINVAL;
    }
#undef MI355_VEC_CASE
    MI355_HIP_TRY(hipGetLastError());
    return MI355_SPMV_OK;
}

template <typename off_t, typename val_t>
int launch_vector(const Plan& p, const off_t* Ap, const val_t* Ax, const val_t* x, val_t* y, hipStream_t s) {
    if (p.n_rows == 0) return MI355_SPMV_OK;
    // 16-byte loads need 16-byte-aligned Aj / Ax / x (hipMalloc gives 256); a caller
    // that passes an offset view gets the 4-byte-per-lane form instead.
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.Aj) | reinterpret_cast<uintptr_t>(Ax) |
                           reinterpret_cast<uintptr_t>(x)) & 15u) == 0;
    // (MI355_SPMV_PLAIN: tuning / tests — a block keeps the whole plan's order; small_plain: the plan's own choice for a
    // small matrix, which a block inherits with its lanes per row: the same sums bit for bit)
    const bool force_plain = (p.knob.plain != 0 && !p.is_block) || p.small_plain;
    if (aligned && p.nnz >= 4 && !force_plain) {
        const ApView view{Ap, sizeof(off_t) == 8 ? 1 : 0};
        if (p.sweep) return launch_vector_sweep<val_t>(p, view, Ax, x, y, s);
        return p.block_threads == kHugeBlock   ? launch_vector_window<kHugeBlock, val_t>(p, view, Ax, x, y, s)
               : p.block_threads == kWideBlock ? launch_vector_window<kWideBlock, val_t>(p, view, Ax, x, y, s)
                                               : launch_vector_window<kBlock, val_t>(p, view, Ax, x, y, s);
    }
    return launch_vector_plain<off_t, val_t>(p, Ap, Ax, x, y, s);
}

// One translation unit per value type (csr_vector_f64.hip includes this file with MI355_TU_F64): the two
// halves of the instantiations compile side by side.
#ifdef MI355_TU_PROBE      // (scripts: one kernel instantiated on its own to read its register use quickly)
#elif !defined(MI355_TU_F64)
template int launch_vector<int32_t, float>(const Plan&, const int32_t*, const float*, const float*, float*, hipStream_t);
template int launch_vector<int64_t, float>(const Plan&, const int64_t*, const float*, const float*, float*, hipStream_t);
#else
template int launch_vector<int32_t, double>(const Plan&, const int32_t*, const double*, const double*, double*, hipStream_t);
template int launch_vector<int64_t, double>(const Plan&, const int64_t*, const double*, const double*, double*, hipStream_t);
#endif

}  // namespace mi355
