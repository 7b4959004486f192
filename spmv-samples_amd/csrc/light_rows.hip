// light_rows.hip — kind LIGHT: dynamic row distribution (LightSpMV rebuilt for gfx950).
//
// Replaces include/spmv/LightSpMV.cuh:111-263 (csrDynamicVector / csrDynamicWarp)
// and its per-call setup (:266-316).  The idea kept from the reference: a
// persistent grid that fetches rows from a global atomic counter, so that rows of
// very different lengths balance across the chip.
//
// What changes for MI355X:
//  * the unit handed out by one atomic is a CHUNK of consecutive rows (equal row counts,
//    or equal weight on a power-law matrix: analyze.hip, decide_balance) taken by a
//    whole workgroup (the reference hands 1 row to a vector or 32/T rows to a warp,
//    LightSpMV.cuh:128-132, :205-209).  One returning device-scope atomic on a
//    single word saturates near 88 dequeues/us on this chip; 1 row per atomic would
//    cost ~10^5 us for 8.3 M rows.  A chunk also gives the workgroup a window of x
//    worth staging through LDS (xwindow.hpp), which is what lifts the gather limit;
//  * the counter is SHARDED: 8 counters, each on its own 128-byte line, each
//    covering one contiguous eighth of the chunks.  A workgroup starts on the shard of
//    its XCD (blockIdx % 8 shares an L2) and walks the other shards when its own
//    runs dry, so placement only affects speed, never results;
//  * lane 0 of the workgroup dequeues, the chunk index reaches the other waves
//    through LDS + barrier; all lanes stay in the loops and rows past the chunk are
//    predicated (the reference's warp-level kernel lets finished vectors leave a
//    loop whose siblings still shuffle, LightSpMV.cuh:212, :246, :260);
//  * the counters live in the plan's scratch, zeroed once at plan creation, and are
//    RE-ARMED BY THE KERNEL: the last workgroup to leave (a ninth counter counts the
//    leavers) stores zeros, so an execute is a single launch with no memset in front
//    of it and can be captured in a hipGraph (the reference mallocs, memsets and frees
//    the counter per call, LightSpMV.cuh:274-276, :314).  A plan therefore serves one
//    stream at a time;
//  * x is read with plain loads / the LDS window (no texture path on CDNA; the
//    reference's texture fetch, LightSpMV.cuh:59-88, has no counterpart).
//
// Exit condition: a workgroup drains its own shard, asks once which other shards still
// hold rows, visits those and leaves; a visit ends on the first dequeue at or past the
// shard's end, which every workgroup reaches whatever the interleaving, so the grid
// always drains.

#include <cstdlib>

#include "common.hpp"
#include "row_dot.hpp"
#include "giant_rows.hpp"
#include "xwindow.hpp"

namespace mi355 {

constexpr int kCounterStride = 16;  // unsigned long long per 128-byte line

__device__ __forceinline__ unsigned long long wave_broadcast_u64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v));
    const unsigned hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}


// Bit s set = shard s still had rows to hand out a moment ago.  Called by a whole wave: lanes 0..7 read
// the eight counters with one returning atomic each, in flight together, so a workgroup whose own shard
// ran dry pays ONE round trip to learn which other shards are worth a visit instead of one failed
// dequeue (a dependent ~2 us round trip) per shard.  A counter only grows during an execute, so "dry" is
// final; a shard reported busy is visited and drained with ordinary dequeues.
__device__ __forceinline__ unsigned shards_with_rows(unsigned long long* __restrict__ counters, int64_t n_units) {
    int lane = threadIdx.x & (kWave - 1);
    // (opaque to the optimiser: the per-lane address and shard size below are loop invariants that it would
    // otherwise compute once, keep in VGPRs across the whole chunk body of the persistent loop, and spill)
    asm volatile("" : "+v"(lane));
    bool has = false;
    if (lane < kXcds) {
        const int64_t size = n_units * (lane + 1) / kXcds - n_units * lane / kXcds;
        has = int64_t(atomicAdd(&counters[lane * kCounterStride], 0ull)) < size;
    }
    return unsigned(__ballot(has)) & ((1u << kXcds) - 1u);
}

// The workgroup has made its last dequeue (on every shard it visited one returned past the shard's end,
// and the issuing thread waited for every returned value).  The last of the gridDim.x workgroups to get here re-arms
// the counters for the next execute.
__device__ __forceinline__ void light_leave(unsigned long long* __restrict__ counters) {
    __syncthreads();
    // Relaxed device-scope atomics are enough (and a fence here would write the XCD's L2 back, +10 us):
    // the counters are only ever touched by atomics, and this thread has consumed the value returned by
    // every dequeue it issued, so they all precede this increment.
    if (threadIdx.x == 0) {
        const unsigned long long left = atomicAdd(&counters[kXcds * kCounterStride], 1ull);
        if (left == gridDim.x - 1)
            for (int i = 0; i <= kXcds; ++i) atomicExch(&counters[i * kCounterStride], 0ull);
    }
}

// (512 threads: two workgroups per CU need 4 waves per SIMD, i.e. <= 128 VGPRs; 256 threads: three workgroups
// per CU need 3 waves per SIMD, <= 168 VGPRs.)  Like the CSR-vector kernel this one does not depend on the
// width of the row offsets: a chunk is walked with 32-bit offsets relative to its first nonzero.
template <int BLOCK, int T, int R, int NSEG, bool ADAPT, typename val_t>
__global__ __launch_bounds__(BLOCK, (BLOCK >= kWideBlock || R == 2 ? 4 : 3)) void light_rows_window_kernel(
    int32_t n_rows, int32_t n_cols, int64_t nnz, const ApView Ap, const int32_t* __restrict__ Aj_arg,
    const val_t* __restrict__ Ax_arg, const val_t* __restrict__ x_arg, val_t* __restrict__ y_arg,
    unsigned long long* __restrict__ counters, ChunkMap cmap, int32_t window_cap, BandHint hint,
    SegmentPlan segs, val_t alpha, val_t beta) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];   // window | bounds | y | flags
    __shared__ int s_red[2];
    __shared__ unsigned long long s_got;
    __shared__ unsigned s_busy;
    ChunkScratch<val_t> scr(s_dyn, window_cap, cmap.rows_cap);
    scr.alpha = alpha;
    scr.beta = beta;
    scr.long_steps = cmap.long_steps;
    scr.giant_len = cmap.giant_len;
    // the body of one chunk (all threads; ends with the results swept to y)
    auto run_chunk = [&](int64_t chunk_begin, int64_t chunk_end) {
        // Opaque copies of the operand pointers, once per chunk: without them every per-thread address the chunk
        // body derives from a kernel argument is a loop invariant of the persistent loop, gets computed once up
        // front, lives in VGPRs across the whole body and is spilled (the same body in the one-chunk-per-workgroup
        // CSR-vector kernel needs ~130 VGPRs; here it needed more than 256).
        int zero = 0;
        asm volatile("" : "+v"(zero));
        zero = __builtin_amdgcn_readfirstlane(zero);     // (a scalar the optimiser cannot see through)
        const int32_t* Aj = Aj_arg + zero;
        const val_t* Ax = Ax_arg + zero;
        const val_t* x = x_arg + zero;
        val_t* y = y_arg + zero;
        bool fits;
        const int64_t base = stage_chunk_bounds<val_t>(scr, chunk_begin, chunk_end, Ap, cmap.rel_limit, fits);
        if (!fits) {          // (uniform) more nonzeros than 32-bit chunk-relative offsets reach
            chunk_rows_wide<BLOCK, val_t>(chunk_begin, chunk_end, Ap, Aj, Ax, x, y, alpha, beta, cmap.giant_len);
            __syncthreads();
            return;
        }
        __syncthreads();      // (also orders the read of s_got before the next dequeue writes it)
        const int32_t* const Aj_c = Aj + base;
        const val_t* const Ax_c = Ax + base;
        const int64_t left = nnz - base;
        const int32_t nnz_c = int32_t(left < kRel32Limit + 32768 ? left : kRel32Limit + 32768);
        // the window is staged inside chunk_rows, behind the first group's stream loads
        if constexpr (NSEG > 1) {
            auto stage = [&] {
                return stage_x_segments<val_t>(chunk_begin, chunk_end, n_cols, x, scr.s_x, window_cap, segs);
            };
            chunk_rows_any<BLOCK, T, R, true, ADAPT, val_t>(chunk_begin, chunk_end, nnz_c, Aj_c, Ax_c, x, y, stage, scr);
        } else {
            auto first_last = [&](int64_t r, int& first, int& last) {
                const int32_t s = scr.s_b[r - chunk_begin], e = scr.s_b[r - chunk_begin + 1];
                if (e <= s) return false;
                first = Aj_c[s];
                last = Aj_c[e - 1];
                return true;
            };
            auto stage = [&] {
                return stage_x_window<val_t>(chunk_begin, chunk_end, n_cols, first_last, x, scr.s_x, window_cap, s_red, hint);
            };
            chunk_rows_any<BLOCK, T, R, NSEG == 1, ADAPT, val_t>(chunk_begin, chunk_end, nnz_c, Aj_c, Ax_c, x, y, stage, scr);
        }
    };
    // static_mode — few chunks per workgroup slot (small matrices): there is nothing to balance, every workgroup
    // takes the chunk of its index and the counters are not touched (the dequeue costs two dependent atomics
    // and a shard poll per workgroup: 32 vs 14 us on 2^17 rows).
    const bool static_mode = cmap.n_chunks <= int64_t(gridDim.x) && !cmap.dequeue_once;
    const int home = blockIdx.x % kXcds;
    if constexpr (!ADAPT) {
        // Equal-row chunks (uniform matrices): ONE dequeue per workgroup, as many workgroups as chunks.  The
        // rows are still handed out by the global counters in arrival order — LightSpMV's scheme — but the loop
        // that keeps a persistent workgroup alive is the hardware dispatcher's: the body then compiles like the
        // CSR-vector kernel's (no loop-carried state: 127 instead of 168+ VGPRs, so the 512-thread plan fits
        // two workgroups per CU), and chunks of equal cost need no stealing.  A workgroup starts on the shard of
        // its XCD (neighbouring chunks share windows of x in that L2) and tries the other shards if it is dry;
        // there are exactly as many workgroups as chunks, so everybody finds one.
        int64_t chunk = -1;
        if (static_mode) {
            chunk = xcd_contiguous_id(blockIdx.x, gridDim.x);
            if (chunk >= cmap.n_chunks) chunk = -1;
        } else {
            for (int visit = 0; visit < kXcds && chunk < 0; ++visit) {   // (uniform over the workgroup)
                const int shard = (home + visit) % kXcds;
                const int64_t shard_begin = cmap.n_chunks * shard / kXcds;          // in chunks
                const int64_t shard_end = cmap.n_chunks * (shard + 1) / kXcds;
                if (threadIdx.x == 0) s_got = atomicAdd(&counters[shard * kCounterStride], 1ull);
                __syncthreads();
                const int64_t c = shard_begin + int64_t(wave_broadcast_u64(s_got));
                if (c < shard_end) chunk = c;
                __syncthreads();          // s_got read by all before the next shard's dequeue overwrites it
            }
        }
        if (chunk >= 0) {
            int64_t chunk_begin, chunk_end;
            cmap.range(chunk, n_rows, chunk_begin, chunk_end);
            if (chunk_begin < chunk_end) run_chunk(chunk_begin, chunk_end);
        }
        if (!static_mode) light_leave(counters);
        return;
    } else {
    // Weight-cut chunks (power-law matrices): a PERSISTENT grid that keeps dequeuing, own shard first, then the
    // shards that still hold chunks — chunks of the same weight take different times (2 000 near-empty rows vs
    // 100 long ones), and stealing across the XCDs' shards is what balances them (R-MAT-24: 2.6 ms against
    // 3.7 ms for one workgroup per chunk in index order).
    // ONE loop with ONE call of the chunk body (two call sites made the compiler keep the body as a real
    // function for the biggest instantiations: captures through scratch memory, a call per chunk).
    unsigned busy = 1u << home;
    int visit = 0;
    bool took_static = false;
    for (;;) {
        int64_t chunk = -1;
        if (static_mode) {
            if (took_static) break;
            took_static = true;
            chunk = xcd_contiguous_id(blockIdx.x, gridDim.x);
            if (chunk >= cmap.n_chunks) break;
        } else {
            while (visit < kXcds) {                              // (everything here is uniform over the workgroup)
                const int shard = (home + visit) % kXcds;
                if ((busy >> shard) & 1u) {
                    const int64_t shard_begin = cmap.n_chunks * shard / kXcds;          // in chunks
                    const int64_t shard_end = cmap.n_chunks * (shard + 1) / kXcds;
                    if (threadIdx.x == 0) s_got = atomicAdd(&counters[shard * kCounterStride], 1ull);
                    __syncthreads();
                    const int64_t c = shard_begin + int64_t(wave_broadcast_u64(s_got));
                    if (c < shard_end) { chunk = c; break; }
                    __syncthreads();      // s_got read by all before the next shard's dequeue overwrites it
                }
                ++visit;
                if (visit == 1) {         // own shard dry: ask the others once, all at the same time
                    if (threadIdx.x < kWave) {
                        const unsigned b = shards_with_rows(counters, cmap.n_chunks);
                        if (threadIdx.x == 0) s_busy = b;
                    }
                    __syncthreads();
                    // (LDS contents are the same for every lane, but the compiler cannot know: a VGPR here would
                    // make the branches on `busy` divergent)
                    busy = __builtin_amdgcn_readfirstlane(s_busy);
                }
            }
            if (chunk < 0) break;
        }
        int64_t chunk_begin, chunk_end;
        cmap.range(chunk, n_rows, chunk_begin, chunk_end);
        if (chunk_begin >= chunk_end) {   // a hub row heavier than a chunk leaves empty chunks behind it
            __syncthreads();              // s_got read by all before the next dequeue overwrites it
            continue;
        }
        run_chunk(chunk_begin, chunk_end);
        __syncthreads();                  // every wave is done with the window before it is refilled
    }
    if (!static_mode) light_leave(counters);
    }
}

// 4-byte-per-lane form for operands that are not 16-byte aligned: one dequeue per wave.
template <int T, typename off_t, typename val_t>
__global__ __launch_bounds__(kBlock) void light_rows_kernel(
    int32_t n_rows, off_t nnz, const off_t* __restrict__ Ap, const int32_t* __restrict__ Aj,
    const val_t* __restrict__ Ax, const val_t* __restrict__ x, val_t* __restrict__ y,
    unsigned long long* __restrict__ counters, int32_t rows_per_chunk, val_t alpha, val_t beta) {
    constexpr int ROWS_PER_STEP = kWave / T;
    const int lane64 = threadIdx.x & (kWave - 1);
    const int lane = threadIdx.x & (T - 1);
    const int vec_in_wave = lane64 / T;
    const int home = blockIdx.x % kXcds;
    unsigned busy = 1u << home;
    for (int visit = 0; visit < kXcds; ++visit) {
        const int shard = (home + visit) % kXcds;
        if (visit == 1) busy = shards_with_rows(counters, n_rows);   // (this kernel's unit is the row)
        if (!((busy >> shard) & 1u)) continue;      // wave-uniform
        const int64_t shard_begin = int64_t(n_rows) * shard / kXcds;
        const int64_t shard_end = int64_t(n_rows) * (shard + 1) / kXcds;
        while (true) {
            unsigned long long got = 0;
            if (lane64 == 0) {
                got = atomicAdd(&counters[shard * kCounterStride], (unsigned long long)rows_per_chunk);
            }
            got = wave_broadcast_u64(got);
            const int64_t chunk_begin = shard_begin + int64_t(got);
            if (chunk_begin >= shard_end) break;  // wave-uniform
            const int64_t chunk_end = min(chunk_begin + rows_per_chunk, shard_end);
            for (int64_t base = chunk_begin; base < chunk_end; base += ROWS_PER_STEP) {
                const int64_t row = base + vec_in_wave;
                const bool live = row < chunk_end;
                off_t start = 0, end = 0;
                if (live) {
                    start = Ap[row];
                    end = Ap[row + 1];
                }
                val_t sum = row_partial<T, off_t, val_t>(start, end, lane, Aj, Ax, x);
                sum = vector_reduce<T, val_t>(sum);
                if (live && lane == 0) y[row] = (beta != val_t(0)) ? alpha * sum + beta * y[row] : alpha * sum;
            }
        }
    }
    light_leave(counters);
}

// The band is wider than any window of x (xwindow.hpp, chunk_rows_sweep; csr_vector.hip has the static twin): one
// 1 024-thread workgroup per CU, a chunk = one group of rows held in registers, handed out by the counters exactly
// as the equal-row chunks of light_rows_window_kernel are (one dequeue per workgroup).
template <int T, int R, typename val_t>
__global__ __launch_bounds__(kHugeBlock, 4) void light_rows_sweep_kernel(
    int32_t n_rows, int32_t n_cols, int64_t nnz, const ApView Ap, const int32_t* __restrict__ Aj,
    const val_t* __restrict__ Ax, const val_t* __restrict__ x, val_t* __restrict__ y,
    unsigned long long* __restrict__ counters, ChunkMap cmap, int32_t window_cap, BandHint hint, val_t alpha, val_t beta) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];   // window | bounds | y | flags
    __shared__ unsigned long long s_got;
    ChunkScratch<val_t> scr(s_dyn, window_cap, cmap.rows_cap);
    scr.alpha = alpha;
    scr.beta = beta;
    const bool static_mode = cmap.n_chunks <= int64_t(gridDim.x) && !cmap.dequeue_once;
    const int home = blockIdx.x % kXcds;
    int64_t chunk = -1;
    if (static_mode) {
        chunk = xcd_contiguous_id(blockIdx.x, gridDim.x);
        if (chunk >= cmap.n_chunks) chunk = -1;
    } else {
        for (int visit = 0; visit < kXcds && chunk < 0; ++visit) {   // (uniform over the workgroup)
            const int shard = (home + visit) % kXcds;
            const int64_t shard_begin = cmap.n_chunks * shard / kXcds;
            const int64_t shard_end = cmap.n_chunks * (shard + 1) / kXcds;
            if (threadIdx.x == 0) s_got = atomicAdd(&counters[shard * kCounterStride], 1ull);
            __syncthreads();
            const int64_t c = shard_begin + int64_t(wave_broadcast_u64(s_got));
            if (c < shard_end) chunk = c;
            __syncthreads();
        }
    }
    int64_t rb = 0, re = 0;
    if (chunk >= 0) cmap.range(chunk, n_rows, rb, re);
    if (rb < re) {
        bool fits;
        const int64_t base = stage_chunk_bounds<val_t>(scr, rb, re, Ap, cmap.rel_limit, fits);
        if (!fits) {
            chunk_rows_wide<kHugeBlock, val_t>(rb, re, Ap, Aj, Ax, x, y, alpha, beta, 0);
        } else {
            __syncthreads();
            const int64_t left = nnz - base;
            const int32_t nnz_c = int32_t(left < kRel32Limit + 32768 ? left : kRel32Limit + 32768);
            chunk_rows_sweep<kHugeBlock, T, R, val_t>(rb, re, nnz_c, Aj + base, Ax + base, x, y, n_cols, window_cap, hint, scr);
        }
    }
    if (!static_mode) light_leave(counters);
}

template <typename val_t> constexpr int light_rows_in_flight() { return sizeof(val_t) == 4 ? 4 : 2; }

#ifndef MI355_TU_F64   // the host-side shape functions live in the fp32 translation unit only
static int64_t light_resident(const Plan& p, int64_t rows) {
    // persistent grid = what stays resident on a CU: bounded by LDS (160 KB: the chunk's layout + ~1 KB static)
    // and by registers (3 workgroups of 256 threads, 2 of 512).  Asking for more than fits leaves the surplus
    // workgroups to start when the others have finished everything (4 asked / 3 resident: 207 vs 200 us).
    if (p.knob.light_blocks_per_cu > 0) return int64_t(kCus) * p.knob.light_blocks_per_cu;
    const size_t val_bytes = p.val_type == MI355_VAL_F64 ? 8 : 4;
    const size_t lds = chunk_lds_bytes(p.window_elems, int(rows), val_bytes) + 1024;
    int64_t per_cu = int64_t(160 * 1024 / lds);
    // registers: the kernels are bounded to 3 waves per SIMD (256 threads) / 4 (512 threads), see the kernel
    const int64_t reg_bound = workgroups_per_cu_by_registers(p);
    if (per_cu > reg_bound) per_cu = reg_bound;
    if (per_cu < 1) per_cu = 1;
    return int64_t(kCus) * per_cu;
}

// Workgroups to launch: a persistent grid (what stays resident) that dequeues chunks when there are several
// chunks per workgroup to balance; one workgroup per chunk, taken by index, when there are at most two per slot
// (the dequeue — two dependent atomics and a poll per workgroup — then costs more than it can balance away).
static int64_t light_grid(const Plan& p, int64_t n_chunks, int64_t resident) {
    // equal-row chunks: one workgroup (and one dequeue) per chunk; weight-cut chunks: the persistent grid
    int64_t blocks = (!p.balanced || n_chunks <= 2 * resident) ? n_chunks : resident;
    return blocks < 1 ? 1 : blocks;
}
// equal-row chunks: take the chunk by index (no counter) when there are at most two per workgroup slot
static bool light_dequeue_once(const Plan& p, int64_t n_chunks, int64_t resident) {
    return !p.balanced && n_chunks > 2 * resident;
}

void shape_light(Plan& p) {
    p.lanes_per_row = pick_lanes_per_row(p.nnz - p.nnz_begin, p.n_rows, p.elems_per_lane);
    const int R = p.val_type == MI355_VAL_F64 ? light_rows_in_flight<double>() : light_rows_in_flight<float>();
    {                                                          // tuning knob (as the CSR-vector kind)
        const int t = p.knob.lanes;
        if (t == 2 || t == 4 || t == 8 || t == 16 || t == 32 || t == 64) p.lanes_per_row = t;
    }
    const int div = p.knob.light_chunk_div;
    // chunks: the static kind's size (halving them cost 6 % on the S32-band target: the
    // window of x is staged per chunk), never below one pass of the workgroup
    shape_chunks(p, R, div > 0 ? div : 1, true, true);
    p.n_tiles = (int64_t(p.n_rows) + p.rows_per_chunk - 1) / p.rows_per_chunk;
    p.grid_blocks = light_grid(p, p.n_tiles, light_resident(p, p.rows_per_chunk));
    p.light_dequeue_once = light_dequeue_once(p, p.n_tiles, light_resident(p, p.rows_per_chunk));
    p.n_kernels = 1;
    snprintf(p.main_kernel, sizeof(p.main_kernel), "light_rows_window_kernel");
}

// after decide_balance: the persistent grid is sized by the chunks there are, window by the rows a chunk may hold
void reshape_light_balanced(Plan& p) {
    if (!p.balanced) return;
    p.n_tiles = p.n_chunks;
    p.block_threads = kBlock;          // (weight-cut chunks are sized for 256 threads)
    p.window_bytes = kWindowBytes;
    p.window_elems = pick_window_elems(p, p.rows_cap);
    if (p.n_seg >= 2) { p.window_elems = 0; p.n_seg = 0; }
    p.grid_blocks = light_grid(p, p.n_chunks, light_resident(p, p.rows_cap));
    p.light_dequeue_once = false;
}

void block_grid_light(Plan& p) {
    p.n_tiles = p.n_chunks;
    const int64_t resident = p.sweep ? int64_t(kCus) : light_resident(p, p.balanced ? p.rows_cap : p.rows_per_chunk);
    p.grid_blocks = light_grid(p, p.n_chunks, resident);
    p.light_dequeue_once = light_dequeue_once(p, p.n_chunks, resident);
    snprintf(p.main_kernel, sizeof(p.main_kernel), p.sweep ? "light_rows_sweep_kernel" : "light_rows_window_kernel");
}

// after shape_sweep said yes (analyze.hip): one workgroup per chunk, one per CU resident
void reshape_light_sweep(Plan& p) {
    if (!p.sweep) return;
    block_grid_light(p);
}

#endif  // MI355_TU_F64

template <int BLOCK, typename val_t>
static int launch_light_window(const Plan& p, const ApView Ap, const val_t* Ax, const val_t* x, val_t* y,
                               hipStream_t s) {
    if constexpr (BLOCK >= kWideBlock) {
        // (see launch_vector_window: no 512- / 1 024-thread kernel without ONE window of x)
        if (p.window_elems <= 0 || p.n_seg >= 2) return launch_light_window<kBlock, val_t>(p, Ap, Ax, x, y, s);
    }
    constexpr int R = light_rows_in_flight<val_t>();
    const BandHint hint{p.band_lo, p.band_hi, p.window_from_band};
    const dim3 grid((unsigned)p.grid_blocks), block(BLOCK);
    const int64_t nnz = p.nnz_read;
    const ChunkMap cmap{p.balanced ? p.chunk_row : nullptr, (int32_t)p.rows_per_chunk, (int32_t)p.rows_cap, p.n_chunks,
                        long_steps_for(p), p.n_giant > 0 ? p.giant_len : int64_t(0),
                        p.knob.rel32_limit > 0 ? p.knob.rel32_limit : kRel32Limit,
                        p.light_dequeue_once ? 1 : 0};
    const size_t lds = chunk_lds_bytes(p.window_elems, p.rows_cap, sizeof(val_t));
    SegmentPlan segs;
    segs.n = p.n_seg;
    for (int i = 0; i < kMaxSegments; ++i) { segs.lo[i] = p.seg_lo[i]; segs.hi[i] = p.seg_hi[i]; }
#define MI355_LIGHT_ARGS s, p.n_rows, p.n_cols, nnz, Ap, p.Aj, Ax, x, y, p.counters, cmap, (int32_t)p.window_elems, hint, segs, (val_t)p.alpha, (val_t)p.beta
    // rows a vector keeps in flight: fp32 with 16 or more lanes per row (rows of 33+ nonzeros) runs with 2 instead of
    // 4 - a long row keeps its lanes' loads busy by itself, and the body then needs ~95 VGPRs instead of ~135 (four
    // 256-thread workgroups per CU instead of three: what a small matrix's single round of chunks is sized for)
    constexpr auto wide_r = [](int tt) constexpr { return (sizeof(val_t) == 4 && tt >= 16) ? 2 : R; };
#define MI355_LIGHT_LAUNCH(TT, NSEG_, ADAPT_)                                                                  \
    do {                                                                                                       \
        if (const int st = allow_dynamic_lds((const void*)light_rows_window_kernel<BLOCK, TT, (wide_r(TT)), NSEG_, ADAPT_, val_t>, lds)) return st; \
        hipLaunchKernelGGL((light_rows_window_kernel<BLOCK, TT, (wide_r(TT)), NSEG_, ADAPT_, val_t>), grid, block, lds, MI355_LIGHT_ARGS); \
    } while (0)
#define MI355_LIGHT_CASE(TT)                                                                                   \
    case TT:                                                                                                   \
        if (p.window_elems > 0 && p.n_seg >= 2) {                                                              \
            /* (several bands: shape_chunks keeps those plans on 256 threads) */                               \
            if constexpr (BLOCK == kBlock) MI355_LIGHT_LAUNCH(TT, kMaxSegments, false);                        \
            else { set_error("light_rows: no 512-thread kernel for a multi-band window"); return MI355_SPMV_EINVAL; } \
        }                                                                                                      \
        else if (p.window_elems > 0) MI355_LIGHT_LAUNCH(TT, 1, false);                                         \
        else if constexpr (BLOCK == kBlock) MI355_LIGHT_LAUNCH(TT, 0, false);                                  \
        break;
    if constexpr (BLOCK == kBlock) if (p.balanced) {   // vector width per chunk (chunk_rows_any); the T of the template is not used
        // (the weight-cut layout holds up to 2 K rows of bounds and results next to the window: may pass 64 KB)
        if (p.window_elems > 0) MI355_LIGHT_LAUNCH(2, 1, true);
        else MI355_LIGHT_LAUNCH(2, 0, true);
        MI355_HIP_TRY(hipGetLastError());
        return p.off_type == MI355_OFF_I64
                   ? launch_giant_rows<int64_t, val_t>(p, static_cast<const int64_t*>(Ap.p), Ax, x, y, s)
                   : launch_giant_rows<int32_t, val_t>(p, static_cast<const int32_t*>(Ap.p), Ax, x, y, s);   // (rows too long for one workgroup, if any)
    }
    switch (p.lanes_per_row) {
        MI355_LIGHT_CASE(2)
        MI355_LIGHT_CASE(4)
        MI355_LIGHT_CASE(8)
        MI355_LIGHT_CASE(16)
        MI355_LIGHT_CASE(32)
        MI355_LIGHT_CASE(64)
        default:
            set_error("light_rows: bad lanes_per_row %d", p.lanes_per_row);
            return MI355_SPMV_EINVAL;
    }
#undef MI355_LIGHT_CASE
#undef MI355_LIGHT_LAUNCH
    MI355_HIP_TRY(hipGetLastError());
    return MI355_SPMV_OK;
}

template <typename val_t>
static int launch_light_sweep(const Plan& p, const ApView Ap, const val_t* Ax, const val_t* x, val_t* y, hipStream_t s) {
    const BandHint hint{p.band_lo, p.band_hi, true};
    const dim3 grid((unsigned)p.grid_blocks), block(kHugeBlock);
    const size_t lds = chunk_lds_bytes(p.window_elems, p.rows_cap, sizeof(val_t));
    const ChunkMap cmap{nullptr, (int32_t)p.rows_per_chunk, (int32_t)p.rows_cap, p.n_chunks, 0, int64_t(0),
                        p.knob.rel32_limit > 0 ? p.knob.rel32_limit : kRel32Limit, p.light_dequeue_once ? 1 : 0};
    const int64_t vectors = kHugeBlock / p.lanes_per_row;     // (rows a vector holds: 4, or 8 in fp32 — csr_vector.hip, sweep_rows_for)
    const int held = int(p.rows_per_chunk / vectors);
    constexpr bool kHasEight = sizeof(val_t) == 4;
    if (p.rows_per_chunk != vectors * held || !(held == kSweepRows || (kHasEight && held == 8 && p.lanes_per_row >= 4)) ||
        p.rows_cap < p.rows_per_chunk || p.window_elems < int(kHugeBlock * 16 / sizeof(val_t))) {
        set_error("light_rows: sweep plan with %lld rows per chunk at %d lanes per row", (long long)p.rows_per_chunk, p.lanes_per_row);
        return MI355_SPMV_EINVAL;
    }
#define MI355_LIGHT_SWEEP(TT, RR)                                                                              \
    do {                                                                                                       \
        if (const int st = allow_dynamic_lds((const void*)light_rows_sweep_kernel<TT, RR, val_t>, lds)) return st; \
        hipLaunchKernelGGL((light_rows_sweep_kernel<TT, RR, val_t>), grid, block, lds, s, p.n_rows, p.n_cols, p.nnz_read, Ap, \
                           p.Aj, Ax, x, y, p.counters, cmap, (int32_t)p.window_elems, hint, (val_t)p.alpha, (val_t)p.beta); \
    } while (0)
#define MI355_LIGHT_CASE(TT)                                                                                   \
    case TT:                                                                                                   \
        if constexpr (kHasEight && TT >= 4) { if (held == 8) { MI355_LIGHT_SWEEP(TT, 8); break; } }             \
        MI355_LIGHT_SWEEP(TT, kSweepRows);                                                                     \
        break;
    switch (p.lanes_per_row) {
        MI355_LIGHT_CASE(2)
        MI355_LIGHT_CASE(4)
        MI355_LIGHT_CASE(8)
        MI355_LIGHT_CASE(16)
        MI355_LIGHT_CASE(32)
        MI355_LIGHT_CASE(64)
        default:
            set_error("light_rows: bad lanes_per_row %d", p.lanes_per_row);
            return MI355_SPMV_EINVAL;
    }
#undef MI355_LIGHT_CASE
    MI355_HIP_TRY(hipGetLastError());
    return MI355_SPMV_OK;
}

template <typename off_t, typename val_t>
static int launch_light_plain(const Plan& p, const off_t* Ap, const val_t* Ax, const val_t* x, val_t* y,
                              hipStream_t s) {
    const dim3 grid((unsigned)p.grid_blocks), block(kBlock);
    const off_t nnz = (off_t)p.nnz;
    const int32_t chunk = (int32_t)p.rows_per_chunk;
#define MI355_LIGHT_CASE(TT)                                                                          \
    case TT:                                                                                          \
        hipLaunchKernelGGL((light_rows_kernel<TT, off_t, val_t>), grid, block, 0, s, p.n_rows, nnz, Ap, \
                           p.Aj, Ax, x, y, p.counters, chunk, (val_t)p.alpha, (val_t)p.beta);         \
        break;
    switch (p.lanes_per_row) {
        MI355_LIGHT_CASE(2)
        MI355_LIGHT_CASE(4)
        MI355_LIGHT_CASE(8)
        MI355_LIGHT_CASE(16)
        MI355_LIGHT_CASE(32)
        MI355_LIGHT_CASE(64)
        default:
            set_error("light_rows: bad lanes_per_row %d", p.lanes_per_row);
            return MI355_SPMV_EINVAL;
    }
#undef MI355_LIGHT_CASE
    MI355_HIP_TRY(hipGetLastError());
    return MI355_SPMV_OK;
}

template <typename off_t, typename val_t>
int launch_light(const Plan& p, const off_t* Ap, const val_t* Ax, const val_t* x, val_t* y, hipStream_t s) {
    if (p.n_rows == 0) return MI355_SPMV_OK;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.Aj) | reinterpret_cast<uintptr_t>(Ax) |
                           reinterpret_cast<uintptr_t>(x)) & 15u) == 0;
    if (aligned && p.nnz >= 4) {
        const ApView view{Ap, sizeof(off_t) == 8 ? 1 : 0};
        if (p.sweep) return launch_light_sweep<val_t>(p, view, Ax, x, y, s);
        return p.block_threads == kHugeBlock   ? launch_light_window<kHugeBlock, val_t>(p, view, Ax, x, y, s)
               : p.block_threads == kWideBlock ? launch_light_window<kWideBlock, val_t>(p, view, Ax, x, y, s)
                                               : launch_light_window<kBlock, val_t>(p, view, Ax, x, y, s);
    }
    return launch_light_plain<off_t, val_t>(p, Ap, Ax, x, y, s);
}

// One translation unit per value type (light_rows_f64.hip includes this file with MI355_TU_F64).
#ifdef MI355_TU_PROBE      // (scripts: one kernel instantiated on its own to read its register use quickly)
#elif !defined(MI355_TU_F64)
template int launch_light<int32_t, float>(const Plan&, const int32_t*, const float*, const float*, float*, hipStream_t);
template int launch_light<int64_t, float>(const Plan&, const int64_t*, const float*, const float*, float*, hipStream_t);
#else
template int launch_light<int32_t, double>(const Plan&, const int32_t*, const double*, const double*, double*, hipStream_t);
template int launch_light<int64_t, double>(const Plan&, const int64_t*, const double*, const double*, double*, hipStream_t);
#endif

}  // namespace mi355
