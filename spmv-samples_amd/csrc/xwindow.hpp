// xwindow.hpp — the dense vector x staged through an LDS window, and the row-chunk
// body shared by the CSR-vector and the dynamic-row kernels.
//
// Why: on gfx950 a wave-wide 4-byte gather whose 64 lanes fall in 64 different
// cache lines (a banded row: neighbouring nonzeros are ~256 columns apart) is
// served at about one line per clock by the CU's vector L1; measured on the
// S32-band target the plain global gather caps the whole SpMV at ~2.5-2.8 TB/s
// while the same kernel with the gather removed streams at 5.4 TB/s
// (tools/exp_spmv.hip).  LDS serves 64 arbitrary 4-byte reads in a few cycles.
// So a workgroup that owns a chunk of consecutive rows first copies the window of
// x those rows touch into LDS with coalesced 16-byte loads, then gathers from LDS.
//
// The window is a software cache, never a correctness assumption:
//   * it is placed from a cheap SAMPLE — the first and last column of 32 rows spread
//     over the chunk (a good guess when columns are sorted and the structure is
//     smooth; the interface promises neither: the reference loader keeps file
//     order, load.hpp:443-471);
//   * every gathered column is range-checked; a column outside the window is
//     loaded from global memory instead (predicated, skipped when no lane needs it);
//   * if the sampled span exceeds the LDS capacity the window is centred on it.
// The reference has no counterpart (it reads x through the texture path or plain
// loads: LightSpMV.cuh:59-88, cusp_warp_reduce.cuh:41-48).
#pragma once

#include <climits>
#include <type_traits>

#include "common.hpp"
#include "row_dot.hpp"

namespace mi355 {

constexpr int kSamples = 32;   // rows sampled per chunk to place the window

template <typename val_t>
struct XWindow {
    const val_t* s_x;   // LDS copy of x[lo, lo+len)
    int32_t lo;
    int32_t len;
    // LDS index of column `col` (0 when it is outside the window); returns whether it is inside
    __device__ __forceinline__ bool find(int32_t col, unsigned& idx) const {
        const unsigned rel = unsigned(col - lo);
        const bool in = rel < unsigned(len);
        idx = in ? rel : 0u;
        return in;
    }
};

// Several windows at once, for matrices whose columns sit in a few far-apart bands (a 3-D
// stencil: the planes k-1, k, k+1 are ~nx*ny columns apart; one window cannot span them but
// three narrow ones hold everything).  Segment s holds x[lo[s], lo[s]+len[s]) at s_x + off[s].
constexpr int kMaxSegments = 4;
struct SegmentPlan {                 // bands [lo, hi] of (column - row), from the plan's probe
    int n;
    int64_t lo[kMaxSegments], hi[kMaxSegments];
};
template <typename val_t>
struct XWindowN {
    const val_t* s_x;
    int32_t lo[kMaxSegments], len[kMaxSegments], off[kMaxSegments];
    // derived by finish(): the segments are disjoint and ascending, so the one a column can be in is the last
    // whose start it reaches — one compare and two selects per segment instead of a full range test each
    int32_t thr[kMaxSegments];    // start of segment s (INT32_MAX for an empty one)
    int32_t shift[kMaxSegments];  // lo[s] - off[s]: LDS index = col - shift
    int32_t end[kMaxSegments];    // lo[s] + len[s]
    __device__ __forceinline__ void finish() {
#pragma unroll
        for (int s = 0; s < kMaxSegments; ++s) {
            thr[s] = len[s] > 0 ? lo[s] : INT32_MAX;
            shift[s] = lo[s] - off[s];
            end[s] = len[s] > 0 ? lo[s] + len[s] : INT32_MIN;
        }
    }
    __device__ __forceinline__ bool find(int32_t col, unsigned& idx) const {
        int32_t sh = shift[0], en = end[0];
#pragma unroll
        for (int s = 1; s < kMaxSegments; ++s) {
            const bool at = col >= thr[s];
            sh = at ? shift[s] : sh;
            en = at ? end[s] : en;
        }
        const bool in = (col >= thr[0]) & (col < en);
        idx = in ? unsigned(col - sh) : 0u;
        return in;
    }
};

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = min(v, __shfl_xor(v, o, kWave));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o, kWave));
    return v;
}

// Workgroup-wide: sample the column range of rows [rb, re), copy that window of x
// into s_x (capacity `cap` elements).  All threads of the workgroup must call (any size); ends
// with a barrier.  s_red: 2 ints of LDS scratch.
// BandHint: the band [lo, hi] of (column - row) the plan's probe saw (analyze.hip).  When the
// whole band of a chunk fits the window (use == true) the window is placed from it and
// the per-chunk sample — two dependent loads and a barrier — is skipped.
struct BandHint {
    int64_t lo, hi;
    bool use;
};

// first_last(r, first, last): the first and last stored column of row r (false when the row is empty) — the
// merge kernel reads them through Ap, the row-chunk kernels through the bounds they already hold in LDS.
template <typename val_t, typename FirstLast>
__device__ __forceinline__ XWindow<val_t> stage_x_window(int64_t rb, int64_t re, int32_t n_cols,
                                                         FirstLast&& first_last,
                                                         const val_t* __restrict__ x, val_t* s_x,
                                                         int32_t cap, int* s_red,
                                                         const BandHint hint = BandHint{0, 0, false}) {
    // (opaque copy of the thread index: inside a persistent loop the optimiser otherwise hoists this function's
    // per-thread addresses out of the loop, keeps them in VGPRs across the whole chunk body and spills them)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    int lo = INT32_MAX, hi = -1;
    if (cap <= 0) {        // the plan decided against a window: no sampling
        XWindow<val_t> none;
        none.s_x = s_x;
        none.lo = 0;
        none.len = 0;
        __syncthreads();   // callers rely on this function being a workgroup barrier
        return none;
    }
    if (hint.use && re > rb) {
        const int64_t l = rb + hint.lo, h = re - 1 + hint.hi;
        lo = int(l < 0 ? 0 : (l >= n_cols ? n_cols - 1 : l));
        hi = int(h < 0 ? 0 : (h >= n_cols ? n_cols - 1 : h));
    } else {
    // kSamples rows spread evenly over the chunk, first and last row included.  NOT
    // every row: the first and last column of a 32-nonzero row sit in the two cache
    // lines that hold the whole row of Aj, so sampling every row re-reads all of Aj
    // (measured: -25 % on the S32-band target).
    if (tid < kSamples && re > rb) {
        const int64_t r = rb + ((re - 1 - rb) * tid) / (kSamples - 1);
        int first, last;
        if (first_last(r, first, last)) {
            lo = min(first, last);
            hi = max(first, last);
        }
    }
    if (tid < kWave) {   // the samples live in wave 0 (kSamples <= kWave)
        lo = wave_min(lo);
        hi = wave_max(hi);
        if (tid == 0) {
            s_red[0] = lo;
            s_red[1] = hi;
        }
    }
    __syncthreads();
    lo = s_red[0];
    hi = s_red[1];
    }
    XWindow<val_t> win;
    win.s_x = s_x;
    if (hi < 0) {          // no sampled row has a nonzero: no window, every gather goes to global
        win.lo = 0;
        win.len = 0;
        __syncthreads();
        return win;
    }
    constexpr int PER16 = 16 / int(sizeof(val_t));   // elements per 16-byte load
    // The samples may miss the extremes: widen the sampled span by an eighth (+64) on
    // each side; if that exceeds the capacity, centre the window on the span.
    const int span = hi - lo + 1;
    if (cap <= 0 || span / 16 > cap) {   // columns scattered far beyond what LDS can hold: not worth a window
        win.lo = 0;
        win.len = 0;
        __syncthreads();
        return win;
    }
    int len = hint.use ? span + (PER16 - 1) : span + 2 * (span / 8 + 64);   // (+: the start is rounded down to 16 bytes)
    if (len > cap) len = cap;
    if (len > n_cols) len = n_cols;
    int64_t start = (int64_t(lo) + hi + 1 - len) / 2;
    if (start + len > n_cols) start = n_cols - len;
    if (start < 0) start = 0;
    lo = int(start) & ~(PER16 - 1);                  // x is 16-byte aligned (checked on the host)
    if (lo + len > n_cols) len = n_cols - lo;
    win.lo = lo;
    win.len = len;
    using v16 = typename std::conditional<sizeof(val_t) == 4, float4v, double __attribute__((ext_vector_type(2)))>::type;
    int full = (min(lo + len, n_cols & ~(PER16 - 1)) - lo) / PER16;   // whole 16-byte groups inside x
    if (full < 0) full = 0;
    const int nthreads = int(blockDim.x);
    for (int g = tid; g < full; g += nthreads) {
        *reinterpret_cast<v16*>(s_x + g * PER16) = *reinterpret_cast<const v16*>(x + lo + g * PER16);
    }
    for (int i = full * PER16 + tid; i < len; i += nthreads) s_x[i] = x[lo + i];
    __syncthreads();
    return win;
}

// Workgroup-wide: stage the segments of `plan` for rows [rb, re) (capacity `cap` elements in
// total, later segments are cut when it runs out).  Ends with a barrier.
template <typename val_t>
__device__ __forceinline__ XWindowN<val_t> stage_x_segments(int64_t rb, int64_t re, int32_t n_cols,
                                                            const val_t* __restrict__ x, val_t* s_x,
                                                            int32_t cap, const SegmentPlan& plan) {
    constexpr int PER16 = 16 / int(sizeof(val_t));
    using v16 = typename std::conditional<sizeof(val_t) == 4, float4v, double __attribute__((ext_vector_type(2)))>::type;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));        // (see stage_x_window)
    XWindowN<val_t> win;
    win.s_x = s_x;
    int off = 0;
#pragma unroll
    for (int s = 0; s < kMaxSegments; ++s) {
        int lo = 0, len = 0;
        if (s < plan.n && re > rb && off < cap && n_cols > 0) {
            int64_t l = rb + plan.lo[s], h = re - 1 + plan.hi[s];
            l = l < 0 ? 0 : l;
            h = h >= n_cols ? int64_t(n_cols) - 1 : h;
            if (h >= l) {
                lo = int(l) & ~(PER16 - 1);
                len = int(h) + 1 - lo;
                if (off + len > cap) len = cap - off;
            }
        }
        win.lo[s] = lo;
        win.len[s] = len;
        win.off[s] = off;
        int full = (min(lo + len, n_cols & ~(PER16 - 1)) - lo) / PER16;   // whole 16-byte groups inside x
        if (full < 0) full = 0;
        for (int g = tid; g < full; g += int(blockDim.x))
            *reinterpret_cast<v16*>(s_x + off + g * PER16) = *reinterpret_cast<const v16*>(x + lo + g * PER16);
        for (int i = full * PER16 + tid; i < len; i += int(blockDim.x)) s_x[off + i] = x[lo + i];
        off += (len + PER16 - 1) & ~(PER16 - 1);
    }
    win.finish();
    __syncthreads();
    return win;
}

// One value of x: from the window when the column is inside it, else from global.
template <typename val_t>
__device__ __forceinline__ val_t window_gather(const XWindow<val_t>& win, const val_t* __restrict__ x,
                                               int32_t col, bool needed) {
    const unsigned rel = unsigned(col - win.lo);
    const bool in = rel < unsigned(win.len);
    val_t v = val_t(0);
    if (in) v = win.s_x[rel];          // (with no window, len == 0 and s_x may have no storage)
    if (needed && !in) v = x[col];
    return v;
}

// LDS scratch of one workgroup for chunk_rows, carved from dynamic LDS behind the window:
//   [ window_elems values of x ][ rows+1 row bounds (int32, chunk-relative) ][ rows results (val_t) ][ rows/32 flag words ]
constexpr int kMaxChunkRows = 2048;   // upper bound of rows per chunk (pick_rows_per_chunk)
constexpr int kLongSteps = 16;        // a row is "long" beyond this many steps of its T-lane vector
constexpr int kHugeRow = 1024;        // a long row beyond this many nonzeros is summed by the whole workgroup
// A chunk is walked with 32-bit offsets relative to its first 16-byte group whatever the width of Ap (the
// merge kernel does the same with its tiles): half the LDS for the bounds, no 64-bit arithmetic or register
// pairs in the loop, and ONE kernel body for both offset widths.  A chunk whose nonzeros span more than this
// (rows of ~10^9 nonzeros that the giant-row pass did not take) goes through chunk_rows_wide instead.
constexpr int64_t kRel32Limit = int64_t(INT32_MAX) - 65536;

// A 64-bit value that is the same in every lane, moved to scalar registers.
__device__ __forceinline__ int64_t uniform_i64(int64_t v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v));
    const unsigned hi = __builtin_amdgcn_readfirstlane(unsigned(uint64_t(v) >> 32));
    return int64_t((uint64_t(hi) << 32) | lo);
}

// ... and a float / double that is the same in every lane (an LDS word every thread reads).
__device__ __forceinline__ float uniform_val(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
__device__ __forceinline__ int32_t uniform_val(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uniform_val(double v) { return __longlong_as_double(uniform_i64(__double_as_longlong(v))); }

// Row offsets of either width behind one kernel signature (a uniform branch per load, in prologues only).
struct ApView {
    const void* p;
    int wide;       // 1 = int64 offsets, 0 = int32
    __device__ __forceinline__ int64_t at(int64_t i) const {
        return wide ? static_cast<const int64_t*>(p)[i] : int64_t(static_cast<const int32_t*>(p)[i]);
    }
};

// How chunk ids map to rows.  Uniform plans: chunk c = rows [c * rows_per_chunk, ...).  nnz-balanced plans
// (analyze.hip, decide_balance): boundaries from a table built at plan creation, so that a chunk of a
// power-law matrix holds a bounded number of nonzeros as well as of rows.
struct ChunkMap {
    const int32_t* table;     // n_chunks + 1 boundaries, or nullptr
    int32_t rows_per_chunk;   // uniform plans
    int32_t rows_cap;         // rows the workgroup's LDS layout holds
    int64_t n_chunks;
    int32_t long_steps;       // a row is "long" (left to the second pass) beyond this many steps of its vector
    int64_t giant_len;        // > 0: a row beyond this many nonzeros is left to the giant-row kernels (giant_rows.hpp)
    int64_t rel_limit;        // nonzero span a chunk may have on the 32-bit path (kRel32Limit; smaller in tests)
    int32_t dequeue_once;     // LIGHT, equal-row chunks: 1 = every workgroup takes its chunk from the counters, 0 = by index
    __device__ __forceinline__ void range(int64_t c, int32_t n_rows, int64_t& rb, int64_t& re) const {
        if (table) {
            rb = table[c];
            re = table[c + 1];
        } else {
            rb = c * rows_per_chunk;
            re = min(rb + rows_per_chunk, int64_t(n_rows));
        }
    }
};

__host__ __device__ inline size_t lds_align16(size_t v) { return (v + 15) & ~size_t(15); }
__host__ __device__ inline size_t chunk_lds_bytes(int window_elems, int rows, size_t val_bytes) {
    return lds_align16(size_t(window_elems) * val_bytes) + lds_align16(size_t(rows + 1) * 4) +
           lds_align16(size_t(rows) * val_bytes) + lds_align16(size_t(rows / 32 + 1) * 4);
}

template <typename val_t>
struct ChunkScratch {
    val_t* s_x;           // window_elems
    int32_t* s_b;         // rows + 1 : Ap[chunk_begin .. chunk_end] - base
    val_t* s_y;           // rows     : results of the chunk, stored to y in one coalesced sweep
    unsigned* long_map;   // rows / 32 + 1 : one bit per row, set = long row, summed in the second pass
    val_t alpha, beta;    // y = alpha * (A x) + beta * y   (1, 0 unless mi355_spmv_plan_set_alpha_beta)
    int long_steps = kLongSteps;
    int64_t giant_len = 0;   // > 0: rows longer than this are summed by giant_rows.hpp, this workgroup stores 0 for them
    int store_rows = -1;     // >= 0: only the first store_rows results go to y (the merge kind's row-parallel runs keep
                             // the partial sum of a row that continues in the next run: it stays in s_y for the caller)
    __device__ ChunkScratch(unsigned char* base, int window_elems, int rows) {
        s_x = reinterpret_cast<val_t*>(base);
        base += lds_align16(size_t(window_elems) * sizeof(val_t));
        s_b = reinterpret_cast<int32_t*>(base);
        base += lds_align16(size_t(rows + 1) * 4);
        s_y = reinterpret_cast<val_t*>(base);
        base += lds_align16(size_t(rows) * sizeof(val_t));
        long_map = reinterpret_cast<unsigned*>(base);
        alpha = val_t(1);
        beta = val_t(0);
    }
};

// Before the barrier that precedes chunk_rows: copy the chunk's row bounds into LDS, relative to
// base = Ap[chunk_begin] & ~3 (returned: the chunk's Aj / Ax start there, 16-byte aligned), and clear the
// long-row flags.  fits = the chunk's nonzeros span less than `limit` (uniform over the workgroup).
template <typename val_t>
__device__ __forceinline__ int64_t stage_chunk_bounds(const ChunkScratch<val_t>& scr, int64_t chunk_begin,
                                                      int64_t chunk_end, const ApView Ap, int64_t limit, bool& fits) {
    const int rows = int(chunk_end - chunk_begin);
    // (every lane loads the same two words; the copies through readfirstlane tell the compiler so: base then
    // lives in an SGPR pair and the chunk's Aj / Ax views are scalar bases with 32-bit lane offsets, instead of
    // 64-bit address arithmetic in VGPR pairs all through the loop)
    const int64_t base = uniform_i64(Ap.at(chunk_begin)) & ~int64_t(3);
    fits = uniform_i64(Ap.at(chunk_end)) - base <= limit;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // (opaque: per-thread LDS addresses stay inside, see stage_x_window)
    if (fits)
        for (int i = tid; i <= rows; i += int(blockDim.x)) scr.s_b[i] = int32_t(Ap.at(chunk_begin + i) - base);
    for (int i = tid; i < rows / 32 + 1; i += int(blockDim.x)) scr.long_map[i] = 0u;
    return base;
}

// One step of 4 nonzeros of one lane: Aj/Ax group at j (16-byte aligned element index).
template <typename val_t>
__device__ __forceinline__ void load_group(int32_t j, int32_t nnz, const int32_t* __restrict__ Aj,
                                           const val_t* __restrict__ Ax, int4v& c,
                                           typename Vec4<val_t>::type& a) {
    using v4 = typename Vec4<val_t>::type;
    if (j + 4 <= nnz) {
        c = stream_load(reinterpret_cast<const int4v*>(Aj + j));
        a = stream_load(reinterpret_cast<const v4*>(Ax + j));
    } else {   // last, partial group of the arrays: never read past nnz
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool in = (j + e) < nnz;
            c[e] = in ? Aj[j + e] : 0;
            a[e] = in ? Ax[j + e] : val_t(0);
        }
    }
}

// The chunk's results: one coalesced sweep from LDS (16-byte nontemporal stores when y is aligned);
// y = alpha * sum + beta * y_old — y is read only when beta != 0 (alpha = 1, beta = 0: plain store).
// The caller has put a barrier between the last write of s_y and this.
template <int BLOCK, typename val_t>
__device__ __forceinline__ void store_chunk_results(const ChunkScratch<val_t>& scr, val_t* __restrict__ y,
                                                    int64_t chunk_begin, int rows, int tid_known = -1) {
    val_t* const yc = y + chunk_begin;
    constexpr int PER16 = 16 / int(sizeof(val_t));
    const val_t alpha = scr.alpha, beta = scr.beta;
    const bool scaled = (alpha != val_t(1)) || (beta != val_t(0));   // uniform
    const int n_store = scr.store_rows >= 0 ? scr.store_rows : rows;
    int tid = tid_known >= 0 ? tid_known : int(threadIdx.x);   // (a caller that has rebuilt the index passes it: chunk_rows)
    asm volatile("" : "+v"(tid));   // (opaque: keeps the sweep's per-thread offsets from being hoisted out of a persistent loop)
    if (!scaled && (reinterpret_cast<uintptr_t>(yc) & 15u) == 0) {
        using v16 = typename std::conditional<sizeof(val_t) == 4, float4v, double __attribute__((ext_vector_type(2)))>::type;
        const int full = n_store / PER16;
        for (int g = tid; g < full; g += BLOCK)
            __builtin_nontemporal_store(*reinterpret_cast<const v16*>(scr.s_y + g * PER16),
                                        reinterpret_cast<v16*>(yc + g * PER16));
        for (int i = full * PER16 + tid; i < n_store; i += BLOCK) yc[i] = scr.s_y[i];
    } else if (!scaled) {
        for (int i = tid; i < n_store; i += BLOCK) yc[i] = scr.s_y[i];
    } else {
        for (int i = tid; i < n_store; i += BLOCK) {
            val_t v = alpha * scr.s_y[i];
            if (beta != val_t(0)) v += beta * yc[i];
            yc[i] = v;
        }
    }
}

// Rows [chunk_begin, chunk_end) by this workgroup: T lanes per row, R rows per vector per
// group, 4 nonzeros per lane per step.  Aj / Ax / nnz are CHUNK-RELATIVE (see stage_chunk_bounds).
// Structure (each point is a measured win on the S32-band target, tools/exp_pipe.hip):
//  * row bounds come from LDS (stage_chunk_bounds), so the only vector-memory traffic in
//    the loop is the Aj/Ax stream itself;
//  * the loop is software-pipelined by hand: the stream loads of group g+1 are issued
//    BEFORE group g is consumed, into a second register set (2x unrolled ping-pong: a
//    register copy of a pending load would force a wait), with branch-free clamped
//    addresses (hipcc puts s_waitcnt vmcnt(0) after conditional loads), so ~8 KB per wave
//    stay in flight while the wave computes;
//  * a column outside the window is fetched AND consumed inside its own branch, so the
//    common path never waits for it;
//  * results go to LDS and leave in one coalesced nontemporal sweep per chunk: one 4-byte
//    store per row straight from the loop cost 13 % of the kernel, although y is 1.5 % of
//    the bytes;
//  * a row longer than kLongSteps steps is not walked by its T lanes (a power-law hub row
//    would serialise the chunk behind one vector — the weakness of the reference's
//    CSR-vector and LightSpMV kernels): its bit is set in an LDS bitmap and a second pass
//    sums every marked row with a whole 64-lane wave, the four waves taking rows in turn.
//  * the window of x is staged AFTER the first group's stream loads are issued (`stage` is the
//    caller's staging function: its loads, LDS writes and closing barrier run while those loads
//    are in flight), so a chunk's prologue costs one memory round trip, not two.
// All BLOCK threads of the workgroup must call (wave-wide shuffles and barriers inside); the caller
// has run stage_chunk_bounds + a barrier; `stage()` returns the window and ends with a barrier.
template <int BLOCK, int T, int R, bool WINDOW, typename val_t, typename StageFn>
__device__ __forceinline__ void chunk_rows(int64_t chunk_begin, int64_t chunk_end, int32_t nnz,
                                           const int32_t* __restrict__ Aj,
                                           const val_t* __restrict__ Ax, const val_t* __restrict__ x,
                                           val_t* __restrict__ y, StageFn&& stage,
                                           const ChunkScratch<val_t>& scr) {
    using v4 = typename Vec4<val_t>::type;
    using off_t = int32_t;                                 // chunk-relative offsets
    constexpr int VECS = BLOCK / T;
    constexpr int WAVES = BLOCK / kWave;
    constexpr int STRIDE = VECS * R;                       // rows per group
    const off_t LONG = off_t(T) * 4 * scr.long_steps;
    const int lane = threadIdx.x & (T - 1);
    const int rows = int(chunk_end - chunk_begin);
    const int n_groups = (rows + STRIDE - 1) / STRIDE;
    const off_t nnz_vec = nnz & ~off_t(3);                 // 16-byte loads stay below this element
    const off_t j_max = nnz_vec - 4;                       // callers guarantee nnz >= 4 (launch_*: plain kernel otherwise)

    // Which rows a vector owns.  A wave holds VW = 64 / T vectors; in one group it covers VW * R consecutive
    // rows, and slot r of vector v is row r * VW + v of them — so ONE load instruction (slot r of every vector)
    // reads VW CONSECUTIVE rows, a contiguous piece of Aj / Ax.  (Round 1 gave a vector R adjacent rows: an
    // instruction then read every R-th row; harmless while rows are whole cache lines (32 nonzeros), but with
    // 27 or 28 per row every line is shared by two rows, i.e. fetched by two different instructions: a single
    // fp64 band fell from 6.6 TB/s at 32 per row to 4.4 at 27.)
    constexpr int VW = kWave / T;
    const int row_in_wave = (threadIdx.x & (kWave - 1)) / T;
    const int wave_rows0 = (threadIdx.x / kWave) * (VW * R);
#ifdef MI355_ROW_MAP_ADJACENT   // (A/B builds only: round 1's mapping, a vector owns R adjacent rows)
    auto row_of = [&](int g, int r) { return g * STRIDE + int(threadIdx.x / T) * R + r; };
#else
    auto row_of = [&](int g, int r) { return g * STRIDE + wave_rows0 + r * VW + row_in_wave; };
#endif
    // A register set holds the loaded columns and values only: the rows' bounds are read from LDS again when the group
    // is consumed (2 R ds_reads) instead of riding along through the previous group's arithmetic — 2 R registers per
    // set that the wide fp32 bodies, held to 128, do not have (they spilled).
    struct Group {
        int4v c[R];
        v4 a[R];
    };
    struct Bounds { off_t lo[R], hi[R]; };
    auto read_bounds = [&](int g, Bounds& B) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row_of(g, r);
            B.lo[r] = scr.s_b[min(row, rows)];
            B.hi[r] = scr.s_b[min(row + 1, rows)];
        }
    };
    auto step0 = [&](off_t lo) { return (lo & ~off_t(3)) + off_t(lane) * 4; };   // first element of this lane's step-0 group
    // issue the step-0 loads of group g (g may be past the end: rows clamp to empty)
    auto issue = [&](int g, Group& G) {
        Bounds B;
        read_bounds(g, B);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const off_t j0 = step0(B.lo[r]);
            off_t jl = j0 < B.hi[r] ? j0 : (B.lo[r] & ~off_t(3));
            jl = jl < j_max ? jl : j_max;
            // straight-line, branch-free: hipcc serialises (vmcnt(0)) around loads in branches
            G.c[r] = stream_load(reinterpret_cast<const int4v*>(Aj + jl));
            G.a[r] = stream_load(reinterpret_cast<const v4*>(Ax + jl));
        }
        // (nothing of the consume that follows may be scheduled above these loads: its first instructions wait for the
        // PREVIOUS group's data, and a load issued behind that wait has lost its head start — seen as s_waitcnt vmcnt(6)
        // in front of the eight prefetch loads, S32-band 177 -> 187 us)
        __builtin_amdgcn_sched_barrier(0);
    };
    // DEPTH register sets in a ring: the loads of DEPTH - 1 groups are in flight while one is reduced.
    // (measured with 4 / 3 sets for the fp32 / fp64 R = 2 bodies: cant stand-in 11.3 -> 12.6 us, C4 stand-in 541 -> 602 us:
    // the extra sets cost more occupancy than the deeper prefetch buys; two everywhere)
    // (round 3, C4 stand-in with 3 sets and 168 registers, which its LDS-bound three workgroups per CU allow: 585 -> 602 us)
    // (round 3 again, on the rewritten body: 3 / 4 sets on the cant stand-in 10.8 -> 13.9 / 12.2 us at 93 / 107 registers)
    constexpr int DEPTH = 2;
    Group G[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d, G[d]);   // in flight while the window is staged
    const auto win = stage();     // (workgroup barrier inside)

    // sum += a[e] * x[c[e]] for the elements k = j+e inside [lo, hi).  WINDOW: x comes from the
    // LDS window: FOUR unconditional ds_reads at clamped addresses, issued back to back, then an fma and a select per
    // element; the rare column outside the window is fetched and consumed under ONE branch per 4 elements, so the
    // common path never waits on vector memory.  The values pass through an empty volatile asm (pin4), i.e. the reads
    // happen whatever the comparisons say: left alone, hipcc sinks each read into its select and emits a BRANCH per
    // element — s_and_saveexec, ds_read, s_waitcnt lgkmcnt(0), v_fmac, s_or exec: sixteen dependent LDS round trips
    // per group of the fp32 body, one after the other (rounds 1-2 shipped exactly that; S32-band 178 -> 1xx us once
    // the reads overlap).  !WINDOW: plain gathers (masked elements hold a legal column of a neighbouring row, or 0,
    // so the address is always in range), pinned the same way.
    auto pin4 = [](val_t (&v)[4]) { asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])); };
    // the window part of a step: lookups and reads only (no use of the values) ...
    auto gather4 = [&](const int4v& c, val_t (&xv)[4], bool (&in)[4]) {
        if constexpr (WINDOW) {
            unsigned idx[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) in[e] = win.find(c[e], idx[e]);
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] = win.s_x[idx[e]];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { in[e] = true; xv[e] = x[c[e]]; }
        }
    };
    // ... and the arithmetic on values that have been pinned
    // (branch-free on purpose: masks are combined with & / |, never && / ||, and every product is formed
    // unconditionally and pinned before its select — written as `(valid && in) ? sum + a * xv : sum` hipcc lowers each
    // element to exec-mask control flow, ~10 scalar instructions around one v_fma)
    auto fold4 = [&](val_t& sum, const int4v& c, const v4& a, const val_t (&xv)[4], const bool (&in)[4], off_t j, off_t lo, off_t hi) {
        const off_t d_lo = lo - j, d_hi = hi - j;
        const int e_lo = d_lo > 0 ? int(d_lo) : 0;                 // lo - j <= 3 wherever this is called
        const int e_hi = d_hi < 4 ? (d_hi > 0 ? int(d_hi) : 0) : 4;
        bool need_any = false;
        bool need[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool valid = bool(int(e >= e_lo) & int(e < e_hi));
            const bool use = bool(int(valid) & int(in[e]));
            val_t t = sum + a[e] * xv[e];
            asm volatile("" : "+v"(t));
            sum = use ? t : sum;
            need[e] = bool(int(valid) & int(!in[e]));
            need_any = bool(int(need_any) | int(need[e]));
        }
        if constexpr (WINDOW) {
            if (need_any) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (need[e]) sum += a[e] * x[c[e]];
                }
            }
        }
    };
    auto accumulate = [&](val_t& sum, const int4v& c, const v4& a, off_t j, off_t lo, off_t hi) {
        val_t xv[4];
        bool in[4];
        gather4(c, xv, in);
        pin4(xv);
        fold4(sum, c, a, xv, in, j, lo, hi);
    };
    auto consume = [&](int g, const Group& G) {
        Bounds B;
        read_bounds(g, B);
        val_t sum[R];
        off_t jn[R], hi[R];
        bool deferred[R];
        bool more = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // (uniform over the T lanes of the vector; a giant row always takes the long-row path, where it is
            // zeroed for the slice kernels, whatever the long-steps knob says)
            const off_t len_r = B.hi[r] - B.lo[r];
            deferred[r] = len_r > LONG || (scr.giant_len > 0 && int64_t(len_r) > scr.giant_len);
            // the 16-byte path covers elements below nnz_vec; a long row is left to pass 2
            hi[r] = deferred[r] ? B.lo[r] : (B.hi[r] < nnz_vec ? B.hi[r] : nnz_vec);
            sum[r] = val_t(0);
        }
        // the window reads of XB rows (8 values) in flight at once, then their arithmetic
        constexpr int XB = R >= 2 ? 2 : 1;
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += XB) {
            val_t xv[XB][4];
            bool in[XB][4];
#pragma unroll
            for (int b = 0; b < XB; ++b) gather4(G.c[r0 + b], xv[b], in[b]);
#pragma unroll
            for (int b = 0; b < XB; ++b) pin4(xv[b]);
#pragma unroll
            for (int b = 0; b < XB; ++b) {
                const int r = r0 + b;
                const off_t j0 = step0(B.lo[r]);
                fold4(sum[r], G.c[r], G.a[r], xv[b], in[b], j0, B.lo[r], hi[r]);
                jn[r] = j0 + off_t(T) * 4;
                more |= jn[r] < hi[r];
            }
        }
        while (more) {                                     // rows longer than one step (4T nonzeros)
            int4v c2[R];
            v4 a2[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                off_t jl = jn[r] < hi[r] ? jn[r] : (B.lo[r] & ~off_t(3));
                jl = jl < j_max ? jl : j_max;
                c2[r] = stream_load(reinterpret_cast<const int4v*>(Aj + jl));
                a2[r] = stream_load(reinterpret_cast<const v4*>(Ax + jl));
            }
            more = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                accumulate(sum[r], c2[r], a2[r], jn[r], B.lo[r], hi[r]);   // (clamped loads are masked by hi)
                jn[r] += off_t(T) * 4;
                more |= jn[r] < hi[r];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) sum[r] = vector_reduce<T, val_t>(sum[r]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = row_of(g, r);
                if (row >= rows) continue;
                if (deferred[r]) {
                    unsigned rel = unsigned(row);
                    asm volatile("" : "+v"(rel));   // (opaque: the bit masks of the R rows are otherwise precomputed and kept live)
                    atomicOr(&scr.long_map[rel >> 5], 1u << (rel & 31));
                    continue;
                }
                if (B.hi[r] > nnz_vec) {                   // the last (partial) group of the arrays
                    for (off_t k = (B.lo[r] > nnz_vec ? B.lo[r] : nnz_vec); k < B.hi[r]; ++k)
                        sum[r] += Ax[k] * x[Aj[k]];
                }
                scr.s_y[row] = sum[r];
            }
        }
    };

    if constexpr (DEPTH == 2) {
        for (int g = 0; g < n_groups; g += 2) {
            issue(g + 1, G[1]);
            consume(g, G[0]);
            issue(g + 2, G[0]);
            consume(g + 1, G[1]);
        }
    } else {
        for (int g = 0; g < n_groups; g += DEPTH) {
            // (fully unrolled: every G[...] index below is a constant; groups past the end load clamped addresses
            // and store nothing)
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                issue(g + d + DEPTH - 1, G[(d + DEPTH - 1) % DEPTH]);
                consume(g + d, G[d]);
            }
        }
    }

    // second pass: every marked row by one whole wave (64 lanes x 4 nonzeros x 2 groups per step)
    __syncthreads();
    const int lane64 = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int tid2 = threadIdx.x;
    const int words = (rows + 31) >> 5;
    int turn = 0;                                   // marked rows are dealt to the waves in turn
    bool any_huge = false;                          // same in every thread: it depends on LDS contents only
    for (int w = 0; w < words; ++w) {
        unsigned bits = scr.long_map[w];            // same value in every lane
        while (bits) {
            const int bpos = __ffs(bits) - 1;
            bits &= bits - 1;
            const int local = (w << 5) + bpos;
            const off_t start = scr.s_b[local], end = scr.s_b[local + 1];
            if (end - start > off_t(kHugeRow)) {                     // left to the whole workgroup, below
                any_huge = true;
                continue;
            }
            if ((turn++ & (WAVES - 1)) != wave) continue;   // wave-uniform
            val_t sum = val_t(0);
            for (off_t j = (start & ~off_t(3)) + off_t(lane64) * 4; j < end; j += off_t(kWave) * 8) {
                int4v c0, c1;
                v4 a0, a1;
                const off_t j1 = j + off_t(kWave) * 4;
                load_group<val_t>(j, nnz, Aj, Ax, c0, a0);
                if (j1 < end) load_group<val_t>(j1, nnz, Aj, Ax, c1, a1);
                else { c1 = int4v{0, 0, 0, 0}; a1 = v4{0, 0, 0, 0}; }
                accumulate(sum, c0, a0, j, start, end);
                accumulate(sum, c1, a1, j1, start, end);
            }
            sum = vector_reduce<kWave, val_t>(sum);
            if (lane64 == 0) scr.s_y[local] = sum;
        }
    }

    // ... and a hub row (more than kHugeRow nonzeros: one wave would need many dependent steps) by all the
    // waves, partial sums folded through LDS in wave order
    if (any_huge) {
        __shared__ val_t s_part[WAVES];
        for (int w = 0; w < words; ++w) {
            unsigned bits = scr.long_map[w];
            while (bits) {
                const int bpos = __ffs(bits) - 1;
                bits &= bits - 1;
                const int local = (w << 5) + bpos;
                const off_t start = scr.s_b[local], end = scr.s_b[local + 1];
                if (end - start <= off_t(kHugeRow)) continue;            // uniform over the workgroup
                if (scr.giant_len > 0 && int64_t(end - start) > scr.giant_len) {   // split across workgroups elsewhere
                    if (tid2 == 0) scr.s_y[local] = val_t(0);
                    continue;
                }
                // the main loop's pipeline again: R slabs of BLOCK x 4 nonzeros in flight, the next
                // R issued before the current ones are consumed, branch-free clamped addresses (a slab past
                // the row re-reads the row's first line and is masked by hi_v)
                val_t sum = val_t(0);
                constexpr int64_t SLAB = int64_t(BLOCK) * 4;                    // (64-bit: the steps may pass 2^31)
                const off_t hi_v = end < nnz_vec ? end : nnz_vec;
                const off_t first = start & ~off_t(3);
                struct Slabs { int4v c[R]; v4 a[R]; };
                auto issue_slabs = [&](int64_t it, Slabs& S) {
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        const int64_t j = it + u * SLAB + int64_t(tid2) * 4;
                        off_t jl = j < int64_t(hi_v) ? off_t(j) : first;
                        jl = jl < j_max ? jl : j_max;
                        S.c[u] = stream_load(reinterpret_cast<const int4v*>(Aj + jl));
                        S.a[u] = stream_load(reinterpret_cast<const v4*>(Ax + jl));
                    }
                };
                auto eat_slabs = [&](int64_t it, const Slabs& S) {
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        const int64_t j = it + u * SLAB + int64_t(tid2) * 4;
                        accumulate(sum, S.c[u], S.a[u], j < int64_t(hi_v) ? off_t(j) : hi_v, start, hi_v);
                    }
                };
                {
                    Slabs S0, S1;
                    issue_slabs(first, S0);
                    for (int64_t it = first; it < int64_t(hi_v); it += 2 * R * SLAB) {   // uniform over the workgroup
                        issue_slabs(it + R * SLAB, S1);
                        eat_slabs(it, S0);
                        issue_slabs(it + 2 * R * SLAB, S0);
                        eat_slabs(it + R * SLAB, S1);
                    }
                }
                if (tid2 == 0 && end > nnz_vec)                                  // the arrays' last, partial group
                    for (off_t k = (start > nnz_vec ? start : nnz_vec); k < end; ++k) sum += Ax[k] * x[Aj[k]];
                sum = vector_reduce<kWave, val_t>(sum);
                if (lane64 == 0) s_part[wave] = sum;
                __syncthreads();
                if (tid2 == 0) {
                    val_t total = s_part[0];
#pragma unroll
                    for (int i = 1; i < WAVES; ++i) total += s_part[i];
                    scr.s_y[local] = total;
                }
                __syncthreads();
            }
        }
    }

    __syncthreads();
    store_chunk_results<BLOCK, val_t>(scr, y, chunk_begin, rows, tid2);
}

// chunk_rows with the vector width chosen PER CHUNK (nnz-balanced plans): a power-law matrix has chunks of
// 2 000 near-empty rows and chunks of 100 rows x 120 nonzeros; one width for all of them leaves the second
// kind walking 16 dependent steps per row.  Three widths (2, 8, 32 lanes: rows of <= 8, 32, 128 nonzeros in
// one step) from the chunk's own mean row length, read from the bounds already in LDS.
// REGULAR (the merge kind's row-parallel runs: rows alike): the width that takes a mean row in ONE step (8 / 16 / 32 /
// 64 / 128 nonzeros) — a second, dependent step per row costs such a matrix more than idle lanes do (rows of 64 +- 16 with 8
// lanes: three steps, 740 us; with 32 lanes: see merge_path.hip).
template <int BLOCK, int T, int R, bool WINDOW, bool ADAPT, typename val_t, typename StageFn, bool REGULAR = false>
__device__ __forceinline__ void chunk_rows_any(int64_t chunk_begin, int64_t chunk_end, int32_t nnz,
                                               const int32_t* __restrict__ Aj,
                                               const val_t* __restrict__ Ax, const val_t* __restrict__ x,
                                               val_t* __restrict__ y, StageFn&& stage,
                                               const ChunkScratch<val_t>& scr) {
    if constexpr (!ADAPT) {
        chunk_rows<BLOCK, T, R, WINDOW, val_t>(chunk_begin, chunk_end, nnz, Aj, Ax, x, y, stage, scr);
    } else {
        const int rows = int(chunk_end - chunk_begin);
        const int32_t mean = (scr.s_b[rows] - scr.s_b[0]) / int32_t(rows > 0 ? rows : 1);   // uniform over the workgroup
        if constexpr (REGULAR) {       // five widths: one step of 8 / 16 / 32 / 64 / 128 nonzeros
            if (mean <= 8) chunk_rows<BLOCK, 2, R, WINDOW, val_t>(chunk_begin, chunk_end, nnz, Aj, Ax, x, y, stage, scr);
            else if (mean <= 16) chunk_rows<BLOCK, 4, R, WINDOW, val_t>(chunk_begin, chunk_end, nnz, Aj, Ax, x, y, stage, scr);
            else if (mean <= 32) chunk_rows<BLOCK, 8, R, WINDOW, val_t>(chunk_begin, chunk_end, nnz, Aj, Ax, x, y, stage, scr);
            else if (mean <= 64) chunk_rows<BLOCK, 16, R, WINDOW, val_t>(chunk_begin, chunk_end, nnz, Aj, Ax, x, y, stage, scr);
            else chunk_rows<BLOCK, 32, R, WINDOW, val_t>(chunk_begin, chunk_end, nnz, Aj, Ax, x, y, stage, scr);
            return;
        }
        if (mean <= 16) chunk_rows<BLOCK, 2, R, WINDOW, val_t>(chunk_begin, chunk_end, nnz, Aj, Ax, x, y, stage, scr);
        else if (mean <= 64) chunk_rows<BLOCK, 8, R, WINDOW, val_t>(chunk_begin, chunk_end, nnz, Aj, Ax, x, y, stage, scr);
        else chunk_rows<BLOCK, 32, R, WINDOW, val_t>(chunk_begin, chunk_end, nnz, Aj, Ax, x, y, stage, scr);
    }
}

// A band too wide for ANY window (more columns than a CU's LDS holds): the window SWEEPS the band.
// The plain gather is bound by line fills — every lane of a gather instruction pulls its own 128-byte line
// into the CU for 4 useful bytes (measured: ~96 cycles per 64-lane gather at a 131 K-column band) — while a
// staged window moves whole lines at the same fill rate with every byte used.  So a chunk is exactly ONE group
// of rows (BLOCK / T vectors x R rows: its step-0 loads of Aj / Ax stay in registers), the band of the chunk is
// cut into windows of `cap` columns, and for every window in turn the workgroup stages it and each lane adds
// the elements whose column falls inside.  Per chunk: BLOCK * R * 4 nonzero slots (2 MB of line fills as plain
// gathers at 16 K slots) against passes * cap * sizeof(val_t) staged bytes — the plan takes this shape only
// while the second is well below the first (analyze.hip, shape_chunks).
// The band is the probe's (a sample): a column outside the swept span is gathered from global memory, a row
// longer than one step (4 T nonzeros; the plan picks T from the longest row it saw) finishes with plain
// gathers.  With sorted columns a lane adds its elements in the order chunk_rows does.
// Same contract as chunk_rows: all BLOCK threads call, the caller has run stage_chunk_bounds + a barrier.
template <int BLOCK, int T, int R, typename val_t>
__device__ __forceinline__ void chunk_rows_sweep(int64_t chunk_begin, int64_t chunk_end, int32_t nnz,
                                                 const int32_t* __restrict__ Aj, const val_t* __restrict__ Ax,
                                                 const val_t* __restrict__ x, val_t* __restrict__ y,
                                                 int32_t n_cols, int32_t cap, const BandHint hint,
                                                 const ChunkScratch<val_t>& scr) {
    using v4 = typename Vec4<val_t>::type;
    using off_t = int32_t;
    constexpr int PER16 = 16 / int(sizeof(val_t));
    constexpr int VW = kWave / T;
    // (opaque copy of the thread index: inside the kernel's loop over chunks the optimiser otherwise hoists every
    // per-thread value below out of the loop, keeps them all live across the chunk and spills)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & (T - 1);
    const int rows = int(chunk_end - chunk_begin);          // <= BLOCK / T * R (the plan's rows per chunk)
    const off_t nnz_vec = nnz & ~off_t(3);
    const off_t j_max = nnz_vec - 4;
    const int row_in_wave = (tid & (kWave - 1)) / T;
    const int wave_rows0 = (tid / kWave) * (VW * R);
    auto row_of = [&](int r) { return wave_rows0 + r * VW + row_in_wave; };   // (chunk_rows' mapping, group 0)

    int4v c[R];
    v4 a[R];
    val_t sum[R];
    // (the row bounds are read from LDS again after the sweep instead of being held in registers across it)
    auto bounds = [&](int r, off_t& lo, off_t& hi, off_t& j) {
        const int row = row_of(r);
        lo = scr.s_b[min(row, rows)];
        hi = scr.s_b[min(row + 1, rows)];
        j = (lo & ~off_t(3)) + off_t(lane) * 4;
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
        off_t lo, hi, j;
        bounds(r, lo, hi, j);
        off_t jl = j < hi ? j : (lo & ~off_t(3));
        jl = jl < j_max ? jl : j_max;
        c[r] = stream_load(reinterpret_cast<const int4v*>(Aj + jl));
        a[r] = stream_load(reinterpret_cast<const v4*>(Ax + jl));
        const off_t top = hi < nnz_vec ? hi : nnz_vec;       // (the arrays' last, partial group: scalar, below)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool mine = j + e >= lo && j + e < top;
            c[r][e] = mine ? c[r][e] : INT32_MAX;            // (no column: a slot that is not the row's, here and below)
        }
        sum[r] = val_t(0);
    }

    // the columns the chunk's rows may hold according to the plan's band, whole 16-byte groups of x
    int64_t l = chunk_begin + hint.lo, h = chunk_end - 1 + hint.hi;
    l = l < 0 ? 0 : l;
    h = h >= n_cols ? int64_t(n_cols) - 1 : h;
    // (the last n_cols % PER16 columns of x are not part of any staged window — a window is whole 16-byte groups,
    // written by LDS-DMA — and fall to the gathers below with everything else outside the swept span; h < l: an empty
    // sweep, everything falls to the gathers)
    const int whole_end = (n_cols & ~(PER16 - 1)) - 1;
    const int c_lo = int(l) & ~(PER16 - 1), c_hi = int(h) < whole_end ? int(h) : whole_end;
    // (Tried and dropped, rounds 2-3: staging through registers — 10 loads per thread and batch, 4 for the eight-row
    // body: a 150 KB window was three dependent round trips; the loads of window p + 1 held in registers across the
    // consume of window p — 203 vs 190 us at two passes, 432 vs 370 at seven; a persistent workgroup per CU walking its
    // share of the chunks — 194 vs 187, 438 vs 358; touching the Aj / Ax lines of the chunk one round of the chip ahead
    // while sweeping — 414 vs 370 us on 2^22 rows at two passes, 570 vs 508 at four.)
    cap = (cap / (BLOCK * PER16)) * (BLOCK * PER16);           // whole rounds of the workgroup's 16-byte groups (the launchers check cap >= one round)
    for (int w0 = c_lo; w0 <= c_hi; w0 += cap) {               // uniform over the workgroup; cap is a multiple of PER16
        const int len = min(cap, c_hi + 1 - w0);
        // 16-byte groups that hold the window's columns: its last, partial group is staged whole (c_hi + 1 need not be a
        // multiple of PER16; the group lies inside x, c_hi stops at the last whole group of x)
        int full = (min((w0 + len + PER16 - 1) & ~(PER16 - 1), n_cols & ~(PER16 - 1)) - w0) / PER16;
        full = full > 0 ? full : 0;
        if (w0 != c_lo) __syncthreads();                       // the previous window is still being read
        // The window goes from global memory STRAIGHT into LDS (global_load_lds_dwordx4: no destination registers, so
        // every 16-byte group of the window is in flight at once whatever the chunk holds in registers).  Through
        // registers the eight-row fp32 body had 4 loads per thread and batch: a window of 150 KB was three DEPENDENT
        // round trips to L2, and staging — not the Aj / Ax stream, which this structure reads at 7.8 TB/s, nor the
        // consume — was more than half the kernel (321 us; 144 without the passes; 338 with the passes and half the
        // consumes).  A wave-instruction writes LDS at a wave-uniform base + 16 bytes x lane, which is the window's
        // own layout.  cap is a whole number of such rounds of the workgroup (analyze.hip: shape_sweep), so every lane
        // of every instruction has a slot inside the window's LDS: a group past the window's end re-loads the last
        // whole group of x into a slot nothing reads.  The instructions count on vmcnt: the wait + the barrier below
        // order them before every wave's ds_reads.
        {
            const int wave_first = (tid & ~(kWave - 1));             // this wave's first thread
            const int rounds = cap / (BLOCK * PER16);
            const int last = full > 0 ? full - 1 : 0;
            for (int u = 0; u < rounds; ++u) {                       // uniform
                const int g = u * BLOCK + tid;
                const int gs = g < full ? g : last;
                const val_t* src = x + w0 + gs * PER16;
                val_t* dst = scr.s_x + (u * BLOCK + wave_first) * PER16;     // wave-uniform: + lane * 16 bytes by the hardware
                if (full > 0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            val_t xv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned rel = unsigned(c[r][e] - w0);
                xv[e] = scr.s_x[rel < unsigned(len) ? rel : 0u];
            }
            // (the four values go through an empty volatile asm, i.e. the reads happen whatever the comparison says:
            // left alone, hipcc turns the selects below into a branch per element — ds_read, wait, fma under
            // s_cbranch_execz, sixteen times in a row)
            asm volatile("" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = unsigned(c[r][e] - w0) < unsigned(len);
                sum[r] = in ? (sum[r] + a[r][e] * xv[e]) : sum[r];
            }
        }
    }
    // what the probe's sample of the band missed: columns outside the swept span
    {
        bool any = false;
        unsigned out = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c[r][e] != INT32_MAX && (c[r][e] < c_lo || c[r][e] > c_hi)) { out |= 1u << (r * 4 + e); any = true; }
        if (any) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((out >> (r * 4 + e)) & 1u) sum[r] += a[r][e] * x[c[r][e]];
        }
    }
    // ... and rows longer than one step: the remaining steps with plain gathers
#pragma unroll
    for (int r = 0; r < R; ++r) {
        off_t lo, hi, j;
        bounds(r, lo, hi, j);
        const off_t top = hi < nnz_vec ? hi : nnz_vec;
        for (off_t jj = j + off_t(T) * 4; jj < top; jj += off_t(T) * 4) {
            const off_t jl = jj < j_max ? jj : j_max;           // (jj < top <= nnz_vec: jl == jj)
            const int4v c2 = stream_load(reinterpret_cast<const int4v*>(Aj + jl));
            const v4 a2 = stream_load(reinterpret_cast<const v4*>(Ax + jl));
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (jj + e < top) sum[r] += a2[e] * x[c2[e]];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) sum[r] = vector_reduce<T, val_t>(sum[r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row_of(r);
            if (row >= rows) continue;
            const off_t lo = scr.s_b[row], hi = scr.s_b[row + 1];
            if (hi > nnz_vec)                                  // the last (partial) group of the arrays
                for (off_t k = (lo > nnz_vec ? lo : nnz_vec); k < hi; ++k) sum[r] += Ax[k] * x[Aj[k]];
            scr.s_y[row] = sum[r];
        }
    }
    __syncthreads();
    store_chunk_results<BLOCK, val_t>(scr, y, chunk_begin, rows);
}

// A chunk whose nonzeros span more than the 32-bit path can index (see kRel32Limit): one wave per row,
// 4-byte loads, 64-bit indices, results straight to y.  Slow, and only ever reached by matrices with rows
// of ~10^9 nonzeros; summation order = the fallback kernels' (row_dot.hpp).
template <int BLOCK, typename val_t>
__device__ __forceinline__ void chunk_rows_wide(int64_t chunk_begin, int64_t chunk_end, const ApView Ap,
                                                const int32_t* __restrict__ Aj, const val_t* __restrict__ Ax,
                                                const val_t* __restrict__ x, val_t* __restrict__ y, val_t alpha,
                                                val_t beta, int64_t giant_len) {
    const int lane64 = threadIdx.x & (kWave - 1);
    for (int64_t row = chunk_begin + threadIdx.x / kWave; row < chunk_end; row += BLOCK / kWave) {   // wave-uniform
        const int64_t start = Ap.at(row), end = Ap.at(row + 1);
        val_t sum = val_t(0);
        if (!(giant_len > 0 && end - start > giant_len))           // (a giant row: 0 here, the slice kernels add it)
            for (int64_t k = start + lane64; k < end; k += kWave) sum += Ax[k] * x[Aj[k]];
        sum = vector_reduce<kWave, val_t>(sum);
        if (lane64 == 0) y[row] = (beta != val_t(0)) ? alpha * sum + beta * y[row] : alpha * sum;
    }
}

// Rows per workgroup chunk: ~32 K nonzeros (256 KB of fp32 stream) per chunk, a
// multiple of the rows one pass of the workgroup covers.
inline int64_t pick_rows_per_chunk(int64_t nnz, int64_t n_rows, int lanes_per_row, int rows_in_flight,
                                   int block_threads = kBlock, int64_t nnz_per_chunk = 32768, int per_cu = 4) {
    const int64_t pass = int64_t(block_threads / lanes_per_row) * rows_in_flight;
    const int64_t mean = n_rows > 0 ? (nnz + n_rows - 1) / n_rows : 1;
    int64_t rows = nnz_per_chunk / (mean > 0 ? mean : 1);
    // small matrices: prefer one chunk per workgroup slot (per_cu a CU) over long chunks
    const int64_t fill = (n_rows + int64_t(kCus) * per_cu - 1) / (int64_t(kCus) * per_cu);
    if (rows > fill) rows = fill;
    rows = (rows + pass - 1) / pass * pass;
    if (rows < pass) rows = pass;
    if (rows > kMaxChunkRows) rows = kMaxChunkRows / pass * pass > 0 ? kMaxChunkRows / pass * pass : pass;
    return rows;
}

constexpr int kWindowBytes = 36 * 1024;   // LDS window of x per workgroup: 4 workgroups per CU

}  // namespace mi355
