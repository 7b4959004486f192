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
// into s_x (capacity `cap` elements).  All kBlock threads must call; ends with a
// barrier.  s_red: 2 * (kBlock / kWave) ints of LDS scratch.
// BandHint: the band [lo, hi] of (column - row) the plan's probe saw (analyze.hip).  When the
// whole band of a chunk fits the window (use == true) the window is placed from it and
// the per-chunk sample — two dependent loads and a barrier — is skipped.
struct BandHint {
    int64_t lo, hi;
    bool use;
};

template <typename off_t, typename val_t>
__device__ __forceinline__ XWindow<val_t> stage_x_window(int64_t rb, int64_t re, int32_t n_cols,
                                                         const off_t* __restrict__ Ap,
                                                         const int32_t* __restrict__ Aj,
                                                         const val_t* __restrict__ x, val_t* s_x,
                                                         int32_t cap, int* s_red,
                                                         const BandHint hint = BandHint{0, 0, false}) {
    const int tid = threadIdx.x;
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
        const off_t s = Ap[r], e = Ap[r + 1];
        if (e > s) {
            const int first = Aj[s], last = Aj[e - 1];
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
    int len = hint.use ? span : span + 2 * (span / 8 + 64);
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
    for (int g = tid; g < full; g += kBlock) {
        *reinterpret_cast<v16*>(s_x + g * PER16) = *reinterpret_cast<const v16*>(x + lo + g * PER16);
    }
    for (int i = full * PER16 + tid; i < len; i += kBlock) s_x[i] = x[lo + i];
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

// LDS scratch of one workgroup for chunk_rows.
constexpr int kMaxChunkRows = 8192;   // upper bound of rows per chunk (pick_rows_per_chunk)
constexpr int kLongSteps = 16;        // a row is "long" beyond this many steps of its T-lane vector
struct ChunkScratch {
    // one bit per row of the chunk: set = long row, summed in the second pass.
    // Zeroed by the caller (zero_long_map) before the barrier that precedes chunk_rows.
    unsigned* long_map;   // [kMaxChunkRows / 32]
};
__device__ __forceinline__ void zero_long_map(unsigned* long_map) {
    for (int i = threadIdx.x; i < kMaxChunkRows / 32; i += kBlock) long_map[i] = 0u;
}

// One step of 4 nonzeros of one lane: Aj/Ax group at j (16-byte aligned element index).
template <typename off_t, typename val_t>
__device__ __forceinline__ void load_group(off_t j, off_t nnz, const int32_t* __restrict__ Aj,
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

// Rows [chunk_begin, chunk_end) by this workgroup: T lanes per row, R rows per
// vector in flight (R x the bytes in flight of one row: the loads of the R rows are
// issued back to back before any is consumed), 4 nonzeros per lane per step.
// A row longer than kLongSteps steps is not walked by its T lanes (a power-law hub
// row would serialise the whole chunk behind one vector — the weakness of the
// reference's CSR-vector and LightSpMV kernels on such inputs): its bit is set in an
// LDS bitmap and a second pass sums every marked row with a whole 64-lane wave,
// 512 nonzeros per step, the four waves taking marked rows in turn.
// All kBlock threads must call (wave-wide shuffles and one barrier inside).
template <int T, int R, typename off_t, typename val_t>
__device__ __forceinline__ void chunk_rows(int64_t chunk_begin, int64_t chunk_end, off_t nnz,
                                           const off_t* __restrict__ Ap, const int32_t* __restrict__ Aj,
                                           const val_t* __restrict__ Ax, const val_t* __restrict__ x,
                                           val_t* __restrict__ y, const XWindow<val_t>& win,
                                           const ChunkScratch& scr) {
    using v4 = typename Vec4<val_t>::type;
    constexpr int VECS = kBlock / T;
    constexpr off_t LONG = off_t(T) * 4 * kLongSteps;
    const int lane = threadIdx.x & (T - 1);
    const int vec = threadIdx.x / T;
    for (int64_t base = chunk_begin; base < chunk_end; base += VECS * R) {
        const int64_t row0 = base + int64_t(vec) * R;
        off_t bound[R + 1];
#pragma unroll
        for (int r = 0; r <= R; ++r) {
            const int64_t row = row0 + r;
            bound[r] = Ap[row < chunk_end ? row : chunk_end];   // rows past the chunk become empty
        }
        off_t j[R];
        val_t sum[R];
        bool deferred[R];
        bool more = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            deferred[r] = (bound[r + 1] - bound[r]) > LONG;     // uniform over the T lanes of the vector
            if (deferred[r] && lane == 0) {
                const unsigned rel = unsigned(row0 + r - chunk_begin);
                atomicOr(&scr.long_map[rel >> 5], 1u << (rel & 31));
            }
            j[r] = deferred[r] ? bound[r + 1] : (bound[r] & ~off_t(3)) + off_t(lane) * 4;
            sum[r] = val_t(0);
            more |= j[r] < bound[r + 1];
        }
        while (more) {
            int4v c[R];
            v4 a[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (j[r] < bound[r + 1]) {
                    load_group<off_t, val_t>(j[r], nnz, Aj, Ax, c[r], a[r]);
                } else {
                    c[r] = int4v{0, 0, 0, 0};
                    a[r] = v4{0, 0, 0, 0};
                }
            }
            more = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const off_t k = j[r] + e;
                    const bool valid = (k >= bound[r]) && (k < bound[r + 1]);
                    const val_t xv = window_gather<val_t>(win, x, c[r][e], valid);
                    sum[r] = valid ? (sum[r] + a[r][e] * xv) : sum[r];
                }
                j[r] += off_t(T) * 4;
                more |= j[r] < bound[r + 1];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) sum[r] = vector_reduce<T, val_t>(sum[r]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (row0 + r < chunk_end && !deferred[r]) y[row0 + r] = sum[r];
            }
        }
    }

    // second pass: every marked row by one whole wave (64 lanes x 4 nonzeros x 2 groups per step)
    __syncthreads();
    const int lane64 = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int words = int((chunk_end - chunk_begin + 31) >> 5);
    int turn = 0;                                   // marked rows are dealt to the waves in turn
    for (int w = 0; w < words; ++w) {
        unsigned bits = scr.long_map[w];            // same value in every lane
        while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            if ((turn++ & (kBlock / kWave - 1)) != wave) continue;   // wave-uniform
            const int64_t row = chunk_begin + (int64_t(w) << 5) + b;
            const off_t start = Ap[row], end = Ap[row + 1];
            val_t sum = val_t(0);
            for (off_t j = (start & ~off_t(3)) + off_t(lane64) * 4; j < end; j += off_t(kWave) * 8) {
                int4v c0, c1;
                v4 a0, a1;
                const off_t j1 = j + off_t(kWave) * 4;
                load_group<off_t, val_t>(j, nnz, Aj, Ax, c0, a0);
                if (j1 < end) load_group<off_t, val_t>(j1, nnz, Aj, Ax, c1, a1);
                else { c1 = int4v{0, 0, 0, 0}; a1 = v4{0, 0, 0, 0}; }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool valid = (j + e >= start) && (j + e < end);
                    const val_t xv = window_gather<val_t>(win, x, c0[e], valid);
                    sum = valid ? (sum + a0[e] * xv) : sum;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool valid = (j1 + e < end);
                    const val_t xv = window_gather<val_t>(win, x, c1[e], valid);
                    sum = valid ? (sum + a1[e] * xv) : sum;
                }
            }
            sum = vector_reduce<kWave, val_t>(sum);
            if (lane64 == 0) y[row] = sum;
        }
    }
}

// Rows per workgroup chunk: ~32 K nonzeros (256 KB of fp32 stream) per chunk, a
// multiple of the rows one pass of the workgroup covers.
inline int64_t pick_rows_per_chunk(int64_t nnz, int64_t n_rows, int lanes_per_row, int rows_in_flight) {
    const int64_t pass = int64_t(kBlock / lanes_per_row) * rows_in_flight;
    const int64_t mean = n_rows > 0 ? (nnz + n_rows - 1) / n_rows : 1;
    int64_t rows = 32768 / (mean > 0 ? mean : 1);
    // small matrices: prefer >= 4 chunks per CU over long chunks
    const int64_t fill = (n_rows + int64_t(kCus) * 4 - 1) / (int64_t(kCus) * 4);
    if (rows > fill) rows = fill;
    rows = (rows + pass - 1) / pass * pass;
    if (rows < pass) rows = pass;
    if (rows > kMaxChunkRows) rows = kMaxChunkRows / pass * pass > 0 ? kMaxChunkRows / pass * pass : pass;
    return rows;
}

constexpr int kWindowBytes = 36 * 1024;   // LDS window of x per workgroup: 4 workgroups per CU

}  // namespace mi355
