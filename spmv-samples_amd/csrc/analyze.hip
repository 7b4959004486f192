// analyze.hip — structure probe run once at plan creation.
//
// Decides ONE launch-shape question: is an LDS window of x worth its LDS?  A window
// costs 36 KB per workgroup (occupancy) and pays only if it can hold most of the
// columns a chunk of rows touches.  256 rows spread over the matrix are sampled
// (first and last column of each); the band [min(col - row), max(col - row)] over
// the samples, plus the rows a workgroup owns, is the span a window would have to
// cover.  The result never affects correctness: the kernels range-check every
// column against the window they actually staged (xwindow.hpp).
// The reference has no counterpart (no plan, no LDS staging of x).

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <new>

#include "common.hpp"
#include "xwindow.hpp"

namespace mi355 {

constexpr int kProbePerRow = 32;   // (column - row) samples per probed row

template <typename off_t>
__global__ __launch_bounds__(kBlock) void probe_kernel(int32_t n_rows, const off_t* __restrict__ Ap,
                                                       const int32_t* __restrict__ Aj, long long* out) {
    // out[0], out[1]: band [lo, hi] over the first/last column of 256 rows;
    // out[2 + 32 t + i]: (column - row) at 32 positions spread over row t's nonzeros (LLONG_MAX = none)
    __shared__ long long s_lo[kBlock / kWave], s_hi[kBlock / kWave];
    const int tid = threadIdx.x;
    long long lo = LLONG_MAX, hi = LLONG_MIN;
    for (int i = 0; i < kProbePerRow; ++i) out[2 + tid * kProbePerRow + i] = LLONG_MAX;
    if (n_rows > 0) {
        const int64_t r = (int64_t(n_rows - 1) * tid) / (kBlock - 1);
        const off_t s = Ap[r], e = Ap[r + 1];
        if (e > s) {
            const long long first = Aj[s], last = Aj[e - 1];
            lo = min(first, last) - r;
            hi = max(first, last) - r;
            const off_t len = e - s;
            const int take = len < kProbePerRow ? int(len) : kProbePerRow;
            for (int i = 0; i < take; ++i) {
                const off_t k = take > 1 ? s + ((len - 1) * i) / (take - 1) : s;
                out[2 + tid * kProbePerRow + i] = (long long)Aj[k] - r;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, kWave));
        hi = max(hi, __shfl_xor(hi, o, kWave));
    }
    if ((tid & (kWave - 1)) == 0) {
        s_lo[tid / kWave] = lo;
        s_hi[tid / kWave] = hi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kBlock / kWave; ++w) {
            lo = min(lo, s_lo[w]);
            hi = max(hi, s_hi[w]);
        }
        out[0] = lo;
        out[1] = hi;
    }
}

int probe_structure(Plan& p) {
    p.probe_ok = false;
    p.band_lo = p.band_hi = 0;
    p.probe_n = 0;
    p.n_seg = 0;
    if (p.n_rows <= 0 || p.nnz <= 0) return MI355_SPMV_OK;
    constexpr size_t N = 2 + size_t(kBlock) * kProbePerRow;
    static_assert(size_t(kBlock) * kProbePerRow <= sizeof(p.probe_off) / sizeof(p.probe_off[0]), "probe buffer");
    long long* d_out = nullptr;
    MI355_HIP_TRY(hipMalloc(&d_out, N * sizeof(long long)));
    if (p.off_type == MI355_OFF_I32)
        hipLaunchKernelGGL((probe_kernel<int32_t>), dim3(1), dim3(kBlock), 0, nullptr, p.n_rows,
                           static_cast<const int32_t*>(p.Ap), p.Aj, d_out);
    else
        hipLaunchKernelGGL((probe_kernel<int64_t>), dim3(1), dim3(kBlock), 0, nullptr, p.n_rows,
                           static_cast<const int64_t*>(p.Ap), p.Aj, d_out);
    long long* h = new (std::nothrow) long long[N];
    hipError_t e = h ? hipGetLastError() : hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMemcpy(h, d_out, N * sizeof(long long), hipMemcpyDeviceToHost);   // synchronises
    (void)hipFree(d_out);
    if (e != hipSuccess) {
        delete[] h;
        set_error("probe_structure: %s", hipGetErrorString(e));
        return MI355_SPMV_EHIP;
    }
    if (h[0] <= h[1]) {
        p.band_lo = h[0];
        p.band_hi = h[1];
        p.probe_ok = true;
        for (size_t i = 2; i < N; ++i)
            if (h[i] != LLONG_MAX) p.probe_off[p.probe_n++] = h[i];
        std::sort(p.probe_off, p.probe_off + p.probe_n);
    }
    delete[] h;
    return MI355_SPMV_OK;
}

// Cluster the sampled offsets into bands: a gap wider than the rows a workgroup owns starts a
// new band (splitting costs `rows` extra columns per band, keeping the gap costs the gap).
// Returns the columns a workgroup would have to hold: sum of (band width + rows).
static int64_t cluster_bands(Plan& p, int64_t rows) {
    p.n_seg = 0;
    if (p.probe_n == 0) return 0;
    int64_t lo[64], hi[64];
    int n = 0;
    lo[0] = hi[0] = p.probe_off[0];
    for (int i = 1; i < p.probe_n; ++i) {
        if (p.probe_off[i] - hi[n] > rows && n + 1 < 64) {
            ++n;
            lo[n] = p.probe_off[i];
        }
        hi[n] = p.probe_off[i];
    }
    ++n;
    while (n > kMaxSegments) {   // merge across the narrowest gap
        int best = 0;
        for (int i = 1; i + 1 < n; ++i)
            if (lo[i + 1] - hi[i] < lo[best + 1] - hi[best]) best = i;
        hi[best] = hi[best + 1];
        for (int i = best + 1; i + 1 < n; ++i) { lo[i] = lo[i + 1]; hi[i] = hi[i + 1]; }
        --n;
    }
    int64_t need = 0;
    for (int i = 0; i < n; ++i) {
        // the samples may miss a band's edges: widen each by 1/16 of its width + 8 columns
        const int64_t pad = (hi[i] - lo[i]) / 16 + 8;
        p.seg_lo[i] = lo[i] - pad;
        p.seg_hi[i] = hi[i] + pad;
        need += (p.seg_hi[i] - p.seg_lo[i] + 1) + rows;
    }
    p.n_seg = n;
    return need;
}

// Window size (elements) for a workgroup that owns `rows_per_workgroup` consecutive
// rows: the full 36 KB when the sampled band plus those rows fits within 1.5x of
// it (the kernels centre a too-small window on the span), else none.
// MI355_SPMV_WINDOW=0|1 forces the choice (tuning / tests).
int pick_window_elems(Plan& p, int64_t rows_per_workgroup) {
    const int val_bytes = p.val_type == MI355_VAL_F64 ? 8 : 4;
    const int cap = kWindowBytes / val_bytes;
    p.window_from_band = false;
    const char* force = getenv("MI355_SPMV_WINDOW");
    if (force) return atoi(force) ? cap : 0;
    if (!p.probe_ok) return 0;
    const int64_t span = (p.band_hi - p.band_lo + 1) + rows_per_workgroup;
    // the band (plus the chunk's rows) all but fits: no need to sample every chunk
    const char* band = getenv("MI355_SPMV_WINDOW_FROM_BAND");
    p.window_from_band = band ? atoi(band) != 0 : span <= int64_t(cap) * 9 / 8;
    p.n_seg = 0;
    if (span <= int64_t(cap) * 3 / 2) return cap;
    // one window cannot hold the band: do a few narrow ones?  (MI355_SPMV_SEGMENTS=0 disables)
    const char* seg = getenv("MI355_SPMV_SEGMENTS");
    if (!(seg && atoi(seg) == 0)) {
        const int64_t need = cluster_bands(p, rows_per_workgroup);
        // up to 1.25x: the tail of the last band is cut and falls back to global loads; the
        // row-based kinds then shrink their chunk so that everything fits (segment_rows_fit)
        if (p.n_seg >= 2 && need <= int64_t(cap) * 5 / 4) return cap;
    }
    p.n_seg = 0;
    return 0;
}

// Rows per workgroup for which the plan's bands fit the window exactly:
// sum_k (width_k + rows) <= cap.  0 when there are no segments.
int64_t segment_rows_fit(const Plan& p) {
    if (p.n_seg < 2 || p.window_elems <= 0) return 0;
    int64_t width = 0;
    for (int i = 0; i < p.n_seg; ++i) width += p.seg_hi[i] - p.seg_lo[i] + 1 + 4;   // +4: 16-byte rounding
    const int64_t fit = (int64_t(p.window_elems) - width) / p.n_seg;
    return fit > 0 ? fit : 0;
}

}  // namespace mi355
