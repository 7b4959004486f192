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

#include <cstdlib>

#include "common.hpp"
#include "xwindow.hpp"

namespace mi355 {

template <typename off_t>
__global__ __launch_bounds__(kBlock) void probe_kernel(int32_t n_rows, const off_t* __restrict__ Ap,
                                                       const int32_t* __restrict__ Aj, long long* out) {
    __shared__ long long s_lo[kBlock / kWave], s_hi[kBlock / kWave];
    const int tid = threadIdx.x;
    long long lo = LLONG_MAX, hi = LLONG_MIN;
    if (n_rows > 0) {
        const int64_t r = (int64_t(n_rows - 1) * tid) / (kBlock - 1);
        const off_t s = Ap[r], e = Ap[r + 1];
        if (e > s) {
            const long long first = Aj[s], last = Aj[e - 1];
            lo = min(first, last) - r;
            hi = max(first, last) - r;
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
    if (p.n_rows <= 0 || p.nnz <= 0) return MI355_SPMV_OK;
    long long* d_out = nullptr;
    MI355_HIP_TRY(hipMalloc(&d_out, 2 * sizeof(long long)));
    if (p.off_type == MI355_OFF_I32)
        hipLaunchKernelGGL((probe_kernel<int32_t>), dim3(1), dim3(kBlock), 0, nullptr, p.n_rows,
                           static_cast<const int32_t*>(p.Ap), p.Aj, d_out);
    else
        hipLaunchKernelGGL((probe_kernel<int64_t>), dim3(1), dim3(kBlock), 0, nullptr, p.n_rows,
                           static_cast<const int64_t*>(p.Ap), p.Aj, d_out);
    long long h[2] = {0, 0};
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);   // synchronises
    (void)hipFree(d_out);
    if (e != hipSuccess) {
        set_error("probe_structure: %s", hipGetErrorString(e));
        return MI355_SPMV_EHIP;
    }
    if (h[0] <= h[1]) {
        p.band_lo = h[0];
        p.band_hi = h[1];
        p.probe_ok = true;
    }
    return MI355_SPMV_OK;
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
    return span <= int64_t(cap) * 3 / 2 ? cap : 0;
}

}  // namespace mi355
