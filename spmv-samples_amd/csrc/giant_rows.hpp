// giant_rows.hpp — rows of more than kGiantRow nonzeros in the row-based kinds (VECTOR, LIGHT).
//
// A row is summed by one workgroup, however the chunks are cut; for a row of millions of nonzeros (a dense
// row in an otherwise banded matrix) that workgroup streams alone while the chip idles (2^21 banded rows + one
// row of 4 M entries: 5.8 ms, merge 0.33 ms).  Plans with weight-cut chunks therefore list such rows at
// creation (find_giant_rows, analyze.hip); the chunk kernel stores 0 for them, and two small kernels follow:
//   giant_slices_kernel   one workgroup per slice of giant_slice_for(giant_len) nonzeros: partial sum -> scratch
//   giant_final_kernel    one thread per giant row: adds the row's partials IN SLICE ORDER into y
// so the result does not depend on which workgroup finishes first (no float atomics: run-to-run bitwise
// reproducible like everything else here).  The reference's row-based kernels have no counterpart: a long
// row serialises its vector / warp (LightSpMV.cuh:128-132, cusp_warp_reduce.cuh:26-57).
#pragma once

#include "common.hpp"
#include "row_dot.hpp"

namespace mi355 {

template <typename off_t, typename val_t>
__global__ __launch_bounds__(kBlock) void giant_slices_kernel(
    int n_giant, const int32_t* __restrict__ giant_row, const int64_t* __restrict__ slice_first,
    const off_t* __restrict__ Ap, const int32_t* __restrict__ Aj, const val_t* __restrict__ Ax,
    const val_t* __restrict__ x, val_t* __restrict__ partial, int64_t slice_len) {
    __shared__ val_t s_part[kBlock / kWave];
    const int64_t slice = blockIdx.x;
    int lo = 0, hi = n_giant;                    // the giant row this slice belongs to: last g with slice_first[g] <= slice
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (slice_first[mid] <= slice) lo = mid;
        else hi = mid;
    }
    const int32_t row = giant_row[lo];
    const int64_t begin = int64_t(Ap[row]) + (slice - slice_first[lo]) * slice_len;
    const int64_t row_end = int64_t(Ap[row + 1]);
    const int64_t end = begin + slice_len < row_end ? begin + slice_len : row_end;
    val_t sum = val_t(0);
    // 4 independent loads per thread per trip; the columns of such a row are all over x: plain gathers
    for (int64_t k = begin + threadIdx.x; k < end; k += int64_t(kBlock) * 4) {
        val_t a[4];
        int32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t kk = k + int64_t(u) * kBlock;
            const bool in = kk < end;
            c[u] = in ? Aj[kk] : 0;
            a[u] = in ? Ax[kk] : val_t(0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) sum += a[u] * x[c[u]];
    }
    sum = vector_reduce<kWave, val_t>(sum);
    if ((threadIdx.x & (kWave - 1)) == 0) s_part[threadIdx.x / kWave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        val_t total = s_part[0];
#pragma unroll
        for (int w = 1; w < kBlock / kWave; ++w) total += s_part[w];
        partial[slice] = total;
    }
}

template <typename val_t>
__global__ __launch_bounds__(kBlock) void giant_final_kernel(int n_giant, const int32_t* __restrict__ giant_row,
                                                             const int64_t* __restrict__ slice_first,
                                                             const val_t* __restrict__ partial, val_t* __restrict__ y,
                                                             val_t alpha) {
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n_giant) return;
    val_t sum = val_t(0);
    for (int64_t s = slice_first[g]; s < slice_first[g + 1]; ++s) sum += partial[s];
    y[giant_row[g]] += alpha * sum;          // the chunk kernel left beta * y_old (0 when beta == 0) there
}

// after the chunk kernel, same stream
template <typename off_t, typename val_t>
static int launch_giant_rows(const Plan& p, const off_t* Ap, const val_t* Ax, const val_t* x, val_t* y, hipStream_t s) {
    if (p.n_giant <= 0) return MI355_SPMV_OK;
    hipLaunchKernelGGL((giant_slices_kernel<off_t, val_t>), dim3((unsigned)p.n_giant_slices), dim3(kBlock), 0, s,
                       p.n_giant, p.giant_row, p.giant_slice_first, Ap, p.Aj, Ax, x,
                       static_cast<val_t*>(p.giant_partial), giant_slice_for(p.giant_len));
    MI355_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL((giant_final_kernel<val_t>), dim3((unsigned)((p.n_giant + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       p.n_giant, p.giant_row, p.giant_slice_first, static_cast<const val_t*>(p.giant_partial), y,
                       (val_t)p.alpha);
    MI355_HIP_TRY(hipGetLastError());
    return MI355_SPMV_OK;
}

}  // namespace mi355
