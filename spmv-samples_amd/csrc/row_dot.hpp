// row_dot.hpp — small pieces shared by the row-based kernels: vector types, the
// nontemporal stream load, the sub-wave reduction, the lanes-per-row rule, and the
// 4-byte-per-lane row dot product of the fallback kernels.
//
// The reference's per-row arithmetic (cusp_warp_reduce.cuh:26-57, LightSpMV.cuh:147-170):
// lane l of a T-lane vector reads one 4-byte Aj and one Ax element per step, stride T, and
// the T partial sums are folded by a shuffle tree.  On gfx950 a 4-byte-per-lane stream
// reaches roughly half the HBM rate of a 16-byte-per-lane one, so the main kernels
// (xwindow.hpp) give a lane 4 consecutive nonzeros per step — one global_load_dwordx4 of
// Aj, one or two of Ax — starting at the row start rounded DOWN to a multiple of 4 so that
// every load is 16-byte aligned, with elements outside [start, end) masked.  (The
// reference has the same idea for one case only: the aligned sweep for T == 32 and rows
// longer than 32, cusp_warp_reduce.cuh:33-44.)  Summation order: per lane ascending, then
// a shuffle-down tree over T lanes — the shape of SURVEY Appendix A.1 with T up to 64.
#pragma once

#include "common.hpp"

namespace mi355 {

typedef int int4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef double double4v __attribute__((ext_vector_type(4)));

template <typename val_t> struct Vec4;
template <> struct Vec4<float> { using type = float4v; };
template <> struct Vec4<double> { using type = double4v; };
template <> struct Vec4<int32_t> { using type = int4v; };   // (integer values: the generalized merge kind)

// Aj / Ax are read exactly once per SpMV: stream them past the caches (nontemporal) so
// that the lines of x, which ARE re-used, stay resident.  Measured on the two-stream
// read pattern of the kernels: 6.9 TB/s nontemporal vs 6.15 TB/s plain.
template <typename V>
__device__ __forceinline__ V stream_load(const V* p) {
#ifdef MI355_STREAM_PLAIN   // (A/B builds only)
    return *p;
#else
    return __builtin_nontemporal_load(p);
#endif
}

// 4-byte-per-lane partial sum of lane `lane` (0..T-1) of the vector that owns [start, end):
// the reference's form, used only when Aj/Ax/x are not 16-byte aligned.
template <int T, typename off_t, typename val_t>
__device__ __forceinline__ val_t row_partial(off_t start, off_t end, int lane, const int32_t* __restrict__ Aj,
                                             const val_t* __restrict__ Ax, const val_t* __restrict__ x) {
    val_t sum = val_t(0);
    for (off_t j = start + lane; j < end; j += T) sum += Ax[j] * x[Aj[j]];
    return sum;
}

// Fold the T lane partials of every vector in the wave; lane 0 of each vector ends up
// with the row sum.  All 64 lanes must execute this.
template <int T, typename val_t>
__device__ __forceinline__ val_t vector_reduce(val_t v) {
#pragma unroll
    for (int o = T / 2; o >= 1; o >>= 1) {
        v += __shfl_down(v, o, T);
    }
    return v;
}

// T (lanes per row) from the mean row length.  With 4 nonzeros per lane per step a T-lane
// vector covers 4T nonzeros per step; pick the smallest T whose step covers the mean row,
// so that a typical row is one load per lane and a wave holds 64/T rows in flight.
// (Reference rule, 1 element per lane and T <= 32: cusp_warp_reduce.cuh:100-127;
// LightSpMV.cuh:354-370.)
inline int pick_lanes_per_row(int64_t nnz, int64_t n_rows, int elems) {
    const int64_t mean = n_rows > 0 ? (nnz + n_rows - 1) / n_rows : 0;
    int t = 2;
    while (t < 64 && int64_t(t) * elems < mean) t <<= 1;
    return t;
}

}  // namespace mi355
