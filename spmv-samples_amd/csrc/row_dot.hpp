// row_dot.hpp — the per-row dot product shared by the CSR-vector and the
// dynamic-row kernels: T lanes (a sub-wave "vector", T in 2..64) own one row.
//
// What the reference does here (cusp_warp_reduce.cuh:26-57, LightSpMV.cuh:147-170):
// lane l reads one 4-byte Aj and one Ax element per step, stride T, and the T
// partial sums are folded by a 32-lane shuffle tree.  On gfx950 a 4-byte-per-lane
// stream reaches roughly half the HBM rate of a 16-byte-per-lane one, so here a
// lane owns ELEMS = 4 consecutive nonzeros per step (one global_load_dwordx4 of
// Aj, one or two of Ax), the sweep starts at the row start rounded DOWN to a
// multiple of 4 so every load is 16-byte aligned, and elements outside
// [start, end) are masked.  (The reference has the same idea for one case only:
// the aligned sweep for T == 32 and rows longer than 32, cusp_warp_reduce.cuh:33-44.)
// Summation order: per lane ascending, then a shuffle-down tree over T lanes —
// the shape of SURVEY Appendix A.1 with T up to 64.
#pragma once

#include "common.hpp"

namespace mi355 {

typedef int int4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef double double4v __attribute__((ext_vector_type(4)));

template <typename val_t> struct Vec4;
template <> struct Vec4<float> { using type = float4v; };
template <> struct Vec4<double> { using type = double4v; };

// Aj / Ax are read exactly once per SpMV: stream them past the caches
// (nontemporal) so that the lines of x, which ARE re-used, stay resident.
template <typename V>
__device__ __forceinline__ V stream_load(const V* p) {
    return __builtin_nontemporal_load(p);
}

// Partial sum of lane `lane` (0..T-1) of the vector that owns [start, end).
template <int T, int ELEMS, typename off_t, typename val_t>
__device__ __forceinline__ val_t row_partial(off_t start, off_t end, off_t nnz, int lane,
                                             const int32_t* __restrict__ Aj,
                                             const val_t* __restrict__ Ax,
                                             const val_t* __restrict__ x) {
    val_t sum = val_t(0);
    if constexpr (ELEMS == 4) {
        using v4 = typename Vec4<val_t>::type;
        off_t j = (start & ~off_t(3)) + off_t(lane) * 4;
        for (; j < end; j += off_t(T) * 4) {
            int4v c;
            v4 a;
            if (j + 4 <= nnz) {
                c = stream_load(reinterpret_cast<const int4v*>(Aj + j));
                a = stream_load(reinterpret_cast<const v4*>(Ax + j));
            } else {
                // last, partial group of the arrays: never read past nnz
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = (j + e) < nnz;
                    c[e] = in ? Aj[j + e] : 0;
                    a[e] = in ? Ax[j + e] : val_t(0);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const off_t k = j + e;
                const bool valid = (k >= start) && (k < end);
                // a masked element still holds an in-range column of a neighbouring
                // row (or 0), so the gather address is always legal.
                const val_t xv = x[c[e]];
                sum = valid ? (sum + a[e] * xv) : sum;
            }
        }
    } else {
        for (off_t j = start + lane; j < end; j += T) {
            sum += Ax[j] * x[Aj[j]];
        }
    }
    return sum;
}

// Fold the T lane partials of every vector in the wave; lane 0 of each vector
// ends up with the row sum.  All 64 lanes must execute this.
template <int T, typename val_t>
__device__ __forceinline__ val_t vector_reduce(val_t v) {
#pragma unroll
    for (int o = T / 2; o >= 1; o >>= 1) {
        v += __shfl_down(v, o, T);
    }
    return v;
}

// T (lanes per row) from the mean row length.  With 4 nonzeros per lane per
// step a T-lane vector covers 4T nonzeros per step; pick the smallest T whose
// step covers the mean row, so that a typical row is one load per lane and a
// wave holds 64/T rows in flight.  (Reference rule, 1 element per lane and
// T <= 32: cusp_warp_reduce.cuh:100-127; LightSpMV.cuh:354-370.)
inline int pick_lanes_per_row(int64_t nnz, int64_t n_rows, int elems) {
    const int64_t mean = n_rows > 0 ? (nnz + n_rows - 1) / n_rows : 0;
    int t = 2;
    while (t < 64 && int64_t(t) * elems < mean) t <<= 1;
    return t;
}

}  // namespace mi355
