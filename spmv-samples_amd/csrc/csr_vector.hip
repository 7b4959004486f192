// csr_vector.hip — kind VECTOR: CSR-vector SpMV with per-row sub-wave reduction.
//
// Replaces the reference's CUSP CSR-vector family on the hot path
// (include/spmv/cusp/cusp_warp_reduce.cuh:11-59 kernel, :93-133 width selection,
// include/spmv/cusp/utils.cuh:38-47 shuffle tree).  Written for gfx950: 64-lane
// waves, 256-thread workgroups, 16-byte-per-lane loads of Aj/Ax (row_dot.hpp),
// workgroup -> row-block mapping that keeps neighbouring row blocks on one XCD so
// that the window of x they share is served by that XCD's L2.
//
// One vector of T lanes per row, 256/T rows per workgroup, grid = ceil(rows / that):
// the reference's launch shape (cusp_warp_reduce.cuh:70-87) scaled to wave64.

#include "common.hpp"
#include "row_dot.hpp"

namespace mi355 {

template <int T, int ELEMS, typename off_t, typename val_t>
__global__ __launch_bounds__(kBlock) void csr_vector_kernel(
    int32_t n_rows, off_t nnz, const off_t* __restrict__ Ap, const int32_t* __restrict__ Aj,
    const val_t* __restrict__ Ax, const val_t* __restrict__ x, val_t* __restrict__ y) {
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
    val_t sum = row_partial<T, ELEMS, off_t, val_t>(start, end, nnz, lane, Aj, Ax, x);
    sum = vector_reduce<T, val_t>(sum);
    if (live && lane == 0) y[row] = sum;
}

void shape_vector(Plan& p) {
    p.lanes_per_row = pick_lanes_per_row(p.nnz, p.n_rows, p.elems_per_lane);
    const int rows_per_block = kBlock / p.lanes_per_row;
    p.grid_blocks = (int64_t(p.n_rows) + rows_per_block - 1) / rows_per_block;
    if (p.grid_blocks < 1) p.grid_blocks = 1;
    p.n_kernels = 1;
    snprintf(p.main_kernel, sizeof(p.main_kernel), "csr_vector_kernel");
}

template <int ELEMS, typename off_t, typename val_t>
static int launch_vector_t(const Plan& p, const off_t* Ap, const val_t* Ax, const val_t* x, val_t* y,
                           hipStream_t s) {
    const dim3 grid((unsigned)p.grid_blocks), block(kBlock);
    const off_t nnz = (off_t)p.nnz;
#define MI355_VEC_CASE(TT)                                                                          \
    case TT:                                                                                        \
        hipLaunchKernelGGL((csr_vector_kernel<TT, ELEMS, off_t, val_t>), grid, block, 0, s, p.n_rows, \
                           nnz, Ap, p.Aj, Ax, x, y);                                                \
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
    // 16-byte loads need 16-byte-aligned Aj / Ax (hipMalloc gives 256); a caller
    // that passes an offset view gets the 4-byte-per-lane form instead.
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.Aj) | reinterpret_cast<uintptr_t>(Ax)) & 15u) == 0;
    if (p.elems_per_lane == 4 && aligned) return launch_vector_t<4, off_t, val_t>(p, Ap, Ax, x, y, s);
    return launch_vector_t<1, off_t, val_t>(p, Ap, Ax, x, y, s);
}

template int launch_vector<int32_t, float>(const Plan&, const int32_t*, const float*, const float*, float*, hipStream_t);
template int launch_vector<int32_t, double>(const Plan&, const int32_t*, const double*, const double*, double*, hipStream_t);
template int launch_vector<int64_t, float>(const Plan&, const int64_t*, const float*, const float*, float*, hipStream_t);
template int launch_vector<int64_t, double>(const Plan&, const int64_t*, const double*, const double*, double*, hipStream_t);

}  // namespace mi355
