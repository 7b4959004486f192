// light_rows.hip — kind LIGHT: dynamic row distribution (LightSpMV rebuilt for gfx950).
//
// Replaces include/spmv/LightSpMV.cuh:111-263 (csrDynamicVector / csrDynamicWarp)
// and its per-call setup (:266-316).  The idea kept from the reference: a
// persistent grid whose vectors fetch rows from a global atomic counter, so that
// rows of very different lengths balance across the chip.
//
// What changes for MI355X:
//  * one dequeue per WAVE, not per vector: lane 0 adds `rows_per_chunk` to the
//    counter and the value is broadcast with a scalar readfirstlane; all 64 lanes
//    stay in the loop and out-of-range vectors are predicated (the reference's
//    warp-level kernel lets finished vectors leave a loop whose siblings still
//    shuffle, LightSpMV.cuh:212, :246, :260);
//  * one returning device-scope atomic on a single word saturates near 88
//    dequeues/us on this chip, and 1 row per atomic (the reference at T = 8)
//    would be ~10^5 us for 8.3 M rows, so a dequeue hands out a chunk of rows
//    sized so the whole SpMV needs ~8 dequeues per resident wave, and the counter
//    is SHARDED: 8 counters, each on its own 128-byte line, each covering one
//    contiguous eighth of the rows.  A wave starts on the shard of its XCD
//    (blockIdx % 8 shares an L2) and walks the other shards when its own runs
//    dry, so placement only affects speed, never results;
//  * the counters live in the plan's scratch and are zeroed by a memset node on
//    the stream each call (the reference mallocs, memsets and frees them per
//    call, LightSpMV.cuh:274-276, :314);
//  * x is read with plain global loads (no texture path on CDNA; the
//    reference's texture fetch, LightSpMV.cuh:59-88, has no counterpart);
//  * the dot product is row_dot.hpp's 16-byte-per-lane form.
//
// Exit condition: every wave leaves after visiting all 8 shards, each visit ends
// on the first dequeue at or past the shard's end — reached by every wave
// whatever the interleaving, so the grid always drains.

#include "common.hpp"
#include "row_dot.hpp"

namespace mi355 {

constexpr int kCounterStride = 16;  // unsigned long long per 128-byte line

__device__ __forceinline__ unsigned long long wave_broadcast_u64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v));
    const unsigned hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}

template <int T, int ELEMS, typename off_t, typename val_t>
__global__ __launch_bounds__(kBlock) void light_rows_kernel(
    int32_t n_rows, off_t nnz, const off_t* __restrict__ Ap, const int32_t* __restrict__ Aj,
    const val_t* __restrict__ Ax, const val_t* __restrict__ x, val_t* __restrict__ y,
    unsigned long long* __restrict__ counters, int32_t rows_per_chunk) {
    constexpr int ROWS_PER_STEP = kWave / T;
    const int lane64 = threadIdx.x & (kWave - 1);
    const int lane = threadIdx.x & (T - 1);
    const int vec_in_wave = lane64 / T;
    const int home = blockIdx.x % kXcds;

    for (int visit = 0; visit < kXcds; ++visit) {
        const int shard = (home + visit) % kXcds;
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
                val_t sum = row_partial<T, ELEMS, off_t, val_t>(start, end, nnz, lane, Aj, Ax, x);
                sum = vector_reduce<T, val_t>(sum);
                if (live && lane == 0) y[row] = sum;
            }
        }
    }
}

void shape_light(Plan& p) {
    p.lanes_per_row = pick_lanes_per_row(p.nnz, p.n_rows, p.elems_per_lane);
    const int rows_per_step = kWave / p.lanes_per_row;
    // persistent grid: up to 8 workgroups (32 waves) per CU, fewer for small inputs
    int64_t blocks = (int64_t(p.n_rows) + int64_t(rows_per_step) * 4 * 4 - 1) / (int64_t(rows_per_step) * 4 * 4);
    if (blocks > int64_t(kCus) * 8) blocks = int64_t(kCus) * 8;
    if (blocks < 1) blocks = 1;
    p.grid_blocks = blocks;
    const int64_t waves = blocks * (kBlock / kWave);
    int64_t chunk = int64_t(p.n_rows) / (waves * 8);
    chunk = (chunk + rows_per_step - 1) / rows_per_step * rows_per_step;
    if (chunk < rows_per_step) chunk = rows_per_step;
    if (chunk > 4096) chunk = 4096;
    p.rows_per_chunk = chunk;
    p.n_tiles = (int64_t(p.n_rows) + chunk - 1) / chunk;
    p.n_kernels = 1;
    snprintf(p.main_kernel, sizeof(p.main_kernel), "light_rows_kernel");
}

template <int ELEMS, typename off_t, typename val_t>
static int launch_light_t(const Plan& p, const off_t* Ap, const val_t* Ax, const val_t* x, val_t* y,
                          hipStream_t s) {
    const dim3 grid((unsigned)p.grid_blocks), block(kBlock);
    const off_t nnz = (off_t)p.nnz;
    const int32_t chunk = (int32_t)p.rows_per_chunk;
#define MI355_LIGHT_CASE(TT)                                                                         \
    case TT:                                                                                         \
        hipLaunchKernelGGL((light_rows_kernel<TT, ELEMS, off_t, val_t>), grid, block, 0, s, p.n_rows, \
                           nnz, Ap, p.Aj, Ax, x, y, p.counters, chunk);                              \
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
    MI355_HIP_TRY(hipMemsetAsync(p.counters, 0, sizeof(unsigned long long) * kCounterStride * kXcds, s));
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.Aj) | reinterpret_cast<uintptr_t>(Ax)) & 15u) == 0;
    if (p.elems_per_lane == 4 && aligned) return launch_light_t<4, off_t, val_t>(p, Ap, Ax, x, y, s);
    return launch_light_t<1, off_t, val_t>(p, Ap, Ax, x, y, s);
}

template int launch_light<int32_t, float>(const Plan&, const int32_t*, const float*, const float*, float*, hipStream_t);
template int launch_light<int32_t, double>(const Plan&, const int32_t*, const double*, const double*, double*, hipStream_t);
template int launch_light<int64_t, float>(const Plan&, const int64_t*, const float*, const float*, float*, hipStream_t);
template int launch_light<int64_t, double>(const Plan&, const int64_t*, const double*, const double*, double*, hipStream_t);

}  // namespace mi355
