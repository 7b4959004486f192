/* mi355_spmv.h — C ABI of the MI355X-native CSR SpMV engine (libmi355spmv.so).
 *
 * Drop-in boundary.  The reference (peakcrosser7/spmv-samples) has no FFI of its
 * own: its operator boundary is the header-only C++ template
 *
 *   template <index_t, offset_t, mat_value_t, vec_x_value_t, vec_y_value_t>
 *   void SpMV(const std::string& kind_str, index_t n_rows, index_t n_cols,
 *             offset_t nnz, const offset_t* Ap, const index_t* Aj,
 *             const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y);
 *                                          (reference include/spmv.h:29-48)
 *
 * and every kind behind it is `SpMV_<kind>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y)`
 * with that same 8-argument device-pointer signature, registered by one
 * `X("label", SpMV_<kind>)` row of the SPMV_KINDS X-macro (include/spmv.h:18-27,
 * README.md:28-45).  The entry points below are what such a kind binds: plain
 * pointers and sizes, no C++ or torch types.  spmv-samples_amd/host/spmv/mi355.hpp
 * holds the template kinds that call them; INTEGRATION.md shows the two lines a
 * maintainer of the reference adds.
 *
 * Conventions carried over from the reference:
 *   - Ap, Aj, Ax, x, y are DEVICE pointers owned by the caller (main.cu:48-74).
 *   - index_t is 32-bit (main.cu:15); offset_t is 32- or 64-bit; values are
 *     float or double, the same type for A, x and y (main.cu:17).
 *   - y is fully overwritten (beta = 0), empty rows give 0 (cpu_navie.hpp:10-15).
 *   - The reference aborts on a device error (common.cuh:13-23).  The C ABI never
 *     aborts: it returns a status; the C++ kinds turn non-zero into abort().
 *
 * Symbol suffix:  _<i32|i64>_<f32|f64>  =  offset_t width, value type.
 */
#ifndef MI355_SPMV_H
#define MI355_SPMV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_SPMV_VERSION 100 /* 0.1.0 */

/* status codes */
enum {
    MI355_SPMV_OK = 0,
    MI355_SPMV_EINVAL = 1,  /* bad size / null pointer / unknown enum          */
    MI355_SPMV_ENOTSUP = 2, /* type combination or kind not built              */
    MI355_SPMV_EHIP = 3,    /* a HIP runtime call failed (see _last_error)     */
    MI355_SPMV_ENOMEM = 4,  /* scratch allocation failed                       */
    MI355_SPMV_ENODEV = 5   /* no gfx950 device visible                        */
};

/* kinds — the three hot variants of the reference, rebuilt for wave64:
 *   VECTOR  CSR-vector, per-row sub-wave reduction
 *           (replaces SpMV_cusp_warp_reduce, include/spmv/cusp/cusp_warp_reduce.cuh:138-147)
 *   MERGE   merge-path load-balanced SpMV: search -> tile -> fix-up
 *           (replaces SpMV_merge_based / SpMV_merge_based_generalized,
 *            include/spmv/merge_based/merge_based.cuh:22-56,
 *            include/spmv/merge_genl/merge_genl.cuh:41-79)
 *   LIGHT   dynamic row distribution from global atomic row counters
 *           (replaces SpMV_light_vector / SpMV_light_warp,
 *            include/spmv/LightSpMV.cuh:379-416)                              */
enum { MI355_KIND_VECTOR = 0, MI355_KIND_MERGE = 1, MI355_KIND_LIGHT = 2, MI355_KIND_COUNT = 3 };
enum { MI355_OFF_I32 = 0, MI355_OFF_I64 = 1 };
enum { MI355_VAL_F32 = 0, MI355_VAL_F64 = 1 };

/* semirings of the generalized merge kind.  The reference's
 * SpMV_merge_based_generalized takes a functor_t with initialize / combine / reduce
 * (include/spmv/merge_genl/merge_genl.cuh:19-38; CPU twin include/spmv/cpu_navie.hpp:20-34)
 * and ships (+, *); a C ABI enumerates them instead:
 *   PLUS_TIMES  y[r] = sum_k  Ax[k] * x[Aj[k]]           (identity 0)      — every other entry point
 *   MIN_PLUS    y[r] = min_k (Ax[k] + x[Aj[k]])          (identity +inf)   — shortest-path relaxation
 *   MAX_TIMES   y[r] = max_k (Ax[k] * x[Aj[k]])          (identity -inf)   — or-and on {0,1}, widest path */
enum { MI355_SEMIRING_PLUS_TIMES = 0, MI355_SEMIRING_MIN_PLUS = 1, MI355_SEMIRING_MAX_TIMES = 2,
       MI355_SEMIRING_COUNT = 3 };

/* plan flags */
enum {
    MI355_PLAN_DEFAULT = 0,
    /* Keep results that depend only on Ap (merge-path tile coordinates) across
     * executes instead of recomputing them every call.  Valid while the caller
     * leaves Ap unchanged.  Off by default: a default plan holds scratch memory
     * and launch shapes only, and every execute runs every kernel of the kind. */
    MI355_PLAN_REUSE_STRUCTURE = 1
};

/* ---- one-shot entry points -------------------------------------------------
 * Same 8 arguments as the reference's SpMV_<kind> (include/spmv.h:29-34) plus a
 * hipStream_t (NULL = the null stream, which is what the reference launches on).
 * Allocates its scratch, runs, synchronises the stream, frees the scratch — the
 * per-call life cycle of the reference kinds (LightSpMV.cuh:274-276, :314;
 * merge_based.cuh:34-56).  Returns a status code.                              */
#define MI355_SPMV_DECLARE(KIND, SUF, OFF, VAL)                                          \
    int mi355_spmv_##KIND##_##SUF(int32_t n_rows, int32_t n_cols, OFF nnz, const OFF* Ap, \
                                  const int32_t* Aj, const VAL* Ax, const VAL* x, VAL* y, \
                                  void* stream);
#define MI355_SPMV_DECLARE_KIND(KIND)                 \
    MI355_SPMV_DECLARE(KIND, i32_f32, int32_t, float)  \
    MI355_SPMV_DECLARE(KIND, i32_f64, int32_t, double) \
    MI355_SPMV_DECLARE(KIND, i64_f32, int64_t, float)  \
    MI355_SPMV_DECLARE(KIND, i64_f64, int64_t, double)

MI355_SPMV_DECLARE_KIND(vector) /* replaces SpMV_cusp_warp_reduce  cusp_warp_reduce.cuh:138 */
MI355_SPMV_DECLARE_KIND(merge)  /* replaces SpMV_merge_based[_generalized] merge_based.cuh:22, merge_genl.cuh:41 */
MI355_SPMV_DECLARE_KIND(light)  /* replaces SpMV_light_vector/_warp  LightSpMV.cuh:379, :400 */

/* generalized merge-path SpMV: the reference's SpMV_merge_based_generalized
 * (include/spmv/merge_genl/merge_genl.cuh:41-79) with the semiring as an argument.  */
#define MI355_SPMV_DECLARE_GENL(SUF, OFF, VAL)                                                  \
    int mi355_spmv_merge_genl_##SUF(int semiring, int32_t n_rows, int32_t n_cols, OFF nnz,      \
                                    const OFF* Ap, const int32_t* Aj, const VAL* Ax,            \
                                    const VAL* x, VAL* y, void* stream);
MI355_SPMV_DECLARE_GENL(i32_f32, int32_t, float)
MI355_SPMV_DECLARE_GENL(i32_f64, int32_t, double)
MI355_SPMV_DECLARE_GENL(i64_f32, int64_t, float)
MI355_SPMV_DECLARE_GENL(i64_f64, int64_t, double)

/* ---- plan entry points -----------------------------------------------------
 * The reference re-creates scratch on every call (quirks 7-9 of SURVEY.md §2c);
 * a plan keeps scratch and launch shapes across the timing loop of main.cu:102-113.
 * create: sizes + structure pointers (Ap/Aj are retained, not copied, and must
 *         outlive the plan; their CONTENTS are read here — a structure probe and,
 *         for VECTOR / LIGHT, the chunk boundaries — so they must already be valid
 *         and must not change while the plan lives).  Synchronises the device.
 * execute: asynchronous on `stream`, no host sync, no allocation, kernels only
 *         (hipGraph-capturable: tests/test_gpu_parity.py).  A plan serves ONE
 *         stream at a time (its scratch — merge coordinates and carries, the LIGHT
 *         counters — belongs to the execute in flight); concurrent executes need
 *         one plan each.
 * destroy: frees scratch (hipFree: waits for the device).                      */
typedef struct mi355_spmv_plan mi355_spmv_plan;

int mi355_spmv_plan_create(mi355_spmv_plan** plan, int kind, int off_type, int val_type,
                           int32_t n_rows, int32_t n_cols, int64_t nnz, const void* Ap,
                           const int32_t* Aj, int flags);
int mi355_spmv_plan_execute(mi355_spmv_plan* plan, const void* Ax, const void* x, void* y,
                            void* stream);
int mi355_spmv_plan_destroy(mi355_spmv_plan* plan);
/* y = alpha * A x + beta * y for the following executes (any kind, (+, *) semiring; default 1, 0:
 * y overwritten).  SURVEY §8(f)-4: the reference's cuSPARSE kind passes alpha = 1, beta = 0
 * (include/spmv/cusparse.cuh:42-43) and its vendored CUB carries the same two scalars, disabled
 * (merge_based/agent_spmv_orig.cuh:425-433, dispatch_spmv_orig.cuh:802).  y is read only when beta != 0. */
int mi355_spmv_plan_set_alpha_beta(mi355_spmv_plan* plan, double alpha, double beta);
/* MERGE plans only (ENOTSUP otherwise): choose the semiring of the following executes.  */
int mi355_spmv_plan_set_semiring(mi355_spmv_plan* plan, int semiring);
/* Block the host until `stream` has drained (hipStreamSynchronize), so that a
 * host-only C++ caller can bracket Timer::kernel_stop() the way the reference's
 * kinds do with cudaDeviceSynchronize() (cusp_warp_reduce.cuh:131) without
 * including HIP headers.                                                        */
int mi355_spmv_stream_synchronize(void* stream);

/* Launch shape chosen by the plan (for reports and tests).                     */
typedef struct mi355_spmv_plan_info {
    int32_t kind, off_type, val_type;
    int32_t lanes_per_row;    /* T: sub-wave width (VECTOR, LIGHT); 0 for MERGE    */
    int32_t elems_per_lane;   /* nonzeros one lane loads per step (4 = 16-B loads) */
    int32_t block_threads;
    int64_t grid_blocks;      /* blocks of the main kernel                         */
    int64_t tile_items;       /* MERGE: merge items per tile                       */
    int64_t n_tiles;          /* MERGE: tiles; LIGHT: row chunks                   */
    int64_t rows_per_chunk;   /* LIGHT: rows per dequeue                           */
    int64_t scratch_bytes;    /* device scratch held by the plan                   */
    int32_t n_kernels;        /* kernels launched per execute                      */
    int32_t window_elems;     /* elements of x staged through LDS per workgroup (0 = none) */
    int32_t window_segments;  /* 1 = one window; 2..4 = that many column bands staged side by side */
    char main_kernel[64];     /* substring of the dominant kernel's symbol name    */
    int32_t balanced_chunks;  /* VECTOR, LIGHT: 1 = row chunks cut by weight (nonzeros + mean row length per
                                 row) because equal-row chunks were uneven, 0 = equal-row chunks           */
    int32_t rows_cap;         /* VECTOR, LIGHT: most rows a chunk can hold                                 */
    int64_t n_chunks;         /* VECTOR, LIGHT: row chunks                                                 */
} mi355_spmv_plan_info;
int mi355_spmv_plan_get_info(const mi355_spmv_plan* plan, mi355_spmv_plan_info* info);

/* MERGE only, for parity tests: copy the tile start coordinates the search
 * kernel produced by the last execute to HOST arrays of n_tiles+1 entries
 * (synchronises).  Integers: compared bit-exactly with the oracle's restatement
 * of thread_search.cuh:15-49.                                                  */
int mi355_spmv_plan_merge_coords(mi355_spmv_plan* plan, int64_t* tile_row, int64_t* tile_nnz);

/* ---- misc ------------------------------------------------------------------ */
int mi355_spmv_version(void);
const char* mi355_spmv_status_string(int status);
/* Message of the last failing call on this thread ("" if none).               */
const char* mi355_spmv_last_error(void);
/* Number of visible gfx950 devices (0 if none / no HIP runtime).              */
int mi355_spmv_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355_SPMV_H */
