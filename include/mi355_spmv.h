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

#define MI355_SPMV_VERSION 310 /* 0.3.1 */

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
enum { MI355_KIND_VECTOR = 0, MI355_KIND_MERGE = 1, MI355_KIND_LIGHT = 2, MI355_KIND_COUNT = 3,
       /* plan_create / plan_acquire / the mi355_spmv_auto_* one-shots / dist_create_local only: the library picks.  The
        * reference leaves the choice of kind to the command line (main.cu:26-30); a caller that has no opinion gets
        * MERGE when the row lengths are skewed (the heaviest run of rows a workgroup would take holds more than twice
        * the mean: power-law matrices, where merge-path is ahead of the row kinds on every measured config) and for
        * integer values, VECTOR otherwise.  A heuristic on structure, not a measurement (bench.py measures);
        * plan_get_info().kind reports what was picked.                                                            */
       MI355_KIND_AUTO = 100 };
enum { MI355_OFF_I32 = 0, MI355_OFF_I64 = 1 };
enum { MI355_VAL_F32 = 0, MI355_VAL_F64 = 1,
       /* 32-bit integers: the MERGE kind only (every semiring; two's-complement wrap-around, exact whatever the
        * reduction order).  The reference's generalized merge kind is a template over the value types and its
        * functor (merge_genl.cuh:19-38, :134-150); this is the integer / boolean instance of it.  VECTOR / LIGHT
        * plans and alpha / beta return MI355_SPMV_ENOTSUP for it.                                              */
       MI355_VAL_I32 = 2 };

/* semirings of the generalized merge kind.  The reference's
 * SpMV_merge_based_generalized takes a functor_t with initialize / combine / reduce
 * (include/spmv/merge_genl/merge_genl.cuh:19-38; CPU twin include/spmv/cpu_navie.hpp:20-34)
 * and ships (+, *); a C ABI cannot take a C++ functor: the tuned kernels enumerate these (any OTHER functor, and any
 * mix of the five types, goes through mi355_spmv_functor_* below — its text compiled at run time):
 *   PLUS_TIMES  y[r] = sum_k  Ax[k] * x[Aj[k]]           (identity 0)      — every other entry point
 *   MIN_PLUS    y[r] = min_k (Ax[k] + x[Aj[k]])          (identity +inf; INT32_MAX for integers) — shortest-path relaxation
 *   MAX_TIMES   y[r] = max_k (Ax[k] * x[Aj[k]])          (identity -inf)   — widest / most reliable path
 *   MAX_PLUS    y[r] = max_k (Ax[k] + x[Aj[k]])          (identity -inf)   — longest path, Viterbi
 *   OR_AND      y[r] = OR_k (Ax[k] != 0 AND x[Aj[k]] != 0) as 1.0 / 0.0  (identity 0)
 *                                                                         — boolean SpMV: one BFS / reachability step */
enum { MI355_SEMIRING_PLUS_TIMES = 0, MI355_SEMIRING_MIN_PLUS = 1, MI355_SEMIRING_MAX_TIMES = 2,
       MI355_SEMIRING_MAX_PLUS = 3, MI355_SEMIRING_OR_AND = 4, MI355_SEMIRING_COUNT = 5 };

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
MI355_SPMV_DECLARE_KIND(auto)   /* MI355_KIND_AUTO: one of the three, picked from the structure */

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
MI355_SPMV_DECLARE_GENL(i32_i32, int32_t, int32_t)   /* integer values: identities 0 / INT32_MAX / INT32_MIN, OR_AND as 1 / 0 */
MI355_SPMV_DECLARE_GENL(i64_i32, int64_t, int32_t)

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
/* The reference's operator has separate matrix / x / y value types (include/spmv.h:29-34; its generalized merge
 * kind computes in the y type, merge_genl.cuh:29-31).  Built here: all three equal (every kind), and an fp32
 * MATRIX under fp64 x and y for the MERGE kind — values are widened as they meet x, products and sums are fp64
 * (the mixed-precision case that halves the matrix stream).  Other combinations return MI355_SPMV_ENOTSUP.
 * execute then takes Ax as float*, x / y as double*.                                                         */
int mi355_spmv_plan_create_typed(mi355_spmv_plan** plan, int kind, int off_type, int mat_type, int x_type,
                                 int y_type, int32_t n_rows, int32_t n_cols, int64_t nnz, const void* Ap,
                                 const int32_t* Aj, int flags);
int mi355_spmv_merge_f32mat_f64vec_i32(int32_t n_rows, int32_t n_cols, int32_t nnz, const int32_t* Ap,
                                       const int32_t* Aj, const float* Ax, const double* x, double* y, void* stream);
int mi355_spmv_merge_f32mat_f64vec_i64(int32_t n_rows, int32_t n_cols, int64_t nnz, const int64_t* Ap,
                                       const int32_t* Aj, const float* Ax, const double* x, double* y, void* stream);
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
    char knobs[160];          /* the MI355_* tuning variables that were set when the plan was created, "NAME=value ..." */
} mi355_spmv_plan_info;
int mi355_spmv_plan_get_info(const mi355_spmv_plan* plan, mi355_spmv_plan_info* info);

/* ---- row-block plans --------------------------------------------------------
 * rows are independent, so a matrix is cut into contiguous row blocks (one or more per GPU)
 * and each block is an ordinary CSR SpMV.  For the concatenated y to equal the one-GPU y BIT FOR BIT with
 * the row-local kinds (VECTOR, LIGHT), a block must sum every row exactly as the whole matrix's plan does:
 * same lanes per row, same workgroup size, same chunks (hence the same per-chunk vector width and the same
 * window of x), same long-row and giant-row treatment, same 16-byte phase of every row start.  So:
 *   mi355_spmv_plan_get_shape   the launch-shape decisions of a plan, as plain data (can be sent to other
 *                               processes);
 *   mi355_spmv_plan_partition   nnz-balanced cut points that fall on the plan's chunk boundaries;
 *   mi355_spmv_plan_create_block  a plan for rows [row_begin, row_begin + n_rows) of that matrix which inherits
 *                               the shape.  Its arrays are a 16-byte-aligned VIEW of the whole CSR:
 *                               Aj / Ax start at element (Ap_whole[row_begin] & ~3) and Ap[i] =
 *                               Ap_whole[row_begin + i] - (Ap_whole[row_begin] & ~3), so Ap[0] is 0..3 (the
 *                               "phase") and nnz is the END offset Ap[n_rows].  No copy of Aj / Ax is needed on
 *                               the device that holds the whole matrix.  Unless the block ends where the whole
 *                               matrix ends, Aj / Ax must be READABLE up to the next multiple of 4 elements
 *                               past nnz (a view of the whole arrays is; a copy is padded): the tail of the
 *                               block's last row is then read in whole 16-byte groups, as the whole plan does.
 * MERGE blocks take the same arrays but are shaped on their own (tile boundaries move with the cut anyway;
 * results stay inside the parity bound); shape may then be NULL.                              */
typedef struct mi355_spmv_plan_shape {
    int32_t struct_bytes;             /* sizeof(mi355_spmv_plan_shape) of the library that filled it */
    int32_t kind, off_type, val_type;
    int32_t n_rows, n_cols;           /* the whole matrix */
    int64_t nnz;
    int32_t lanes_per_row, elems_per_lane, block_threads;
    int32_t balanced_chunks, rows_cap, giant_rows_enabled;
    int64_t rows_per_chunk, n_chunks, bal_k, bal_q, giant_len;
    int32_t window_elems, window_bytes, window_from_band, window_segments, probe_ok, long_steps;
    int64_t band_lo, band_hi, seg_lo[4], seg_hi[4];
    int32_t window_sweep;             /* 1 = the band is wider than any window: one group of rows per chunk, the window sweeps it */
    int32_t small_plain;              /* 1 = VECTOR / LIGHT on a small matrix: the plain one-pass kernel (lanes_per_row lanes, 4-byte loads) */
} mi355_spmv_plan_shape;
int mi355_spmv_plan_get_shape(const mi355_spmv_plan* plan, mi355_spmv_plan_shape* shape);
/* parts + 1 entries each: row_cuts[p] = first row of block p, chunk_cuts[p] = its first chunk in the plan's
 * numbering (0 for MERGE), nnz_cuts[p] = Ap[row_cuts[p]].  Cuts are multiples of 4 rows (or n_rows), chosen so
 * that nnz_cuts[p] ~ p * nnz / parts.  Reads Ap on the device (synchronises).                              */
int mi355_spmv_plan_partition(const mi355_spmv_plan* plan, int parts, int64_t* row_cuts, int64_t* chunk_cuts,
                              int64_t* nnz_cuts);
int mi355_spmv_plan_create_block(mi355_spmv_plan** plan, int kind, int off_type, int val_type,
                                 const mi355_spmv_plan_shape* whole, int64_t row_begin, int64_t chunk_begin,
                                 int64_t n_chunks, int64_t nnz_begin_whole,
                                 int32_t n_rows, int32_t n_cols, int64_t nnz_end, const void* Ap,
                                 const int32_t* Aj, int flags);

/* ---- multi-GPU: row blocks, x replicated, allgatherv(y) over RCCL/xGMI -------------------------------
 * The reference is single-device (main.cu:53, common.cuh:8); the north star adds one node of 8 GPUs:
 * contiguous nnz-balanced row blocks, x replicated, the y slices concatenated on every GPU.  RCCL has no
 * allgatherv: block p is broadcast from its owner into its displacement of every GPU's y (grouped
 * ncclBroadcast calls on a dedicated stream).  Each GPU's rows are cut into `sub_blocks` blocks so that the
 * slice of block s travels while block s + 1 is being computed.  RCCL (librccl.so.1) is loaded on first use,
 * and only when more than one GPU takes part: libmi355spmv.so itself links the HIP runtime only.
 *
 * LOCAL mode — one process drives all the GPUs (the reference's single-threaded harness, `--ngpu N`):
 *   create_local   Ap / Aj of the WHOLE matrix on the current device ("home"); blocks are dealt to
 *                  `devices` in order (parts = n_devices * sub_blocks); remote blocks get copies of
 *                  their slice of Ap / Aj, the home device's blocks are views.
 *   scatter_values / replicate_x   refresh the remote copies of Ax / x from home-device arrays
 *   execute(d, Ax, x, y, stream)   Ax / x / y on the home device; Ax or x may be NULL = "unchanged since the
 *                  last scatter / replicate"; non-NULL pointers are scattered / replicated first (the drop-in
 *                  semantics of SpMV(kind, ...), which hands over home-device arrays on every call).  y (n_rows)
 *                  receives the full result on the home device; every other GPU holds it too (device_y).
 * RANK mode — one process per GPU (torch.distributed / MPI style launch):
 *   unique_id      128 bytes made by ONE rank, distributed by the caller's own means
 *   create_rank    this rank's blocks (parts_per_rank consecutive blocks of the global cut list), its slice
 *                  of the structure; `whole` = the shape of the whole matrix's plan when bitwise identity
 *                  with a one-GPU run is wanted (NULL: every block is shaped on its own)
 *   execute(d, Ax, x, y, stream)   Ax = this rank's values (view described above), x = this GPU's copy of
 *                  x, y = this GPU's full-length y.
 * With one GPU (n_devices == 1 / world == 1) no communicator is made and execute is the blocks' plain
 * executes on the caller's stream.
 * STATUS: world > 1 is EXPERIMENTAL — it has run only against an emulation of RCCL (several "GPUs" on one device,
 * tests/cpp/fake_rccl.cpp): schedule, counts, displacements and stream order are tested, RCCL itself and xGMI are not. */
typedef struct mi355_spmv_dist mi355_spmv_dist;
int mi355_spmv_dist_create_local(mi355_spmv_dist** dist, int kind, int off_type, int val_type,
                                 int32_t n_rows, int32_t n_cols, int64_t nnz, const void* Ap,
                                 const int32_t* Aj, int n_devices, const int* devices, int sub_blocks,
                                 int flags);
int mi355_spmv_dist_unique_id(void* id128);
int mi355_spmv_dist_create_rank(mi355_spmv_dist** dist, int kind, int off_type, int val_type,
                                int rank, int world, const void* id128, int parts_per_rank,
                                const int64_t* row_cuts, const int64_t* chunk_cuts, const int64_t* nnz_cuts,
                                const mi355_spmv_plan_shape* whole, int32_t n_cols,
                                int32_t n_rows_local, int64_t nnz_end_local, const void* Ap_local,
                                const int32_t* Aj_local, int flags);
int mi355_spmv_dist_scatter_values(mi355_spmv_dist* dist, const void* Ax, void* stream);
int mi355_spmv_dist_replicate_x(mi355_spmv_dist* dist, const void* x, void* stream);
int mi355_spmv_dist_execute(mi355_spmv_dist* dist, const void* Ax, const void* x, void* y, void* stream);
/* The same with flags, so that a scaling report can tell compute from exchange (north star: "1/2/4/8-GPU GFLOP/s and
 * bandwidth-fraction scaling reported"; SURVEY §8(e): "report compute-only and end-to-end scaling"):
 *   MI355_DIST_EXEC_SKIP_EXCHANGE   the blocks' kernels with the whole stream choreography, no RCCL call: afterwards
 *                                   every GPU holds ITS rows of y only;
 *   MI355_DIST_EXEC_EXCHANGE_ONLY   no kernel: the allgatherv of whatever y holds.
 * Collective like execute: every rank / GPU passes the same flags.                                                   */
enum { MI355_DIST_EXEC_DEFAULT = 0, MI355_DIST_EXEC_SKIP_EXCHANGE = 1, MI355_DIST_EXEC_EXCHANGE_ONLY = 2 };
int mi355_spmv_dist_execute_ex(mi355_spmv_dist* dist, const void* Ax, const void* x, void* y, void* stream,
                               int exec_flags);
/* How the y slices travel.  RCCL has no allgatherv (SURVEY §5 last row names three ways to make one); all three are
 * built behind this one API, per sub-block s, on the communication stream:
 *   BCAST      one group of in-place ncclBroadcast, root = owner of block (root, s): a ring per root;
 *   SENDRECV   one group of ncclSend / ncclRecv, every GPU sends its block to every peer and receives theirs:
 *              point to point, all xGMI links at once;
 *   ALLGATHER  one ncclAllGather: in place in y when the blocks of sub-block s are equal and adjacent (one block per
 *              GPU on a uniform matrix), else through a staging buffer padded to the largest block (one pack and
 *              one unpack kernel around it).
 * AUTO (default, or MI355_DIST_EXCHANGE=auto): when more than one GPU takes part, create times MI355_DIST_TRIALS (5)
 * exchanges of each on scratch buffers, agrees on the maximum over ranks (ncclAllReduce) and keeps the fastest;
 * dist_get_info reports the choice and the trial times.  set_exchange changes it later (collective: every rank the
 * same value, no execute in flight).                                                                               */
enum { MI355_DIST_EXCHANGE_AUTO = 0, MI355_DIST_EXCHANGE_BCAST = 1, MI355_DIST_EXCHANGE_SENDRECV = 2,
       MI355_DIST_EXCHANGE_ALLGATHER = 3, MI355_DIST_EXCHANGE_COUNT = 4 };
int mi355_spmv_dist_set_exchange(mi355_spmv_dist* dist, int exchange);
typedef struct mi355_spmv_dist_info {
    int32_t world, rank, sub_blocks, local_mode;
    int32_t exchange;          /* MI355_DIST_EXCHANGE_* in use (never AUTO; BCAST when world == 1: unused)        */
    int32_t auto_picked;       /* 1 = chosen by the timed trial at create                                        */
    int32_t allgather_in_place;/* 1 = every sub-block's ALLGATHER goes straight into y (no staging)              */
    int32_t reserved0;
    float trial_us[MI355_DIST_EXCHANGE_COUNT];   /* per exchange of all sub-blocks, max over ranks; 0 = not timed */
    int64_t max_block_rows;    /* largest block (the padded count of ALLGATHER)                                  */
    int64_t staging_bytes;     /* per GPU, ALLGATHER through staging                                             */
    char exchange_name[16];
} mi355_spmv_dist_info;
int mi355_spmv_dist_get_info(const mi355_spmv_dist* dist, mi355_spmv_dist_info* info);
/* A dist handle holds COPIES of the structure (a rebased Ap per block; Aj on the other GPUs), so unlike the one-shot
 * plan cache it is wrong for a matrix rewritten in place.  For callers with the reference's semantics — SpMV(kind,
 * ...) reads its arrays on every call — this compares a fingerprint of the caller's Ap (every offset) and Aj (a
 * strided sample of 64 K entries) with the one taken at create: one small kernel and one 16-byte read-back
 * (synchronises `stream`).  LOCAL mode only.  *changed = 1: destroy the handle and create it again.             */
int mi355_spmv_dist_structure_changed(mi355_spmv_dist* dist, const void* Ap, const int32_t* Aj, void* stream,
                                      int* changed);
int mi355_spmv_dist_set_alpha_beta(mi355_spmv_dist* dist, double alpha, double beta);
/* number of blocks, and the global cut rows (parts + 1 entries)                */
int mi355_spmv_dist_parts(const mi355_spmv_dist* dist);
int mi355_spmv_dist_cuts(const mi355_spmv_dist* dist, int64_t* row_cuts);
/* launch shape of this process's block `part` (0 .. devices_of_this_process * sub_blocks - 1)             */
int mi355_spmv_dist_part_info(const mi355_spmv_dist* dist, int part, mi355_spmv_plan_info* info);
/* LOCAL mode: GPU `device_index`'s (position in `devices`) copies of y / x      */
void* mi355_spmv_dist_device_y(mi355_spmv_dist* dist, int device_index);
void* mi355_spmv_dist_device_x(mi355_spmv_dist* dist, int device_index);
int mi355_spmv_dist_destroy(mi355_spmv_dist* dist);

/* Re-read the MI355_* tuning variables from the environment (they are otherwise parsed once per process;
 * plans keep the values they were created under and report the non-default ones in plan_info.knobs). */
int mi355_spmv_knobs_reload(void);

/* The one-shot entry points keep their last few plans, found again by the pointers and sizes of Ap / Aj, the types, the
 * kind and the device (the reference's harness calls a kind 2 000 times in a row on one matrix, main.cu:102-113; plan
 * creation is a third of such a call on the target).  Safe if the arrays were rewritten in place: a kept plan holds
 * launch-shape decisions only (plans with giant rows, whose row list is structure, are never kept).
 * MI355_SPMV_PLAN_CACHE=0 disables it; this call destroys the kept plans and frees their scratch.                  */
int mi355_spmv_cache_release(void);
/* The same for a caller that wants its own timer between the steps (the C++ mirror of the reference boundary,
 * host/spmv/mi355.hpp): acquire = a kept plan for this matrix or a new default plan; release = hand it back after the
 * stream it ran on has been synchronised (executed_ok = 0 after a failed execute: the plan is destroyed).  A plan whose
 * semiring was changed is handed back with it (the next acquirer sets its own); alpha / beta other than 1 / 0 are not kept. */
int mi355_spmv_plan_acquire(mi355_spmv_plan** plan, int kind, int off_type, int val_type, int32_t n_rows,
                            int32_t n_cols, int64_t nnz, const void* Ap, const int32_t* Aj);
int mi355_spmv_plan_release(mi355_spmv_plan* plan, int executed_ok);

/* MERGE only, for parity tests: copy the tile start coordinates the search
 * kernel produced by the last execute to HOST arrays of n_tiles+1 entries
 * (synchronises).  Integers: compared bit-exactly with the oracle's restatement
 * of thread_search.cuh:15-49.                                                  */
int mi355_spmv_plan_merge_coords(mi355_spmv_plan* plan, int64_t* tile_row, int64_t* tile_nnz);

/* ---- a generalized SpMV whose functor is the CALLER'S code --------------------
 * The reference's SpMV_merge_based_generalized is a template over a functor_t with three static members
 * (include/spmv/merge_genl/merge_genl.cuh:19-38; CPU twin include/spmv/cpu_navie.hpp:20-34)
 *     y_t initialize();   y_t combine(const mat_t& nonzero, const x_t& x);   y_t reduce(const y_t& lhs, const y_t& rhs);
 * and over five independent types (include/spmv.h:29-34).  A C ABI cannot take a C++ type, but it can take its TEXT:
 *   source        C++ source that defines the functor (and any type it needs); __host__ __device__ __forceinline__
 *                 are understood, so a functor written for the reference is passed as it is (hiprtc has no
 *                 <cmath>: INFINITY and NAN are defined in front of the text)
 *   functor_type  the type to use, as written in C++: "MyFunctor", "MergeFunctor<float, float, double>"
 *   off_type      MI355_OFF_I32 / MI355_OFF_I64;  index_t is int
 *   mat_type, x_type, y_type   the three value types as C++ type names: "float", "double", "int", "long long", or a
 *                 trivially copyable struct the source defines (an (value, index) pair for an arg-max, say)
 * compile: the text is compiled for gfx950 at run time (hiprtc, bound on first use; no device needed);
 *          MI355_SPMV_EINVAL = the text did not compile, mi355_spmv_functor_compile_log() holds the compiler's output
 *          (this thread's last compile), MI355_SPMV_ENOTSUP = no libhiprtc.so on this machine.
 * spmv:    y[r] = reduce over the row of combine(Ax[k], x[Aj[k]]) starting from initialize(), every row written
 *          (an empty row gets initialize()).  Asynchronous on `stream`, no scratch, no host synchronisation.  As in the
 *          reference's device code, reduce must be associative and commutative and initialize() its identity.
 * This is the GENERAL path (T lanes per row, plain gathers of x); the five enumerated semirings of the merge kind
 * above are the tuned one.                                                                                          */
typedef struct mi355_spmv_functor mi355_spmv_functor;
int mi355_spmv_functor_compile(mi355_spmv_functor** functor, const char* source, const char* functor_type, int off_type,
                               const char* mat_type, const char* x_type, const char* y_type);
const char* mi355_spmv_functor_compile_log(void);
int mi355_spmv_functor_spmv(mi355_spmv_functor* functor, int32_t n_rows, int32_t n_cols, int64_t nnz, const void* Ap,
                            const int32_t* Aj, const void* Ax, const void* x, void* y, void* stream);
int mi355_spmv_functor_destroy(mi355_spmv_functor* functor);

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
