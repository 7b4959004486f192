// common.hpp — shared declarations of the MI355X CSR SpMV engine (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/mi355_spmv.h"

namespace mi355 {

constexpr int kWave = 64;            // gfx950 wavefront width
constexpr int kBlock = 256;          // 4 waves per workgroup
constexpr int kMaxGiantRows = 1024;   // giant rows a plan handles (more: they stay with their workgroup)
constexpr int64_t kGiantRow = 65536;   // a row beyond this many nonzeros is cut into slices of kGiantSlice
constexpr int64_t kGiantSlice = 32768;
// slice of a giant row when the plan's threshold is below the default (small power-law matrices: the threshold follows the
// matrix, analyze.hip find_giant_rows): half the threshold, so that a row just beyond it is already shared by two workgroups
inline int64_t giant_slice_for(int64_t giant_len) {
    int64_t s = (giant_len / 2) & ~int64_t(1023);
    if (s < 2048) s = 2048;
    return s < kGiantSlice ? s : kGiantSlice;
}
constexpr int kWideBlock = 512;      // VECTOR / LIGHT on big uniform matrices: 8 waves, chunks twice as long
constexpr int kHugeBlock = 1024;     // VECTOR, band too wide for two workgroups per CU: ONE 16-wave workgroup with ~150 KB of LDS
constexpr int kSweepRows = 4;        // rows a vector of the sweep kernel holds at least (its whole chunk stays in registers)
// ... and 8 for fp32 where a chunk of 8 rows per vector stays within the 2 048 rows a chunk may have (T >= 4): the
// same window passes then serve twice the nonzeros — staging the band, not the Aj / Ax stream, is what a swept chunk
// waits for (band of 65 537 columns, 2^22 rows x 32: 371 -> 321 us; 131 073 columns: 508 -> 411).  fp64 keeps 4: eight
// rows of doubles do not fit the 128 registers a 1 024-thread workgroup has per lane.
// Window of x of a sweeping 1 024-thread workgroup, in elements: what is left of ~155 KB of the CU's LDS next to the
// chunk's own arrays (`fixed_bytes`), in whole ROUNDS of the workgroup's 16-byte groups (16 KB) — the window is staged by
// global_load_lds, whose every wave-instruction fills 64 x 16 bytes of LDS (xwindow.hpp, chunk_rows_sweep).
inline int64_t sweep_window_cap(int64_t val_bytes, int64_t fixed_bytes) {
    const int64_t round = int64_t(kHugeBlock) * 16;
    const int64_t rounds = (155 * 1024 - fixed_bytes) / round;
    return rounds > 0 ? rounds * round / val_bytes : 0;
}
inline int sweep_rows_for(int val_type, int lanes_per_row) {
    return (val_type == MI355_VAL_F32 && (kHugeBlock / lanes_per_row) * 8 <= 2048) ? 8 : kSweepRows;
}
// VECTOR below this many nonzeros: the chunked kernels are a prologue (bounds, window, barriers), a group or two of rows and
// an epilogue — ~7.5 us however small the matrix — while the plain CSR-vector kernel (one pass, no LDS, no barrier) is done in
// 2.7-6.5 us (kernel traces, S32-band shape: 2^12 rows 7.5 vs 2.7 us, 2^14 8.3 vs 3.2, 2^15 8.5 vs 4.3, 2^16 9.6 vs 6.2; even
// at 2^17 rows = 4 M nonzeros, 11.0 both).  rocSPARSE's general kernel on the same boxes: 3.0 / 6.6 us at 2^14 / 2^16.
// The cant stand-in (4.0 M nonzeros, 64 per row): 11.7 us chunked, 10.2 plain at 16 lanes per row (11.5 at 32, 13.7 at 64).
constexpr int64_t kSmallPlainNnz = 4100000;
constexpr int kXcds = 8;             // XCDs per MI355X, each with a private L2
constexpr int kCus = 256;            // compute units per MI355X

void set_error(const char* fmt, ...);

#define MI355_HIP_TRY(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ::mi355::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,             \
                               hipGetErrorString(_e));                                   \
            return (_e == hipErrorOutOfMemory) ? MI355_SPMV_ENOMEM : MI355_SPMV_EHIP;    \
        }                                                                                \
    } while (0)

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an
// L2).  Rows/tiles that are neighbours in memory gather overlapping windows of
// x, so give every XCD one CONTIGUOUS range of work: logical id = f(blockIdx).
// Bijective for any grid size.  Speed only; results never depend on placement.
__device__ __forceinline__ unsigned xcd_contiguous_id(unsigned bid, unsigned nblocks) {
    const unsigned per = nblocks / kXcds, rem = nblocks % kXcds;
    const unsigned xcd = bid % kXcds, k = bid / kXcds;
    // XCDs [0, rem) own per+1 blocks, the rest own per.
    const unsigned base = xcd * per + (xcd < rem ? xcd : rem);
    return base + k;
}

// Tuning / test knobs: every MI355_* environment variable the plan path honours, parsed ONCE per process
// (mi355_spmv_knobs_reload re-reads them), copied into each plan at creation and echoed by
// mi355_spmv_plan_get_info (`knobs`: the non-default ones), so a stray variable is visible in every report.
// -1 / 0 = not set.  None of them can change a result beyond the summation order.
struct Knobs {
    int lanes = 0;             // MI355_SPMV_LANES            T of VECTOR / LIGHT (2..64)
    int block = 0;             // MI355_SPMV_BLOCK            256 | 512 threads (VECTOR / LIGHT)
    int64_t rows_per_chunk = 0;// MI355_SPMV_ROWS_PER_CHUNK
    int window = -1;           // MI355_SPMV_WINDOW           0 = never stage x in LDS, 1 = always
    int window_from_band = -1; // MI355_SPMV_WINDOW_FROM_BAND 0 = sample every chunk, 1 = place from the probe's band
    int segments = -1;         // MI355_SPMV_SEGMENTS         0 = no multi-band windows
    int sweep = -1;            // MI355_SPMV_SWEEP            0 = never sweep a wide band with the window, 1 = whenever legal
    int balance = -1;          // MI355_SPMV_BALANCE          0 = equal-row chunks, 1 = weight-cut chunks
    int long_steps = 0;        // MI355_SPMV_LONG_STEPS       steps after which a row goes to the long-row pass
    int giant = -1;            // MI355_SPMV_GIANT            0 = no giant-row slices
    int64_t giant_row = 0;     // MI355_SPMV_GIANT_ROW        nonzeros beyond which a row is giant (>= 4096)
    int plain = 0;             // MI355_SPMV_PLAIN            1 = the 4-byte-per-lane fallback kernels
    int small = -1;            // MI355_SPMV_SMALL            0 = small matrices keep the chunked kernels too (VECTOR / LIGHT)
    int64_t rel32_limit = 0;   // MI355_SPMV_REL32_LIMIT      tests: nonzero span beyond which a chunk leaves the 32-bit path
    int light_blocks_per_cu = 0;   // MI355_LIGHT_BLOCKS_PER_CU
    int light_chunk_div = 0;   // MI355_LIGHT_CHUNK_DIV
    int merge_block = 0;       // MI355_MERGE_BLOCK           256 | 512
    int merge_tps = 0;         // MI355_MERGE_TPS             tiles per run
    int merge_search_lanes = 0;// MI355_MERGE_SEARCH_LANES    1 | 4 | 16
    int merge_rows = -1;       // MI355_MERGE_ROWS            0 = never the row-parallel run kernel, 1 = always
    int merge_fused = -1;      // MI355_MERGE_FUSED           0 = never the single-launch small-grid kernel, 1 = whenever legal
    int plan_cache = -1;       // MI355_SPMV_PLAN_CACHE       0 = the one-shot entry points make and destroy a plan per call
    int merge_wide_window = -1;// MI355_MERGE_WIDE_WINDOW     0 = fp64 keeps the 36 KB window budget (no second try with 56 KB)
    int merge_segments = -1;   // MI355_MERGE_SEGMENTS        0 = a regular several-band matrix keeps the item walk (no segmented runs)
    int dist_exchange = 0;     // MI355_DIST_EXCHANGE         auto | bcast | sendrecv | allgather (MI355_DIST_EXCHANGE_*; auto = timed trial at create)
    int dist_trials = 0;       // MI355_DIST_TRIALS           exchanges timed per candidate by the auto pick (default 5)
    int dist_shared_device = 0;// MI355_DIST_SHARED_DEVICE    tests: 1 = a device may be listed twice (an emulated RCCL: several "GPUs" on one)
    char rccl_lib[200] = "";   // MI355_SPMV_RCCL_LIB         the RCCL to dlopen instead of librccl.so.1 (tests: tests/cpp/libfakerccl.so)
    char text[160] = "";       // the non-default ones, "NAME=value ..." (as read)
};
const Knobs& knobs();          // parsed on first use
void knobs_reload();

struct Plan {
    int kind, off_type, val_type, flags;   // val_type: the type of x, y and of all arithmetic
    int mat_type;              // the type the matrix values are stored in (= val_type, or F32 under F64 vectors: MERGE)
    int32_t n_rows, n_cols;
    int64_t nnz;               // END offset of the nonzeros: Ap[n_rows] (= their count unless nnz_begin > 0)
    int64_t nnz_read;          // elements of Aj / Ax the 16-byte loads may touch: nnz, or nnz rounded up to a multiple
                               // of 4 in a row-block plan that is not the last block (the view continues into the
                               // next block, so the tail of its last row is read by whole groups exactly as the
                               // whole matrix's plan reads it: same summation order)
    int64_t nnz_begin;         // Ap[0]: 0, or 1..3 in a row-block plan whose arrays are a 16-byte-aligned view
                               // of a larger CSR (plan_create_block): elements below it belong to no row
    Knobs knob;                // the knobs this plan was shaped under
    // row-block plans (mi355_spmv_plan_create_block): launch shape inherited from the whole matrix's plan
    bool is_block;
    int64_t block_row_begin;   // first row of the block in the whole matrix
    int64_t block_chunk_begin; // first chunk of the block in the whole plan's chunk numbering
    int64_t block_weight_off;  // weight-cut plans: (global nnz offset of the block - nnz_begin) + bal_k * block_row_begin
    bool giant_enabled;        // weight-cut plans: whether rows beyond giant_len are cut into slices
    const void* Ap;
    const int32_t* Aj;
    // launch shape
    int lanes_per_row;     // T
    int elems_per_lane;    // 1 or 4
    int64_t grid_blocks;
    // merge-path
    int64_t tile_items, n_tiles;
    int64_t tiles_per_super, n_super;   // consecutive tiles one workgroup walks; number of such groups
    bool coords_valid;
    bool merge_rows;            // MERGE: regular matrix -> runs are summed row-parallel (merge_rows_kernel)
    int64_t probe_len_min, probe_len_max;   // shortest / longest of the probe's sampled rows (valid when probe_ok)
    int probe_short_rows = 0;               // ... and how many of the 256 fill less than 3/4 of the step the longest needs
    int semiring;               // MERGE: MI355_SEMIRING_* (0 = plus-times)
    double alpha, beta;         // y = alpha * A x + beta * y (1, 0 by default)
    // structure probe (plan creation): band of (column - row) seen on sampled rows
    int64_t band_lo, band_hi;   // valid when probe_ok
    bool probe_ok;
    int block_threads;          // VECTOR / LIGHT: 256, or 512 for big uniform matrices (chunks twice as long)
    int window_bytes;           // LDS budget of the x window per workgroup (pick_window_elems)
    int window_elems;           // LDS window of x per workgroup, in elements; 0 = no window
    bool window_from_band;      // place the window from band_lo/band_hi instead of sampling per chunk
    int mr_block = 256;         // MERGE, row-parallel runs: workgroup size (512: the band needs ~78 KB of LDS) and rows per piece of a run
    int mr_piece_rows = 1984;
    int mr_sweep_lanes = 0;     // MERGE, row-parallel runs on a band wider than any window: lanes per row of the sweeping body (0 = not swept)
    bool sweep = false;         // VECTOR: the band is wider than any window — one group of rows per chunk, the window sweeps the band (chunk_rows_sweep)
    // multi-band plan: up to 4 bands of (column - row) found by clustering the probe's samples
    int n_seg;
    int64_t seg_lo[4], seg_hi[4];
    int probe_n;                 // sampled (column - row) offsets (sorted ascending once probe_sorted)
    bool probe_sorted;
    int64_t probe_off[8192];
    // row chunks of VECTOR / LIGHT
    int64_t rows_per_chunk;     // uniform plan: every chunk has this many rows
    bool balanced;              // nnz-balanced plan: chunk c = rows [chunk_row[c], chunk_row[c+1])
    int64_t n_chunks;
    int rows_cap;               // rows the LDS layout of a workgroup holds (>= any chunk)
    int64_t bal_k, bal_q;       // a row weighs (its nonzeros + bal_k), a chunk holds <= bal_q of weight
    int32_t* chunk_row;         // [n_chunks + 1], device (balanced plans only)
    // giant rows (balanced plans): rows beyond kGiantRow nonzeros are cut into slices summed by separate workgroups
    int n_giant;                // 0 = none
    int64_t giant_len;          // rows beyond this many nonzeros are giant (kGiantRow, or MI355_SPMV_GIANT_ROW)
    int64_t n_giant_slices;
    int32_t giant_row_host[kMaxGiantRows];
    int64_t giant_slice_first_host[kMaxGiantRows + 1];
    int32_t* giant_row;         // [n_giant], device
    int64_t* giant_slice_first; // [n_giant + 1], device
    void* giant_partial;        // [n_giant_slices] of value type, device
    // scratch
    void* scratch;
    size_t scratch_bytes;
    size_t scratch_capacity;   // bytes of the allocation behind `scratch` (>= scratch_bytes when reused)
    int32_t* tile_row;     // [n_tiles + 1]
    int64_t* tile_nnz;     // [n_tiles + 1]
    int32_t* carry_row;    // [n_super]
    void* carry_val;       // [n_super] of value type
    unsigned long long* counters;  // LIGHT: kXcds shards, one 128-B line each
    bool light_dequeue_once;       // LIGHT, equal-row chunks: one workgroup and one dequeue per chunk (else by index)
    int n_kernels;
    bool small_plain = false;   // VECTOR / LIGHT: a matrix small enough for the plain one-pass kernel to win (capi.hip, plan_create_impl)
    char main_kernel[64];
};

// A launch that asks for more dynamic LDS than the default 64 KB cap must raise the kernel's limit first
// (gfx950: up to 160 KB per workgroup).  Remembered per kernel, so the call happens once.
int allow_dynamic_lds(const void* kernel, size_t bytes);

// kernel launchers (one translation unit per kind)
template <typename off_t, typename val_t>
int launch_vector(const Plan& p, const off_t* Ap, const val_t* Ax, const val_t* x, val_t* y, hipStream_t s);
template <typename off_t, typename val_t, typename mat_t>
int launch_merge(Plan& p, const off_t* Ap, const mat_t* Ax, const val_t* x, val_t* y, hipStream_t s);
template <typename off_t, typename val_t>
int launch_light(const Plan& p, const off_t* Ap, const val_t* Ax, const val_t* x, val_t* y, hipStream_t s);

int merge_compute_coords(Plan& p);   // MERGE: run the search kernel now (null stream, synchronises)
int probe_structure(Plan& p);
int pick_window_elems(Plan& p, int64_t rows_per_workgroup);
int64_t segment_rows_fit(const Plan& p);
void shape_chunks(Plan& p, int rows_in_flight, int64_t chunk_div, bool allow_wide, bool allow_huge = false);
bool shape_sweep(Plan& p);   // VECTOR / LIGHT: band wider than any window, after decide_balance (analyze.hip)   // VECTOR / LIGHT: block size, chunk, window
int workgroups_per_cu_by_registers(const Plan& p);   // VECTOR / LIGHT: what the kernels' launch bounds allow
int long_steps_for(const Plan& p);   // steps of its vector after which a row is left to the long-row pass
int decide_balance(Plan& p);       // VECTOR / LIGHT, after shape_*: uniform or nnz-balanced chunks
int build_chunk_table(Plan& p);    // after the scratch is allocated
int find_giant_rows(Plan& p);      // balanced plans: rows beyond kGiantRow nonzeros (synchronises)
void shape_vector(Plan& p);
void shape_merge(Plan& p);
void shape_light(Plan& p);
void reshape_vector_balanced(Plan& p);
void reshape_light_balanced(Plan& p);
void block_grid_vector(Plan& p);   // row-block plans: grid / names from the inherited shape and the block's n_chunks
void block_grid_light(Plan& p);
void reshape_light_sweep(Plan& p);   // LIGHT: after shape_sweep said yes
// nnz-balanced cuts on the plan's chunk boundaries (analyze.hip; reads Ap on the device, synchronises)
int partition_plan(const Plan& p, int parts, int64_t* row_cuts, int64_t* chunk_cuts, int64_t* nnz_cuts);

}  // namespace mi355
