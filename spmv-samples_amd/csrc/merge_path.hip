// merge_path.hip — kind MERGE: merge-path load-balanced CSR SpMV for gfx950.
//
// Replaces the reference's vendored CUB pipeline:
//   search   include/spmv/merge_based/thread_search.cuh:15-49,
//            dispatch_spmv_orig.cuh:104-148            -> merge_search_kernel
//   tile     include/spmv/merge_based/agent_spmv_orig.cuh:454-757,
//            dispatch_spmv_orig.cuh:154-191            -> merge_tile_kernel
//   fix-up   include/spmv/merge_based/agent_segment_fixup.cuh:97-384,
//            dispatch_spmv_orig.cuh:197-229            -> merge_fixup_kernel
// and the offset_t-typed twin include/spmv/merge_genl/ (the only reference merge
// variant that accepts 64-bit offsets): both offset widths are instantiated here.
//
// The decomposition is the reference's: the merge of the row-end offsets
// Ap[1..n_rows] with the counting sequence 0..nnz-1 is cut into tiles of equal item
// count; a tile owns the rows that END inside it and the nonzeros inside it,
// whatever the row lengths.  What is rebuilt for MI355X:
//  * a workgroup walks a RUN of consecutive tiles (a "super-tile", ~32 K items) so
//    that (a) the window of x those tiles touch is staged through LDS once
//    (xwindow.hpp — the plain global gather, not the Aj/Ax stream, bounds SpMV on
//    this chip), (b) the next tile's Aj/Ax are already in flight in registers while
//    the current tile is walked, (c) the partial sum of a row that crosses tiles is
//    carried in a register instead of through memory: only one carry per
//    super-tile reaches the fix-up kernel;
//  * nonzeros are streamed with 16-byte-per-lane loads from the tile start
//    rounded down to a multiple of 4 (the reference reads 4 bytes per lane,
//    strided by the block, agent_spmv_orig.cuh:474-506);
//  * row ends are staged tile-relative as 32-bit integers whatever offset_t is,
//    so a 64-bit-offset matrix costs no extra LDS;
//  * the walk bounds each thread by the tile's real item count instead of
//    padding the row-end list and clamping the carry (agent_spmv_orig.cuh:543-548,
//    :744-753);
//  * the reference's cub::BlockScan of (key, value) pairs with ReduceByKeyOp
//    (agent_spmv_orig.cuh:616-629) becomes a flag-segmented wave64 scan
//    (six __shfl_up steps) plus a 4-entry cross-wave pass, written here;
//  * a row closed inside one thread is stored to y straight from the walk (no
//    per-item key/value register arrays, no LDS scatter pass);
//  * the fix-up is deterministic: one thread per distinct carried row sums that
//    row's carries in order and adds once, instead of float atomics
//    (agent_segment_fixup.cuh:228-271), so results do not change
//    from run to run;
//  * super-tile ids are remapped so each XCD walks a contiguous range of the matrix.
//
// n_cols == 1 (the reference's special kernel, dispatch_spmv_orig.cuh:68-96,
// :572-597) needs no special case here.

#include <climits>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "row_dot.hpp"
#include "xwindow.hpp"

namespace mi355 {

// merge items per thread (IPT) is 8 or 16; a tile has kBlock * IPT - 4 items so that
// (tile nnz + 3) / 4 <= kBlock * IPT / 4 sixteen-byte groups, IPT / 4 per thread
constexpr int kMergeSuperItems = 32768;                   // items one workgroup walks at most

// ---- semirings --------------------------------------------------------------
// The reference's generalized merge kind takes a functor with initialize / combine / reduce
// (include/spmv/merge_genl/merge_genl.cuh:19-38, agent_spmv_orig.cuh:98-124; CPU twin
// include/spmv/cpu_navie.hpp:20-34) and ships one instance, (+, *).  A C ABI cannot take a C++
// functor, so the semirings are enumerated (include/mi355_spmv.h).  min/max never round, so
// MIN_PLUS and MAX_TIMES results are bit-exact whatever the reduction order.
// "infinities" of a value type: +-inf for floating point, the extreme integers for int32 (min-plus / max-plus on
// integer weights: an identity only ever meets reduce(), never combine(), so it cannot overflow)
template <typename val_t> struct Extreme {
    __device__ static __forceinline__ val_t hi() { return val_t(INFINITY); }
    __device__ static __forceinline__ val_t lo() { return val_t(-INFINITY); }
};
template <> struct Extreme<int32_t> {
    __device__ static __forceinline__ int32_t hi() { return INT32_MAX; }
    __device__ static __forceinline__ int32_t lo() { return INT32_MIN; }
};
template <int S, typename val_t> struct Semiring;
template <typename val_t> struct Semiring<MI355_SEMIRING_PLUS_TIMES, val_t> {
    __device__ static __forceinline__ val_t identity() { return val_t(0); }
    __device__ static __forceinline__ val_t combine(val_t a, val_t x) { return a * x; }
    __device__ static __forceinline__ val_t reduce(val_t u, val_t v) { return u + v; }
};
template <typename val_t> struct Semiring<MI355_SEMIRING_MIN_PLUS, val_t> {
    __device__ static __forceinline__ val_t identity() { return Extreme<val_t>::hi(); }
    __device__ static __forceinline__ val_t combine(val_t a, val_t x) { return a + x; }
    __device__ static __forceinline__ val_t reduce(val_t u, val_t v) { return v < u ? v : u; }
};
template <typename val_t> struct Semiring<MI355_SEMIRING_MAX_TIMES, val_t> {
    __device__ static __forceinline__ val_t identity() { return Extreme<val_t>::lo(); }
    __device__ static __forceinline__ val_t combine(val_t a, val_t x) { return a * x; }
    __device__ static __forceinline__ val_t reduce(val_t u, val_t v) { return u < v ? v : u; }
};

template <typename val_t> struct Semiring<MI355_SEMIRING_MAX_PLUS, val_t> {
    __device__ static __forceinline__ val_t identity() { return Extreme<val_t>::lo(); }
    __device__ static __forceinline__ val_t combine(val_t a, val_t x) { return a + x; }
    __device__ static __forceinline__ val_t reduce(val_t u, val_t v) { return u < v ? v : u; }
};
template <typename val_t> struct Semiring<MI355_SEMIRING_OR_AND, val_t> {   // booleans carried as 0.0 / 1.0
    __device__ static __forceinline__ val_t identity() { return val_t(0); }
    __device__ static __forceinline__ val_t combine(val_t a, val_t x) { return (a != val_t(0) && x != val_t(0)) ? val_t(1) : val_t(0); }
    __device__ static __forceinline__ val_t reduce(val_t u, val_t v) { return (u != val_t(0) || v != val_t(0)) ? val_t(1) : val_t(0); }
};

// ---- K6: tile start coordinates ------------------------------------------------
// The split of diagonal d is the first p in [lo, hi] with Ap[p + 1] > d - p - 1.  The reference finds it
// by bisection, one thread per diagonal (thread_search.cuh:15-49): ~log2(n_rows) DEPENDENT loads, and the
// few workgroups that hold all the diagonals issue them uncoalesced.  Here L lanes share a diagonal: one
// probe per lane cuts the range (L+1)-fold per round (the predicate is monotone, so the number of probes
// below the split, a popcount of the group's ballot bits, names the sub-range), the chain is
// log_{L+1}(n_rows) + 1 loads long and the probes of a round are spread over L times as many CUs.
// L = 16 for a few thousand diagonals (latency-bound), smaller L as the diagonals alone fill the chip
// (L = 1 is the bisection); launch_merge picks L from the measured crossovers.
// The split of diagonal `diag`, searched by a group of kSearchLanes consecutive lanes (k = lane's index in its
// group, shift = the group's first lane in the wave).  Every lane of the wave must call; groups whose search has
// ended probe nothing new.  Returns the rows consumed (lo); the nonzeros consumed are nnz_begin + diag - lo.
template <int kSearchLanes, typename off_t>
__device__ __forceinline__ int64_t merge_search_group(int64_t diag, int32_t n_rows, int64_t nnz_begin, int64_t nnz,
                                                      const off_t* __restrict__ Ap, int k, int shift) {
    const int64_t count = nnz - nnz_begin;
    int64_t lo = diag - count > 0 ? diag - count : 0;
    int64_t hi = diag < n_rows ? diag : n_rows;
    while (__any(lo < hi)) {                                    // a finished group probes nothing new: c = 0
        const int64_t n = hi - lo;
        const bool last = n <= kSearchLanes;                    // every remaining position probed: c is the answer
        auto probe = [&](int j) { return last ? lo + j : lo + (int64_t(j + 1) * n) / (kSearchLanes + 1); };
        const int64_t q = probe(k);
        const int64_t qc = q < hi ? q : hi - 1;                 // clamped into [-1, n_rows): the load is in range
        const bool below = (q < hi) & (int64_t(Ap[qc + 1]) - nnz_begin <= diag - qc - 1);
        const int c = __popcll((__ballot(below) >> shift) & ((kSearchLanes == 64 ? ~0ull : (1ull << kSearchLanes) - 1)));
        if (last) {
            lo += c;
            hi = lo;
        } else {
            const int64_t new_lo = c > 0 ? probe(c - 1) + 1 : lo;
            hi = c < kSearchLanes ? probe(c) : hi;
            lo = new_lo;
        }
    }
    return lo;
}

template <int kSearchLanes, typename off_t>
__global__ __launch_bounds__(kBlock) void merge_search_kernel(
    int32_t n_rows, int64_t nnz_begin, int64_t nnz, const off_t* __restrict__ Ap, int64_t tile_items, int64_t n_tiles,
    int32_t* __restrict__ tile_row, int64_t* __restrict__ tile_nnz) {
    // (nnz_begin = Ap[0], nnz = Ap[n_rows]: the counting sequence of the merge is nnz_begin .. nnz - 1; a
    // row-block view of a larger CSR starts at 1..3, everything else at 0)
    const int64_t gid = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t t_raw = gid / kSearchLanes;
    const int64_t t = t_raw <= n_tiles ? t_raw : n_tiles;       // surplus groups repeat the last diagonal
    const int k = int(gid) & (kSearchLanes - 1);
    const int shift = (threadIdx.x & (kWave - 1)) & ~(kSearchLanes - 1);
    const int64_t items = int64_t(n_rows) + (nnz - nnz_begin);
    int64_t diag = t * tile_items;
    if (diag > items) diag = items;
    const int64_t lo = merge_search_group<kSearchLanes, off_t>(diag, n_rows, nnz_begin, nnz, Ap, k, shift);
    if (k == 0 && t_raw <= n_tiles) {
        tile_row[t] = int32_t(lo);
        tile_nnz[t] = nnz_begin + diag - lo;
    }
}

// ---- K7: one run of consecutive tiles per workgroup ---------------------------------
// SEARCH: the run's tile coordinates are found HERE (16 lanes per diagonal, up to 16 diagonals at once by the
// whole workgroup) instead of by a search kernel in front: one launch and one kernel boundary fewer per SpMV
// (the reference has the same option, agent_spmv_orig.cuh:697-719).  The workgroup's first lane group also stores
// them where the search kernel would have, so plan_merge_coords / MI355_PLAN_REUSE_STRUCTURE see the same arrays.
// mat_t: the type the matrix values are STORED in — val_t, or float under double vectors (the reference keeps the
// matrix / x / y types apart, include/spmv.h:29-34; its generalized merge kind computes in the y type,
// merge_genl.cuh:29-31): a value is widened when it meets x, products and sums are val_t throughout.
template <int BLOCK, int IPT, bool VEC, bool WINDOW, int S, bool SEARCH, typename off_t, typename val_t, typename mat_t = val_t>
__global__ __launch_bounds__(BLOCK) void merge_tile_kernel(
    int32_t n_rows, int32_t n_cols, int64_t nnz_begin, int64_t nnz, const off_t* __restrict__ Ap, const int32_t* __restrict__ Aj,
    const mat_t* __restrict__ Ax, const val_t* __restrict__ x, val_t* __restrict__ y,
    int32_t* __restrict__ tile_row_g, int64_t* __restrict__ tile_nnz_g, int64_t tile_items,
    int32_t* __restrict__ carry_row, val_t* __restrict__ carry_val, int64_t n_tiles, int32_t tiles_per_super,
    int32_t window_cap, BandHint hint, val_t alpha, val_t beta) {
    constexpr int G = IPT / 4;
    using v4 = typename Vec4<val_t>::type;
    using m4 = typename Vec4<mat_t>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];   // window_cap values of x
    val_t* s_x = reinterpret_cast<val_t*>(s_dyn);
    __shared__ __attribute__((aligned(32))) val_t s_nz[BLOCK * IPT];   // products, index = nnz - (y0 & ~3)
    __shared__ int s_re[BLOCK * IPT + 1];                               // tile-relative row ends
    __shared__ val_t s_wave_sum[BLOCK / kWave];
    __shared__ int s_wave_flag[BLOCK / kWave];
    __shared__ int s_red[2];

    const unsigned sup = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int tid = threadIdx.x;
    const int lane64 = tid & (kWave - 1);
    const int wave = tid / kWave;
    const int64_t first = int64_t(sup) * tiles_per_super;
    const int64_t last = min(first + tiles_per_super, n_tiles);

    // tile coordinates first .. last of this run: from the search kernel's arrays, or searched here
    constexpr int kMaxRun = 32;                       // (launch_merge: tiles_per_super < kMaxRun when SEARCH)
    __shared__ int32_t s_tile_row[SEARCH ? kMaxRun + 1 : 1];
    __shared__ int64_t s_tile_nnz[SEARCH ? kMaxRun + 1 : 1];
    if constexpr (SEARCH) {
        const int64_t items = int64_t(n_rows) + (nnz - nnz_begin);
        const int n_diag = int(last - first) + 1;
        // one pass of the whole workgroup: 16 lanes per diagonal for up to BLOCK / 16 diagonals, else 8 lanes
        auto pass = [&](auto lanes_tag) {
            constexpr int L = decltype(lanes_tag)::value;
            const int d = tid / L;
            int64_t diag = (first + min(d, n_diag - 1)) * tile_items;           // surplus groups repeat the last diagonal
            if (diag > items) diag = items;
            const int64_t lo = merge_search_group<L, off_t>(diag, n_rows, nnz_begin, nnz, Ap, tid & (L - 1),
                                                            (tid & (kWave - 1)) & ~(L - 1));
            if ((tid & (L - 1)) == 0 && d < n_diag) {
                s_tile_row[d] = int32_t(lo);
                s_tile_nnz[d] = nnz_begin + diag - lo;
                if (d < n_diag - 1 || last == n_tiles) {                        // (the next run stores its own first one)
                    tile_row_g[first + d] = int32_t(lo);
                    tile_nnz_g[first + d] = nnz_begin + diag - lo;
                }
            }
        };
        if (n_diag <= BLOCK / 16) pass(std::integral_constant<int, 16>{});      // (uniform over the workgroup)
        else pass(std::integral_constant<int, 8>{});
        __syncthreads();
    }
    const int32_t* const tile_row = SEARCH ? s_tile_row - first : tile_row_g;   // (indexed by absolute tile number below)
    const int64_t* const tile_nnz = SEARCH ? s_tile_nnz - first : tile_nnz_g;

    int x0 = tile_row[first], x1 = tile_row[first + 1];
    int64_t y0 = tile_nnz[first], y1 = tile_nnz[first + 1];
    const int64_t row_lo = x0;
    const int64_t row_hi = min(int64_t(tile_row[last]) + 1, int64_t(n_rows));

    // registers holding the Aj/Ax groups of the tile about to be processed
    int4v c[G];
    m4 a[G];
    // branch-free: addresses are clamped below the last whole 16-byte group of the arrays (hipcc
    // serialises loads it finds in branches); the few nonzeros at or past nnz_vec are redone below
    const int64_t nnz_vec = nnz & ~int64_t(3);
    const int64_t j_max = nnz_vec - 4;                 // VEC launches guarantee nnz >= 4
    auto issue = [&](int64_t ya) {
        const int64_t base = ya & ~int64_t(3);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            int64_t j = base + 4 * int64_t(tid + g * BLOCK);
            j = j < j_max ? j : j_max;
            c[g] = stream_load(reinterpret_cast<const int4v*>(Aj + j));
            a[g] = stream_load(reinterpret_cast<const m4*>(Ax + j));
        }
    };
    if constexpr (VEC) issue(y0);
    // the window of x for all rows this run touches (incl. the row left open at its end), staged while the
    // first tile's stream is in flight
    auto first_last = [&](int64_t r, int& fc, int& lc) {
        const off_t s = Ap[r], e = Ap[r + 1];
        if (e <= s) return false;
        fc = Aj[s];
        lc = Aj[e - 1];
        return true;
    };
    const XWindow<val_t> win =
        stage_x_window<val_t>(row_lo, row_hi, n_cols, first_last, x, s_x, window_cap, s_red, hint);
    // the first BLOCK row ends of a tile are fetched one tile ahead as well (a tile with more
    // rows than that — mean row length below 8 — loads the rest when it gets there)
    auto fetch_row_end = [&](int xa, int xe) -> int64_t {
        return (tid < xe - xa) ? int64_t(Ap[int64_t(xa) + tid + 1]) : int64_t(0);
    };
    int64_t re_next = fetch_row_end(x0, x1);

    using SR = Semiring<S, val_t>;  // S = 0: the ordinary (+, *) of every other kind
    // y = alpha * (A x) + beta * y for the ordinary semiring (the fix-up adds alpha * carry); 1, 0 otherwise
    auto put = [&](int64_t row, val_t v) {
        if constexpr (S == MI355_SEMIRING_PLUS_TIMES) {
            v = alpha * v;
            if (beta != val_t(0)) v += beta * y[row];
        }
        y[row] = v;
    };
    val_t block_carry = SR::identity();   // sum so far of the row left open by the previous tile of this run
    for (int64_t t = first; t < last; ++t) {
        int x2 = x1;
        int64_t y2 = y1;
        if (t + 1 < last) {
            x2 = tile_row[t + 2];
            y2 = tile_nnz[t + 2];
        }
        const int tr = x1 - x0;            // rows that end in this tile
        const int tn = int(y1 - y0);       // nonzeros in this tile
        const int shift = VEC ? int(y0 & 3) : 0;

        // (1) products a*x for the tile's nonzeros
        if constexpr (VEC) {
            if constexpr (WINDOW) {
                // (this kernel is bound by instruction issue: the lookup is a clamp (v_min), an address and one
                // compare per element; which elements are REAL nonzeros of the tile is only worked out in the rare
                // branch where some column fell outside the window)
                const unsigned len_m1 = unsigned(win.len > 0 ? win.len - 1 : 0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    v4 p;
                    bool any_out = false;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned rel = unsigned(c[g][e] - win.lo);
                        any_out |= rel >= unsigned(win.len);
                        p[e] = SR::combine(val_t(a[g][e]), win.s_x[min(rel, len_m1)]);
                    }
                    if (any_out) {                       // rare: loaded and consumed inside the branch
                        const int rel0 = 4 * (tid + g * BLOCK) - shift;   // tile-relative index of element 0
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bool out = unsigned(c[g][e] - win.lo) >= unsigned(win.len);
                            if (out && (rel0 + e >= 0) && (rel0 + e < tn)) p[e] = SR::combine(val_t(a[g][e]), x[c[g][e]]);
                        }
                    }
                    *reinterpret_cast<v4*>(&s_nz[4 * (tid + g * BLOCK)]) = p;
                }
            } else {
                val_t xv[G][4];
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[g][e] = x[c[g][e]];   // every column is a loaded Aj entry: in range
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    v4 p;
#pragma unroll
                    for (int e = 0; e < 4; ++e) p[e] = SR::combine(val_t(a[g][e]), xv[g][e]);
                    *reinterpret_cast<v4*>(&s_nz[4 * (tid + g * BLOCK)]) = p;
                }
            }
            if (y1 > nnz_vec) {   // uniform, at most one tile: the nonzeros past the last whole group
                // Their slots were just written — with the products of the clamped group — by the thread whose
                // 16-byte store covers them, usually in another wave: the corrective stores must come after
                // that store, hence the barrier (without it the last row of a matrix with nnz % 4 != 0 was
                // wrong about once in 25 processes).
                __syncthreads();
                const int64_t k = (y0 > nnz_vec ? y0 : nnz_vec) + tid;
                if (k < y1) s_nz[int(k - y0) + shift] = SR::combine(val_t(Ax[k]), x[Aj[k]]);
            }
            // the next tile's stream goes in flight now and lands while this tile is walked
            if (t + 1 < last) issue(y1);
        } else {
            // Aj / Ax not 16-byte aligned (an offset view): 4-byte-per-lane form
            for (int i = tid; i < tn; i += BLOCK) {
                const int32_t col = Aj[y0 + i];
                s_nz[i] = SR::combine(val_t(Ax[y0 + i]), window_gather<val_t>(win, x, col, true));
            }
        }
        // (2) row ends, relative to y0; the row still open at the tile end never ends here
        if (tid <= tr) s_re[tid] = (tid < tr) ? int(re_next - y0) : INT_MAX;
        for (int i = tid + BLOCK; i <= tr; i += BLOCK) {
            s_re[i] = (i < tr) ? int(int64_t(Ap[int64_t(x0) + i + 1]) - y0) : INT_MAX;
        }
        if (t + 1 < last) re_next = fetch_row_end(x1, x2);
        __syncthreads();

        // (3) this thread's piece of the merge path: items [d0, d1) of the tile
        const int items = tr + tn;
        const int d0 = min(tid * IPT, items);
        const int d1 = min(d0 + IPT, items);
        int lo = max(d0 - tn, 0), hi = min(d0, tr);
        while (lo < hi) {
            const int p = (lo + hi) >> 1;
            if (s_re[p] <= d0 - p - 1) lo = p + 1;
            else hi = p;
        }
        int cx = lo, cy = d0 - lo;

        // (4) walk: a nonzero extends the running row sum, a row end closes it
        val_t run = SR::identity(), first_val = SR::identity();
        int first_end = -1;
        int re = s_re[cx];
        const int cnt = d1 - d0;
        // Rows longer than a thread's IPT items (the banded target, FEM matrices, stencils) put at most ONE row
        // end among them: the items are then `before` nonzeros, the row end, and the rest nonzeros of the next
        // row — contiguous in s_nz.  When that holds for every thread of the wave, the item-by-item walk (IPT
        // dependent LDS reads and branches) is replaced by IPT independent reads and two masked sums, in the
        // same order, so the result is the same bit for bit.  Otherwise the wave walks.
        const int before = re - cy;                                   // nonzeros ahead of my first row end
        const bool has_end = before < cnt;
        const int re2 = (has_end && cx + 1 <= tr) ? s_re[cx + 1] : INT_MAX;
        const bool simple = !has_end || (re2 - re >= cnt - before - 1);
        if (__all(simple)) {
            const int n_nz = cnt - (has_end ? 1 : 0);                 // my nonzeros: s_nz[cy + shift .. + n_nz)
            val_t v[IPT];
#pragma unroll
            for (int k = 0; k < IPT; ++k) v[k] = s_nz[min(cy + shift + k, BLOCK * IPT - 1)];
            val_t head = SR::identity(), tail = SR::identity();
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                head = (k < before && k < n_nz) ? SR::reduce(head, v[k]) : head;
                tail = (k >= before && k < n_nz) ? SR::reduce(tail, v[k]) : tail;
            }
            if (has_end) {
                first_end = cx;                           // may continue a row opened by earlier threads
                first_val = head;
                run = tail;
            } else {
                run = head;
            }
        } else {
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                if (k < cnt) {
                    if (cy < re) {
                        run = SR::reduce(run, s_nz[cy + shift]);
                        ++cy;
                    } else {
                        if (first_end < 0) {
                            first_end = cx;                   // may continue a row opened by earlier threads
                            first_val = run;
                        } else {
                            put(int64_t(x0) + cx, run);       // opened and closed inside this thread
                        }
                        run = SR::identity();
                        ++cx;
                        re = s_re[cx];
                    }
                }
            }
        }

        // (5) carry-in = sum of the open-row tails of the preceding threads back to the
        //     last thread that closed a row (or the previous tile's carry): a
        //     flag-segmented inclusive scan
        val_t sv = run;
        int sf = first_end >= 0 ? 1 : 0;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const val_t ov = __shfl_up(sv, d, kWave);
            const int of = __shfl_up(sf, d, kWave);
            if (lane64 >= d) {
                if (!sf) sv = SR::reduce(ov, sv);
                sf |= of;
            }
        }
        if (lane64 == kWave - 1) {
            s_wave_sum[wave] = sv;
            s_wave_flag[wave] = sf;
        }
        __syncthreads();
        val_t prefix = block_carry;  // block-inclusive value at the end of the previous wave
        val_t total = block_carry;   // ... and at the end of the last wave: the row still open at the end of the tile
#pragma unroll
        for (int w = 0; w < BLOCK / kWave; ++w) {
            const val_t ws = s_wave_sum[w];
            total = s_wave_flag[w] ? ws : SR::reduce(total, ws);
            if (w + 1 == wave) prefix = total;
        }
        const val_t incl = sf ? sv : SR::reduce(prefix, sv);
        val_t carry_in = __shfl_up(incl, 1, kWave);
        if (lane64 == 0) carry_in = prefix;
        if (first_end >= 0) put(int64_t(x0) + first_end, SR::reduce(carry_in, first_val));
        // No third barrier (round 1 passed the tile's carry through one more LDS word): every thread folds the
        // wave totals itself, and the next tile cannot disturb this one — it writes s_nz / s_re before ITS first
        // barrier and the wave totals after it, and no wave gets there before every wave has left the second
        // barrier of this tile with its reads of s_nz / s_re done (they precede that barrier).
        block_carry = total;
        x0 = x1; y0 = y1;
        x1 = x2; y1 = y2;
    }
    if (tid == 0) {
        // the row still open at the end of the run: x0 now holds tile_row[last] (== n_rows: none)
        carry_row[sup] = x0;
        carry_val[sup] = block_carry;
    }
}

// ---- K7r: a run of a REGULAR matrix, row-parallel ------------------------------------------------------
// The merge-path cut decides WHICH nonzeros and row ends a workgroup owns (the run: a fixed number of merge items,
// whatever the row lengths — the reference's decomposition, agent_spmv_orig.cuh:697-719); how the workgroup sums them
// is free.  On a matrix whose rows are alike and not short (plan: shape_merge, merge_rows) the item-by-item machinery
// of K7 — products through LDS, a search per thread, a segmented scan per tile — costs ~460 instructions per wave and
// tile and holds the kernel at 4.4 TB/s.  Here the run is handed to the row-chunk body of the CSR-vector kind
// (xwindow.hpp: T lanes per row, 16-byte loads straight into registers, window of x, results swept from LDS): the
// rows that END in the run are stored, the partial sum of the row still open at its end becomes the run's carry, and a
// row that began in an earlier run contributes the part that lies in this one (the fix-up adds the earlier carries,
// exactly as for K7).  The workgroup finds its two diagonals itself.  Runs with more rows than the LDS layout holds
// (stretches of empty rows) are walked in pieces.
constexpr int kMergeRowsCap = 1984;    // rows per piece: bounds + results fit 16 KB next to the window

// SEARCH: the workgroup finds its two diagonals itself (grids of a few rounds); else run_row / run_nnz hold the run
// boundaries, found by the search kernel on n_super + 1 diagonals (every workgroup of a big grid would otherwise pay
// the chain of dependent loads at its start).
// TS > 0: the band is wider than any window (plan: shape_merge, mr_sweep_lanes) — a piece is then ONE group of rows of the
// 1 024-thread workgroup, TS lanes per row and R rows per vector held in registers, and the window sweeps the band
// (xwindow.hpp, chunk_rows_sweep: the CSR-vector kind's body for such bands; plain gathers ran the run at 1.6 TB/s).
// NSEG > 1: the columns sit in several far-apart bands (the 3-D stencil) — each band gets its own segment of the window,
// staged per piece of the run (xwindow.hpp, stage_x_segments: the CSR-vector kind's multi-band plan).
template <int NSEG> struct SegmentArg { static const SegmentPlan& pick(const SegmentPlan& s, const struct NoSegments&) { return s; } };
struct NoSegments {};   // (the one-window variants take no segment list: 68 bytes of kernel arguments cost the sweep variants their last scalar registers)
template <> struct SegmentArg<1> { static const NoSegments& pick(const SegmentPlan&, const NoSegments& n) { return n; } };
template <int BLOCK, int R, bool WINDOW, bool SEARCH, typename off_t, typename val_t, int TS = 0, int NSEG = 1>
__global__ __launch_bounds__(BLOCK, (BLOCK >= kWideBlock ? 4 : 3)) void merge_rows_kernel(
    int32_t n_rows, int32_t n_cols, int64_t nnz_begin, int64_t nnz, const off_t* __restrict__ Ap,
    const int32_t* __restrict__ Aj_arg, const val_t* __restrict__ Ax_arg, const val_t* __restrict__ x_arg,
    val_t* __restrict__ y_arg, int64_t tile_items, const int32_t* __restrict__ run_row, const int64_t* __restrict__ run_nnz,
    int32_t* __restrict__ carry_row, val_t* __restrict__ carry_val,
    int64_t n_tiles, int32_t tiles_per_super, int32_t window_cap, BandHint hint, val_t alpha, val_t beta, int32_t piece_rows,
    typename std::conditional<(NSEG > 1), SegmentPlan, NoSegments>::type segs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];   // window | bounds | y | flags
    __shared__ int s_red[2];
    __shared__ int64_t s_diag[4];            // (row, nnz) of the run's first and last diagonal
    ChunkScratch<val_t> scr(s_dyn, window_cap, piece_rows);
    scr.alpha = alpha;
    scr.beta = beta;
    const unsigned sup = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int tid = threadIdx.x;
    const int64_t first = int64_t(sup) * tiles_per_super;
    const int64_t last = min(first + tiles_per_super, n_tiles);
    if constexpr (!SEARCH) {
        if (tid == 0) {
            s_diag[0] = run_row[sup];
            s_diag[1] = run_nnz[sup];
            s_diag[2] = run_row[sup + 1];
            s_diag[3] = run_nnz[sup + 1];
        }
        __syncthreads();
    } else {   // the two diagonals, 16 lanes each (lanes 0..31 of wave 0; the other lanes of that wave repeat the second)
        const int64_t items = int64_t(n_rows) + (nnz - nnz_begin);
        if (tid < kWave) {
            const int which = min(tid / 16, 1);
            int64_t diag = (which ? last : first) * tile_items;
            if (diag > items) diag = items;
            const int64_t lo = merge_search_group<16, off_t>(diag, n_rows, nnz_begin, nnz, Ap, tid & 15, tid & 48);
            if ((tid & 15) == 0 && tid < 32) {
                s_diag[2 * which] = lo;
                s_diag[2 * which + 1] = nnz_begin + diag - lo;
            }
        }
        __syncthreads();
    }
    // rows [row_lo, row_last) END in this run; row_last (if it exists) is open at its end.  The four coordinates stay in
    // LDS and are read again by every piece (scalar loads): held in scalar registers across the pieces they were what
    // pushed the eight-row sweeping variants past their register budget.  Row counts of a run fit 32 bits.
    int n_store_all, n_all;
    {
        const int64_t row_lo0 = uniform_i64(s_diag[0]), row_last0 = uniform_i64(s_diag[2]);
        n_store_all = int(row_last0 - row_lo0);
        n_all = n_store_all + (row_last0 < n_rows ? 1 : 0);
    }
    val_t carry = val_t(0);
    for (int pb = 0; pb < n_all; pb += piece_rows) {                 // (uniform; one piece unless the run holds more rows than the layout)
        const int pe = min(pb + piece_rows, n_all);
        const int rows = pe - pb;
        const int64_t row_lo = uniform_i64(s_diag[0]), y_first = uniform_i64(s_diag[1]), y_last = uniform_i64(s_diag[3]);
        const int64_t base = y_first & ~int64_t(3);
        const int64_t left = nnz - base;
        const int32_t nnz_c = int32_t(left < kRel32Limit ? left : kRel32Limit);
        // opaque copies of the operand pointers, once per piece (see light_rows.hip: keeps per-thread addresses from
        // being hoisted out of this loop and spilled)
        int zero = 0;
        asm volatile("" : "+v"(zero));
        zero = __builtin_amdgcn_readfirstlane(zero);
        const int32_t* const Aj_c = Aj_arg + zero + base;
        const val_t* const Ax_c = Ax_arg + zero + base;
        const val_t* const x = x_arg + zero;
        val_t* const y = y_arg + zero;
        // bounds of the piece's rows, clipped to the run's nonzeros [y_first, y_last), relative to base
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        for (int i = t2; i <= rows; i += BLOCK) {
            int64_t b = int64_t(Ap[min(row_lo + pb + i, int64_t(n_rows))]);
            b = b < y_first ? y_first : (b > y_last ? y_last : b);
            scr.s_b[i] = int32_t(b - base);
        }
        for (int i = t2; i < rows / 32 + 1; i += BLOCK) scr.long_map[i] = 0u;
        scr.store_rows = min(pe, n_store_all) - pb;                   // the open row's partial stays in s_y
        __syncthreads();
        const int64_t rb = row_lo + pb, re = row_lo + pe;
        if constexpr (TS > 0) {
            chunk_rows_sweep<BLOCK, TS, R, val_t>(rb, re, nnz_c, Aj_c, Ax_c, x, y, n_cols, window_cap, hint, scr);
            __syncthreads();
            if (pe == n_all && n_all > n_store_all) carry = uniform_val(scr.s_y[rows - 1]);
            __syncthreads();
            continue;
        }
        if constexpr (NSEG > 1) {
            auto stage = [&] { return stage_x_segments<val_t>(rb, re, n_cols, x, scr.s_x, window_cap, segs); };
            chunk_rows_any<BLOCK, 2, R, true, true, val_t, decltype(stage)&, true>(rb, re, nnz_c, Aj_c, Ax_c, x, y, stage, scr);
        } else {
            auto first_last = [&](int64_t r, int& fc, int& lc) {
                const int32_t s = scr.s_b[r - rb], e = scr.s_b[r - rb + 1];
                if (e <= s) return false;
                fc = Aj_c[s];
                lc = Aj_c[e - 1];
                return true;
            };
            auto stage = [&] { return stage_x_window<val_t>(rb, re, n_cols, first_last, x, scr.s_x, window_cap, s_red, hint); };
            chunk_rows_any<BLOCK, 2, R, WINDOW, true, val_t, decltype(stage)&, true>(rb, re, nnz_c, Aj_c, Ax_c, x, y, stage, scr);
        }
        __syncthreads();
        if (pe == n_all && n_all > n_store_all) carry = uniform_val(scr.s_y[rows - 1]);   // (one LDS word: scalar register)
        __syncthreads();                                                       // ... read before the next piece refills it
    }
    if (tid == 0) {
        carry_row[sup] = int32_t(s_diag[2]);      // the run's last row; == n_rows: no row is open (the fix-up skips it)
        carry_val[sup] = carry;
    }
}

// ---- K8: add the carries of rows that straddle runs ------------------------------------
template <int S, typename val_t>
__global__ __launch_bounds__(kBlock) void merge_fixup_kernel(
    int64_t n_carries, int32_t n_rows, const int32_t* __restrict__ carry_row,
    const val_t* __restrict__ carry_val, val_t* __restrict__ y, val_t alpha) {
    const int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (t >= n_carries) return;
    // the neighbours and the value are fetched with carry_row[t]: two dependent round trips (then y[r]), not four
    const int32_t r = carry_row[t];
    const int32_t r_prev = t > 0 ? carry_row[t - 1] : -1;
    const int32_t r_next = t + 1 < n_carries ? carry_row[t + 1] : -1;
    val_t s = carry_val[t];
    if (r >= n_rows || r_prev == r) return;      // no carry, or not the first run carrying row r
    using SR = Semiring<S, val_t>;
    if (r_next == r)
        for (int64_t u = t + 1; u < n_carries && carry_row[u] == r; ++u) s = SR::reduce(s, carry_val[u]);
    if constexpr (S == MI355_SEMIRING_PLUS_TIMES) s = alpha * s;
    y[r] = SR::reduce(y[r], s);
}

// ---- host side -----------------------------------------------------------------------
// Whether the tile kernel searches its own coordinates (MI355_MERGE_FUSED = 0 | 1 overrides): always, when a run
// is short enough for the workgroup to search all its diagonals in two passes.
static bool merge_search_in_kernel(const Plan& p) {
    if (p.knob.merge_fused == 0) return false;
    if (p.tiles_per_super + 1 > 32) return false;         // (one pass of 256 threads searches 32 diagonals)
    if (p.knob.merge_fused > 0) return true;
    // Measured (us, fused / search kernel in front): web-Google stand-in (1 473 runs) 46.6 / 49.5, cant stand-in 19.3 /
    // 20.3 — but S32-band (4 233 runs of 16 tiles) 282 / 258 and R-MAT-24 (8 700 runs) 2 516 / 2 483: on a grid of
    // many rounds every workgroup pays the search's chain of dependent loads at its start, and the search kernel's
    // ~9 us are a few per cent.  Fused where the grid is at most ~two rounds of the chip.
    return p.n_super <= int64_t(kCus) * 8;
}

// Row-parallel runs (merge_rows_kernel) for a matrix whose rows all but fill ONE step of a vector of 2, 4, 8, 16 or 32
// lanes (the widths the run body picks from: 8, 16, 32, 64, 128 nonzeros per step) — the probe's 256 sampled rows all
// hold between three quarters of such a step and the whole of it — and big enough for runs of 16 K+ items (on a small
// matrix a run is a tile or two: the window and the two diagonals cost more than they are worth — cant stand-in 40.6 us
// against 19.5 with the item walk).  Measured on 2^27 nonzeros (us, runs / item walk; scripts/gpu_r02_merge_regular.py,
// profiles/r02_row_length_scan.txt): fixed 8 per row 347 / 410, 16: 285 / 315, 27: 251 / 275, 32: 218 / 257, 48: 239 / 256,
// 64: 205 / 245, 100: 229 / 242, 128: 195 / 237 — but 40: 262 / 256 (a step of 64 is 62 % full), and rows of VARYING length
// lose at every mean (24 +- 6: 415 / 341, 64 +- 16: 479 / 381, 128 +- 32: 502 / 380): those keep the item walk, at
// 4.8-5.5 TB/s.  MI355_MERGE_ROWS = 0 | 1 overrides.
[[maybe_unused]] static bool merge_rows_regular(const Plan& p);
[[maybe_unused]] static bool merge_rows_wanted(const Plan& p) {
    if (p.knob.merge_rows >= 0) return p.knob.merge_rows != 0;
    if (p.tiles_per_super * p.tile_items < 16000) return false;      // (8 tiles of 2 044 items and up)
    return merge_rows_regular(p);
}
[[maybe_unused]] static bool merge_rows_regular(const Plan& p) {
    if (!p.probe_ok || p.n_rows <= 0 || p.val_type == MI355_VAL_I32) return false;
    // (... or all but an eighth of them do: the boundary rows of a stencil — the nlpkkt stand-in's 27-point rows are
    // 18, 12 or 8 long on the faces, edges and corners of its box — cost their vectors a few idle lanes, nothing more)
    for (const int64_t step : {8, 16, 32, 64, 128})
        if (p.probe_len_max <= step && (p.probe_len_min * 4 >= step * 3 || p.probe_short_rows * 8 <= kBlock)) return true;
    return false;
}

#if !defined(MI355_TU_F64) && !defined(MI355_TU_I32)   // the host-side shape functions live in the fp32 translation unit only
void shape_merge(Plan& p) {
    // tuning knobs: MI355_MERGE_TPS = tiles per run (and MI355_SPMV_WINDOW = 0|1, analyze.hip)
    // 256 threads x 8 items or (MI355_MERGE_BLOCK=512) 512 threads x 4 items: the same 2 044-item tiles
    p.block_threads = p.knob.merge_block == kWideBlock ? kWideBlock : kBlock;
    const int ipt = p.block_threads == kWideBlock ? 4 : 8;   // (16: S32-band 251 vs 259 us, web-Google stand-in 52.4 vs 46.4: not kept)
    p.lanes_per_row = 0;
    p.elems_per_lane = ipt;            // reported as items per thread for this kind
    p.tile_items = int64_t(p.block_threads) * ipt - 4;
    const int64_t items = int64_t(p.n_rows) + (p.nnz - p.nnz_begin);
    p.n_tiles = (items + p.tile_items - 1) / p.tile_items;
    // runs of up to ~32 K items, but at least ~4 runs per CU when the matrix allows
    int64_t tps = p.n_tiles / (int64_t(kCus) * 4);
    const int64_t cap = kMergeSuperItems / p.tile_items;
    if (tps > cap) tps = cap;
    if (p.knob.merge_tps > 0) tps = p.knob.merge_tps;
    if (tps < 1) tps = 1;
    // A REGULAR mid-size matrix (one to four runs of 8 tiles per CU) takes runs of 8 tiles rather than the two to seven
    // the rule above gives it: 16 K items are what the row-parallel runs and their window of x need to pay
    // (S32-band shape, us, before / after: 2^17 rows 20.5 / 19.4, 2^18 37.0 / 25.1, 2^19 47.7 / 41.3 — that one by the
    // threshold in merge_rows_wanted alone).  Taken back below if no window placed from the band serves such a run.
    const int64_t tps_small = tps;
    const bool bumped = p.knob.merge_tps <= 0 && p.knob.merge_rows < 0 && tps < 8 && p.n_tiles >= 8 * int64_t(kCus) &&
                        p.block_threads == kBlock && merge_rows_regular(p);
    if (bumped) tps = 8;
    p.tiles_per_super = tps;
    p.n_super = (p.n_tiles + tps - 1) / tps;
    p.grid_blocks = p.n_super;
    // a window of x only pays when a run is long enough to amortise staging it, and
    // when the band the probe saw (plus the rows of a run) fits
    bool several_bands = false;
    int segment_piece = 0;             // rows per piece of a row-parallel run with one window segment per band (0: not that plan)
    {
        const int64_t mean1 = 1 + (p.n_rows > 0 ? (p.nnz - p.nnz_begin) / p.n_rows : 0);
        const int64_t rows_per_run = tps * p.tile_items / mean1 + 1;
        p.window_elems = (tps * p.tile_items >= 8192) ? pick_window_elems(p, rows_per_run) : 0;
        // fp64 halves what the 36 KB budget (three workgroups per CU) holds: the S32-band shape in fp64 ran on plain
        // gathers at 2.4 TB/s.  Second try with 56 KB (two workgroups per CU next to the kernel's 16-24 KB of own LDS).
        if (p.window_elems == 0 && p.n_seg < 2 && p.val_type == MI355_VAL_F64 && p.knob.window < 0 && tps * p.tile_items >= 8192 &&
            p.knob.merge_wide_window != 0) {
            p.window_bytes = 56 * 1024;
            p.window_elems = pick_window_elems(p, rows_per_run);
            if (p.window_elems == 0 || p.n_seg >= 2) p.window_bytes = 0;
        }
        // Several far-apart bands (the 3-D stencil).  A REGULAR matrix of that kind takes row-parallel runs with a segment
        // of the window per band, staged per piece of a run — the CSR-vector kind's multi-band plan: the piece is as many
        // rows as the bands leave room for, a run is one piece.  (Round 2 had this at 681 us against the item walk's 727
        // on the C4 stand-in, with spilling kernels, and dropped it; the chunk body of round 3 fits its registers.)
        // MI355_MERGE_SEGMENTS=0 keeps the item walk on plain gathers, as every other several-band matrix does.
        if (p.n_seg >= 2 && p.block_threads == kBlock && p.knob.merge_segments != 0 && p.knob.merge_tps <= 0 &&
            p.knob.window < 0 && merge_rows_wanted(p)) {
            int64_t piece = segment_rows_fit(p);
            if (piece > kMergeRowsCap) piece = kMergeRowsCap;
            piece &= ~int64_t(3);
            int64_t t2 = piece * mean1 / p.tile_items;
            if (t2 > kMergeSuperItems / p.tile_items) t2 = kMergeSuperItems / p.tile_items;
            const int64_t n_super = t2 >= 1 ? (p.n_tiles + t2 - 1) / t2 : 0;
            if (piece >= 256 && t2 >= 1 && n_super >= int64_t(kCus) * 2) {
                int64_t need = 0;                         // (LDS is occupancy: what the bands need with that many rows)
                for (int i = 0; i < p.n_seg; ++i) need += p.seg_hi[i] - p.seg_lo[i] + 1 + 4 + piece;
                need = (need + 3) & ~int64_t(3);
                if (need < p.window_elems) p.window_elems = int(need);
                p.tiles_per_super = t2;
                p.n_super = n_super;
                p.grid_blocks = n_super;
                segment_piece = int(piece);
            }
        }
        if (bumped && !(p.window_elems > 0 && p.n_seg < 2 && p.window_from_band)) {   // no window for runs of 8 tiles: the shorter runs
            tps = tps_small;
            p.tiles_per_super = tps;
            p.n_super = (p.n_tiles + tps - 1) / tps;
            p.grid_blocks = p.n_super;
            p.window_bytes = 0;
            p.window_elems = (tps * p.tile_items >= 8192) ? pick_window_elems(p, tps * p.tile_items / mean1 + 1) : 0;
            segment_piece = 0;
        }
        several_bands = p.n_seg >= 2 && segment_piece == 0;
        if (several_bands) { p.window_elems = 0; p.n_seg = 0; }   // several bands: the item walk keeps to global gathers
    }
    p.n_kernels = (p.n_super > 1 ? 2 : 1) + ((merge_search_in_kernel(p) && p.block_threads == kBlock) ? 0 : 1);
    // (a matrix whose columns sit in several far-apart bands — the 3-D stencil — keeps the item walk: row-parallel runs
    // on plain gathers measured 720 us against 650-700 on the C4 stand-in, and with the bands staged per piece of a run
    // 681 against 727 on one box, with four spilling kernels: not kept)
    p.merge_rows = p.block_threads == kBlock && !several_bands && merge_rows_wanted(p);
    p.mr_block = kBlock;
    p.mr_piece_rows = segment_piece > 0 ? segment_piece : kMergeRowsCap;
    if (segment_piece > 0 && !p.merge_rows) { p.window_elems = 0; p.n_seg = 0; }   // (cannot happen: merge_rows_wanted held above)
    // The band does not fit the window of a 256-thread workgroup (fp64 on the S32-band shape: 8 193 columns + the rows of a run):
    // two workgroups of 512 threads per CU may take ~78 KB each, as the CSR-vector kind's wide plan does; the run is then
    // as long as the rows the band leaves room for, and walked in one piece.
    if (p.merge_rows && segment_piece == 0 && p.knob.merge_wide_window != 0 && p.knob.window < 0 && p.knob.merge_tps <= 0 && p.probe_ok &&
        !(p.window_elems > 0 && p.window_from_band)) {
        const int64_t vb = p.val_type == MI355_VAL_F64 ? 8 : 4;
        const int64_t band = p.band_hi - p.band_lo + 1;
        const int64_t mean1 = 1 + (p.n_rows > 0 ? (p.nnz - p.nnz_begin) / p.n_rows : 0);
        // ... and a band too wide for that gets ONE workgroup of 1 024 threads per CU with ~155 KB (the CSR-vector kind's
        // third plan): fp32, 32 769 columns, 32 per row: 343 -> see profiles/r02_shape_sweep.txt
        const struct { int block; int64_t lds; int64_t min_piece; } tries[2] = {{kWideBlock, 78 * 1024, 256}, {kHugeBlock, 155 * 1024, 512}};
        for (const auto& t : tries) {
            int64_t piece = (t.lds - vb * (band + 8) - 4) * 8 / (8 * (2 * vb + 4) + 1);   // val (band + rows + 8) + 4 (rows + 1) + val rows + rows / 8
            piece &= ~int64_t(3);
            if (piece > kMergeRowsCap) piece = kMergeRowsCap;
            if (!(band > 0 && piece >= t.min_piece)) continue;
            const Plan saved = p;
            int64_t t2 = piece * mean1 / p.tile_items;
            if (t2 > kMergeSuperItems / p.tile_items) t2 = kMergeSuperItems / p.tile_items;
            if (t2 < 1) t2 = 1;
            p.tiles_per_super = t2;
            p.n_super = (p.n_tiles + t2 - 1) / t2;
            p.grid_blocks = p.n_super;
            p.window_bytes = int(vb * (band + piece + 8));
            p.window_elems = pick_window_elems(p, piece);
            if (p.window_elems > 0 && p.n_seg < 2 && p.window_from_band && p.n_super >= int64_t(kCus) * 2) {
                p.mr_block = t.block;
                p.mr_piece_rows = int(piece);
                break;
            }
            p = saved;
        }
    }
    // Still no window: the band is wider than one CU's LDS.  The CSR-vector kind sweeps such a band with the window
    // (analyze.hip, shape_sweep); a run here does the same — a piece = one group of rows of a 1 024-thread workgroup held in
    // registers, 4 T nonzeros per row in one step — under the same rule: the staged bytes of a piece stay below half the
    // line fills its nonzeros would cost as plain gathers.  Rows of up to 8 nonzeros (T = 2) keep the gathers.
    p.mr_sweep_lanes = 0;
    if (p.merge_rows && p.window_elems == 0 && p.probe_ok && p.knob.sweep != 0 && p.knob.window < 0 && p.knob.merge_tps <= 0 &&
        p.knob.merge_wide_window != 0 && p.probe_len_max > 8 && p.probe_len_max <= 128) {
        const int64_t vb = p.val_type == MI355_VAL_F64 ? 8 : 4;
        const int64_t band = p.band_hi - p.band_lo + 1;
        const int64_t mean1 = 1 + (p.n_rows > 0 ? (p.nnz - p.nnz_begin) / p.n_rows : 0);
        int t = 4;
        while (t < 32 && 4 * t < p.probe_len_max) t *= 2;
        const int64_t piece = int64_t(kHugeBlock / t) * sweep_rows_for(p.val_type, t);
        const int64_t fixed = int64_t(chunk_lds_bytes(0, int(piece), size_t(vb)));
        const int64_t cap = sweep_window_cap(vb, fixed);
        const int64_t span = band + piece + 8;
        const int64_t passes = cap > 0 ? (span + cap - 1) / cap : 0;
        int64_t t2 = piece * mean1 / p.tile_items;
        if (t2 > kMergeSuperItems / p.tile_items) t2 = kMergeSuperItems / p.tile_items;
        if (t2 < 1) t2 = 1;
        const int64_t n_super = (p.n_tiles + t2 - 1) / t2;
        const bool pays = span * vb <= 64 * (mean1 - 1) * piece;
        if (band > 0 && passes >= 1 && passes <= 16 && n_super >= int64_t(kCus) * 2 && (pays || p.knob.sweep == 1)) {
            p.tiles_per_super = t2;
            p.n_super = n_super;
            p.grid_blocks = n_super;
            p.mr_block = kHugeBlock;
            p.mr_piece_rows = int(piece);
            p.mr_sweep_lanes = t;
            p.window_bytes = int(cap * vb);
            p.window_elems = int(cap);
            p.window_from_band = true;
            p.n_seg = 0;
        }
    }
    if (p.merge_rows) {
        p.n_kernels = (p.n_super > 1 ? 2 : 1) + (merge_search_in_kernel(p) ? 0 : 1);
        snprintf(p.main_kernel, sizeof(p.main_kernel), "merge_rows_kernel");
        return;
    }
    p.coords_valid = false;
    snprintf(p.main_kernel, sizeof(p.main_kernel), "merge_tile_kernel");
}

// The tile coordinates on demand (mi355_spmv_plan_merge_coords on a plan whose executes do not produce them: the
// row-parallel run kernel only ever finds its own two diagonals).
int merge_compute_coords(Plan& p) {
    if (p.n_rows == 0 || p.n_tiles == 0) return MI355_SPMV_OK;
    const unsigned g = unsigned(((p.n_tiles + 1) * 4 + kBlock - 1) / kBlock);
    if (p.off_type == MI355_OFF_I32)
        hipLaunchKernelGGL((merge_search_kernel<4, int32_t>), dim3(g), dim3(kBlock), 0, nullptr, p.n_rows, p.nnz_begin, p.nnz,
                           static_cast<const int32_t*>(p.Ap), p.tile_items, p.n_tiles, p.tile_row, p.tile_nnz);
    else
        hipLaunchKernelGGL((merge_search_kernel<4, int64_t>), dim3(g), dim3(kBlock), 0, nullptr, p.n_rows, p.nnz_begin, p.nnz,
                           static_cast<const int64_t*>(p.Ap), p.tile_items, p.n_tiles, p.tile_row, p.tile_nnz);
    MI355_HIP_TRY(hipGetLastError());
    MI355_HIP_TRY(hipStreamSynchronize(nullptr));
    return MI355_SPMV_OK;
}

#endif

template <typename off_t, typename val_t, typename mat_t>
int launch_merge(Plan& p, const off_t* Ap, const mat_t* Ax, const val_t* x, val_t* y, hipStream_t s) {
    if (p.n_rows == 0 || p.n_tiles == 0) return MI355_SPMV_OK;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.Aj) | reinterpret_cast<uintptr_t>(Ax) |
                           reinterpret_cast<uintptr_t>(x)) & 15u) == 0;
    const bool reuse = (p.flags & MI355_PLAN_REUSE_STRUCTURE) && p.coords_valid;
    const bool vec = aligned && p.nnz >= 4;
    const bool wide = p.block_threads == kWideBlock && p.semiring == MI355_SEMIRING_PLUS_TIMES;
    // the tile kernel finds its run's coordinates itself (no search kernel in front) on the 16-byte path with 256 threads
    // regular matrix: row-parallel runs (plus-times, one value type, 16-byte path); the kernel searches its own diagonals
    if constexpr (std::is_same<val_t, mat_t>::value && std::is_floating_point<val_t>::value) {
        if (p.merge_rows && vec && p.semiring == MI355_SEMIRING_PLUS_TIMES) {
            constexpr int RR = sizeof(val_t) == 4 ? 4 : 2;
            const int32_t capw = (int32_t)p.window_elems;
            const size_t lds = chunk_lds_bytes(capw, p.mr_piece_rows, sizeof(val_t));
            const BandHint hint_r{p.band_lo, p.band_hi, p.window_from_band};
            const dim3 grid_r((unsigned)p.n_super);
            // run boundaries: searched in the kernel on small grids, by the search kernel (n_super + 1 diagonals of
            // tiles_per_super tiles each; the last one clamps to the end of the merge) on big ones
            const bool in_kernel = merge_search_in_kernel(p);
            if (!in_kernel) {
                const int64_t diagonals = p.n_super + 1;
                const unsigned gs = unsigned((diagonals * 4 + kBlock - 1) / kBlock);
                hipLaunchKernelGGL((merge_search_kernel<4, off_t>), dim3(gs), dim3(kBlock), 0, s, p.n_rows, p.nnz_begin, p.nnz, Ap,
                                   p.tile_items * p.tiles_per_super, p.n_super, p.tile_row, p.tile_nnz);
                MI355_HIP_TRY(hipGetLastError());
                p.coords_valid = false;       // (the arrays now hold RUN boundaries, not tile coordinates)
            }
            // (rows a vector keeps in flight: the 512-thread kernel is held to 128 VGPRs, which the fp32 body with 4 rows exceeds)
            SegmentPlan segs;
            segs.n = p.n_seg;
            for (int i = 0; i < kMaxSegments; ++i) { segs.lo[i] = p.seg_lo[i]; segs.hi[i] = p.seg_hi[i]; }
            const NoSegments no_segs;
#define MI355_MERGE_ROWS_LAUNCH_N(BLOCK_, WIN_, SEARCH_, NSEG_)                                                    \
    do {                                                                                                           \
        constexpr int RR_ = (BLOCK_ >= kWideBlock || NSEG_ > 1) ? 2 : RR;   /* (several bands: the fp32 body with 4 rows spills) */ \
        const auto& segs_ = SegmentArg<NSEG_>::pick(segs, no_segs);                                                 \
        if (const int st = allow_dynamic_lds((const void*)merge_rows_kernel<BLOCK_, RR_, WIN_, SEARCH_, off_t, val_t, 0, NSEG_>, lds + 1024)) return st; \
        hipLaunchKernelGGL((merge_rows_kernel<BLOCK_, RR_, WIN_, SEARCH_, off_t, val_t, 0, NSEG_>), grid_r, dim3(BLOCK_), lds, s, p.n_rows, \
                           p.n_cols, p.nnz_begin, p.nnz, Ap, p.Aj, Ax, x, y, p.tile_items, p.tile_row, p.tile_nnz,   \
                           p.carry_row, static_cast<val_t*>(p.carry_val), p.n_tiles, (int32_t)p.tiles_per_super,    \
                           capw, hint_r, (val_t)p.alpha, (val_t)p.beta, (int32_t)p.mr_piece_rows, segs_);           \
    } while (0)
#define MI355_MERGE_ROWS_LAUNCH(BLOCK_, WIN_, SEARCH_) MI355_MERGE_ROWS_LAUNCH_N(BLOCK_, WIN_, SEARCH_, 1)
            if (p.mr_sweep_lanes > 0 && capw > 0) {           // the window sweeps the band: one group of rows per piece
                constexpr int RS = sizeof(val_t) == 4 ? 8 : kSweepRows;
                const BandHint hint_s{p.band_lo, p.band_hi, true};
                if (p.mr_piece_rows != (kHugeBlock / p.mr_sweep_lanes) * RS || capw < int32_t(kHugeBlock * 16 / sizeof(val_t))) {
                    set_error("merge: sweep plan with %d rows per piece at %d lanes per row", p.mr_piece_rows, p.mr_sweep_lanes);
                    return MI355_SPMV_EINVAL;
                }
#define MI355_MERGE_SWEEP_LAUNCH(TS_, SEARCH_)                                                                     \
    do {                                                                                                           \
        if (const int st = allow_dynamic_lds((const void*)merge_rows_kernel<kHugeBlock, RS, true, SEARCH_, off_t, val_t, TS_>, lds + 1024)) return st; \
        hipLaunchKernelGGL((merge_rows_kernel<kHugeBlock, RS, true, SEARCH_, off_t, val_t, TS_>), grid_r, dim3(kHugeBlock), lds, s, p.n_rows, \
                           p.n_cols, p.nnz_begin, p.nnz, Ap, p.Aj, Ax, x, y, p.tile_items, p.tile_row, p.tile_nnz,   \
                           p.carry_row, static_cast<val_t*>(p.carry_val), p.n_tiles, (int32_t)p.tiles_per_super,    \
                           capw, hint_s, (val_t)p.alpha, (val_t)p.beta, (int32_t)p.mr_piece_rows, no_segs);         \
    } while (0)
                switch (p.mr_sweep_lanes) {
                    case 4:  if (in_kernel) MI355_MERGE_SWEEP_LAUNCH(4, true); else MI355_MERGE_SWEEP_LAUNCH(4, false); break;
                    case 8:  if (in_kernel) MI355_MERGE_SWEEP_LAUNCH(8, true); else MI355_MERGE_SWEEP_LAUNCH(8, false); break;
                    case 16: if (in_kernel) MI355_MERGE_SWEEP_LAUNCH(16, true); else MI355_MERGE_SWEEP_LAUNCH(16, false); break;
                    case 32: if (in_kernel) MI355_MERGE_SWEEP_LAUNCH(32, true); else MI355_MERGE_SWEEP_LAUNCH(32, false); break;
                    default: set_error("merge: bad sweep width %d", p.mr_sweep_lanes); return MI355_SPMV_EINVAL;
                }
#undef MI355_MERGE_SWEEP_LAUNCH
            }
            else if (p.mr_block == kHugeBlock && capw > 0) {       // (the wide run kernels exist around ONE window of x)
                if (in_kernel) MI355_MERGE_ROWS_LAUNCH(kHugeBlock, true, true); else MI355_MERGE_ROWS_LAUNCH(kHugeBlock, true, false);
            }
            else if (p.mr_block == kWideBlock && capw > 0) {
                if (in_kernel) MI355_MERGE_ROWS_LAUNCH(kWideBlock, true, true); else MI355_MERGE_ROWS_LAUNCH(kWideBlock, true, false);
            }
            else if (capw > 0 && p.n_seg >= 2) {                   // several bands: a segment of the window each
                if (in_kernel) MI355_MERGE_ROWS_LAUNCH_N(kBlock, true, true, kMaxSegments); else MI355_MERGE_ROWS_LAUNCH_N(kBlock, true, false, kMaxSegments);
            }
            else if (capw > 0) { if (in_kernel) MI355_MERGE_ROWS_LAUNCH(kBlock, true, true); else MI355_MERGE_ROWS_LAUNCH(kBlock, true, false); }
            else { if (in_kernel) MI355_MERGE_ROWS_LAUNCH(kBlock, false, true); else MI355_MERGE_ROWS_LAUNCH(kBlock, false, false); }
#undef MI355_MERGE_ROWS_LAUNCH
#undef MI355_MERGE_ROWS_LAUNCH_N
            MI355_HIP_TRY(hipGetLastError());
            if (p.n_super > 1) {
                const unsigned g = unsigned((p.n_super + kBlock - 1) / kBlock);
                hipLaunchKernelGGL((merge_fixup_kernel<MI355_SEMIRING_PLUS_TIMES, val_t>), dim3(g), dim3(kBlock), 0, s, p.n_super,
                                   p.n_rows, p.carry_row, static_cast<const val_t*>(p.carry_val), y, (val_t)p.alpha);
                MI355_HIP_TRY(hipGetLastError());
            }
            return MI355_SPMV_OK;
        }
    }
    const bool fused = !reuse && vec && !wide && merge_search_in_kernel(p);
    if (!reuse && !fused) {
        const int64_t diagonals = p.n_tiles + 1;
        // measured (us, L = 1 / 4 / 16): 2 946 diagonals 7.3 / 6.0 / 4.4, 68 K 11.9 / 8.5 / 13.3, 139 K 14.0 / 16.0 / 30.9
        const int forced = p.knob.merge_search_lanes;
        const int lanes = forced > 0 ? forced : diagonals <= 16384 ? 16 : diagonals <= 98304 ? 4 : 1;
        const unsigned g = unsigned((diagonals * lanes + kBlock - 1) / kBlock);
        if (lanes >= 16)
            hipLaunchKernelGGL((merge_search_kernel<16, off_t>), dim3(g), dim3(kBlock), 0, s, p.n_rows, p.nnz_begin, p.nnz, Ap,
                               p.tile_items, p.n_tiles, p.tile_row, p.tile_nnz);
        else if (lanes >= 4)
            hipLaunchKernelGGL((merge_search_kernel<4, off_t>), dim3(g), dim3(kBlock), 0, s, p.n_rows, p.nnz_begin, p.nnz, Ap,
                               p.tile_items, p.n_tiles, p.tile_row, p.tile_nnz);
        else
            hipLaunchKernelGGL((merge_search_kernel<1, off_t>), dim3(g), dim3(kBlock), 0, s, p.n_rows, p.nnz_begin, p.nnz, Ap,
                               p.tile_items, p.n_tiles, p.tile_row, p.tile_nnz);
        MI355_HIP_TRY(hipGetLastError());
    }
    if (!reuse) p.coords_valid = true;
    const int32_t cap = (aligned && p.nnz >= 4) ? (int32_t)p.window_elems : 0;
    const size_t dyn = size_t(cap) * sizeof(val_t);
    const BandHint hint{p.band_lo, p.band_hi, p.window_from_band};
    const dim3 grid((unsigned)p.n_super);
#define MI355_MERGE_ARGS dyn, s, p.n_rows, p.n_cols, p.nnz_begin, p.nnz, Ap, p.Aj, Ax, x, y, p.tile_row, p.tile_nnz, p.tile_items, p.carry_row, \
                       static_cast<val_t*>(p.carry_val), p.n_tiles, (int32_t)p.tiles_per_super, cap, hint, (val_t)p.alpha, (val_t)p.beta
#define MI355_MERGE_ALLOW(K_)                                                                                 \
    if (dyn > 40 * 1024)                                                                                      \
        if (const int st_ = allow_dynamic_lds((const void*)K_, dyn + 24 * 1024)) return st_;   /* (dynamic + the kernel's own LDS may pass 64 KB) */
#define MI355_MERGE_LAUNCH(VEC_, WIN_, S_)                                                                    \
    do {                                                                                                      \
        if constexpr (S_ == MI355_SEMIRING_PLUS_TIMES) {                                                      \
            if (wide) {                                                                                       \
                MI355_MERGE_ALLOW((merge_tile_kernel<kWideBlock, 4, VEC_, WIN_, S_, false, off_t, val_t, mat_t>)) \
                hipLaunchKernelGGL((merge_tile_kernel<kWideBlock, 4, VEC_, WIN_, S_, false, off_t, val_t, mat_t>), grid, dim3(kWideBlock), MI355_MERGE_ARGS); \
                break;                                                                                        \
            }                                                                                                 \
        }                                                                                                     \
        if constexpr (VEC_) {                                                                                 \
            if (fused) {                                                                                      \
                MI355_MERGE_ALLOW((merge_tile_kernel<kBlock, 8, VEC_, WIN_, S_, true, off_t, val_t, mat_t>))  \
                hipLaunchKernelGGL((merge_tile_kernel<kBlock, 8, VEC_, WIN_, S_, true, off_t, val_t, mat_t>), grid, dim3(kBlock), MI355_MERGE_ARGS); \
                break;                                                                                        \
            }                                                                                                 \
        }                                                                                                     \
        MI355_MERGE_ALLOW((merge_tile_kernel<kBlock, 8, VEC_, WIN_, S_, false, off_t, val_t, mat_t>))         \
        hipLaunchKernelGGL((merge_tile_kernel<kBlock, 8, VEC_, WIN_, S_, false, off_t, val_t, mat_t>), grid, dim3(kBlock), MI355_MERGE_ARGS); \
    } while (0)
#define MI355_MERGE_SEMIRING(S_)                                                    \
    do {                                                                            \
        if (!vec) MI355_MERGE_LAUNCH(false, false, S_);                             \
        else if (cap > 0) MI355_MERGE_LAUNCH(true, true, S_);                       \
        else MI355_MERGE_LAUNCH(true, false, S_);                                   \
        MI355_HIP_TRY(hipGetLastError());                                           \
        if (p.n_super > 1) {                                                        \
            const unsigned g = unsigned((p.n_super + kBlock - 1) / kBlock);         \
            hipLaunchKernelGGL((merge_fixup_kernel<S_, val_t>), dim3(g), dim3(kBlock), 0, s, p.n_super, p.n_rows, \
                               p.carry_row, static_cast<const val_t*>(p.carry_val), y, (val_t)p.alpha);      \
            MI355_HIP_TRY(hipGetLastError());                                       \
        }                                                                           \
    } while (0)
    switch (p.semiring) {
        case MI355_SEMIRING_PLUS_TIMES: MI355_MERGE_SEMIRING(MI355_SEMIRING_PLUS_TIMES); break;
        case MI355_SEMIRING_MIN_PLUS:   MI355_MERGE_SEMIRING(MI355_SEMIRING_MIN_PLUS); break;
        case MI355_SEMIRING_MAX_TIMES:  MI355_MERGE_SEMIRING(MI355_SEMIRING_MAX_TIMES); break;
        case MI355_SEMIRING_MAX_PLUS:   MI355_MERGE_SEMIRING(MI355_SEMIRING_MAX_PLUS); break;
        case MI355_SEMIRING_OR_AND:     MI355_MERGE_SEMIRING(MI355_SEMIRING_OR_AND); break;
        default:
            set_error("merge: unknown semiring %d", p.semiring);
            return MI355_SPMV_EINVAL;
    }
#undef MI355_MERGE_SEMIRING
#undef MI355_MERGE_LAUNCH
#undef MI355_MERGE_ALLOW
#undef MI355_MERGE_ARGS
    return MI355_SPMV_OK;
}

// One translation unit per value type (merge_path_f64.hip / merge_path_i32.hip include this file with MI355_TU_F64 /
// MI355_TU_I32): the three thirds of the instantiations compile side by side.
#if defined(MI355_TU_F64)
template int launch_merge<int32_t, double, double>(Plan&, const int32_t*, const double*, const double*, double*, hipStream_t);
template int launch_merge<int64_t, double, double>(Plan&, const int64_t*, const double*, const double*, double*, hipStream_t);
// fp32 matrix under fp64 vectors (mi355_spmv_plan_create_typed)
template int launch_merge<int32_t, double, float>(Plan&, const int32_t*, const float*, const double*, double*, hipStream_t);
template int launch_merge<int64_t, double, float>(Plan&, const int64_t*, const float*, const double*, double*, hipStream_t);
#elif defined(MI355_TU_I32)
// 32-bit integer values (MI355_VAL_I32: every semiring, exact)
template int launch_merge<int32_t, int32_t, int32_t>(Plan&, const int32_t*, const int32_t*, const int32_t*, int32_t*, hipStream_t);
template int launch_merge<int64_t, int32_t, int32_t>(Plan&, const int64_t*, const int32_t*, const int32_t*, int32_t*, hipStream_t);
#else
template int launch_merge<int32_t, float, float>(Plan&, const int32_t*, const float*, const float*, float*, hipStream_t);
template int launch_merge<int64_t, float, float>(Plan&, const int64_t*, const float*, const float*, float*, hipStream_t);
#endif

}  // namespace mi355
