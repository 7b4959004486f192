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

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <mutex>
#include <new>
#include <unordered_map>
#include <utility>

#include "common.hpp"
#include "xwindow.hpp"

namespace mi355 {

constexpr int kProbePerRow = 32;   // (column - row) samples per probed row

// ---- knobs --------------------------------------------------------------------------------------
static std::mutex g_knobs_mutex;
static Knobs g_knobs;
static bool g_knobs_ready = false;

static void parse_knobs(Knobs& k) {
    k = Knobs();
    size_t used = 0;
    auto note = [&](const char* name, const char* v) {
        const int w = snprintf(k.text + used, sizeof(k.text) - used, "%s%s=%s", used ? " " : "", name, v);
        if (w > 0) used = std::min(sizeof(k.text) - 1, used + size_t(w));
    };
    auto geti = [&](const char* name, int& dst) {
        if (const char* v = getenv(name)) { dst = atoi(v); note(name, v); }
    };
    auto getl = [&](const char* name, int64_t& dst) {
        if (const char* v = getenv(name)) { dst = atoll(v); note(name, v); }
    };
    geti("MI355_SPMV_LANES", k.lanes);
    geti("MI355_SPMV_BLOCK", k.block);
    getl("MI355_SPMV_ROWS_PER_CHUNK", k.rows_per_chunk);
    geti("MI355_SPMV_WINDOW", k.window);
    geti("MI355_SPMV_WINDOW_FROM_BAND", k.window_from_band);
    geti("MI355_SPMV_SEGMENTS", k.segments);
    geti("MI355_SPMV_SWEEP", k.sweep);
    geti("MI355_SPMV_BALANCE", k.balance);
    geti("MI355_SPMV_LONG_STEPS", k.long_steps);
    geti("MI355_SPMV_GIANT", k.giant);
    getl("MI355_SPMV_GIANT_ROW", k.giant_row);
    geti("MI355_SPMV_PLAIN", k.plain);
    geti("MI355_SPMV_SMALL", k.small);
    getl("MI355_SPMV_REL32_LIMIT", k.rel32_limit);
    geti("MI355_LIGHT_BLOCKS_PER_CU", k.light_blocks_per_cu);
    geti("MI355_LIGHT_CHUNK_DIV", k.light_chunk_div);
    geti("MI355_MERGE_BLOCK", k.merge_block);
    geti("MI355_MERGE_TPS", k.merge_tps);
    geti("MI355_MERGE_SEARCH_LANES", k.merge_search_lanes);
    geti("MI355_MERGE_FUSED", k.merge_fused);
    geti("MI355_MERGE_WIDE_WINDOW", k.merge_wide_window);
    geti("MI355_MERGE_SEGMENTS", k.merge_segments);
    geti("MI355_SPMV_PLAN_CACHE", k.plan_cache);
    geti("MI355_MERGE_ROWS", k.merge_rows);
    geti("MI355_DIST_TRIALS", k.dist_trials);
    geti("MI355_DIST_SHARED_DEVICE", k.dist_shared_device);
    if (const char* v = getenv("MI355_DIST_EXCHANGE")) {
        k.dist_exchange = !strcmp(v, "bcast") ? MI355_DIST_EXCHANGE_BCAST : !strcmp(v, "sendrecv") ? MI355_DIST_EXCHANGE_SENDRECV
                          : !strcmp(v, "allgather") ? MI355_DIST_EXCHANGE_ALLGATHER : MI355_DIST_EXCHANGE_AUTO;
        note("MI355_DIST_EXCHANGE", v);
    }
    if (const char* v = getenv("MI355_SPMV_RCCL_LIB")) {
        snprintf(k.rccl_lib, sizeof(k.rccl_lib), "%s", v);
        note("MI355_SPMV_RCCL_LIB", strrchr(v, '/') ? strrchr(v, '/') + 1 : v);
    }
    if (k.window > 1) k.window = 1;
    if (k.balance > 1) k.balance = 1;
}

const Knobs& knobs() {
    std::lock_guard<std::mutex> lock(g_knobs_mutex);
    if (!g_knobs_ready) { parse_knobs(g_knobs); g_knobs_ready = true; }
    return g_knobs;
}

void knobs_reload() {
    std::lock_guard<std::mutex> lock(g_knobs_mutex);
    parse_knobs(g_knobs);
    g_knobs_ready = true;
}

template <typename off_t>
__global__ __launch_bounds__(kBlock) void probe_kernel(int32_t n_rows, const off_t* __restrict__ Ap,
                                                       const int32_t* __restrict__ Aj, long long* out) {
    // out[0], out[1]: band [lo, hi] over the first/last column of 256 rows;
    // out[2 + 32 t + i]: (column - row) at 32 positions spread over row t's nonzeros (LLONG_MAX = none);
    // out[2 + 32 * 256], out[3 + 32 * 256]: shortest / longest of the 256 rows
    __shared__ long long s_lo[kBlock / kWave], s_hi[kBlock / kWave], s_lmin[kBlock / kWave], s_lmax[kBlock / kWave];
    __shared__ int s_short;
    const int tid = threadIdx.x;
    long long lo = LLONG_MAX, hi = LLONG_MIN, lmin = LLONG_MAX, lmax = 0, mine = 0;
    for (int i = 0; i < kProbePerRow; ++i) out[2 + tid * kProbePerRow + i] = LLONG_MAX;
    if (n_rows > 0) {
        const int64_t r = (int64_t(n_rows - 1) * tid) / (kBlock - 1);
        const off_t s = Ap[r], e = Ap[r + 1];
        lmin = lmax = mine = (long long)(e - s);
        if (e > s) {
            const long long first = Aj[s], last = Aj[e - 1];
            lo = min(first, last) - r;
            hi = max(first, last) - r;
            const off_t len = e - s;
            const int take = len < kProbePerRow ? int(len) : kProbePerRow;
            for (int i = 0; i < take; ++i) {
                const off_t k = take > 1 ? s + ((len - 1) * i) / (take - 1) : s;
                out[2 + tid * kProbePerRow + i] = (long long)Aj[k] - r;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, kWave));
        hi = max(hi, __shfl_xor(hi, o, kWave));
        lmin = min(lmin, __shfl_xor(lmin, o, kWave));
        lmax = max(lmax, __shfl_xor(lmax, o, kWave));
    }
    if ((tid & (kWave - 1)) == 0) {
        s_lo[tid / kWave] = lo;
        s_hi[tid / kWave] = hi;
        s_lmin[tid / kWave] = lmin;
        s_lmax[tid / kWave] = lmax;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kBlock / kWave; ++w) {
            lo = min(lo, s_lo[w]);
            hi = max(hi, s_hi[w]);
            lmin = min(lmin, s_lmin[w]);
            lmax = max(lmax, s_lmax[w]);
        }
        out[0] = lo;
        out[1] = hi;
        out[2 + kBlock * kProbePerRow] = lmin;
        out[3 + kBlock * kProbePerRow] = lmax;
        s_lmax[0] = lmax;
        s_short = 0;
    }
    // how many of the sampled rows fill less than three quarters of the step (8, 16, 32, 64 or 128 nonzeros: the widths
    // a vector of lanes covers at once) that the longest of them needs: a stencil's boundary rows are a few per cent
    // of the sample, a matrix of VARYING row lengths half of it (merge_path.hip, merge_rows_wanted)
    __syncthreads();
    {
        const long long longest = s_lmax[0];
        long long step = 8;
        while (step < longest && step < 128) step *= 2;
        const bool is_short = n_rows > 0 && mine * 4 < step * 3;
        const int n = __popcll(__ballot(is_short));
        if ((tid & (kWave - 1)) == 0 && n) atomicAdd(&s_short, n);
    }
    __syncthreads();
    if (tid == 0) out[4 + kBlock * kProbePerRow] = s_short;
}

// Device scratch of the plan-time kernels (probe samples, heaviest-chunk word): one allocation per device
// for the life of the process instead of a hipMalloc + hipFree (an implicit device synchronisation) per plan —
// the one-shot entry points create a plan per call, like the reference's kinds.  plan_create holds the lock
// while it uses the buffer.
constexpr size_t kAnalysisWords = 2 + size_t(kBlock) * kProbePerRow + 2 + 2;   // band, samples, row lengths, 2 spare words
static std::mutex g_analysis_mutex;
static long long* g_analysis_buf[64] = {};

static long long* analysis_buffer() {   // call with g_analysis_mutex held
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!g_analysis_buf[dev]) {
        void* ptr = nullptr;
        if (hipMalloc(&ptr, kAnalysisWords * sizeof(long long)) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        g_analysis_buf[dev] = static_cast<long long*>(ptr);
    }
    return g_analysis_buf[dev];
}

int probe_structure(Plan& p) {
    p.probe_ok = false;
    p.band_lo = p.band_hi = 0;
    p.probe_n = 0;
    p.n_seg = 0;
    if (p.n_rows <= 0 || p.nnz <= 0) return MI355_SPMV_OK;
    constexpr size_t NS = 2 + size_t(kBlock) * kProbePerRow;   // band + samples
    constexpr size_t N = NS + 3;                                // + shortest / longest sampled row, + the count of short ones
    static_assert(size_t(kBlock) * kProbePerRow <= sizeof(p.probe_off) / sizeof(p.probe_off[0]), "probe buffer");
    p.probe_len_min = p.probe_len_max = 0;
    p.probe_short_rows = 0;
    std::lock_guard<std::mutex> lock(g_analysis_mutex);
    long long* d_out = analysis_buffer();
    if (!d_out) { set_error("probe_structure: no device scratch"); return MI355_SPMV_ENOMEM; }
    if (p.off_type == MI355_OFF_I32)
        hipLaunchKernelGGL((probe_kernel<int32_t>), dim3(1), dim3(kBlock), 0, nullptr, p.n_rows,
                           static_cast<const int32_t*>(p.Ap), p.Aj, d_out);
    else
        hipLaunchKernelGGL((probe_kernel<int64_t>), dim3(1), dim3(kBlock), 0, nullptr, p.n_rows,
                           static_cast<const int64_t*>(p.Ap), p.Aj, d_out);
    long long* h = new (std::nothrow) long long[N];
    hipError_t e = h ? hipGetLastError() : hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMemcpy(h, d_out, N * sizeof(long long), hipMemcpyDeviceToHost);   // synchronises
    if (e != hipSuccess) {
        delete[] h;
        set_error("probe_structure: %s", hipGetErrorString(e));
        return MI355_SPMV_EHIP;
    }
    if (h[0] <= h[1]) {
        p.band_lo = h[0];
        p.band_hi = h[1];
        p.probe_ok = true;
        for (size_t i = 2; i < NS; ++i)
            if (h[i] != LLONG_MAX) p.probe_off[p.probe_n++] = h[i];
        p.probe_len_min = h[NS];
        p.probe_len_max = h[NS + 1];
        p.probe_short_rows = int(h[NS + 2]);
        p.probe_sorted = false;    // sorted on first use (cluster_bands): most plans never need the samples
    }
    delete[] h;
    return MI355_SPMV_OK;
}

// ---- uniform or nnz-balanced chunks (VECTOR, LIGHT) ---------------------------------------------
// Chunks of equal ROW count are right for matrices whose rows are alike (the S32-band target, FEM
// matrices, stencils): no table, no extra load.  On a power-law matrix they are not: the 2 048 rows
// that hold the hubs of the web-Google stand-in carry 8 % of all nonzeros, one workgroup walks them
// while the chip idles (vector 763 us, light 575 us vs merge 49 us; R-MAT-24: 20.4 / 14.4 ms vs 2.4 ms).
// The plan therefore measures the heaviest uniform chunk once (two reads of Ap per chunk) and, when it is
// more than twice the mean,
// cuts the rows by WEIGHT instead: a row weighs (its nonzeros + k), k = mean row length, and chunk c
// starts at the first row r with Ap[r] + k r >= c Q.  That is the merge-path diagonal cut with rows
// weighted k instead of 1 (thread_search.cuh:15-49) at chunk granularity: a chunk holds at most Q / k rows
// and at most Q nonzeros plus one row's overshoot, and the boundaries come from a binary search per
// chunk at plan creation (chunk_table_kernel) instead of a per-launch search kernel.
template <typename off_t>
__global__ __launch_bounds__(kBlock) void chunk_max_kernel(int32_t n_rows, const off_t* __restrict__ Ap,
                                                           int64_t rows_per_chunk, int64_t n_chunks,
                                                           unsigned long long* out) {
    unsigned long long m = 0;
    for (int64_t c = int64_t(blockIdx.x) * kBlock + threadIdx.x; c < n_chunks; c += int64_t(gridDim.x) * kBlock) {
        const int64_t rb = c * rows_per_chunk;
        const int64_t re = min(rb + rows_per_chunk, int64_t(n_rows));
        const unsigned long long w = (unsigned long long)(Ap[re] - Ap[rb]);
        m = w > m ? w : m;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long other = __shfl_xor(m, o, kWave);
        m = other > m ? other : m;
    }
    if ((threadIdx.x & (kWave - 1)) == 0 && m) atomicMax(out, m);
}

template <typename off_t>
__global__ __launch_bounds__(kBlock) void chunk_table_kernel(int32_t n_rows, const off_t* __restrict__ Ap, int64_t k,
                                                             int64_t q, int64_t n_chunks,
                                                             int32_t* __restrict__ chunk_row, int64_t weight_off,
                                                             int64_t chunk_off) {
    // weight_off / chunk_off: a row-block plan numbers weights and chunks as the WHOLE matrix's plan does
    // (weight of local row r = Ap[r] + k r + weight_off, local chunk c = whole chunk c + chunk_off), so that its
    // boundaries are the whole plan's; both 0 otherwise.  A block starts on a multiple of 4 rows of the whole.
    const int64_t c = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (c > n_chunks) return;
    if (c == n_chunks) {
        chunk_row[c] = n_rows;
        return;
    }
    const int64_t target = (c + chunk_off) * q - weight_off;   // first r in [0, n_rows] with Ap[r] + k r >= target
    int64_t lo = 0, hi = n_rows;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (int64_t(Ap[mid]) + k * mid >= target) hi = mid;
        else lo = mid + 1;
    }
    chunk_row[c] = int32_t(lo & ~int64_t(3));   // multiples of 4 rows keep the y sweep on 16-byte stores
}

int allow_dynamic_lds(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return MI355_SPMV_OK;
    static std::mutex mu;
    static std::unordered_map<const void*, size_t> allowed;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = allowed[kernel];
    if (have >= bytes) return MI355_SPMV_OK;
    MI355_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    have = bytes;
    return MI355_SPMV_OK;
}

// Workgroups of the plan's size a CU holds by REGISTERS (the launch bounds of the VECTOR / LIGHT kernels): two of
// 512 threads; four of 256 when the body keeps 2 rows per vector in flight (fp64; fp32 with 16+ lanes per row),
// three with 4 rows (fp32, up to 8 lanes per row, and the per-chunk-width kernels of weight-cut plans).
int workgroups_per_cu_by_registers(const Plan& p) {
    if (p.block_threads == kHugeBlock) return 1;
    if (p.block_threads == kWideBlock) return 2;
    if (p.balanced) return p.val_type == MI355_VAL_F64 ? 4 : 3;
    return (p.val_type == MI355_VAL_F64 || p.lanes_per_row >= 16) ? 4 : 3;
}

int long_steps_for(const Plan& p) {
    if (p.knob.long_steps > 0) return p.knob.long_steps;
    // measured on the power-law stand-ins (us, 1 / 2 / 4 / 8 / 16 steps): web-Google 139 / 120 / 99 / 104 / 105,
    // R-MAT-24 light 2 820 at 4 vs 3 200 at 16; uniform plans keep the long chain (their rows rarely need it)
    return p.balanced ? 4 : kLongSteps;
}

int decide_balance(Plan& p) {
    p.balanced = false;
    p.chunk_row = nullptr;
    p.rows_cap = int(p.rows_per_chunk);
    p.n_chunks = p.rows_per_chunk > 0 ? (int64_t(p.n_rows) + p.rows_per_chunk - 1) / p.rows_per_chunk : 0;
    if (p.n_chunks < 1) p.n_chunks = 1;
    const int ev = p.knob.balance;                          // 0 = never, 1 = always, -1 = measure
    if (p.n_rows <= 0 || p.nnz <= 0 || ev == 0) return MI355_SPMV_OK;
    bool want = ev > 0;
    if (!want && p.n_chunks >= 2) {
        std::lock_guard<std::mutex> lock(g_analysis_mutex);
        long long* buf = analysis_buffer();
        if (!buf) { set_error("decide_balance: no device scratch"); return MI355_SPMV_ENOMEM; }
        unsigned long long* d_max = reinterpret_cast<unsigned long long*>(buf + kAnalysisWords - 1);
        hipError_t e = hipMemsetAsync(d_max, 0, sizeof(unsigned long long), nullptr);
        const unsigned g = unsigned(std::min<int64_t>((p.n_chunks + kBlock - 1) / kBlock, 1024));
        if (e == hipSuccess) {
            if (p.off_type == MI355_OFF_I32)
                hipLaunchKernelGGL((chunk_max_kernel<int32_t>), dim3(g), dim3(kBlock), 0, nullptr, p.n_rows,
                                   static_cast<const int32_t*>(p.Ap), p.rows_per_chunk, p.n_chunks, d_max);
            else
                hipLaunchKernelGGL((chunk_max_kernel<int64_t>), dim3(g), dim3(kBlock), 0, nullptr, p.n_rows,
                                   static_cast<const int64_t*>(p.Ap), p.rows_per_chunk, p.n_chunks, d_max);
            e = hipGetLastError();
        }
        unsigned long long h_max = 0;
        if (e == hipSuccess) e = hipMemcpy(&h_max, d_max, sizeof(h_max), hipMemcpyDeviceToHost);   // synchronises
        if (e != hipSuccess) {
            set_error("decide_balance: %s", hipGetErrorString(e));
            return MI355_SPMV_EHIP;
        }
        const double mean = double(p.nnz) / double(p.n_chunks);
        want = double(h_max) > 2.0 * mean + 1024.0;
    }
    if (!want) return MI355_SPMV_OK;
    // chunk weight: twice what a chunk of rows_per_chunk mean rows weighs, rows capped so that the LDS
    // layout (bounds + results of rows_cap rows) stays what a uniform plan of kMaxChunkRows rows takes
    int64_t r0 = p.rows_per_chunk;
    if (r0 > kMaxChunkRows / 2) r0 = kMaxChunkRows / 2;
    if (r0 < 4) r0 = 4;
    p.bal_k = (p.nnz - p.nnz_begin + p.n_rows - 1) / p.n_rows;
    if (p.bal_k < 1) p.bal_k = 1;
    p.bal_q = 2 * p.bal_k * r0;
    const int64_t weight = (p.nnz - p.nnz_begin) + p.bal_k * int64_t(p.n_rows);
    p.n_chunks = (weight + p.bal_q - 1) / p.bal_q;
    if (p.n_chunks < 1) p.n_chunks = 1;
    p.rows_cap = int(2 * r0 + 4);              // Q / k rows, + 3 for the round-down of the boundaries
    p.balanced = true;
    // Whole rounds (as shape_chunks does for equal-row chunks): 862 chunks on 768 workgroup slots are two rounds of
    // full-size chunks, the second one an eighth full; cut the same weight into 2 x 768 smaller chunks instead.
    // Only ever makes chunks smaller, so rows_cap holds.
    if (p.knob.rows_per_chunk <= 0) {
        const int64_t slots = int64_t(kCus) * workgroups_per_cu_by_registers(p);
        const int64_t rounds = (p.n_chunks + slots - 1) / slots;
        if (rounds <= 4 && p.n_chunks > slots / 2 && p.n_chunks % slots != 0) {
            int64_t q = (weight + rounds * slots - 1) / (rounds * slots);
            if (q < 8 * p.bal_k) q = 8 * p.bal_k;
            if (q < p.bal_q) {
                p.bal_q = q;
                p.n_chunks = (weight + q - 1) / q;
            }
        }
    }
    return MI355_SPMV_OK;
}

// ---- giant rows (giant_rows.hpp) ------------------------------------------------------------------
template <typename off_t>
__global__ __launch_bounds__(kBlock) void giant_scan_kernel(int32_t n_rows, const off_t* __restrict__ Ap, int cap,
                                                            int64_t giant_len, long long* out) {   // out[0] = count, then (row, length) pairs
    for (int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x; r < n_rows; r += int64_t(gridDim.x) * kBlock) {
        const int64_t len = int64_t(Ap[r + 1]) - int64_t(Ap[r]);
        if (len > giant_len) {
            const unsigned long long i = atomicAdd(reinterpret_cast<unsigned long long*>(out), 1ull);
            if (i < (unsigned long long)cap) {
                out[1 + 2 * i] = r;
                out[2 + 2 * i] = len;
            }
        }
    }
}

int find_giant_rows(Plan& p) {
    p.n_giant = 0;
    p.n_giant_slices = 0;
    // (a row-block plan does what the whole matrix's plan decided: p.giant_enabled / p.giant_len are inherited)
    if (!p.balanced || (p.is_block ? !p.giant_enabled : p.knob.giant == 0)) { p.giant_enabled = false; return MI355_SPMV_OK; }
    // A row is giant when it alone is more than an eighth of a CU's fair share of the matrix (a hub of 30 K nonzeros in a
    // 4 M-nonzero R-MAT kept ONE workgroup busy for most of the kernel: 75 us against merge's 30), between 4 K and 64 K.
    if (!p.is_block) {
        int64_t fair = ((p.nnz - p.nnz_begin) / (int64_t(kCus) * 8) + 1023) & ~int64_t(1023);
        fair = fair < 4096 ? 4096 : (fair > kGiantRow ? kGiantRow : fair);
        p.giant_len = p.knob.giant_row >= 4096 ? p.knob.giant_row : fair;
    }
    p.giant_enabled = false;
    static_assert(1 + 2 * size_t(kMaxGiantRows) <= kAnalysisWords, "analysis buffer");
    std::lock_guard<std::mutex> lock(g_analysis_mutex);
    long long* buf = analysis_buffer();
    if (!buf) { set_error("find_giant_rows: no device scratch"); return MI355_SPMV_ENOMEM; }
    static thread_local long long h[1 + 2 * kMaxGiantRows];
    auto scan = [&](int64_t giant_len) -> int {          // rows longer than giant_len -> h (count, then (row, length) pairs)
        hipError_t e = hipMemsetAsync(buf, 0, sizeof(long long), nullptr);
        if (e == hipSuccess) {
            const unsigned g = unsigned(std::min<int64_t>((int64_t(p.n_rows) + kBlock - 1) / kBlock, 2048));
            if (p.off_type == MI355_OFF_I32)
                hipLaunchKernelGGL((giant_scan_kernel<int32_t>), dim3(g), dim3(kBlock), 0, nullptr, p.n_rows,
                                   static_cast<const int32_t*>(p.Ap), kMaxGiantRows, giant_len, buf);
            else
                hipLaunchKernelGGL((giant_scan_kernel<int64_t>), dim3(g), dim3(kBlock), 0, nullptr, p.n_rows,
                                   static_cast<const int64_t*>(p.Ap), kMaxGiantRows, giant_len, buf);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);   // synchronises
        if (e != hipSuccess) {
            set_error("find_giant_rows: %s", hipGetErrorString(e));
            return MI355_SPMV_EHIP;
        }
        return MI355_SPMV_OK;
    };
    if (const int st = scan(p.giant_len)) return st;
    // The slices cost two more launches (~6 us).  With the threshold lowered for a small matrix they are only taken when the
    // longest row is on the critical path by more than that: a workgroup walks ~1 K nonzeros of a hub per us, so the hub must
    // outweigh a workgroup slot's share of the whole matrix (nonzeros + mean-row-length per row, over the slots in use) by 10 K+.
    // R-MAT-18 (30 K hub, share 11 K): 75 -> 55 us; R-MAT-20 (69 K, 44 K): 200 -> 173; R-MAT-16 (13 K, 8 K) and the web-Google
    // stand-in lost 6 us each with the slices and keep the default threshold.
    if (!p.is_block && p.knob.giant_row < 4096 && p.giant_len < kGiantRow) {
        long long longest = 0;
        if (h[0] > 0 && h[0] <= kMaxGiantRows)
            for (long long i = 0; i < h[0]; ++i) longest = std::max(longest, h[2 + 2 * i]);
        const int64_t nnz = p.nnz - p.nnz_begin;
        const int64_t slots = std::max<int64_t>(1, std::min<int64_t>(p.n_chunks, int64_t(kCus) * 3));   // (a small matrix has fewer chunks than slots)
        const int64_t share = (nnz + (p.n_rows > 0 ? nnz / p.n_rows : 0) * int64_t(p.n_rows)) / slots;
        if (longest < share + 10240) {
            p.giant_len = kGiantRow;
            if (const int st = scan(p.giant_len)) return st;
        }
    }
    const long long count = h[0];
    if (count > kMaxGiantRows) return MI355_SPMV_OK;   // too many to be "a few dense rows": they stay with their workgroups
    p.giant_enabled = true;                            // (a block of this matrix may hold some even if this one holds none)
    if (count <= 0) return MI355_SPMV_OK;
    const int64_t slice = giant_slice_for(p.giant_len);
    static thread_local std::pair<long long, long long> rows[kMaxGiantRows];
    for (long long i = 0; i < count; ++i) rows[i] = {h[1 + 2 * i], h[2 + 2 * i]};
    std::sort(rows, rows + count);                                // the device appended them in any order
    p.giant_slice_first_host[0] = 0;
    for (long long i = 0; i < count; ++i) {
        p.giant_row_host[i] = int32_t(rows[i].first);
        p.giant_slice_first_host[i + 1] = p.giant_slice_first_host[i] + (rows[i].second + slice - 1) / slice;
    }
    p.n_giant = int(count);
    p.n_giant_slices = p.giant_slice_first_host[count];
    return MI355_SPMV_OK;
}

int build_chunk_table(Plan& p) {
    if (p.n_giant > 0) {
        MI355_HIP_TRY(hipMemcpy(p.giant_row, p.giant_row_host, sizeof(int32_t) * size_t(p.n_giant), hipMemcpyHostToDevice));
        MI355_HIP_TRY(hipMemcpy(p.giant_slice_first, p.giant_slice_first_host, sizeof(int64_t) * size_t(p.n_giant + 1),
                                hipMemcpyHostToDevice));
    }
    if (!p.balanced) return MI355_SPMV_OK;
    const unsigned g = unsigned((p.n_chunks + 1 + kBlock - 1) / kBlock);
    if (p.off_type == MI355_OFF_I32)
        hipLaunchKernelGGL((chunk_table_kernel<int32_t>), dim3(g), dim3(kBlock), 0, nullptr, p.n_rows,
                           static_cast<const int32_t*>(p.Ap), p.bal_k, p.bal_q, p.n_chunks, p.chunk_row,
                           p.is_block ? p.block_weight_off : -p.nnz_begin, p.is_block ? p.block_chunk_begin : int64_t(0));
    else
        hipLaunchKernelGGL((chunk_table_kernel<int64_t>), dim3(g), dim3(kBlock), 0, nullptr, p.n_rows,
                           static_cast<const int64_t*>(p.Ap), p.bal_k, p.bal_q, p.n_chunks, p.chunk_row,
                           p.is_block ? p.block_weight_off : -p.nnz_begin, p.is_block ? p.block_chunk_begin : int64_t(0));
    MI355_HIP_TRY(hipGetLastError());
    MI355_HIP_TRY(hipStreamSynchronize(nullptr));   // the first execute may come on any stream
    return MI355_SPMV_OK;
}

// ---- partition into row blocks (multi-GPU; include/mi355_spmv.h "row-block plans") -----------------
// Boundary b of `parts` blocks: the first UNIT whose first row starts at or after nonzero b * nnz / parts.
// A unit is a chunk of the plan (VECTOR / LIGHT: a block then owns whole chunks, so it can reproduce the whole
// plan's per-chunk decisions) or 4 rows (MERGE: any multiple of 4 rows keeps the 16-byte phase rule simple).
template <typename off_t>
__global__ __launch_bounds__(kBlock) void partition_kernel(int32_t n_rows, const off_t* __restrict__ Ap,
                                                           const int32_t* __restrict__ table, int64_t rows_per_unit,
                                                           int64_t n_units, int parts, long long* out) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b > parts) return;
    auto unit_row = [&](int64_t u) -> int64_t {
        if (u >= n_units) return n_rows;
        return table ? int64_t(table[u]) : min(u * rows_per_unit, int64_t(n_rows));
    };
    const int64_t first = int64_t(Ap[0]), last = int64_t(Ap[n_rows]);
    int64_t u = 0;
    if (b == parts) u = n_units;
    else if (b > 0) {
        const int64_t target = first + int64_t((__int128)(last - first) * b / parts);
        int64_t lo = 0, hi = n_units;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (int64_t(Ap[unit_row(mid)]) >= target) hi = mid;
            else lo = mid + 1;
        }
        u = lo;
    }
    const int64_t r = unit_row(u);
    out[3 * b + 0] = r;
    out[3 * b + 1] = u;
    out[3 * b + 2] = int64_t(Ap[r]);
}

int partition_plan(const Plan& p, int parts, int64_t* row_cuts, int64_t* chunk_cuts, int64_t* nnz_cuts) {
    if (p.n_rows <= 0) {
        for (int b = 0; b <= parts; ++b) { row_cuts[b] = 0; chunk_cuts[b] = 0; nnz_cuts[b] = p.nnz_begin; }
        return MI355_SPMV_OK;
    }
    if (size_t(parts + 1) * 3 > kAnalysisWords) { set_error("plan_partition: too many parts"); return MI355_SPMV_EINVAL; }
    const bool chunked = p.kind != MI355_KIND_MERGE;
    const int32_t* table = (chunked && p.balanced) ? p.chunk_row : nullptr;
    const int64_t rows_per_unit = chunked ? (p.rows_per_chunk > 0 ? p.rows_per_chunk : 4) : 4;
    const int64_t n_units = table ? p.n_chunks : (int64_t(p.n_rows) + rows_per_unit - 1) / rows_per_unit;
    std::lock_guard<std::mutex> lock(g_analysis_mutex);
    long long* buf = analysis_buffer();
    if (!buf) { set_error("plan_partition: no device scratch"); return MI355_SPMV_ENOMEM; }
    const unsigned g = unsigned((parts + 1 + kBlock - 1) / kBlock);
    if (p.off_type == MI355_OFF_I32)
        hipLaunchKernelGGL((partition_kernel<int32_t>), dim3(g), dim3(kBlock), 0, nullptr, p.n_rows,
                           static_cast<const int32_t*>(p.Ap), table, rows_per_unit, n_units, parts, buf);
    else
        hipLaunchKernelGGL((partition_kernel<int64_t>), dim3(g), dim3(kBlock), 0, nullptr, p.n_rows,
                           static_cast<const int64_t*>(p.Ap), table, rows_per_unit, n_units, parts, buf);
    hipError_t e = hipGetLastError();
    long long* h = new (std::nothrow) long long[size_t(parts + 1) * 3];
    if (!h) e = hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMemcpy(h, buf, size_t(parts + 1) * 3 * sizeof(long long), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        delete[] h;
        set_error("plan_partition: %s", hipGetErrorString(e));
        return MI355_SPMV_EHIP;
    }
    for (int b = 0; b <= parts; ++b) {
        row_cuts[b] = h[3 * b];
        chunk_cuts[b] = chunked ? h[3 * b + 1] : 0;
        nnz_cuts[b] = h[3 * b + 2];
        if (b > 0 && row_cuts[b] < row_cuts[b - 1]) {   // (monotone by construction; keep it so whatever Ap holds)
            row_cuts[b] = row_cuts[b - 1]; chunk_cuts[b] = chunk_cuts[b - 1]; nnz_cuts[b] = nnz_cuts[b - 1];
        }
    }
    delete[] h;
    return MI355_SPMV_OK;
}

// Cluster the sampled offsets into bands: a gap wider than the rows a workgroup owns starts a
// new band (splitting costs `rows` extra columns per band, keeping the gap costs the gap).
// Returns the columns a workgroup would have to hold: sum of (band width + rows).
static int64_t cluster_bands(Plan& p, int64_t rows) {
    p.n_seg = 0;
    if (p.probe_n == 0) return 0;
    if (!p.probe_sorted) {
        std::sort(p.probe_off, p.probe_off + p.probe_n);
        p.probe_sorted = true;
    }
    int64_t lo[64], hi[64];
    int n = 0;
    lo[0] = hi[0] = p.probe_off[0];
    for (int i = 1; i < p.probe_n; ++i) {
        if (p.probe_off[i] - hi[n] > rows && n + 1 < 64) {
            ++n;
            lo[n] = p.probe_off[i];
        }
        hi[n] = p.probe_off[i];
    }
    ++n;
    while (n > kMaxSegments) {   // merge across the narrowest gap
        int best = 0;
        for (int i = 1; i + 1 < n; ++i)
            if (lo[i + 1] - hi[i] < lo[best + 1] - hi[best]) best = i;
        hi[best] = hi[best + 1];
        for (int i = best + 1; i + 1 < n; ++i) { lo[i] = lo[i + 1]; hi[i] = hi[i + 1]; }
        --n;
    }
    int64_t need = 0;
    for (int i = 0; i < n; ++i) {
        // the samples may miss a band's edges: widen each by 1/16 of its width + 8 columns
        const int64_t pad = (hi[i] - lo[i]) / 16 + 8;
        p.seg_lo[i] = lo[i] - pad;
        p.seg_hi[i] = hi[i] + pad;
        need += (p.seg_hi[i] - p.seg_lo[i] + 1) + rows;
    }
    p.n_seg = n;
    return need;
}

// Window size (elements) for a workgroup that owns `rows_per_workgroup` consecutive rows: what the
// sampled band plus those rows needs, up to the plan's LDS budget (p.window_bytes), when that is within
// 1.5x of the budget (the kernels centre a too-small window on the span); else several narrow bands; else
// none.  MI355_SPMV_WINDOW=0|1 forces the choice (tuning / tests).
int pick_window_elems(Plan& p, int64_t rows_per_workgroup) {
    const int val_bytes = p.val_type == MI355_VAL_F64 ? 8 : 4;
    const int cap = (p.window_bytes > 0 ? p.window_bytes : kWindowBytes) / val_bytes;
    p.window_from_band = false;
    if (p.knob.window >= 0) return p.knob.window ? cap : 0;
    if (!p.probe_ok) return 0;
    const int64_t span = (p.band_hi - p.band_lo + 1) + rows_per_workgroup;
    // the band (plus the chunk's rows) all but fits: no need to sample every chunk
    p.window_from_band = p.knob.window_from_band >= 0 ? p.knob.window_from_band != 0 : span <= int64_t(cap) * 9 / 8;
    p.n_seg = 0;
    if (span <= int64_t(cap) * 3 / 2) {
        // LDS is occupancy: take what the span needs (a sampled window is widened by an eighth + 64 per side)
        const int64_t want = p.window_from_band ? span + 8 : span + 2 * (span / 8 + 64) + 8;
        const int64_t elems = (std::min<int64_t>(want, cap) + 3) & ~int64_t(3);
        return int(std::min<int64_t>(elems, cap));
    }
    // one window cannot hold the band: do a few narrow ones?  (MI355_SPMV_SEGMENTS=0 disables)
    if (p.knob.segments != 0) {
        const int64_t need = cluster_bands(p, rows_per_workgroup);
        // up to 1.25x: the tail of the last band is cut and falls back to global loads; the
        // row-based kinds then shrink their chunk so that everything fits (segment_rows_fit)
        if (p.n_seg >= 2 && need <= int64_t(cap) * 5 / 4) return cap;
    }
    p.n_seg = 0;
    return 0;
}

// LDS a workgroup may take: 3 workgroups of 256 threads per CU (what the 36 KB window was sized for),
// or 2 of 512 threads (up to 64 KB each, ~1 KB of it static).
static int window_budget(const Plan& p, int block_threads, int64_t rows) {
    if (block_threads == kBlock) return kWindowBytes;
    const size_t val_bytes = p.val_type == MI355_VAL_F64 ? 8 : 4;
    const int64_t avail = 63 * 1024 - int64_t(chunk_lds_bytes(0, int(rows), val_bytes));   // <= 64 KB per launch
    return int(avail < 0 ? 0 : avail);
}

// VECTOR / LIGHT: workgroup size, rows per chunk and the window of x.  R = rows a vector keeps in flight,
// div = the kind's chunk divisor (LIGHT's tuning knob).
// Workgroup size: 512 threads own a chunk twice as long (64 K nonzeros) — the window of x is staged half as
// often per row and 2 x 8 waves sit on a CU instead of 3 x 4 (LDS-bound either way): 190 -> 178 us on the
// S32-band target (LIGHT: 199 -> 195 us once its kernel is held to 128 VGPRs; at 151 only one such workgroup
// fits a CU and it lost, 237 us).
void shape_chunks(Plan& p, int R, int64_t div, bool allow_wide, bool allow_huge) {
    if (p.val_type == MI355_VAL_F32 && p.lanes_per_row >= 16 && R > 2) R = 2;   // (the kernels' rule: launch_*_window, wide_r)
    int64_t rows_before_rounding = 0;        // of the last shape(): the chunk before it was shrunk to whole rounds
    auto shape = [&](int block_threads, int64_t nnz_per_chunk, int64_t rows_wanted = 0) {
        p.block_threads = block_threads;
        const int64_t pass = int64_t(block_threads / p.lanes_per_row) * R;
        int64_t rows = pick_rows_per_chunk(p.nnz, p.n_rows, p.lanes_per_row, R, block_threads, nnz_per_chunk,
                                           workgroups_per_cu_by_registers(p));
        if (div > 1) rows = (rows / div + pass - 1) / pass * pass;
        if (rows_wanted > 0) rows = std::min<int64_t>((rows_wanted + 3) & ~int64_t(3), kMaxChunkRows);
        if (p.knob.rows_per_chunk > 0) {
            int64_t r = p.knob.rows_per_chunk;
            r = (r + pass - 1) / pass * pass;
            if (r >= pass && r <= kMaxChunkRows) rows = r;
        }
        if (rows < pass) rows = pass;
        p.rows_per_chunk = rows;
        rows_before_rounding = rows;
        p.window_bytes = window_budget(p, block_threads, rows);
        p.window_elems = pick_window_elems(p, rows);
        if (const int64_t fit = segment_rows_fit(p)) {   // several bands: shrink the chunk until they all fit
            if (fit < p.rows_per_chunk && fit >= pass) p.rows_per_chunk = fit / pass * pass;
            // ... and take only the LDS the bands need with that many rows (LDS is occupancy)
            int64_t need = 0;
            for (int i = 0; i < p.n_seg; ++i) need += p.seg_hi[i] - p.seg_lo[i] + 1 + 4 + p.rows_per_chunk;
            need = (need + 3) & ~int64_t(3);
            if (need < p.window_elems) p.window_elems = int(need);
        }
        // Whole rounds: with a few chunks per workgroup slot, a last round that is a third full costs a quarter
        // of the kernel (2^20 rows: 1 024 chunks on 768 slots).  Shrink the chunk so that the count is a multiple
        // of the slots the plan's LDS and registers leave on the chip.
        if (p.knob.rows_per_chunk <= 0 && p.n_seg < 2) {
            const size_t val_bytes = p.val_type == MI355_VAL_F64 ? 8 : 4;
            const size_t lds = chunk_lds_bytes(p.window_elems, int(p.rows_per_chunk), val_bytes) + 1024;
            int64_t per_cu = int64_t(160 * 1024 / lds);
            const int64_t reg_bound = workgroups_per_cu_by_registers(p);
            if (per_cu > reg_bound) per_cu = reg_bound;
            const int64_t slots = int64_t(kCus) * (per_cu > 0 ? per_cu : 1);
            const int64_t n_chunks = (int64_t(p.n_rows) + p.rows_per_chunk - 1) / p.rows_per_chunk;
            const int64_t rounds = (n_chunks + slots - 1) / slots;
            if (n_chunks > slots / 2 && rounds < 8 && n_chunks % slots != 0) {
                int64_t r = (int64_t(p.n_rows) + rounds * slots - 1) / (rounds * slots);
                r = (r + 3) & ~int64_t(3);
                if (r >= pass && r < p.rows_per_chunk) {
                    const int64_t shrink = p.rows_per_chunk - r;
                    p.rows_per_chunk = r;
                    if (p.window_elems > shrink && p.window_from_band) p.window_elems -= int(shrink & ~int64_t(3));
                }
            }
        }
    };
    const bool force = p.knob.block > 0;
    if (allow_wide && (!force || p.knob.block == kWideBlock)) {   // (the knob cannot force a workgroup size the kind has no kernel for)
        shape(kWideBlock, 65536);   // (3 072- and 3 584-row chunks with 79 KB of LDS measured worse: 189-194 vs 180 us)
        const int64_t mean = p.n_rows > 0 ? (p.nnz + p.n_rows - 1) / p.n_rows : 1;
        // keep it when (a) the ">= 4 chunks per CU" rule left the chunk long and (b) one window placed from the
        // band serves it (with 64-bit offsets / fp64 / several bands the 64 KB a launch may take is better spent
        // on three workgroups of 256: C4 stand-in 660 us vs 824 us)
        // ("long" is judged before the chunk was shrunk to whole rounds: rows of 20-24 nonzeros reach the 2 048-row cap
        // at 41-49 K nonzeros, were shrunk a little, then failed the 48 K test and fell to 256 threads — 302 us against
        // 245 / 201 for 18 / 26 per row on either side)
        const bool long_chunk = p.rows_per_chunk * mean >= 49152 || rows_before_rounding >= kMaxChunkRows;
        // (a forced 512 is honoured unless the window needs several bands: those kernels exist for 256 threads only)
        if ((force && p.n_seg < 2) || (long_chunk && p.window_elems > 0 && p.n_seg < 2 && p.window_from_band)) return;
        // A small matrix whose chunks all run at once — one round of the chip — also keeps the 512 threads when its rows
        // are long (16+ lanes per row: the R = 2 bodies): the same rows by half as many workgroups, i.e. half the
        // prologues (bounds, window, barriers) in a kernel that is nothing but its prologue and four groups of rows.
        // cant stand-in: 10.7-10.8 us against 10.9-11.2 with 976 workgroups of 256 (rounds 2 and 3, three boxes).
        // ... and whatever the row length when every workgroup still gets TWO groups of rows or more to pipeline: two
        // workgroups of 512 per CU — or, where that leaves them a single group each, one per CU with twice the rows.
        // S32-band shape, fp32, T = 8 (us, rule / the 256-thread plan with ~3 workgroups per CU it replaces;
        // scripts/probes/mid_size_knobs.sh): 2^17 rows 10.6 / 11.6, 2^18 15.9 / 19.0, 2^19 27.4 / 29.2 — the
        // mid-size matrices (35-140 MB) where a kernel is one round of the chip.
        if (!force && p.knob.rows_per_chunk <= 0 && div <= 1 && p.window_elems > 0 && p.n_seg < 2 && p.window_from_band) {
            const int64_t pass = int64_t(kWideBlock / p.lanes_per_row) * R;
            const int64_t n_chunks = p.rows_per_chunk > 0 ? (int64_t(p.n_rows) + p.rows_per_chunk - 1) / p.rows_per_chunk : 0;
            if (n_chunks >= kCus && n_chunks <= int64_t(kCus) * 2) {
                if (p.lanes_per_row >= 16 || p.rows_per_chunk >= 2 * pass) return;
                const int64_t twice = (int64_t(p.n_rows) + kCus - 1) / kCus;
                if (twice >= 2 * pass && twice <= kMaxChunkRows) {
                    shape(kWideBlock, 65536, twice);
                    if (p.window_elems > 0 && p.n_seg < 2 && p.window_from_band) return;
                }
            }
        }
    }
    shape(kBlock, 32768);
    // (several bands with two 512-thread workgroups of 78 KB per CU and 1 664-row chunks: measured 722 vs 700 us on the
    // C4 stand-in — the 256-thread plan stays)
    // A band too wide for either budget (fp64 halves what 36 KB holds: the S32-band shape in fp64 ran on plain
    // gathers, 704 us): gfx950 lets a workgroup take more than the default 64 KB of LDS, and two workgroups of
    // 512 threads with ~78 KB each still fit a CU.  The chunk is then as long as the band leaves room for.
    if (allow_wide && !force && p.probe_ok && !(p.window_elems > 0 && p.n_seg < 2 && p.window_from_band) &&
        p.knob.window < 0 && p.knob.rows_per_chunk <= 0) {
        const int64_t off_bytes = 4, val_bytes = p.val_type == MI355_VAL_F64 ? 8 : 4;   // (bounds are chunk-relative int32 in LDS)
        const int64_t band = p.band_hi - p.band_lo + 1;
        const int64_t pass = int64_t(kWideBlock / p.lanes_per_row) * R;
        // val (band + rows + 8) + off (rows + 1) + val rows + rows / 8 <= 78 KB
        int64_t rows = (78 * 1024 - val_bytes * (band + 8) - off_bytes) * 8 / (8 * (2 * val_bytes + off_bytes) + 1);
        rows = rows / pass * pass;
        if (rows > kMaxChunkRows) rows = kMaxChunkRows / pass * pass;
        // (a matrix too small for a full round of such chunks takes shorter ones — the band still fits: 2^18 rows x 64
        // in fp64 fell to the 1 024-thread plan at 3.9 TB/s for want of 512 chunks of 704 rows)
        if (rows > 0 && (p.n_rows + rows - 1) / rows < int64_t(kCus) * 2) {
            const int64_t fewer = ((int64_t(p.n_rows) + int64_t(kCus) * 2 - 1) / (int64_t(kCus) * 2) + pass - 1) / pass * pass;
            if (fewer < rows) rows = fewer;
        }
        const int64_t n_chunks = rows > 0 ? (p.n_rows + rows - 1) / rows : 0;
        if (band > 0 && rows >= pass && rows >= 256 && n_chunks >= int64_t(kCus) * 2 - 8) {   // (>= one full round of the chip)
            const Plan saved = p;
            p.block_threads = kWideBlock;
            p.rows_per_chunk = rows;
            p.window_bytes = int(val_bytes * (band + rows + 8));
            p.window_elems = pick_window_elems(p, rows);
            if (!(p.window_elems > 0 && p.n_seg < 2 && p.window_from_band)) p = saved;   // (several bands etc.: keep the 256-thread plan)
        }
    }
    // Still no window: the band is wider than two workgroups per CU can hold (round 1: every kind fell to the
    // plain-gather rate, 1.6-2.5 TB/s, once the band passed ~17 K columns in fp32 / ~9 K in fp64).  ONE workgroup
    // of 1 024 threads per CU can take ~155 of the CU's 160 KB: twice the band.  Its prologue is not hidden by a
    // neighbour, so this is only worth it where the alternative is the plain gather.
    if (allow_huge && !force && p.probe_ok && p.window_elems == 0 && p.knob.window < 0 && p.knob.rows_per_chunk <= 0) {
        const int64_t val_bytes = p.val_type == MI355_VAL_F64 ? 8 : 4;
        const int64_t band = p.band_hi - p.band_lo + 1;
        const int64_t pass = int64_t(kHugeBlock / p.lanes_per_row) * R;
        // val (band + rows + 8) + 4 (rows + 1) + val rows + rows / 8 <= 155 KB
        int64_t rows = (155 * 1024 - val_bytes * (band + 8) - 4) * 8 / (8 * (2 * val_bytes + 4) + 1);
        rows = rows / pass * pass;
        if (rows > kMaxChunkRows) rows = kMaxChunkRows / pass * pass;
        const int64_t n_chunks = rows > 0 ? (p.n_rows + rows - 1) / rows : 0;
        if (band > 0 && rows >= pass && rows >= 512 && n_chunks >= int64_t(kCus) * 2) {
            const Plan saved = p;
            p.block_threads = kHugeBlock;
            p.rows_per_chunk = rows;
            p.window_bytes = int(val_bytes * (band + rows + 8));
            p.window_elems = pick_window_elems(p, rows);
            if (!(p.window_elems > 0 && p.n_seg < 2 && p.window_from_band)) p = saved;
        }
    }
}

// VECTOR / LIGHT, after decide_balance said "equal-row chunks" and no window of x was found: the band is wider than one
// CU's LDS — let the window sweep it (chunk_rows_sweep).  A chunk is one group of rows of the 1 024-thread
// workgroup, every row one step of its vector (T from the longest row the probe saw), and the chunk's band is
// staged in `passes` windows.  Worth it while the staged bytes stay well below the line fills the same
// nonzeros cost as plain gathers (128 bytes each, some of them L1 hits).  MI355_SPMV_SWEEP=0|1 forces the choice.
bool shape_sweep(Plan& p) {
    p.sweep = false;
    if (p.balanced || p.knob.block > 0 || !p.probe_ok || p.window_elems != 0 || p.knob.window >= 0 ||
        p.knob.rows_per_chunk > 0 || p.knob.sweep == 0 || p.probe_len_max <= 0 || p.probe_len_max > 4 * kWave)
        return false;
    const int64_t val_bytes = p.val_type == MI355_VAL_F64 ? 8 : 4;
    const int64_t band = p.band_hi - p.band_lo + 1;
    int t = p.lanes_per_row;
    while (t < kWave && 4 * t < p.probe_len_max) t *= 2;
    const int64_t rows = int64_t(kHugeBlock / t) * sweep_rows_for(p.val_type, t);
    const int64_t fixed = int64_t(chunk_lds_bytes(0, int(rows), size_t(val_bytes)));
    const int64_t cap = sweep_window_cap(val_bytes, fixed);
    const int64_t span = band + rows + 8;
    const int64_t passes = cap > 0 ? (span + cap - 1) / cap : 0;
    const int64_t mean = p.n_rows > 0 ? (p.nnz - p.nnz_begin) / p.n_rows : 0;
    const int64_t n_chunks = (p.n_rows + rows - 1) / rows;
    const bool pays = span * val_bytes <= 64 * mean * rows;     // staged bytes vs half the gathers' line fills
    if (!(band > 0 && passes >= 1 && passes <= 16 && n_chunks >= int64_t(kCus) * 2 && (pays || p.knob.sweep == 1))) return false;
    p.sweep = true;
    p.lanes_per_row = t;
    p.block_threads = kHugeBlock;
    p.rows_per_chunk = rows;
    p.rows_cap = int(rows);
    p.n_chunks = n_chunks;
    p.grid_blocks = n_chunks;
    p.n_tiles = n_chunks;
    p.window_bytes = int(cap * val_bytes);
    p.window_elems = int(cap);
    p.window_from_band = true;
    p.n_seg = 0;
    snprintf(p.main_kernel, sizeof(p.main_kernel), p.kind == MI355_KIND_LIGHT ? "light_rows_sweep_kernel" : "csr_vector_sweep_kernel");
    return true;
}

// Rows per workgroup for which the plan's bands fit the window exactly:
// sum_k (width_k + rows) <= cap.  0 when there are no segments.
int64_t segment_rows_fit(const Plan& p) {
    if (p.n_seg < 2 || p.window_elems <= 0) return 0;
    int64_t width = 0;
    for (int i = 0; i < p.n_seg; ++i) width += p.seg_hi[i] - p.seg_lo[i] + 1 + 4;   // +4: 16-byte rounding
    const int64_t fit = (int64_t(p.window_elems) - width) / p.n_seg;
    return fit > 0 ? fit : 0;
}

}  // namespace mi355
