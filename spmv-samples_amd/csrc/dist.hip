// dist.hip — mi355_spmv_dist_*: one CSR SpMV across the GPUs of a node (north star).
//
// The reference is single-device (main.cu:53 cudaSetDevice(USED_DEVICE), include/common.cuh:8) and has no
// collective anywhere.  Rows of y = A x are independent (cpu_navie.hpp:9-16), so:
//   * the rows are cut into contiguous nnz-balanced BLOCKS on chunk boundaries of the whole matrix's plan
//     (mi355_spmv_plan_partition); a GPU owns `sub_blocks` consecutive blocks; every block has its own plan that
//     inherits the whole plan's launch shape (mi355_spmv_plan_create_block), so VECTOR / LIGHT results are the
//     one-GPU results bit for bit;
//   * x is replicated; every GPU holds a full-length y and computes its blocks straight into their
//     displacements (no staging copy);
//   * allgatherv(y): RCCL has none.  Three ways to make one are built behind the same API (SURVEY §5 last row:
//     a ring per root costs ~7 x the slice over ONE xGMI link, a direct exchange uses the 7 links at once), per
//     sub-block s on a dedicated communication stream per GPU that waits on the event recorded after the block's
//     kernels, so that sub-block s travels over xGMI while sub-block s + 1 is computed:
//       BCAST      one ncclGroup of in-place ncclBroadcasts (root = owner, same pointer on every GPU);
//       SENDRECV   one ncclGroup of ncclSend / ncclRecv: every GPU sends its block to every peer;
//       ALLGATHER  one ncclAllGather, in place in y when the blocks are equal and adjacent, else through a
//                  staging buffer padded to the largest block (pack kernel, collective, unpack kernel).
//     With more than one GPU, create TIMES them (a few exchanges each on scratch buffers, maximum over ranks by
//     ncclAllReduce) and keeps the fastest; MI355_DIST_EXCHANGE / mi355_spmv_dist_set_exchange override.
// Two ways to drive it: LOCAL (one process, all GPUs: ncclCommInitAll) and RANK (one process per GPU:
// ncclCommInitRank with a caller-distributed unique id).  With one GPU nothing of RCCL is touched and an
// execute is the blocks' plain executes on the caller's stream.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on first multi-GPU create): libmi355spmv.so keeps the HIP
// runtime as its only link dependency, and a process that already carries an RCCL (torch's) shares it.
// MI355_SPMV_RCCL_LIB names another library with the same entry points: the test suite's emulation
// (tests/cpp/fake_rccl.cpp, several "GPUs" on the one device of a test box — with MI355_DIST_SHARED_DEVICE=1 a
// device may then be listed twice), which is how the N > 1 schedule runs where no second GPU exists.

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only; no symbol of it is linked

#include <chrono>
#include <mutex>
#include <new>
#include <vector>

#include "common.hpp"

namespace mi355 {

// ---- RCCL, bound on first use ---------------------------------------------------------------------
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;
static std::mutex g_rccl_mutex;
static bool g_rccl_tried = false;
static char g_rccl_bound[200] = "";    // the MI355_SPMV_RCCL_LIB the table was bound under
static char g_rccl_error[256] = "";

static void rccl_bind(const Knobs& k) {
    RcclApi api;
    g_rccl_error[0] = 0;
    if (k.rccl_lib[0]) {
        api.handle = dlopen(k.rccl_lib, RTLD_NOW | RTLD_LOCAL);
    } else {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
    }
    if (!api.handle) {
        const char* why = dlerror();
        snprintf(g_rccl_error, sizeof(g_rccl_error), "mi355_spmv_dist: %s not found (%s)",
                 k.rccl_lib[0] ? k.rccl_lib : "librccl.so.1", why ? why : "?");
        g_rccl = RcclApi();
        return;
    }
    bool ok = true;
    const char* missing = "";
    auto bind = [&](const char* sym) -> void* {
        void* f = dlsym(api.handle, sym);
        if (!f) { ok = false; missing = sym; }
        return f;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(bind("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(bind("ncclCommInitRank"));
    api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(bind("ncclCommInitAll"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(bind("ncclCommDestroy"));
    api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(bind("ncclBroadcast"));
    api.Send = reinterpret_cast<decltype(api.Send)>(bind("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(bind("ncclRecv"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(bind("ncclAllGather"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(bind("ncclAllReduce"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(bind("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(bind("ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(bind("ncclGetErrorString"));
    if (!ok) {
        snprintf(g_rccl_error, sizeof(g_rccl_error), "mi355_spmv_dist: the RCCL library lacks %s", missing);
        dlclose(api.handle);
        api = RcclApi();
    }
    g_rccl = api;
}

// One bind per process and library name, under a lock (the one-shot tests create handles from four threads).  A
// changed MI355_SPMV_RCCL_LIB (after mi355_spmv_knobs_reload: the test suite switching to its emulation) binds again;
// the previous library stays loaded — handles made under it must be destroyed before the switch.
static RcclApi* rccl_api() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    const Knobs k = knobs();
    if (!g_rccl_tried || strcmp(g_rccl_bound, k.rccl_lib) != 0) {
        rccl_bind(k);
        g_rccl_tried = true;
        snprintf(g_rccl_bound, sizeof(g_rccl_bound), "%s", k.rccl_lib);
    }
    if (!g_rccl.handle) {
        set_error("%s", g_rccl_error);
        return nullptr;
    }
    return &g_rccl;
}

#define MI355_RCCL_TRY(api, expr)                                                               \
    do {                                                                                        \
        ncclResult_t _r = (expr);                                                               \
        if (_r != ncclSuccess) {                                                                \
            ::mi355::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, (api)->GetErrorString(_r)); \
            return MI355_SPMV_EHIP;                                                             \
        }                                                                                       \
    } while (0)

// ncclGroupStart ... ncclGroupEnd: a call that fails in between must still close the group (an open group makes
// every later RCCL call of the thread part of it)
struct RcclGroup {
    RcclApi* api;
    bool open = false;
    explicit RcclGroup(RcclApi* a) : api(a) {}
    ncclResult_t start() { const ncclResult_t r = api->GroupStart(); open = r == ncclSuccess; return r; }
    ncclResult_t end() { open = false; return api->GroupEnd(); }
    ~RcclGroup() { if (open) (void)api->GroupEnd(); }
};

// out[i] = Ap[i] - base for i in [0, n]: the offsets of a row block, relative to the 16-byte-aligned element
// its view of Aj / Ax starts at (so out[0] is the block's phase, 0..3)
template <typename off_t>
__global__ __launch_bounds__(kBlock) void rebase_kernel(const off_t* __restrict__ Ap, int64_t n_plus_1, int64_t base,
                                                        off_t* __restrict__ out) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n_plus_1; i += int64_t(gridDim.x) * kBlock)
        out[i] = off_t(int64_t(Ap[i]) - base);
}

// ALLGATHER through staging: up to 64 pieces of y copied to / from the padded staging buffer by one launch
// (blockIdx.y = piece).  Offsets and counts in 4-byte words; 16-byte copies when `vec` (every base 16-byte aligned,
// which row cuts at multiples of 4 rows give), the last words of a piece one by one.
constexpr int kMaxPieces = 64;
struct Pieces {
    int n;
    int64_t dst[kMaxPieces], src[kMaxPieces], words[kMaxPieces];
};
__global__ __launch_bounds__(kBlock) void pieces_copy_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src,
                                                             const Pieces pc, int vec) {
    const int p = blockIdx.y;
    if (p >= pc.n) return;
    uint32_t* const d = dst + pc.dst[p];
    const uint32_t* const s = src + pc.src[p];
    const int64_t words = pc.words[p];
    const int64_t tid = int64_t(blockIdx.x) * kBlock + threadIdx.x, step = int64_t(gridDim.x) * kBlock;
    int64_t done = 0;
    if (vec) {
        const int64_t groups = words / 4;
        for (int64_t g = tid; g < groups; g += step)
            reinterpret_cast<uint4*>(d)[g] = reinterpret_cast<const uint4*>(s)[g];
        done = groups * 4;
    }
    for (int64_t i = done + tid; i < words; i += step) d[i] = s[i];
}

// fingerprint of the structure arrays (mi355_spmv_dist_structure_changed): a sum of mixed words over every
// offset and over a strided sample of Aj; out[0] += by atomics (the order of a sum of integers does not matter)
__device__ __forceinline__ unsigned long long mix64(unsigned long long v) {
    v ^= v >> 33; v *= 0xff51afd7ed558ccdull; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ull; v ^= v >> 33;
    return v;
}
template <typename off_t>
__global__ __launch_bounds__(kBlock) void fingerprint_kernel(const off_t* __restrict__ Ap, int64_t n_plus_1,
                                                             const int32_t* __restrict__ Aj, int64_t nnz, int64_t aj_stride,
                                                             unsigned long long* out) {
    unsigned long long acc = 0;
    const int64_t tid = int64_t(blockIdx.x) * kBlock + threadIdx.x, step = int64_t(gridDim.x) * kBlock;
    for (int64_t i = tid; i < n_plus_1; i += step) acc += mix64((unsigned long long)(Ap[i]) + 0x9e3779b97f4a7c15ull * (unsigned long long)(i + 1));
    for (int64_t k = tid * aj_stride; k < nnz; k += step * aj_stride) acc += mix64((unsigned long long)(unsigned(Aj[k])) ^ (0xd6e8feb86659fd93ull * (unsigned long long)(k + 1)));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && acc) atomicAdd(out, acc);
}

}  // namespace mi355

using namespace mi355;

namespace {

struct Part {                 // one row block and its plan
    int dev = 0;              // index into mi355_spmv_dist::devs
    int64_t row_begin = 0, n_rows = 0;
    int64_t elem_lo = 0;      // element of the SOURCE arrays its view starts at (multiple of 4)
    int64_t nnz_end = 0;      // END offset inside the view (= elements the view must hold)
    void* Ap = nullptr;       // owned, on the block's device
    int32_t* Aj_own = nullptr;   // owned copy (the other GPUs), else null: view of the source Aj
    void* Ax_own = nullptr;      // owned copy (the other GPUs), else null: view of the caller's Ax
    mi355_spmv_plan* plan = nullptr;
};

struct Dev {
    int device = 0;           // HIP ordinal
    hipStream_t compute = nullptr, comm = nullptr;
    hipStream_t side = nullptr;      // odd sub-blocks run here: the ramp of block s + 1 overlaps the drain of block s
    hipEvent_t start = nullptr, done = nullptr, side_done = nullptr;
    std::vector<hipEvent_t> part_done;
    void* x = nullptr;        // the other GPUs (LOCAL mode): replicated x / full-length y, owned
    void* y = nullptr;
    void* stage = nullptr;    // ALLGATHER through staging: world * max_block_rows values
    ncclComm_t nccl = nullptr;
};

}  // namespace

struct mi355_spmv_dist {
    bool local_mode = true;
    int kind = 0, off_type = 0, val_type = 0;
    int64_t n_rows = 0;       // whole matrix
    int32_t n_cols = 0;
    int64_t nnz = 0;          // whole matrix (LOCAL mode)
    int world = 1;            // GPUs taking part (LOCAL: devices of this process; RANK: ranks)
    int rank = 0;             // RANK mode
    int sub_blocks = 1;       // blocks per GPU
    int home = 0;             // LOCAL: index in devs of the device holding the caller's arrays
    const int32_t* Aj_src = nullptr;
    std::vector<int64_t> row_cuts, chunk_cuts, nnz_cuts;   // all blocks of the matrix, world * sub_blocks + 1
    std::vector<Part> parts;  // this process's blocks, in order (device-major)
    std::vector<Dev> devs;    // this process's GPUs
    // LOCAL mode with several devices: what the last scatter_values / replicate_x were given (the home
    // device's blocks read the caller's arrays in place)
    const void* Ax_home = nullptr;
    const void* x_home = nullptr;
    // exchange
    int exchange = MI355_DIST_EXCHANGE_BCAST;
    bool auto_picked = false;
    float trial_us[MI355_DIST_EXCHANGE_COUNT] = {0, 0, 0, 0};
    int64_t max_block_rows = 0;              // padded count of ALLGATHER through staging (multiple of 4)
    std::vector<char> gather_in_place;       // per sub-block: its blocks are equal and adjacent
    // fingerprint of Ap / Aj at create (LOCAL mode; structure_changed)
    unsigned long long fingerprint = 0;
    unsigned long long* fp_dev = nullptr;    // one word on the home device
};

namespace {

size_t off_bytes(const mi355_spmv_dist& d) { return d.off_type == MI355_OFF_I64 ? 8 : 4; }
size_t val_bytes(const mi355_spmv_dist& d) { return d.val_type == MI355_VAL_F64 ? 8 : 4; }
int rank_of(const mi355_spmv_dist& d, int dev_index) { return d.local_mode ? dev_index : d.rank; }
int64_t block_rows(const mi355_spmv_dist& d, int g) { return d.row_cuts[size_t(g) + 1] - d.row_cuts[size_t(g)]; }

struct DeviceGuard {          // restore the caller's current device on every exit path
    int saved = 0;
    DeviceGuard() { (void)hipGetDevice(&saved); }
    ~DeviceGuard() { (void)hipSetDevice(saved); }
};

int destroy_impl(mi355_spmv_dist* d) {
    if (!d) return MI355_SPMV_OK;
    DeviceGuard guard;
    RcclApi* api = d->world > 1 ? rccl_api() : nullptr;
    for (Part& p : d->parts) {
        (void)hipSetDevice(d->devs[p.dev].device);
        if (p.plan) (void)mi355_spmv_plan_destroy(p.plan);
        if (p.Ap) (void)hipFree(p.Ap);
        if (p.Aj_own) (void)hipFree(p.Aj_own);
        if (p.Ax_own) (void)hipFree(p.Ax_own);
    }
    for (Dev& v : d->devs) {
        (void)hipSetDevice(v.device);
        if (v.nccl && api) (void)api->CommDestroy(v.nccl);
        if (v.compute) (void)hipStreamDestroy(v.compute);
        if (v.comm) (void)hipStreamDestroy(v.comm);
        if (v.side) (void)hipStreamDestroy(v.side);
        if (v.side_done) (void)hipEventDestroy(v.side_done);
        if (v.start) (void)hipEventDestroy(v.start);
        if (v.done) (void)hipEventDestroy(v.done);
        for (hipEvent_t e : v.part_done) (void)hipEventDestroy(e);
        if (v.x) (void)hipFree(v.x);
        if (v.y) (void)hipFree(v.y);
        if (v.stage) (void)hipFree(v.stage);
    }
    if (d->fp_dev && !d->devs.empty()) {
        (void)hipSetDevice(d->devs[size_t(d->home)].device);
        (void)hipFree(d->fp_dev);
    }
    delete d;
    return MI355_SPMV_OK;
}

// Blocks [first, first + count) of d->row_cuts from a source CSR that lives on the CURRENT device:
// Ap_src[i] is the offset of whole-matrix row (src_row0 + i) relative to source element 0, which is element
// src_elem0 of the whole arrays (a multiple of 4).  dev_of(block) gives the target device index; src_dev is the
// index of the device the source lives on (blocks of every other index get copies, also when — an emulated
// RCCL — the two are the same physical device).
template <typename DevOf>
int make_parts(mi355_spmv_dist* d, int first, int count, const void* Ap_src, const int32_t* Aj_src,
               int64_t src_row0, int64_t src_elem0, const mi355_spmv_plan_shape* whole, int flags, int src_dev,
               DevOf dev_of) {
    const int src_device = d->devs[size_t(src_dev)].device;
    const size_t ob = off_bytes(*d);
    for (int b = first; b < first + count; ++b) {
        // (the Part joins d->parts FIRST: whatever it owns by the time a call below fails is freed by destroy_impl)
        d->parts.emplace_back();
        Part& p = d->parts.back();
        p.dev = dev_of(b);
        p.row_begin = d->row_cuts[b];
        p.n_rows = d->row_cuts[b + 1] - d->row_cuts[b];
        const int64_t whole_lo = d->nnz_cuts[b] & ~int64_t(3);      // whole-array element the view starts at
        p.elem_lo = whole_lo - src_elem0;
        p.nnz_end = d->nnz_cuts[b + 1] - whole_lo;
        const int target = d->devs[p.dev].device;
        const bool remote = p.dev != src_dev;
        // row offsets of the block, rebased: made on the source device, then (other GPUs) copied over
        MI355_HIP_TRY(hipSetDevice(src_device));
        MI355_HIP_TRY(hipMalloc(&p.Ap, size_t(p.n_rows + 1) * ob));
        {
            const unsigned g = unsigned(std::min<int64_t>((p.n_rows + 1 + kBlock - 1) / kBlock, 4096));
            const char* src = static_cast<const char*>(Ap_src) + size_t(p.row_begin - src_row0) * ob;
            if (d->off_type == MI355_OFF_I32)
                hipLaunchKernelGGL((rebase_kernel<int32_t>), dim3(g), dim3(kBlock), 0, nullptr,
                                   reinterpret_cast<const int32_t*>(src), p.n_rows + 1, p.elem_lo, static_cast<int32_t*>(p.Ap));
            else
                hipLaunchKernelGGL((rebase_kernel<int64_t>), dim3(g), dim3(kBlock), 0, nullptr,
                                   reinterpret_cast<const int64_t*>(src), p.n_rows + 1, p.elem_lo, static_cast<int64_t*>(p.Ap));
            MI355_HIP_TRY(hipGetLastError());
            MI355_HIP_TRY(hipStreamSynchronize(nullptr));
        }
        if (remote) {
            void* const on_src = p.Ap;
            p.Ap = nullptr;
            const size_t elems = size_t((p.nnz_end + 3) & ~int64_t(3)) + 4;   // whole 16-byte groups + one spare
            hipError_t e = hipSetDevice(target);
            if (e == hipSuccess) e = hipMalloc(&p.Ap, size_t(p.n_rows + 1) * ob);
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p.Aj_own), elems * sizeof(int32_t));
            if (e == hipSuccess) e = hipMalloc(&p.Ax_own, elems * val_bytes(*d));
            if (e == hipSuccess) e = hipMemset(p.Aj_own, 0, elems * sizeof(int32_t));
            if (e == hipSuccess) e = hipMemset(p.Ax_own, 0, elems * val_bytes(*d));
            if (e == hipSuccess) e = hipMemcpy(p.Ap, on_src, size_t(p.n_rows + 1) * ob, hipMemcpyDeviceToDevice);
            if (e == hipSuccess && p.nnz_end > 0)
                e = hipMemcpy(p.Aj_own, Aj_src + p.elem_lo, size_t(p.nnz_end) * sizeof(int32_t), hipMemcpyDeviceToDevice);
            (void)hipSetDevice(src_device);
            (void)hipFree(on_src);
            MI355_HIP_TRY(e);
        }
        MI355_HIP_TRY(hipSetDevice(target));
        const int32_t* Aj_b = p.Aj_own ? p.Aj_own : Aj_src + p.elem_lo;
        const int64_t n_chunks = d->chunk_cuts[b + 1] - d->chunk_cuts[b];
        const int st = mi355_spmv_plan_create_block(&p.plan, d->kind, d->off_type, d->val_type,
                                                    d->kind == MI355_KIND_MERGE ? nullptr : whole, p.row_begin,
                                                    d->chunk_cuts[b], n_chunks, d->nnz_cuts[b], int32_t(p.n_rows),
                                                    d->n_cols, p.nnz_end, p.Ap, Aj_b, flags);
        if (st != MI355_SPMV_OK) return st;
    }
    MI355_HIP_TRY(hipSetDevice(src_device));
    return MI355_SPMV_OK;
}

int make_streams(mi355_spmv_dist* d) {
    for (Dev& v : d->devs) {
        MI355_HIP_TRY(hipSetDevice(v.device));
        if (d->world > 1 || d->sub_blocks > 1) MI355_HIP_TRY(hipEventCreateWithFlags(&v.start, hipEventDisableTiming));
        if (d->sub_blocks > 1) {
            MI355_HIP_TRY(hipStreamCreateWithFlags(&v.side, hipStreamNonBlocking));
            MI355_HIP_TRY(hipEventCreateWithFlags(&v.side_done, hipEventDisableTiming));
        }
        if (d->world > 1) {
            MI355_HIP_TRY(hipStreamCreateWithFlags(&v.compute, hipStreamNonBlocking));
            MI355_HIP_TRY(hipStreamCreateWithFlags(&v.comm, hipStreamNonBlocking));
            MI355_HIP_TRY(hipEventCreateWithFlags(&v.done, hipEventDisableTiming));
            v.part_done.resize(size_t(d->sub_blocks));
            for (hipEvent_t& e : v.part_done) MI355_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
    }
    return MI355_SPMV_OK;
}

// Once the cut lists are known: which sub-blocks ALLGATHER can exchange in place, and the padded block size
void shape_exchange(mi355_spmv_dist* d) {
    d->gather_in_place.assign(size_t(d->sub_blocks), 0);
    int64_t widest = 0;
    for (int g = 0; g < d->world * d->sub_blocks; ++g) widest = std::max(widest, block_rows(*d, g));
    d->max_block_rows = (widest + 3) & ~int64_t(3);
    for (int s = 0; s < d->sub_blocks; ++s) {
        const int64_t cnt = block_rows(*d, s);
        bool ok = cnt > 0;
        for (int r = 0; r < d->world && ok; ++r) {
            const int g = r * d->sub_blocks + s;
            ok = block_rows(*d, g) == cnt && d->row_cuts[size_t(g)] == d->row_cuts[size_t(s)] + int64_t(r) * cnt;
        }
        d->gather_in_place[size_t(s)] = ok ? 1 : 0;
    }
}

bool all_in_place(const mi355_spmv_dist& d) {
    for (char c : d.gather_in_place) if (!c) return false;
    return true;
}

// ALLGATHER through staging needs world * max_block_rows values per GPU (allocated when that exchange is first used)
int ensure_staging(mi355_spmv_dist* d) {
    if (all_in_place(*d)) return MI355_SPMV_OK;
    for (Dev& v : d->devs) {
        if (v.stage) continue;
        MI355_HIP_TRY(hipSetDevice(v.device));
        MI355_HIP_TRY(hipMalloc(&v.stage, size_t(d->world) * size_t(d->max_block_rows) * val_bytes(*d) + 16));
    }
    return MI355_SPMV_OK;
}

int launch_pieces(void* dst, const void* src, const Pieces& pc, int64_t most_words, hipStream_t s) {
    if (pc.n <= 0) return MI355_SPMV_OK;
    bool vec = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0;
    for (int i = 0; i < pc.n; ++i) vec = vec && ((pc.dst[i] | pc.src[i]) & 3) == 0;
    const unsigned gx = unsigned(std::min<int64_t>(std::max<int64_t>((most_words / 4 + kBlock - 1) / kBlock, 1), 512));
    hipLaunchKernelGGL(pieces_copy_kernel, dim3(gx, unsigned(pc.n)), dim3(kBlock), 0, s, static_cast<uint32_t*>(dst),
                       static_cast<const uint32_t*>(src), pc, vec ? 1 : 0);
    MI355_HIP_TRY(hipGetLastError());
    return MI355_SPMV_OK;
}

// The allgatherv of sub-block s, enqueued on every local GPU's communication stream (which already waits for the
// block's kernels).  y_of(i) = GPU i's full-length y.
template <typename YOf>
int exchange_sub_block(mi355_spmv_dist* d, RcclApi* api, int s, YOf y_of) {
    const int n_dev = int(d->devs.size());
    const size_t vb = val_bytes(*d);
    const int wpv = int(vb / 4);                                   // 4-byte words per value
    const ncclDataType_t dt = d->val_type == MI355_VAL_F64 ? ncclFloat64 : ncclFloat32;
    auto blk = [&](int r) { return r * d->sub_blocks + s; };
    RcclGroup group(api);
    switch (d->exchange) {
    case MI355_DIST_EXCHANGE_SENDRECV: {
        // every GPU sends its block to every peer and receives theirs: point to point, all links at once
        MI355_RCCL_TRY(api, group.start());
        for (int i = 0; i < n_dev; ++i) {
            Dev& v = d->devs[size_t(i)];
            char* const yv = static_cast<char*>(y_of(i));
            const int me = rank_of(*d, i);
            const int64_t mine = block_rows(*d, blk(me));
            for (int r = 0; r < d->world; ++r) {
                if (r == me) continue;
                const int64_t theirs = block_rows(*d, blk(r));
                if (mine > 0)
                    MI355_RCCL_TRY(api, api->Send(yv + size_t(d->row_cuts[size_t(blk(me))]) * vb, size_t(mine), dt, r, v.nccl, v.comm));
                if (theirs > 0)
                    MI355_RCCL_TRY(api, api->Recv(yv + size_t(d->row_cuts[size_t(blk(r))]) * vb, size_t(theirs), dt, r, v.nccl, v.comm));
            }
        }
        MI355_RCCL_TRY(api, group.end());
        return MI355_SPMV_OK;
    }
    case MI355_DIST_EXCHANGE_ALLGATHER: {
        if (d->gather_in_place[size_t(s)]) {
            // equal, adjacent blocks: the slices ARE the layout of an allgather (sendbuff = recvbuff + rank * count)
            const int64_t cnt = block_rows(*d, blk(0));
            MI355_RCCL_TRY(api, group.start());
            for (int i = 0; i < n_dev; ++i) {
                Dev& v = d->devs[size_t(i)];
                char* const base = static_cast<char*>(y_of(i)) + size_t(d->row_cuts[size_t(blk(0))]) * vb;
                MI355_RCCL_TRY(api, api->AllGather(base + size_t(rank_of(*d, i)) * size_t(cnt) * vb, base, size_t(cnt), dt, v.nccl, v.comm));
            }
            MI355_RCCL_TRY(api, group.end());
            return MI355_SPMV_OK;
        }
        const int64_t pad = d->max_block_rows;
        for (int i = 0; i < n_dev; ++i) {          // pack: this GPU's block into its slot of the staging buffer
            Dev& v = d->devs[size_t(i)];
            const int me = rank_of(*d, i);
            Pieces pc;
            pc.n = 0;
            if (block_rows(*d, blk(me)) > 0) {
                pc.dst[0] = int64_t(me) * pad * wpv;
                pc.src[0] = d->row_cuts[size_t(blk(me))] * wpv;
                pc.words[0] = block_rows(*d, blk(me)) * wpv;
                pc.n = 1;
            }
            MI355_HIP_TRY(hipSetDevice(v.device));
            if (const int st = launch_pieces(v.stage, y_of(i), pc, pc.n ? pc.words[0] : 0, v.comm)) return st;
        }
        MI355_RCCL_TRY(api, group.start());
        for (int i = 0; i < n_dev; ++i) {
            Dev& v = d->devs[size_t(i)];
            char* const st = static_cast<char*>(v.stage);
            MI355_RCCL_TRY(api, api->AllGather(st + size_t(rank_of(*d, i)) * size_t(pad) * vb, st, size_t(pad), dt, v.nccl, v.comm));
        }
        MI355_RCCL_TRY(api, group.end());
        for (int i = 0; i < n_dev; ++i) {          // unpack: every other GPU's block to its displacement of y
            Dev& v = d->devs[size_t(i)];
            const int me = rank_of(*d, i);
            Pieces pc;
            pc.n = 0;
            int64_t most = 0;
            for (int r = 0; r < d->world; ++r) {
                const int64_t cnt = block_rows(*d, blk(r));
                if (r == me || cnt <= 0) continue;
                pc.dst[pc.n] = d->row_cuts[size_t(blk(r))] * wpv;
                pc.src[pc.n] = int64_t(r) * pad * wpv;
                pc.words[pc.n] = cnt * wpv;
                most = std::max(most, pc.words[pc.n]);
                ++pc.n;
            }
            MI355_HIP_TRY(hipSetDevice(v.device));
            if (const int st = launch_pieces(y_of(i), v.stage, pc, most, v.comm)) return st;
        }
        return MI355_SPMV_OK;
    }
    default: {
        // block (root, s) travels from GPU `root` into the same displacement of every GPU's y — in place on the
        // root.  One group: all of them progress together.
        MI355_RCCL_TRY(api, group.start());
        for (int i = 0; i < n_dev; ++i) {
            Dev& v = d->devs[size_t(i)];
            char* const yv = static_cast<char*>(y_of(i));
            for (int root = 0; root < d->world; ++root) {
                const int64_t cnt = block_rows(*d, blk(root));
                if (cnt <= 0) continue;
                char* at = yv + size_t(d->row_cuts[size_t(blk(root))]) * vb;
                MI355_RCCL_TRY(api, api->Broadcast(at, at, size_t(cnt), dt, root, v.nccl, v.comm));
            }
        }
        MI355_RCCL_TRY(api, group.end());
        return MI355_SPMV_OK;
    }
    }
}

// After the communicator is up (world > 1): choose the exchange.  A forced one (knob) is taken as it is; AUTO times
// each candidate — `reps` exchanges of all sub-blocks on scratch y buffers, host clock between two synchronisations,
// the collectives themselves keep the ranks in step — and every rank takes the maximum over ranks (ncclAllReduce),
// so that all agree on the same choice.
int pick_exchange(mi355_spmv_dist* d, RcclApi* api) {
    const Knobs& k = knobs();
    if (k.dist_exchange != MI355_DIST_EXCHANGE_AUTO) {
        d->exchange = k.dist_exchange;
        return d->exchange == MI355_DIST_EXCHANGE_ALLGATHER ? ensure_staging(d) : MI355_SPMV_OK;
    }
    if (d->n_rows <= 0) return MI355_SPMV_OK;
    const int n_dev = int(d->devs.size());
    const size_t vb = val_bytes(*d);
    // scratch y where this process has none of its own yet (LOCAL: the home GPU; RANK: the rank's GPU)
    std::vector<void*> tmp(size_t(n_dev), nullptr);
    int st = MI355_SPMV_OK;
    auto cleanup = [&] {
        for (int i = 0; i < n_dev; ++i)
            if (tmp[size_t(i)]) { (void)hipSetDevice(d->devs[size_t(i)].device); (void)hipFree(tmp[size_t(i)]); }
    };
    for (int i = 0; i < n_dev && st == MI355_SPMV_OK; ++i) {
        if (d->devs[size_t(i)].y) continue;
        hipError_t e = hipSetDevice(d->devs[size_t(i)].device);
        if (e == hipSuccess) e = hipMalloc(&tmp[size_t(i)], (size_t(d->n_rows) + 4) * vb);
        if (e == hipSuccess) e = hipMemset(tmp[size_t(i)], 0, (size_t(d->n_rows) + 4) * vb);
        if (e != hipSuccess) { set_error("dist: scratch y for the exchange trial: %s", hipGetErrorString(e)); st = e == hipErrorOutOfMemory ? MI355_SPMV_ENOMEM : MI355_SPMV_EHIP; }
    }
    float* times_dev = nullptr;     // [4] on the first local GPU: the trial times, reduced over ranks
    if (st == MI355_SPMV_OK) {
        hipError_t e = hipSetDevice(d->devs[0].device);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&times_dev), 4 * sizeof(float));
        if (e != hipSuccess) { set_error("dist: %s", hipGetErrorString(e)); st = MI355_SPMV_EHIP; }
    }
    if (st == MI355_SPMV_OK) st = ensure_staging(d);
    auto y_of = [&](int i) { return d->devs[size_t(i)].y ? d->devs[size_t(i)].y : tmp[size_t(i)]; };
    auto sync_all = [&]() -> int {
        for (Dev& v : d->devs) {
            MI355_HIP_TRY(hipSetDevice(v.device));
            MI355_HIP_TRY(hipStreamSynchronize(v.comm));
        }
        return MI355_SPMV_OK;
    };
    const int reps = k.dist_trials > 0 ? k.dist_trials : 5;
    float host_us[MI355_DIST_EXCHANGE_COUNT] = {0, 0, 0, 0};
    const int saved = d->exchange;
    for (int mode = MI355_DIST_EXCHANGE_BCAST; mode < MI355_DIST_EXCHANGE_COUNT && st == MI355_SPMV_OK; ++mode) {
        d->exchange = mode;
        auto exchange_all = [&]() -> int {
            for (int s = 0; s < d->sub_blocks; ++s)
                if (const int e = exchange_sub_block(d, api, s, y_of)) return e;
            return MI355_SPMV_OK;
        };
        st = exchange_all();                                  // warm-up: connections, RCCL's own buffers
        if (st == MI355_SPMV_OK) st = sync_all();
        const auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < reps && st == MI355_SPMV_OK; ++rep) st = exchange_all();
        if (st == MI355_SPMV_OK) st = sync_all();
        host_us[mode] = float(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps);
    }
    d->exchange = saved;
    if (st == MI355_SPMV_OK) {
        // maximum over ranks (LOCAL mode: this process measured all GPUs together already — the reduction over
        // its communicators is then a no-op on equal values, but it keeps one code path)
        hipError_t e = hipSetDevice(d->devs[0].device);
        if (e == hipSuccess) e = hipMemcpy(times_dev, host_us, sizeof(host_us), hipMemcpyHostToDevice);
        if (e != hipSuccess) { set_error("dist: %s", hipGetErrorString(e)); st = MI355_SPMV_EHIP; }
    }
    if (st == MI355_SPMV_OK && !d->local_mode) {
        ncclResult_t r = api->AllReduce(times_dev, times_dev, 4, ncclFloat32, ncclMax, d->devs[0].nccl, d->devs[0].comm);
        if (r != ncclSuccess) { set_error("ncclAllReduce -> %s", api->GetErrorString(r)); st = MI355_SPMV_EHIP; }
        if (st == MI355_SPMV_OK) {
            hipError_t e = hipStreamSynchronize(d->devs[0].comm);
            if (e == hipSuccess) e = hipMemcpy(host_us, times_dev, sizeof(host_us), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { set_error("dist: %s", hipGetErrorString(e)); st = MI355_SPMV_EHIP; }
        }
    }
    if (times_dev) { (void)hipSetDevice(d->devs[0].device); (void)hipFree(times_dev); }
    cleanup();
    if (st != MI355_SPMV_OK) return st;
    int best = MI355_DIST_EXCHANGE_BCAST;
    for (int mode = MI355_DIST_EXCHANGE_BCAST; mode < MI355_DIST_EXCHANGE_COUNT; ++mode) {
        d->trial_us[mode] = host_us[mode];
        if (host_us[mode] < host_us[best]) best = mode;
    }
    d->exchange = best;
    d->auto_picked = true;
    return MI355_SPMV_OK;
}

int take_fingerprint(mi355_spmv_dist* d, const void* Ap, const int32_t* Aj, hipStream_t s, unsigned long long* out) {
    Dev& h = d->devs[size_t(d->home)];
    MI355_HIP_TRY(hipSetDevice(h.device));
    if (!d->fp_dev) MI355_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->fp_dev), 16));
    MI355_HIP_TRY(hipMemsetAsync(d->fp_dev, 0, 16, s));
    const int64_t n1 = d->n_rows + 1;
    const int64_t stride = std::max<int64_t>(1, d->nnz / 65536);
    const unsigned g = unsigned(std::min<int64_t>((n1 + kBlock - 1) / kBlock, 2048));
    if (d->off_type == MI355_OFF_I32)
        hipLaunchKernelGGL((fingerprint_kernel<int32_t>), dim3(g), dim3(kBlock), 0, s, static_cast<const int32_t*>(Ap), n1, Aj, d->nnz, stride, d->fp_dev);
    else
        hipLaunchKernelGGL((fingerprint_kernel<int64_t>), dim3(g), dim3(kBlock), 0, s, static_cast<const int64_t*>(Ap), n1, Aj, d->nnz, stride, d->fp_dev);
    MI355_HIP_TRY(hipGetLastError());
    MI355_HIP_TRY(hipMemcpyAsync(out, d->fp_dev, sizeof(*out), hipMemcpyDeviceToHost, s));
    MI355_HIP_TRY(hipStreamSynchronize(s));
    return MI355_SPMV_OK;
}

}  // namespace

extern "C" {

int mi355_spmv_dist_create_local(mi355_spmv_dist** out, int kind, int off_type, int val_type, int32_t n_rows,
                                 int32_t n_cols, int64_t nnz, const void* Ap, const int32_t* Aj, int n_devices,
                                 const int* devices, int sub_blocks, int flags) {
    if (!out) { set_error("dist_create_local: null pointer"); return MI355_SPMV_EINVAL; }
    *out = nullptr;
    if (n_devices < 1 || n_devices > kMaxPieces || sub_blocks < 1 || sub_blocks > 64) {
        set_error("dist_create_local: n_devices %d / sub_blocks %d out of range", n_devices, sub_blocks);
        return MI355_SPMV_EINVAL;
    }
    DeviceGuard guard;
    int home_device = 0;
    MI355_HIP_TRY(hipGetDevice(&home_device));
    mi355_spmv_dist* d = new (std::nothrow) mi355_spmv_dist();
    if (!d) { set_error("dist_create_local: host allocation failed"); return MI355_SPMV_ENOMEM; }
    d->local_mode = true;
    d->kind = kind; d->off_type = off_type; d->val_type = val_type;
    d->n_rows = n_rows; d->n_cols = n_cols; d->nnz = nnz;
    d->world = n_devices;
    d->sub_blocks = sub_blocks;
    d->Aj_src = Aj;
    d->devs.resize(size_t(n_devices));
    d->home = -1;
    const bool shared_ok = knobs().dist_shared_device != 0;    // (an emulated RCCL: several "GPUs" on one device)
    for (int i = 0; i < n_devices; ++i) {
        d->devs[i].device = devices ? devices[i] : (shared_ok ? home_device : i);   // (emulated RCCL: every "GPU" is the current device)
        if (d->devs[i].device == home_device && d->home < 0) d->home = i;
        for (int j = 0; j < i && !shared_ok; ++j)
            if (d->devs[j].device == d->devs[i].device) {
                set_error("dist_create_local: device %d listed twice", d->devs[i].device);
                destroy_impl(d);
                return MI355_SPMV_EINVAL;
            }
    }
    if (d->home < 0) {
        set_error("dist_create_local: the current device (%d, where Ap / Aj live) is not in the device list", home_device);
        destroy_impl(d);
        return MI355_SPMV_EINVAL;
    }
    // the whole matrix's plan decides the launch shape and where the cuts may fall
    mi355_spmv_plan* whole = nullptr;
    int st = mi355_spmv_plan_create(&whole, kind, off_type, val_type, n_rows, n_cols, nnz, Ap, Aj, flags);
    if (st != MI355_SPMV_OK) { destroy_impl(d); return st; }
    mi355_spmv_plan_shape shape;
    const int parts = n_devices * sub_blocks;
    d->row_cuts.resize(size_t(parts) + 1);
    d->chunk_cuts.resize(size_t(parts) + 1);
    d->nnz_cuts.resize(size_t(parts) + 1);
    st = mi355_spmv_plan_get_shape(whole, &shape);
    if (st == MI355_SPMV_OK) d->kind = kind = shape.kind;       // (MI355_KIND_AUTO: what the whole matrix's plan picked)
    if (st == MI355_SPMV_OK)
        st = mi355_spmv_plan_partition(whole, parts, d->row_cuts.data(), d->chunk_cuts.data(), d->nnz_cuts.data());
    (void)mi355_spmv_plan_destroy(whole);
    if (st == MI355_SPMV_OK) { shape_exchange(d); st = make_streams(d); }
    if (st == MI355_SPMV_OK) {
        (void)hipSetDevice(home_device);
        d->parts.reserve(size_t(parts));
        st = make_parts(d, 0, parts, Ap, Aj, /*src_row0=*/0, /*src_elem0=*/0, &shape, flags, d->home,
                        [&](int b) { return b / sub_blocks; });
    }
    if (st == MI355_SPMV_OK && n_rows > 0) st = take_fingerprint(d, Ap, Aj, nullptr, &d->fingerprint);
    // full-length y and a copy of x on every other GPU; one communicator over all of them
    if (st == MI355_SPMV_OK && n_devices > 1) {
        for (int i = 0; i < n_devices && st == MI355_SPMV_OK; ++i) {
            if (i == d->home) continue;
            hipError_t e = hipSetDevice(d->devs[i].device);
            if (e == hipSuccess) e = hipMalloc(&d->devs[i].x, (size_t(n_cols) + 4) * val_bytes(*d));
            if (e == hipSuccess) e = hipMalloc(&d->devs[i].y, (size_t(n_rows) + 4) * val_bytes(*d));
            if (e != hipSuccess) { set_error("dist_create_local: %s", hipGetErrorString(e)); st = e == hipErrorOutOfMemory ? MI355_SPMV_ENOMEM : MI355_SPMV_EHIP; }
        }
        RcclApi* api = st == MI355_SPMV_OK ? rccl_api() : nullptr;
        if (st == MI355_SPMV_OK && !api) st = MI355_SPMV_ENOTSUP;
        if (st == MI355_SPMV_OK) {
            std::vector<ncclComm_t> comms(size_t(n_devices), nullptr);
            std::vector<int> ids(static_cast<size_t>(n_devices));
            for (int i = 0; i < n_devices; ++i) ids[size_t(i)] = d->devs[i].device;
            const ncclResult_t r = api->CommInitAll(comms.data(), n_devices, ids.data());
            if (r != ncclSuccess) { set_error("ncclCommInitAll -> %s", api->GetErrorString(r)); st = MI355_SPMV_EHIP; }
            for (int i = 0; i < n_devices; ++i) d->devs[i].nccl = comms[size_t(i)];
        }
        if (st == MI355_SPMV_OK) st = pick_exchange(d, api);
    }
    if (st != MI355_SPMV_OK) { destroy_impl(d); return st; }
    *out = d;
    return MI355_SPMV_OK;
}

int mi355_spmv_dist_unique_id(void* id128) {
    if (!id128) { set_error("dist_unique_id: null pointer"); return MI355_SPMV_EINVAL; }
    RcclApi* api = rccl_api();
    if (!api) return MI355_SPMV_ENOTSUP;
    static_assert(sizeof(ncclUniqueId) == 128, "the C ABI promises 128 bytes");
    MI355_RCCL_TRY(api, api->GetUniqueId(static_cast<ncclUniqueId*>(id128)));
    return MI355_SPMV_OK;
}

int mi355_spmv_dist_create_rank(mi355_spmv_dist** out, int kind, int off_type, int val_type, int rank, int world,
                                const void* id128, int parts_per_rank, const int64_t* row_cuts,
                                const int64_t* chunk_cuts, const int64_t* nnz_cuts,
                                const mi355_spmv_plan_shape* whole, int32_t n_cols, int32_t n_rows_local,
                                int64_t nnz_end_local, const void* Ap_local, const int32_t* Aj_local, int flags) {
    if (!out) { set_error("dist_create_rank: null pointer"); return MI355_SPMV_EINVAL; }
    *out = nullptr;
    if (world < 1 || world > kMaxPieces || rank < 0 || rank >= world || parts_per_rank < 1 || parts_per_rank > 64 || !row_cuts || !nnz_cuts) {
        set_error("dist_create_rank: bad rank / world / cuts");
        return MI355_SPMV_EINVAL;
    }
    if (world > 1 && !id128) { set_error("dist_create_rank: no unique id"); return MI355_SPMV_EINVAL; }
    const int parts = world * parts_per_rank;
    const int first = rank * parts_per_rank;
    for (int b = 0; b < parts; ++b)
        if (row_cuts[b + 1] < row_cuts[b] || nnz_cuts[b + 1] < nnz_cuts[b]) {
            set_error("dist_create_rank: the cut lists are not ascending");
            return MI355_SPMV_EINVAL;
        }
    if (row_cuts[first + parts_per_rank] - row_cuts[first] != n_rows_local ||
        nnz_cuts[first + parts_per_rank] - (nnz_cuts[first] & ~int64_t(3)) != nnz_end_local) {
        set_error("dist_create_rank: the local arrays do not match this rank's cuts");
        return MI355_SPMV_EINVAL;
    }
    DeviceGuard guard;
    mi355_spmv_dist* d = new (std::nothrow) mi355_spmv_dist();
    if (!d) { set_error("dist_create_rank: host allocation failed"); return MI355_SPMV_ENOMEM; }
    d->local_mode = false;
    d->kind = kind; d->off_type = off_type; d->val_type = val_type;
    d->n_rows = row_cuts[parts]; d->n_cols = n_cols;
    d->nnz = nnz_cuts[parts] - nnz_cuts[0];
    d->world = world; d->rank = rank; d->sub_blocks = parts_per_rank;
    d->Aj_src = Aj_local;
    d->row_cuts.assign(row_cuts, row_cuts + parts + 1);
    d->nnz_cuts.assign(nnz_cuts, nnz_cuts + parts + 1);
    if (chunk_cuts) d->chunk_cuts.assign(chunk_cuts, chunk_cuts + parts + 1);
    else d->chunk_cuts.assign(size_t(parts) + 1, 0);
    d->devs.resize(1);
    d->home = 0;
    int st = MI355_SPMV_OK;
    {
        const hipError_t e = hipGetDevice(&d->devs[0].device);
        if (e != hipSuccess) { set_error("dist_create_rank: %s", hipGetErrorString(e)); st = MI355_SPMV_EHIP; }
    }
    if (st == MI355_SPMV_OK) { shape_exchange(d); st = make_streams(d); }
    if (st == MI355_SPMV_OK) {
        d->parts.reserve(size_t(parts_per_rank));
        st = make_parts(d, first, parts_per_rank, Ap_local, Aj_local, row_cuts[first], nnz_cuts[first] & ~int64_t(3),
                        whole, flags, 0, [](int) { return 0; });
    }
    if (st == MI355_SPMV_OK && world > 1) {
        RcclApi* api = rccl_api();
        if (!api) st = MI355_SPMV_ENOTSUP;
        else {
            ncclUniqueId id;
            memcpy(&id, id128, sizeof(id));
            const ncclResult_t r = api->CommInitRank(&d->devs[0].nccl, world, id, rank);
            if (r != ncclSuccess) { set_error("ncclCommInitRank -> %s", api->GetErrorString(r)); st = MI355_SPMV_EHIP; }
        }
        if (st == MI355_SPMV_OK) st = pick_exchange(d, api);
    }
    if (st != MI355_SPMV_OK) { destroy_impl(d); return st; }
    *out = d;
    return MI355_SPMV_OK;
}

int mi355_spmv_dist_scatter_values(mi355_spmv_dist* d, const void* Ax, void* stream) {
    if (!d || !Ax) { set_error("dist_scatter_values: null argument"); return MI355_SPMV_EINVAL; }
    if (!d->local_mode) { set_error("dist_scatter_values: LOCAL mode only (a rank passes its own values to execute)"); return MI355_SPMV_ENOTSUP; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t vb = val_bytes(*d);
    for (Part& p : d->parts) {
        if (!p.Ax_own || p.nnz_end <= 0) continue;
        MI355_HIP_TRY(hipMemcpyAsync(p.Ax_own, static_cast<const char*>(Ax) + size_t(p.elem_lo) * vb, size_t(p.nnz_end) * vb,
                                     hipMemcpyDeviceToDevice, s));
    }
    d->Ax_home = Ax;
    return MI355_SPMV_OK;
}

int mi355_spmv_dist_replicate_x(mi355_spmv_dist* d, const void* x, void* stream) {
    if (!d || !x) { set_error("dist_replicate_x: null argument"); return MI355_SPMV_EINVAL; }
    if (!d->local_mode) { set_error("dist_replicate_x: LOCAL mode only (a rank passes its own copy of x to execute)"); return MI355_SPMV_ENOTSUP; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (Dev& v : d->devs)
        if (v.x) MI355_HIP_TRY(hipMemcpyAsync(v.x, x, size_t(d->n_cols) * val_bytes(*d), hipMemcpyDeviceToDevice, s));
    d->x_home = x;
    return MI355_SPMV_OK;
}

int mi355_spmv_dist_execute_ex(mi355_spmv_dist* d, const void* Ax, const void* x, void* y, void* stream, int exec_flags) {
    if (!d) { set_error("dist_execute: null handle"); return MI355_SPMV_EINVAL; }
    if (d->n_rows > 0 && !y) { set_error("dist_execute: null y"); return MI355_SPMV_EINVAL; }
    if (exec_flags != MI355_DIST_EXEC_DEFAULT && exec_flags != MI355_DIST_EXEC_SKIP_EXCHANGE && exec_flags != MI355_DIST_EXEC_EXCHANGE_ONLY) {
        set_error("dist_execute: unknown flags %d", exec_flags);
        return MI355_SPMV_EINVAL;
    }
    const bool compute = exec_flags != MI355_DIST_EXEC_EXCHANGE_ONLY;
    const bool exchange = exec_flags != MI355_DIST_EXEC_SKIP_EXCHANGE;
    hipStream_t user = static_cast<hipStream_t>(stream);
    const size_t vb = val_bytes(*d);
    const int n_dev = int(d->devs.size());
    const bool multi = d->world > 1;
    const bool remote_copies = d->local_mode && n_dev > 1;
    if (remote_copies && compute) {
        // non-NULL arrays are handed to the other GPUs first (the drop-in semantics of SpMV(kind, ...));
        // NULL = unchanged since the last scatter_values / replicate_x
        if (Ax) { const int st = mi355_spmv_dist_scatter_values(d, Ax, stream); if (st != MI355_SPMV_OK) return st; }
        if (x) { const int st = mi355_spmv_dist_replicate_x(d, x, stream); if (st != MI355_SPMV_OK) return st; }
        Ax = d->Ax_home;
        x = d->x_home;
    }
    if (compute && d->nnz_cuts.back() > d->nnz_cuts.front() && (!Ax || !x)) {
        set_error(remote_copies ? "dist_execute: Ax / x == NULL before any scatter_values / replicate_x"
                                : "dist_execute: null Ax or x");
        return MI355_SPMV_EINVAL;
    }
    DeviceGuard guard;
    RcclApi* api = multi ? rccl_api() : nullptr;
    if (multi && !api) return MI355_SPMV_ENOTSUP;
    // everything the caller's stream has queued (its previous execute, the scatter / replicate copies above)
    // comes before this execute's work on the other streams
    const bool two = d->sub_blocks > 1;          // two compute streams per GPU: even / odd sub-blocks
    if (multi || two) {
        Dev& h = d->devs[size_t(d->home)];
        MI355_HIP_TRY(hipSetDevice(h.device));
        MI355_HIP_TRY(hipEventRecord(h.start, user));
        for (int i = 0; i < n_dev; ++i) {
            Dev& v = d->devs[size_t(i)];
            MI355_HIP_TRY(hipSetDevice(v.device));
            if (multi && i != d->home) MI355_HIP_TRY(hipStreamWaitEvent(v.compute, h.start, 0));
            if (multi) MI355_HIP_TRY(hipStreamWaitEvent(v.comm, h.start, 0));
            if (two) MI355_HIP_TRY(hipStreamWaitEvent(v.side, h.start, 0));
        }
    }
    auto y_of = [&](int i) -> void* { return (i == d->home || !d->local_mode) ? y : d->devs[size_t(i)].y; };
    for (int s = 0; s < d->sub_blocks; ++s) {
        for (int i = 0; i < n_dev; ++i) {
            Dev& v = d->devs[size_t(i)];
            Part& p = d->parts[size_t(i * d->sub_blocks + s)];
            const bool is_home = i == d->home;
            hipStream_t cs = (two && (s & 1)) ? v.side : ((is_home || !multi) ? user : v.compute);
            MI355_HIP_TRY(hipSetDevice(v.device));
            if (compute && p.n_rows > 0) {
                // home device (and every rank): views of the caller's arrays; the other GPUs: their copies
                const void* ax = p.Ax_own ? p.Ax_own : (Ax ? static_cast<const void*>(static_cast<const char*>(Ax) + size_t(p.elem_lo) * vb) : nullptr);
                const void* xv = (is_home || !d->local_mode) ? x : v.x;
                char* yv = static_cast<char*>(y_of(i));
                const int st = mi355_spmv_plan_execute(p.plan, ax, xv, yv + size_t(p.row_begin) * vb, cs);
                if (st != MI355_SPMV_OK) return st;
            }
            if (multi) {
                MI355_HIP_TRY(hipEventRecord(v.part_done[size_t(s)], cs));
                MI355_HIP_TRY(hipStreamWaitEvent(v.comm, v.part_done[size_t(s)], 0));
            }
        }
        if (multi && exchange) {
            const int st = exchange_sub_block(d, api, s, y_of);
            if (st != MI355_SPMV_OK) return st;
        }
    }
    if (two && !multi) {
        // one GPU, several blocks: the caller's stream continues behind the side stream's blocks
        Dev& v = d->devs[size_t(d->home)];
        MI355_HIP_TRY(hipSetDevice(v.device));
        MI355_HIP_TRY(hipEventRecord(v.side_done, v.side));
        MI355_HIP_TRY(hipStreamWaitEvent(user, v.side_done, 0));
    }
    if (multi) {
        // the caller's stream continues once every GPU of this process holds the whole y
        for (int i = 0; i < n_dev; ++i) {
            Dev& v = d->devs[size_t(i)];
            MI355_HIP_TRY(hipSetDevice(v.device));
            MI355_HIP_TRY(hipEventRecord(v.done, v.comm));
        }
        MI355_HIP_TRY(hipSetDevice(d->devs[size_t(d->home)].device));
        for (int i = 0; i < n_dev; ++i) MI355_HIP_TRY(hipStreamWaitEvent(user, d->devs[size_t(i)].done, 0));
    }
    return MI355_SPMV_OK;
}

int mi355_spmv_dist_execute(mi355_spmv_dist* d, const void* Ax, const void* x, void* y, void* stream) {
    return mi355_spmv_dist_execute_ex(d, Ax, x, y, stream, MI355_DIST_EXEC_DEFAULT);
}

int mi355_spmv_dist_set_exchange(mi355_spmv_dist* d, int exchange) {
    if (!d) { set_error("dist_set_exchange: null handle"); return MI355_SPMV_EINVAL; }
    if (exchange <= MI355_DIST_EXCHANGE_AUTO || exchange >= MI355_DIST_EXCHANGE_COUNT) {
        set_error("dist_set_exchange: %d is not one of BCAST / SENDRECV / ALLGATHER", exchange);
        return MI355_SPMV_EINVAL;
    }
    DeviceGuard guard;
    d->exchange = exchange;
    d->auto_picked = false;
    if (exchange == MI355_DIST_EXCHANGE_ALLGATHER && d->world > 1) return ensure_staging(d);
    return MI355_SPMV_OK;
}

int mi355_spmv_dist_get_info(const mi355_spmv_dist* d, mi355_spmv_dist_info* info) {
    if (!d || !info) { set_error("dist_get_info: null argument"); return MI355_SPMV_EINVAL; }
    memset(info, 0, sizeof(*info));
    info->world = d->world; info->rank = d->rank; info->sub_blocks = d->sub_blocks; info->local_mode = d->local_mode ? 1 : 0;
    info->exchange = d->exchange;
    info->auto_picked = d->auto_picked ? 1 : 0;
    info->allgather_in_place = all_in_place(*d) ? 1 : 0;
    for (int i = 0; i < MI355_DIST_EXCHANGE_COUNT; ++i) info->trial_us[i] = d->trial_us[i];
    info->max_block_rows = d->max_block_rows;
    info->staging_bytes = (!d->devs.empty() && d->devs[0].stage) ? int64_t(d->world) * d->max_block_rows * int64_t(val_bytes(*d)) : 0;
    const char* names[] = {"auto", "bcast", "sendrecv", "allgather"};
    snprintf(info->exchange_name, sizeof(info->exchange_name), "%s", d->world > 1 ? names[d->exchange] : "none");
    return MI355_SPMV_OK;
}

int mi355_spmv_dist_structure_changed(mi355_spmv_dist* d, const void* Ap, const int32_t* Aj, void* stream, int* changed) {
    if (!d || !changed) { set_error("dist_structure_changed: null argument"); return MI355_SPMV_EINVAL; }
    if (!d->local_mode) { set_error("dist_structure_changed: LOCAL mode only"); return MI355_SPMV_ENOTSUP; }
    *changed = 0;
    if (d->n_rows <= 0) return MI355_SPMV_OK;
    if (!Ap || (d->nnz > 0 && !Aj)) { set_error("dist_structure_changed: null Ap / Aj"); return MI355_SPMV_EINVAL; }
    DeviceGuard guard;
    unsigned long long now = 0;
    const int st = take_fingerprint(d, Ap, Aj, static_cast<hipStream_t>(stream), &now);
    if (st != MI355_SPMV_OK) return st;
    *changed = now != d->fingerprint ? 1 : 0;
    return MI355_SPMV_OK;
}

int mi355_spmv_dist_set_alpha_beta(mi355_spmv_dist* d, double alpha, double beta) {
    if (!d) { set_error("dist_set_alpha_beta: null handle"); return MI355_SPMV_EINVAL; }
    if (beta != 0.0 && d->local_mode && d->devs.size() > 1) {
        // beta * y_old needs y_old on every device that owns rows: only the home device has the caller's y
        set_error("dist_set_alpha_beta: beta != 0 is not supported across devices in LOCAL mode");
        return MI355_SPMV_ENOTSUP;
    }
    for (Part& p : d->parts) {
        const int st = mi355_spmv_plan_set_alpha_beta(p.plan, alpha, beta);
        if (st != MI355_SPMV_OK) return st;
    }
    return MI355_SPMV_OK;
}

int mi355_spmv_dist_parts(const mi355_spmv_dist* d) { return d ? int(d->row_cuts.size()) - 1 : 0; }

int mi355_spmv_dist_cuts(const mi355_spmv_dist* d, int64_t* row_cuts) {
    if (!d || !row_cuts) { set_error("dist_cuts: null argument"); return MI355_SPMV_EINVAL; }
    for (size_t i = 0; i < d->row_cuts.size(); ++i) row_cuts[i] = d->row_cuts[i];
    return MI355_SPMV_OK;
}

int mi355_spmv_dist_part_info(const mi355_spmv_dist* d, int part, mi355_spmv_plan_info* info) {
    if (!d || !info || part < 0 || part >= int(d->parts.size())) { set_error("dist_part_info: bad argument"); return MI355_SPMV_EINVAL; }
    return mi355_spmv_plan_get_info(d->parts[size_t(part)].plan, info);
}

void* mi355_spmv_dist_device_y(mi355_spmv_dist* d, int device_index) {
    if (!d || device_index < 0 || device_index >= int(d->devs.size())) return nullptr;
    return d->devs[size_t(device_index)].y;
}

void* mi355_spmv_dist_device_x(mi355_spmv_dist* d, int device_index) {
    if (!d || device_index < 0 || device_index >= int(d->devs.size())) return nullptr;
    return d->devs[size_t(device_index)].x;
}

int mi355_spmv_dist_destroy(mi355_spmv_dist* d) { return destroy_impl(d); }

}  // extern "C"
