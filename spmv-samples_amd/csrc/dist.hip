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
//   * allgatherv(y): RCCL has none, so block g is broadcast in place from its owner (ncclBroadcast, root =
//     owner, same pointer on every GPU), all roots of one sub-block index grouped into one ncclGroup, on a
//     dedicated communication stream per GPU that waits on the event recorded after the block's kernels:
//     sub-block s travels over xGMI while sub-block s + 1 is computed.
// Two ways to drive it: LOCAL (one process, all GPUs: ncclCommInitAll) and RANK (one process per GPU:
// ncclCommInitRank with a caller-distributed unique id).  With one GPU nothing of RCCL is touched and an
// execute is the blocks' plain executes on the caller's stream.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on first multi-GPU create): libmi355spmv.so keeps the HIP
// runtime as its only link dependency, and a process that already carries an RCCL (torch's) shares it.

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only; no symbol of it is linked

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
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) break;
    }
    if (!api.handle) {
        set_error("mi355_spmv_dist: librccl.so.1 not found (%s)", dlerror());
        return nullptr;
    }
    bool ok = true;
    auto bind = [&](const char* sym) -> void* {
        void* f = dlsym(api.handle, sym);
        if (!f) ok = false;
        return f;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(bind("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(bind("ncclCommInitRank"));
    api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(bind("ncclCommInitAll"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(bind("ncclCommDestroy"));
    api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(bind("ncclBroadcast"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(bind("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(bind("ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(bind("ncclGetErrorString"));
    if (!ok) {
        set_error("mi355_spmv_dist: librccl.so.1 lacks a required symbol");
        dlclose(api.handle);
        api.handle = nullptr;
        return nullptr;
    }
    return &api;
}

#define MI355_RCCL_TRY(api, expr)                                                               \
    do {                                                                                        \
        ncclResult_t _r = (expr);                                                               \
        if (_r != ncclSuccess) {                                                                \
            ::mi355::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, (api)->GetErrorString(_r)); \
            return MI355_SPMV_EHIP;                                                             \
        }                                                                                       \
    } while (0)

// out[i] = Ap[i] - base for i in [0, n]: the offsets of a row block, relative to the 16-byte-aligned element
// its view of Aj / Ax starts at (so out[0] is the block's phase, 0..3)
template <typename off_t>
__global__ __launch_bounds__(kBlock) void rebase_kernel(const off_t* __restrict__ Ap, int64_t n_plus_1, int64_t base,
                                                        off_t* __restrict__ out) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n_plus_1; i += int64_t(gridDim.x) * kBlock)
        out[i] = off_t(int64_t(Ap[i]) - base);
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
    int32_t* Aj_own = nullptr;   // owned copy (remote devices), else null: view of the source Aj
    void* Ax_own = nullptr;      // owned copy (remote devices), else null: view of the caller's Ax
    mi355_spmv_plan* plan = nullptr;
};

struct Dev {
    int device = 0;           // HIP ordinal
    hipStream_t compute = nullptr, comm = nullptr;
    hipStream_t side = nullptr;      // odd sub-blocks run here: the ramp of block s + 1 overlaps the drain of block s
    hipEvent_t start = nullptr, done = nullptr, side_done = nullptr;
    std::vector<hipEvent_t> part_done;
    void* x = nullptr;        // remote devices (LOCAL mode): replicated x / full-length y, owned
    void* y = nullptr;
    ncclComm_t nccl = nullptr;
};

}  // namespace

struct mi355_spmv_dist {
    bool local_mode = true;
    int kind = 0, off_type = 0, val_type = 0;
    int64_t n_rows = 0;       // whole matrix
    int32_t n_cols = 0;
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
};

namespace {

size_t off_bytes(const mi355_spmv_dist& d) { return d.off_type == MI355_OFF_I64 ? 8 : 4; }
size_t val_bytes(const mi355_spmv_dist& d) { return d.val_type == MI355_VAL_F64 ? 8 : 4; }

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
    }
    delete d;
    return MI355_SPMV_OK;
}

// Blocks [first, first + count) of d->row_cuts from a source CSR that lives on the CURRENT device:
// Ap_src[i] is the offset of whole-matrix row (src_row0 + i) relative to source element 0, which is element
// src_elem0 of the whole arrays (a multiple of 4).  dev_of(block) gives the target device index.
template <typename DevOf>
int make_parts(mi355_spmv_dist* d, int first, int count, const void* Ap_src, const int32_t* Aj_src,
               int64_t src_row0, int64_t src_elem0, const mi355_spmv_plan_shape* whole, int flags, DevOf dev_of) {
    int src_device = 0;
    MI355_HIP_TRY(hipGetDevice(&src_device));
    const size_t ob = off_bytes(*d);
    for (int b = first; b < first + count; ++b) {
        Part p;
        p.dev = dev_of(b);
        p.row_begin = d->row_cuts[b];
        p.n_rows = d->row_cuts[b + 1] - d->row_cuts[b];
        const int64_t whole_lo = d->nnz_cuts[b] & ~int64_t(3);      // whole-array element the view starts at
        p.elem_lo = whole_lo - src_elem0;
        p.nnz_end = d->nnz_cuts[b + 1] - whole_lo;
        const int target = d->devs[p.dev].device;
        const bool remote = target != src_device;
        // row offsets of the block, rebased (a small array: always a fresh allocation on the block's device)
        void* tmp = nullptr;
        MI355_HIP_TRY(hipSetDevice(src_device));
        MI355_HIP_TRY(hipMalloc(&tmp, size_t(p.n_rows + 1) * ob));
        {
            const unsigned g = unsigned(std::min<int64_t>((p.n_rows + 1 + kBlock - 1) / kBlock, 4096));
            const char* src = static_cast<const char*>(Ap_src) + size_t(p.row_begin - src_row0) * ob;
            if (d->off_type == MI355_OFF_I32)
                hipLaunchKernelGGL((rebase_kernel<int32_t>), dim3(g), dim3(kBlock), 0, nullptr,
                                   reinterpret_cast<const int32_t*>(src), p.n_rows + 1, p.elem_lo, static_cast<int32_t*>(tmp));
            else
                hipLaunchKernelGGL((rebase_kernel<int64_t>), dim3(g), dim3(kBlock), 0, nullptr,
                                   reinterpret_cast<const int64_t*>(src), p.n_rows + 1, p.elem_lo, static_cast<int64_t*>(tmp));
            MI355_HIP_TRY(hipGetLastError());
            MI355_HIP_TRY(hipStreamSynchronize(nullptr));
        }
        if (!remote) {
            p.Ap = tmp;
        } else {
            const size_t elems = size_t((p.nnz_end + 3) & ~int64_t(3)) + 4;   // whole 16-byte groups + one spare
            MI355_HIP_TRY(hipSetDevice(target));
            MI355_HIP_TRY(hipMalloc(&p.Ap, size_t(p.n_rows + 1) * ob));
            MI355_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p.Aj_own), elems * sizeof(int32_t)));
            MI355_HIP_TRY(hipMalloc(&p.Ax_own, elems * val_bytes(*d)));
            MI355_HIP_TRY(hipMemset(p.Aj_own, 0, elems * sizeof(int32_t)));
            MI355_HIP_TRY(hipMemset(p.Ax_own, 0, elems * val_bytes(*d)));
            MI355_HIP_TRY(hipMemcpy(p.Ap, tmp, size_t(p.n_rows + 1) * ob, hipMemcpyDeviceToDevice));
            if (p.nnz_end > 0)
                MI355_HIP_TRY(hipMemcpy(p.Aj_own, Aj_src + p.elem_lo, size_t(p.nnz_end) * sizeof(int32_t), hipMemcpyDeviceToDevice));
            MI355_HIP_TRY(hipSetDevice(src_device));
            MI355_HIP_TRY(hipFree(tmp));
        }
        MI355_HIP_TRY(hipSetDevice(target));
        const int32_t* Aj_b = p.Aj_own ? p.Aj_own : Aj_src + p.elem_lo;
        const int64_t n_chunks = d->chunk_cuts[b + 1] - d->chunk_cuts[b];
        d->parts.push_back(p);                                            // (owned pointers are now the dist's to free)
        const int st = mi355_spmv_plan_create_block(&d->parts.back().plan, d->kind, d->off_type, d->val_type,
                                                    d->kind == MI355_KIND_MERGE ? nullptr : whole, p.row_begin,
                                                    d->chunk_cuts[b], n_chunks, d->nnz_cuts[b], int32_t(p.n_rows),
                                                    d->n_cols, p.nnz_end, d->parts.back().Ap, Aj_b, flags);
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

}  // namespace

extern "C" {

int mi355_spmv_dist_create_local(mi355_spmv_dist** out, int kind, int off_type, int val_type, int32_t n_rows,
                                 int32_t n_cols, int64_t nnz, const void* Ap, const int32_t* Aj, int n_devices,
                                 const int* devices, int sub_blocks, int flags) {
    if (!out) { set_error("dist_create_local: null pointer"); return MI355_SPMV_EINVAL; }
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64 || sub_blocks < 1 || sub_blocks > 64) {
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
    d->n_rows = n_rows; d->n_cols = n_cols;
    d->world = n_devices;
    d->sub_blocks = sub_blocks;
    d->Aj_src = Aj;
    d->devs.resize(size_t(n_devices));
    d->home = -1;
    for (int i = 0; i < n_devices; ++i) {
        d->devs[i].device = devices ? devices[i] : i;
        if (d->devs[i].device == home_device && d->home < 0) d->home = i;
        for (int j = 0; j < i; ++j)
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
    if (st == MI355_SPMV_OK)
        st = mi355_spmv_plan_partition(whole, parts, d->row_cuts.data(), d->chunk_cuts.data(), d->nnz_cuts.data());
    (void)mi355_spmv_plan_destroy(whole);
    if (st == MI355_SPMV_OK) st = make_streams(d);
    if (st == MI355_SPMV_OK) {
        (void)hipSetDevice(home_device);
        st = make_parts(d, 0, parts, Ap, Aj, /*src_row0=*/0, /*src_elem0=*/0, &shape, flags,
                        [&](int b) { return b / sub_blocks; });
    }
    // full-length y and a copy of x on every remote device; one communicator over all of them
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
    if (world < 1 || rank < 0 || rank >= world || parts_per_rank < 1 || parts_per_rank > 64 || !row_cuts || !nnz_cuts) {
        set_error("dist_create_rank: bad rank / world / cuts");
        return MI355_SPMV_EINVAL;
    }
    if (world > 1 && !id128) { set_error("dist_create_rank: no unique id"); return MI355_SPMV_EINVAL; }
    const int parts = world * parts_per_rank;
    const int first = rank * parts_per_rank;
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
    if (st == MI355_SPMV_OK) st = make_streams(d);
    if (st == MI355_SPMV_OK)
        st = make_parts(d, first, parts_per_rank, Ap_local, Aj_local, row_cuts[first], nnz_cuts[first] & ~int64_t(3),
                        whole, flags, [](int) { return 0; });
    if (st == MI355_SPMV_OK && world > 1) {
        RcclApi* api = rccl_api();
        if (!api) st = MI355_SPMV_ENOTSUP;
        else {
            ncclUniqueId id;
            memcpy(&id, id128, sizeof(id));
            const ncclResult_t r = api->CommInitRank(&d->devs[0].nccl, world, id, rank);
            if (r != ncclSuccess) { set_error("ncclCommInitRank -> %s", api->GetErrorString(r)); st = MI355_SPMV_EHIP; }
        }
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

int mi355_spmv_dist_execute(mi355_spmv_dist* d, const void* Ax, const void* x, void* y, void* stream) {
    if (!d) { set_error("dist_execute: null handle"); return MI355_SPMV_EINVAL; }
    if (d->n_rows > 0 && !y) { set_error("dist_execute: null y"); return MI355_SPMV_EINVAL; }
    hipStream_t user = static_cast<hipStream_t>(stream);
    const size_t vb = val_bytes(*d);
    const int n_dev = int(d->devs.size());
    const bool multi = d->world > 1;
    const bool remote_copies = d->local_mode && n_dev > 1;
    if (remote_copies) {
        // non-NULL arrays are handed to the other devices first (the drop-in semantics of SpMV(kind, ...));
        // NULL = unchanged since the last scatter_values / replicate_x
        if (Ax) { const int st = mi355_spmv_dist_scatter_values(d, Ax, stream); if (st != MI355_SPMV_OK) return st; }
        if (x) { const int st = mi355_spmv_dist_replicate_x(d, x, stream); if (st != MI355_SPMV_OK) return st; }
        Ax = d->Ax_home;
        x = d->x_home;
    }
    if (d->nnz_cuts.back() > d->nnz_cuts.front() && (!Ax || !x)) {
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
    const ncclDataType_t dt = d->val_type == MI355_VAL_F64 ? ncclFloat64 : ncclFloat32;
    for (int s = 0; s < d->sub_blocks; ++s) {
        for (int i = 0; i < n_dev; ++i) {
            Dev& v = d->devs[size_t(i)];
            Part& p = d->parts[size_t(i * d->sub_blocks + s)];
            const bool is_home = i == d->home;
            hipStream_t cs = (two && (s & 1)) ? v.side : ((is_home || !multi) ? user : v.compute);
            MI355_HIP_TRY(hipSetDevice(v.device));
            if (p.n_rows > 0) {
                // home device (and every rank): views of the caller's arrays; remote devices: their copies
                const void* ax = p.Ax_own ? p.Ax_own : (Ax ? static_cast<const void*>(static_cast<const char*>(Ax) + size_t(p.elem_lo) * vb) : nullptr);
                const void* xv = (is_home || !d->local_mode) ? x : v.x;
                char* yv = static_cast<char*>((is_home || !d->local_mode) ? y : v.y);
                const int st = mi355_spmv_plan_execute(p.plan, ax, xv, yv + size_t(p.row_begin) * vb, cs);
                if (st != MI355_SPMV_OK) return st;
            }
            if (multi) {
                MI355_HIP_TRY(hipEventRecord(v.part_done[size_t(s)], cs));
                MI355_HIP_TRY(hipStreamWaitEvent(v.comm, v.part_done[size_t(s)], 0));
            }
        }
        if (multi) {
            // allgatherv of sub-block s: block (root, s) travels from GPU `root` into the same displacement of
            // every GPU's y — in place on the root.  One group: all of them progress together.
            MI355_RCCL_TRY(api, api->GroupStart());
            for (int i = 0; i < n_dev; ++i) {
                Dev& v = d->devs[size_t(i)];
                char* yv = static_cast<char*>((i == d->home || !d->local_mode) ? y : v.y);
                for (int root = 0; root < d->world; ++root) {
                    const int g = root * d->sub_blocks + s;
                    const int64_t cnt = d->row_cuts[size_t(g) + 1] - d->row_cuts[size_t(g)];
                    if (cnt <= 0) continue;
                    char* at = yv + size_t(d->row_cuts[size_t(g)]) * vb;
                    MI355_RCCL_TRY(api, api->Broadcast(at, at, size_t(cnt), dt, root, v.nccl, v.comm));
                }
            }
            MI355_RCCL_TRY(api, api->GroupEnd());
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
