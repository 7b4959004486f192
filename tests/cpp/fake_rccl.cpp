// fake_rccl.cpp — TEST CODE: an emulation of the RCCL entry points mi355_spmv_dist_* binds (csrc/dist.hip, RcclApi),
// so that the N > 1 schedule — communicator set-up, grouped broadcasts / send-recv pairs / all-gathers with their
// roots, counts and displacements, the communication-stream and event choreography — runs on a box with ONE GPU
// (the real library refuses two ranks on one device).  Loaded through MI355_SPMV_RCCL_LIB; never shipped, never
// linked by the product.
//
// A "rank" is a communicator; several live on the same device.  Data moves with hipMemcpyAsync on the stream the
// call was given, ordered with events exactly as a collective orders ranks:
//   * a rank's contribution is ready when everything queued on ITS stream before the call has run;
//   * a rank's stream continues after the call only when its own receives have landed AND every peer that reads its
//     send buffer has read it.
// One host thread may drive all ranks (ncclCommInitAll + ncclGroupStart/End, LOCAL mode) or one thread per rank
// (ncclCommInitRank, RANK mode: the group end then blocks on the other threads, like the real thing's streams do).
// Every call is appended to a log the tests read back (fake_rccl_log_*).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <map>
#include <string>
#include <memory>
#include <mutex>
#include <vector>

namespace {

struct Post {                      // one rank's side of one collective, or one end of a send / recv pair
    bool posted = false;
    const void* send = nullptr;
    void* recv = nullptr;
    size_t bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr;    // recorded on `stream` at the call
    hipEvent_t done = nullptr;     // recorded once this rank's copies are queued
    bool done_posted = false;
};

struct Coll {                      // collective number `seq` of a universe
    std::vector<Post> rank;
    int n_posted = 0, n_done = 0;
};

struct P2P {                       // the k-th message from src to dst
    Post send, recv;
};

struct Universe {
    int world = 0;
    int joined = 0, left = 0;
    std::mutex m;
    std::condition_variable cv;
    std::map<uint64_t, Coll> colls;
    std::map<std::pair<int, int>, std::map<uint64_t, P2P>> p2p;   // (src, dst) -> message number -> pair
};

struct Comm {
    std::shared_ptr<Universe> u;
    int rank = 0, device = 0;
    uint64_t seq = 0;                                   // collectives issued
    std::map<int, uint64_t> sent, received;             // messages issued per peer
};

enum Kind { BCAST = 1, SEND = 2, RECV = 3, ALLGATHER = 4, ALLREDUCE = 5 };

struct Call {
    Kind kind;
    Comm* c;
    const void* send;
    void* recv;
    size_t count;
    ncclDataType_t dt;
    int peer;                                           // root (BCAST) or peer (SEND / RECV)
    ncclRedOp_t op;
    hipStream_t stream;
    uint64_t seq = 0;
    float acc[64] = {};                                 // ALLREDUCE: the reduced words, written in phase_join
};

struct LogEntry { int kind, rank, peer; uint64_t count; const void* send; void* recv; };

// (never destroyed: a handle may still be released by an interpreter's finalisers after the static destructors ran)
std::mutex& g_m = *new std::mutex();
std::map<std::string, std::weak_ptr<Universe>>& g_by_id = *new std::map<std::string, std::weak_ptr<Universe>>();
uint64_t g_next_id = 1;
std::vector<LogEntry>& g_log = *new std::vector<LogEntry>();
thread_local int t_group_depth = 0;
thread_local std::vector<Call> t_calls;

size_t dt_bytes(ncclDataType_t dt) {
    switch (dt) {
        case ncclFloat64: case ncclInt64: case ncclUint64: return 8;
        case ncclFloat32: case ncclInt32: case ncclUint32: return 4;
        case ncclFloat16: case ncclBfloat16: return 2;
        default: return 1;
    }
}

#define HIP_OK(e) do { if ((e) != hipSuccess) return ncclUnhandledCudaError; } while (0)

ncclResult_t new_event(hipEvent_t* ev, hipStream_t s) {
    HIP_OK(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    HIP_OK(hipEventRecord(*ev, s));
    return ncclSuccess;
}

// ---- the three phases of a group ------------------------------------------------------------------------------
ncclResult_t phase_post(Call& k) {
    Universe& u = *k.c->u;
    hipEvent_t ready = nullptr;
    HIP_OK(hipSetDevice(k.c->device));
    if (ncclResult_t r = new_event(&ready, k.stream)) return r;
    std::lock_guard<std::mutex> lock(u.m);
    if (k.kind == SEND || k.kind == RECV) {
        const bool snd = k.kind == SEND;
        const int src = snd ? k.c->rank : k.peer, dst = snd ? k.peer : k.c->rank;
        k.seq = snd ? k.c->sent[k.peer]++ : k.c->received[k.peer]++;
        P2P& m = u.p2p[{src, dst}][k.seq];
        Post& p = snd ? m.send : m.recv;
        p.posted = true; p.send = k.send; p.recv = k.recv; p.bytes = k.count * dt_bytes(k.dt); p.stream = k.stream; p.ready = ready;
    } else {
        k.seq = k.c->seq++;
        Coll& c = u.colls[k.seq];
        if (c.rank.empty()) c.rank.resize(size_t(u.world));
        Post& p = c.rank[size_t(k.c->rank)];
        p.posted = true; p.send = k.send; p.recv = k.recv; p.bytes = k.count * dt_bytes(k.dt); p.stream = k.stream; p.ready = ready;
        ++c.n_posted;
    }
    u.cv.notify_all();
    return ncclSuccess;
}

ncclResult_t phase_copy(Call& k) {
    Universe& u = *k.c->u;
    const int me = k.c->rank;
    HIP_OK(hipSetDevice(k.c->device));
    std::unique_lock<std::mutex> lock(u.m);
    if (k.kind == SEND) return ncclSuccess;                       // the receiver does the copy
    if (k.kind == RECV) {
        P2P& m = u.p2p[{k.peer, me}][k.seq];
        u.cv.wait(lock, [&] { return m.send.posted; });
        if (m.send.bytes != m.recv.bytes) return ncclInvalidArgument;
        const Post snd = m.send;
        lock.unlock();
        HIP_OK(hipStreamWaitEvent(k.stream, snd.ready, 0));
        HIP_OK(hipMemcpyAsync(k.recv, snd.send, snd.bytes, hipMemcpyDeviceToDevice, k.stream));
        hipEvent_t done = nullptr;
        if (ncclResult_t r = new_event(&done, k.stream)) return r;
        lock.lock();
        m.recv.done = done;
        m.recv.done_posted = true;
        u.cv.notify_all();
        return ncclSuccess;
    }
    Coll& c = u.colls[k.seq];
    u.cv.wait(lock, [&] { return c.n_posted == u.world; });
    const std::vector<Post> posts = c.rank;                       // (copies: the table may grow under other threads)
    lock.unlock();
    const size_t bytes = posts[size_t(me)].bytes;
    for (const Post& p : posts)
        if (p.bytes != bytes) return ncclInvalidArgument;         // every rank must pass the same count
    if (k.kind == BCAST) {
        if (me != k.peer) {
            HIP_OK(hipStreamWaitEvent(k.stream, posts[size_t(k.peer)].ready, 0));
            HIP_OK(hipMemcpyAsync(k.recv, posts[size_t(k.peer)].send, bytes, hipMemcpyDeviceToDevice, k.stream));
        } else if (k.recv != k.send) {
            HIP_OK(hipMemcpyAsync(k.recv, k.send, bytes, hipMemcpyDeviceToDevice, k.stream));
        }
    } else if (k.kind == ALLGATHER) {
        for (int r = 0; r < u.world; ++r) {
            char* dst = static_cast<char*>(k.recv) + size_t(r) * bytes;
            if (r == me) {
                if (dst != k.send) HIP_OK(hipMemcpyAsync(dst, k.send, bytes, hipMemcpyDeviceToDevice, k.stream));
                continue;
            }
            HIP_OK(hipStreamWaitEvent(k.stream, posts[size_t(r)].ready, 0));
            HIP_OK(hipMemcpyAsync(dst, posts[size_t(r)].send, bytes, hipMemcpyDeviceToDevice, k.stream));
        }
    } else {                                                      // ALLREDUCE of a few fp32 words, on the host
        if (k.dt != ncclFloat32 || k.count > 64) return ncclInvalidArgument;
        float tmp[64];
        for (int r = 0; r < u.world; ++r) {
            HIP_OK(hipEventSynchronize(posts[size_t(r)].ready));
            HIP_OK(hipMemcpy(tmp, posts[size_t(r)].send, bytes, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < k.count; ++i) {
                if (r == 0) k.acc[i] = tmp[i];
                else if (k.op == ncclMax) k.acc[i] = k.acc[i] > tmp[i] ? k.acc[i] : tmp[i];
                else if (k.op == ncclMin) k.acc[i] = k.acc[i] < tmp[i] ? k.acc[i] : tmp[i];
                else k.acc[i] += tmp[i];
            }
        }
        // (the result is written in phase_join, once EVERY rank has read every send buffer: the call may be in place)
    }
    hipEvent_t done = nullptr;
    if (ncclResult_t r = new_event(&done, k.stream)) return r;
    lock.lock();
    c.rank[size_t(me)].done = done;
    c.rank[size_t(me)].done_posted = true;
    ++c.n_done;
    u.cv.notify_all();
    return ncclSuccess;
}

ncclResult_t phase_join(Call& k) {
    Universe& u = *k.c->u;
    const int me = k.c->rank;
    HIP_OK(hipSetDevice(k.c->device));
    std::unique_lock<std::mutex> lock(u.m);
    if (k.kind == RECV) return ncclSuccess;
    if (k.kind == SEND) {                                         // my buffer is free once the receiver has read it
        P2P& m = u.p2p[{me, k.peer}][k.seq];
        u.cv.wait(lock, [&] { return m.recv.done_posted; });
        const hipEvent_t done = m.recv.done;
        lock.unlock();
        HIP_OK(hipStreamWaitEvent(k.stream, done, 0));
        return ncclSuccess;
    }
    Coll& c = u.colls[k.seq];
    u.cv.wait(lock, [&] { return c.n_done == u.world; });
    const std::vector<Post> posts = c.rank;
    lock.unlock();
    if (k.kind == ALLREDUCE) {        // every rank has read every contribution (and synchronised every stream on the way)
        HIP_OK(hipMemcpy(k.recv, k.acc, k.count * sizeof(float), hipMemcpyHostToDevice));
        return ncclSuccess;
    }
    for (int r = 0; r < u.world; ++r)
        if (r != me && posts[size_t(r)].done) HIP_OK(hipStreamWaitEvent(k.stream, posts[size_t(r)].done, 0));
    return ncclSuccess;
}

ncclResult_t flush_group() {
    std::vector<Call> calls;
    calls.swap(t_calls);
    for (Call& k : calls) if (ncclResult_t r = phase_post(k)) return r;
    for (Call& k : calls) if (ncclResult_t r = phase_copy(k)) return r;
    for (Call& k : calls) if (ncclResult_t r = phase_join(k)) return r;
    return ncclSuccess;
}

ncclResult_t enqueue(Call k) {
    {
        std::lock_guard<std::mutex> lock(g_m);
        g_log.push_back(LogEntry{int(k.kind), k.c->rank, k.peer, uint64_t(k.count), k.send, k.recv});
    }
    t_calls.push_back(k);
    return t_group_depth > 0 ? ncclSuccess : flush_group();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::lock_guard<std::mutex> lock(g_m);
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "fake-rccl-%llu", (unsigned long long)g_next_id++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::shared_ptr<Universe> u;
    {
        std::lock_guard<std::mutex> lock(g_m);
        const std::string key(id.internal, strnlen(id.internal, sizeof(id.internal)));
        u = g_by_id[key].lock();
        if (!u) {
            u = std::make_shared<Universe>();
            u->world = nranks;
            g_by_id[key] = u;
        }
    }
    if (u->world != nranks) return ncclInvalidArgument;
    Comm* c = new Comm();
    c->u = u;
    c->rank = rank;
    HIP_OK(hipGetDevice(&c->device));
    {   // like the real thing: returns when every rank has joined
        std::unique_lock<std::mutex> lock(u->m);
        ++u->joined;
        u->cv.notify_all();
        u->cv.wait(lock, [&] { return u->joined >= u->world; });
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1) return ncclInvalidArgument;
    auto u = std::make_shared<Universe>();
    u->world = ndev;
    u->joined = ndev;
    for (int i = 0; i < ndev; ++i) {
        Comm* c = new Comm();
        c->u = u;
        c->rank = i;
        c->device = devlist ? devlist[i] : i;
        comms[i] = reinterpret_cast<ncclComm_t>(c);
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete reinterpret_cast<Comm*>(comm);       // (events of finished collectives are left to the process's end: test code)
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || root < 0 || root >= c->u->world) return ncclInvalidArgument;
    return enqueue(Call{BCAST, c, send, recv, count, dt, root, ncclSum, s});
}
ncclResult_t ncclSend(const void* send, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || peer < 0 || peer >= c->u->world || peer == c->rank) return ncclInvalidArgument;
    return enqueue(Call{SEND, c, send, nullptr, count, dt, peer, ncclSum, s});
}
ncclResult_t ncclRecv(void* recv, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || peer < 0 || peer >= c->u->world || peer == c->rank) return ncclInvalidArgument;
    return enqueue(Call{RECV, c, nullptr, recv, count, dt, peer, ncclSum, s});
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclInvalidArgument;
    return enqueue(Call{ALLGATHER, c, send, recv, count, dt, -1, ncclSum, s});
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclInvalidArgument;
    return enqueue(Call{ALLREDUCE, c, send, recv, count, dt, -1, op, s});
}
ncclResult_t ncclGroupStart() { ++t_group_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (t_group_depth <= 0) return ncclInvalidUsage;
    if (--t_group_depth > 0) return ncclSuccess;
    return flush_group();
}
const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (fake rccl)";
        case ncclUnhandledCudaError: return "unhandled hip error (fake rccl)";
        case ncclInvalidArgument: return "invalid argument (fake rccl)";
        case ncclInvalidUsage: return "invalid usage (fake rccl)";
        default: return "error (fake rccl)";
    }
}

// ---- what the tests read back ---------------------------------------------------------------------------------
int fake_rccl_log_size(void) { std::lock_guard<std::mutex> lock(g_m); return int(g_log.size()); }
void fake_rccl_log_clear(void) { std::lock_guard<std::mutex> lock(g_m); g_log.clear(); }
// entry i -> {kind (1 bcast, 2 send, 3 recv, 4 allgather, 5 allreduce), rank, root-or-peer, count, send pointer, recv pointer}
int fake_rccl_log_get(int i, long long out[6]) {
    std::lock_guard<std::mutex> lock(g_m);
    if (i < 0 || i >= int(g_log.size())) return 1;
    const LogEntry& e = g_log[size_t(i)];
    out[0] = e.kind; out[1] = e.rank; out[2] = e.peer; out[3] = (long long)e.count;
    out[4] = (long long)reinterpret_cast<uintptr_t>(e.send); out[5] = (long long)reinterpret_cast<uintptr_t>(e.recv);
    return 0;
}

}  // extern "C"
