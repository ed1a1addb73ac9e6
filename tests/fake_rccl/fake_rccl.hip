// tests/fake_rccl/fake_rccl.hip -- TEST INFRASTRUCTURE, never shipped with or linked into libpapof.so.
//
// A stand-in for librccl's point-to-point API with RCCL's SEMANTICS, for a box with ONE GPU: every rank of a
// communicator is a host thread of this process (its own papof handle, arena and streams) on the same device.
// libpapof's RcclTransport (papteam_opticalflow_amd/csrc/tiles.hip) dlopen()s this library instead of librccl when
// PAPOF_RCCL_LIB names it, and then runs exactly the code the driver's 8-GPU node runs: papof_tiles_create ->
// ncclCommInitRank, ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the rank's stream, ncclCommAbort.
//
// What "RCCL's semantics" means here, and what the LOCAL transport of tiles.hip (hipStreamSynchronize + two host
// barriers of all ranks per exchange) cannot show:
//   * ncclGroupEnd RETURNS AT ONCE.  No host thread waits for a peer, no stream is synchronised, there is no group barrier:
//     a group becomes ONE kernel on the caller's stream (as in RCCL), one workgroup (or a few) per send / receive.
//   * Ordering is STREAM ORDER only.  A receive completes -- in stream order -- when the matching send's bytes are in the
//     receive buffer; a send completes when its bytes have been read (RENDEZVOUS: the strictest reading of the API, under
//     which a program that needs eager sends to make progress deadlocks here, as it may on the real library with large messages).
//   * Sends and receives between two ranks match in ISSUE ORDER per (source, destination) pair; byte counts must agree;
//     a send nobody receives, or a receive nobody sends, never completes (bounded here: FAKE_RCCL_TIMEOUT_MS, default 30 s,
//     then the kernel gives up and the error counter -- fake_rccl_error_count() -- is raised).
//   * ncclCommAbort makes the communicator's kernels in flight exit (a host-mapped flag they poll, as RCCL's do).
//
// Device-side handshake (all ranks share the device, so plain device memory connects them): a ring of descriptors per
// ordered pair of ranks; the send side publishes {source pointer, bytes} with an agent-scope release, the receive side
// acquires it, copies, and acknowledges with a release; the send side returns on the acknowledgement.  The L2s of the XCDs
// are not coherent with each other: the acquire / release fences are what make a peer kernel's stores visible while both
// kernels are running (kernel boundaries do the same for everything enqueued before / after the group on each stream).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr int kSlots = 16;      // descriptors in flight per ordered pair of ranks
constexpr int kMaxOps = 40;     // sends + receives of one group
constexpr int kBlock = 256;     // threads per workgroup
constexpr int kMaxCopyBlocks = 8;
constexpr unsigned long long kBytesPerCopyBlock = 1ull << 20;

struct alignas(64) Desc {
    unsigned long long seq;    // message number + 1 once {src, bytes} are valid
    const void* src;
    unsigned long long bytes;
    unsigned long long ack;    // message number + 1 once the bytes have been read
    unsigned nack;             // copy blocks of the current message that have finished
    unsigned pad[7];
};

struct Op {
    Desc* d;
    void* buf;
    unsigned long long bytes;
    unsigned long long seq1;  // message number + 1
    int recv;                 // 0 send, 1 receive
    int blk0, nblk;           // workgroups of the group kernel that serve this operation
    int pad;
};

struct GroupArgs {
    int nops;
    int pad;
    unsigned long long timeout_ticks;  // 100 MHz
    unsigned* err;                     // host-mapped: errors of the whole process (read by the tests)
    const unsigned* abort_flag;        // host-mapped: this communicator has been aborted
    Op op[kMaxOps];
};

// err[0] errors, err[1] sends that gave up, err[2] receives that gave up, err[3] byte counts that disagreed
__device__ inline void raise(const GroupArgs& a, int kind) {
    __hip_atomic_fetch_add(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_add(a.err + kind, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ inline bool wait_for(const unsigned long long* word, unsigned long long want, const GroupArgs& a, int kind) {
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want) {
        __builtin_amdgcn_s_sleep(8);
        if ((++spins & 63u) == 0) {
            if (__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return false;
            if (wall_clock64() - t0 > a.timeout_ticks) {
                raise(a, kind);
                return false;
            }
        }
    }
    return true;
}

__global__ void __launch_bounds__(kBlock) k_group(GroupArgs a) {
    __shared__ int s_ok;
    __shared__ const char* s_src;
    int i = 0;
    while (i + 1 < a.nops && (int)blockIdx.x >= a.op[i].blk0 + a.op[i].nblk) i++;
    const Op op = a.op[i];
    Desc* d = op.d;
    if (!op.recv) {  // ---- send: publish, then wait until the bytes have been read
        if (threadIdx.x == 0) {
            d->src = op.buf;
            d->bytes = op.bytes;
            __hip_atomic_store(&d->seq, op.seq1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            wait_for(&d->ack, op.seq1, a, 1);
        }
        return;
    }
    // ---- receive: wait for the descriptor, copy, acknowledge
    if (threadIdx.x == 0) {
        int ok = wait_for(&d->seq, op.seq1, a, 2) ? 1 : 0;
        if (ok && d->bytes != op.bytes) {  // the two ends disagree about the message
            raise(a, 3);
            ok = 2;  // acknowledged without a copy, so that the sender returns
        }
        s_ok = ok;
        s_src = (const char*)d->src;
    }
    __syncthreads();
    if (s_ok == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every wave: the sender's data, not stale lines of this CU / XCD
    if (s_ok == 1) {
        const int part = (int)blockIdx.x - op.blk0;
        const unsigned long long n8 = op.bytes / 8, per = (n8 + op.nblk - 1) / op.nblk;
        const unsigned long long lo = per * part, hi = lo + per < n8 ? lo + per : n8;
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(s_src);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(op.buf);
        for (unsigned long long k = lo + threadIdx.x; k < hi; k += kBlock) dst[k] = src[k];
        if (part == 0)
            for (unsigned long long k = n8 * 8 + threadIdx.x; k < op.bytes; k += kBlock)
                reinterpret_cast<char*>(op.buf)[k] = s_src[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned before = __hip_atomic_fetch_add(&d->nack, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)before == op.nblk - 1) {
            __hip_atomic_store(&d->nack, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&d->ack, op.seq1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct Group {  // one communicator group: all ranks of one ncclUniqueId
    int n = 0, joined = 0, left = 0;
    int device = 0;
    Desc* desc = nullptr;  // [src][dst][kSlots]
    std::mutex mu;
    std::condition_variable cv;
    bool failed = false;
};

struct Globals {
    std::mutex mu;
    std::map<std::string, std::shared_ptr<Group>> groups;
    unsigned* err = nullptr;  // host-mapped
    unsigned long long next_id = 1;
    long groups_launched = 0, sends = 0, recvs = 0;
};
Globals& G() {
    static Globals g;
    return g;
}

unsigned* err_word() {
    Globals& g = G();
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.err) {
        if (hipHostMalloc((void**)&g.err, 64, hipHostMallocMapped) != hipSuccess) return nullptr;
        std::memset(g.err, 0, 64);
    }
    return g.err;
}

struct Pending {
    Op op;
    hipStream_t stream;
};
thread_local int tl_depth = 0;
thread_local std::vector<Pending> tl_ops;
thread_local ncclComm* tl_comm = nullptr;

}  // namespace

struct ncclComm {
    std::shared_ptr<Group> g;
    int rank = 0, nranks = 1;
    unsigned* abort_flag = nullptr;  // host-mapped
    std::vector<unsigned long long> send_seq, recv_seq;
    unsigned long long timeout_ticks = 0;
};

namespace {

ncclResult_t flush_group() {
    std::vector<Pending> ops;
    ops.swap(tl_ops);
    ncclComm* comm = tl_comm;
    tl_comm = nullptr;
    if (ops.empty()) return ncclSuccess;
    if ((int)ops.size() > kMaxOps) return ncclInvalidUsage;
    GroupArgs a{};
    a.nops = (int)ops.size();
    a.timeout_ticks = comm->timeout_ticks;
    a.err = err_word();
    a.abort_flag = comm->abort_flag;
    if (!a.err) return ncclUnhandledCudaError;
    int blocks = 0;
    for (int i = 0; i < a.nops; i++) {
        if (ops[i].stream != ops[0].stream) return ncclInvalidUsage;  // (RCCL allows it; libpapof never does it)
        a.op[i] = ops[i].op;
        a.op[i].blk0 = blocks;
        blocks += a.op[i].nblk;
    }
    if (hipSetDevice(comm->g->device) != hipSuccess) return ncclUnhandledCudaError;
    hipLaunchKernelGGL(k_group, dim3(blocks), dim3(kBlock), 0, ops[0].stream, a);
    if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
    Globals& g = G();
    std::lock_guard<std::mutex> lk(g.mu);
    g.groups_launched++;
    return ncclSuccess;
}

ncclResult_t enqueue(bool recv, const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm* comm,
                     hipStream_t stream) {
    if (!comm || peer < 0 || peer >= comm->nranks || peer == comm->rank || (!buf && count)) return ncclInvalidArgument;
    size_t es = 0;
    switch (type) {
        case ncclInt8: case ncclUint8: es = 1; break;
        case ncclFloat16: case ncclBfloat16: es = 2; break;
        case ncclInt32: case ncclUint32: case ncclFloat32: es = 4; break;
        case ncclInt64: case ncclUint64: case ncclFloat64: es = 8; break;
        default: return ncclInvalidArgument;
    }
    if (tl_comm && tl_comm != comm) return ncclInvalidUsage;  // one communicator per group is all libpapof does
    tl_comm = comm;
    Pending p{};
    const int src = recv ? peer : comm->rank, dst = recv ? comm->rank : peer;
    unsigned long long& seq = recv ? comm->recv_seq[peer] : comm->send_seq[peer];
    p.op.d = comm->g->desc + ((size_t)src * comm->nranks + dst) * kSlots + (seq % kSlots);
    p.op.buf = const_cast<void*>(buf);
    p.op.bytes = (unsigned long long)count * es;
    p.op.seq1 = ++seq;
    p.op.recv = recv ? 1 : 0;
    p.op.nblk = 1;
    if (recv) {
        unsigned long long nb = (p.op.bytes + kBytesPerCopyBlock - 1) / kBytesPerCopyBlock;
        p.op.nblk = (int)(nb < 1 ? 1 : (nb > kMaxCopyBlocks ? kMaxCopyBlocks : nb));
    }
    p.stream = stream;
    tl_ops.push_back(p);
    {
        Globals& g = G();
        std::lock_guard<std::mutex> lk(g.mu);
        (recv ? g.recvs : g.sends)++;
    }
    if (tl_depth == 0) return flush_group();
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof *id);
    Globals& g = G();
    std::lock_guard<std::mutex> lk(g.mu);
    const unsigned long long v = g.next_id++;
    std::snprintf(id->internal, sizeof id->internal, "fake-rccl-%llu-%lld", v,
                  (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (!err_word()) return ncclUnhandledCudaError;
    const std::string key(id.internal, sizeof id.internal);
    std::shared_ptr<Group> grp;
    {
        Globals& g = G();
        std::lock_guard<std::mutex> lk(g.mu);
        auto it = g.groups.find(key);
        if (it == g.groups.end()) {
            grp = std::make_shared<Group>();
            grp->n = nranks;
            g.groups[key] = grp;
        } else
            grp = it->second;
    }
    auto comm = std::make_unique<ncclComm>();
    {
        std::unique_lock<std::mutex> lk(grp->mu);
        if (grp->n != nranks || grp->failed) return ncclInvalidArgument;
        if (!grp->desc) {  // the first rank to arrive lays the descriptor rings out on ITS current device
            if (hipGetDevice(&grp->device) != hipSuccess) return ncclUnhandledCudaError;
            const size_t bytes = (size_t)nranks * nranks * kSlots * sizeof(Desc);
            if (hipMalloc((void**)&grp->desc, bytes) != hipSuccess || hipMemset(grp->desc, 0, bytes) != hipSuccess ||
                hipDeviceSynchronize() != hipSuccess) {
                grp->failed = true;
                grp->cv.notify_all();
                return ncclUnhandledCudaError;
            }
        }
        grp->joined++;
        grp->cv.notify_all();
        // as the real library: returns when every rank of the group has called in
        if (!grp->cv.wait_for(lk, std::chrono::seconds(60), [&] { return grp->joined >= grp->n || grp->failed; })) {
            grp->failed = true;
            grp->cv.notify_all();
        }
        if (grp->failed) return ncclSystemError;
    }
    comm->g = grp;
    comm->rank = rank;
    comm->nranks = nranks;
    comm->send_seq.assign(nranks, 0);
    comm->recv_seq.assign(nranks, 0);
    if (hipHostMalloc((void**)&comm->abort_flag, 64, hipHostMallocMapped) != hipSuccess) return ncclUnhandledCudaError;
    *comm->abort_flag = 0;
    const char* t = std::getenv("FAKE_RCCL_TIMEOUT_MS");
    const double ms = t && std::atof(t) > 0 ? std::atof(t) : 30000.0;
    comm->timeout_ticks = (unsigned long long)(ms * 1e5);  // 100 MHz
    *out = comm.release();
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclInvalidArgument;
    std::shared_ptr<Group> grp = comm->g;
    bool last = false;
    {
        std::lock_guard<std::mutex> lk(grp->mu);
        last = ++grp->left == grp->n;
    }
    (void)hipSetDevice(grp->device);
    (void)hipDeviceSynchronize();  // (the real library drains the communicator's work too)
    if (last && grp->desc) {
        (void)hipFree(grp->desc);
        grp->desc = nullptr;
        Globals& g = G();
        std::lock_guard<std::mutex> lk(g.mu);
        for (auto it = g.groups.begin(); it != g.groups.end(); ++it)
            if (it->second == grp) {
                g.groups.erase(it);
                break;
            }
    }
    if (comm->abort_flag) (void)hipHostFree(comm->abort_flag);
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) {
    if (!comm) return ncclInvalidArgument;
    __atomic_store_n(comm->abort_flag, 1u, __ATOMIC_SEQ_CST);  // kernels in flight and every later one exit
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->nranks;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
    if (!comm || !rank) return ncclInvalidArgument;
    *rank = comm->rank;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    tl_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (tl_depth <= 0) return ncclInvalidUsage;
    if (--tl_depth > 0) return ncclSuccess;
    return flush_group();
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return enqueue(false, buf, count, type, peer, comm, stream);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return enqueue(true, buf, count, type, peer, comm, stream);
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake rccl)";
        case ncclSystemError: return "system error (fake rccl: a rank never joined)";
        case ncclInvalidArgument: return "invalid argument (fake rccl)";
        case ncclInvalidUsage: return "invalid usage (fake rccl)";
        default: return "error (fake rccl)";
    }
}

// ---- hooks of the tests (not part of the RCCL API) ----
// errors so far in this process: receives / sends that gave up (timeout), byte counts that did not match
unsigned fake_rccl_error_count() {
    unsigned* e = err_word();
    return e ? __atomic_load_n(e, __ATOMIC_SEQ_CST) : 0xffffffffu;
}
void fake_rccl_reset_errors() {
    unsigned* e = err_word();
    if (e)
        for (int i = 0; i < 4; i++) __atomic_store_n(e + i, 0u, __ATOMIC_SEQ_CST);
}
// {errors, sends that gave up, receives that gave up, byte counts that disagreed}
void fake_rccl_error_kinds(unsigned out[4]) {
    unsigned* e = err_word();
    for (int i = 0; i < 4; i++) out[i] = e ? __atomic_load_n(e + i, __ATOMIC_SEQ_CST) : 0xffffffffu;
}
// {group kernels launched, sends, receives}: proof that the messages of a test really went through this library
void fake_rccl_stats(long out[3]) {
    Globals& g = G();
    std::lock_guard<std::mutex> lk(g.mu);
    out[0] = g.groups_launched;
    out[1] = g.sends;
    out[2] = g.recvs;
}
const char* fake_rccl_identity() { return "papof fake rccl: threads of one process on one device, RCCL semantics"; }

}  // extern "C"
