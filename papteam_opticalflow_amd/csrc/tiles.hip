// papteam_opticalflow_amd/csrc/tiles.hip -- ONE frame pair sharded as 2-D tiles over several GPUs (SURVEY.md §8e,
// BASELINE.json configs[4]: "1920x1080 tiled 2x4 across 8 x MI355X, RCCL halo exchange over xGMI each SOR sweep").
//
// Same pipeline as flow_device() (api.hip; reference: OpticalFlow::Coarse2FineFlow, src/OpticalFlow.cpp:735-903) with
// the red-black SOR order (sor.hip), one rank per GPU:
//   * every rank keeps FULL-SIZE planes in global pixel coordinates and launches each per-pixel kernel on a region
//     (Rect): its tile grown by exactly the halo the following stages read.  Cheap read-only inputs are replicated
//     instead of exchanged: both pyramids and the per-level features are built redundantly on every rank (the warp is an
//     unbounded gather of frame-2 features, so they have to be resident everywhere anyway).
//   * what IS exchanged, owner -> needer, packed into one buffer per peer and moved by one RCCL group
//     (ncclSend / ncclRecv on the rank's own stream: no host round trip, no collective that involves non-neighbours):
//       - (du, dv) during a solve.  Ghost zones are S half-sweeps deep: after an exchange (du, dv) are valid on the tile
//         grown by S; half-sweep m of a block then runs on the tile grown by S-1-m (redundant work on a shrinking
//         frame), so one exchange serves S half-sweeps.  Xgmi messages of a few KB are latency-bound, so S trades ~S/2
//         extra pixel rings of arithmetic for 1/S of the exchanges.  Red-black half-sweeps read only the other colour's
//         previous values, hence the tiled result is BIT-IDENTICAL to the one-GPU red-black solve (tested).
//       - (u, v) after each outer iteration: a halo of S+3 pixels (warp S+3 -> 5-tap smoothing, both passes S+1 ->
//         5-point derivatives / phi / Laplacian S -> coefficients on S-1 -> ghost-zone solve);
//       - (u, v) at a level change: the rectangle the bilinear up-sampling of the next level's grown tile reads;
//       - the finished tiles of (u, v, warpI2) to rank 0.
//   * message plans are pure functions of (dims, grid, rank): both ends compute them, nothing is negotiated.
//
// Transports: RCCL (production; librccl is dlopen'ed so that a one-GPU user never needs it, and so that a process
// that already loaded PyTorch's copy shares it) and LOCAL (all ranks are threads of one process on one device and
// messages are device-to-device copies): the same orchestration code, used by the parity tests on a one-GPU box.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is loaded at run time

#include "common.h"
#include "flow_internal.h"

namespace papof {

namespace {

// ---------------------------------------------------------------------------------------------------------
// geometry (host)
// ---------------------------------------------------------------------------------------------------------
struct TileGrid {
    int rows, cols;
    int n() const { return rows * cols; }
};

Rect tile_rect(const TileGrid& g, int rank, int W, int H) {
    const int ty = rank / g.cols, tx = rank - ty * g.cols;
    return Rect{(int)((long long)W * tx / g.cols), (int)((long long)H * ty / g.rows),
                (int)((long long)W * (tx + 1) / g.cols), (int)((long long)H * (ty + 1) / g.rows)};
}

Rect grow(const Rect& r, int d, int W, int H) {  // clamped to the image; an empty tile stays empty
    if (r.empty()) return r;
    return Rect{std::max(0, r.x0 - d), std::max(0, r.y0 - d), std::min(W, r.x1 + d), std::min(H, r.y1 + d)};
}

Rect intersect(const Rect& a, const Rect& b) {
    Rect r{std::max(a.x0, b.x0), std::max(a.y0, b.y0), std::min(a.x1, b.x1), std::min(a.y1, b.y1)};
    if (r.empty()) r = Rect{0, 0, 0, 0};
    return r;
}

// Source rectangle (in the sw x sh plane) that ResizeImage's bilinear sampler (src/ImageProcessing.h:235-253,
// :138-157) reads for the destination pixels of `dst`: x = (j+1)/xr - 1, taps at (int)x and (int)x + 1, clamped.
// One pixel of slack on both sides covers the rounding of the division.
Rect resize_source(const Rect& dst, double xr, double yr, int sw, int sh) {
    if (dst.empty()) return Rect{0, 0, 0, 0};
    const int x0 = (int)std::floor((double)(dst.x0 + 1) / xr - 1) - 1, x1 = (int)std::floor((double)dst.x1 / xr - 1) + 3;
    const int y0 = (int)std::floor((double)(dst.y0 + 1) / yr - 1) - 1, y1 = (int)std::floor((double)dst.y1 / yr - 1) + 3;
    return Rect{std::max(0, x0), std::max(0, y0), std::min(sw, x1), std::min(sh, y1)};
}

// ---------------------------------------------------------------------------------------------------------
// transports
// ---------------------------------------------------------------------------------------------------------
struct Msg {
    int peer;
    double* buf;
    size_t count;  // doubles
};

struct Transport {
    int rank = 0, nranks = 1;
    virtual ~Transport() {}
    // what the transport itself says about the group (RCCL: ncclCommCount / ncclCommUserRank of the communicator)
    virtual int seen(int* n, int* r) const {
        *n = nranks;
        *r = rank;
        return PAPOF_OK;
    }
    // Everything enqueued on h->stream before the call is visible to the sends; everything enqueued after it sees the
    // received data.
    virtual int exchange(papof_handle* h, const std::vector<Msg>& sends, const std::vector<Msg>& recvs) = 0;
    // The exact-order band split (bands_flow) lets a solver kernel of one rank write into the planes and counters of the
    // rank below it: it needs the peers' memory DIRECTLY ADDRESSABLE.  bases(): collective; every rank contributes the base
    // of its arena and of its counter block and receives everybody's, as pointers valid on ITS device.  barrier(): when it
    // returns, everything every rank enqueued before its call has completed.
    virtual int bases(papof_handle*, void*, void*, std::vector<void*>&, std::vector<void*>&) {
        set_last_error_text("this transport cannot address peer memory from a kernel: the exact-order band split runs on the "
                            "LOCAL transport (one process, one device) -- see DESIGN.md 7 for the multi-GPU transport it needs");
        return PAPOF_EINVAL;
    }
    virtual int barrier(papof_handle*) { return PAPOF_EINVAL; }
    virtual bool addresses_peers() const { return false; }  // may a kernel of this rank store into a peer's memory?
    // Host-side launch order of the ranks' solver kernels of one solve (bands_flow): rank r's tasks spin on counters that
    // the rank above publishes, so its kernel is enqueued only after that rank's (turn_wait), and says so itself
    // (turn_done).  With every kernel's producers launched before it, progress does not depend on how the runtime maps the
    // ranks' streams onto hardware queues (a FIFO shared by two ranks' streams could otherwise put a waiter in front of
    // its producer).  Transports whose ranks own a device each need neither.
    virtual int turn_wait(int /*above*/, long /*solve*/) { return PAPOF_OK; }
    virtual void turn_done(long /*solve*/) {}
    // Wait for everything this rank has enqueued on h->stream (the end of a call, a reallocation of the staging buffers).
    virtual int drain(papof_handle* h) {
        PAPOF_HIP(hipStreamSynchronize(h->stream));
        return PAPOF_OK;
    }
};

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

// PAPOF_RCCL_LIB (read at every papof_tiles_unique_id / papof_tiles_create): another library with librccl's point-to-point API.
// The tests hand in tests/fake_rccl/libfake_rccl.so -- the ranks of a group as threads of one process on one device, with
// RCCL's semantics (a group is one kernel on the caller's stream, nothing waits on the host) -- so that THIS transport, the one
// the multi-GPU node runs, is what the multi-rank parity tests exercise on a one-GPU box.  One table of entry points per path.
const RcclApi* rccl() {
    static std::mutex mu;
    static std::map<std::string, RcclApi> apis;  // (node-based: pointers to the entries stay valid)
    const char* const over = std::getenv("PAPOF_RCCL_LIB");
    const std::string key = over && over[0] ? over : "";
    std::lock_guard<std::mutex> lk(mu);
    auto it = apis.find(key);
    if (it == apis.end()) {
        RcclApi a;
        void* lib = nullptr;
        if (!key.empty()) {
            lib = dlopen(key.c_str(), RTLD_NOW | RTLD_LOCAL);
        } else {
            // the soname first: a process that already holds a librccl (PyTorch's) gets that very copy back
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char* n : names)
                if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        }
        if (lib) {
            a.lib = lib;
#define PAPOF_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(lib, name))
            PAPOF_SYM(GetUniqueId, "ncclGetUniqueId");
            PAPOF_SYM(CommInitRank, "ncclCommInitRank");
            PAPOF_SYM(CommDestroy, "ncclCommDestroy");
            PAPOF_SYM(CommAbort, "ncclCommAbort");
            PAPOF_SYM(CommCount, "ncclCommCount");
            PAPOF_SYM(CommUserRank, "ncclCommUserRank");
            PAPOF_SYM(GroupStart, "ncclGroupStart");
            PAPOF_SYM(GroupEnd, "ncclGroupEnd");
            PAPOF_SYM(Send, "ncclSend");
            PAPOF_SYM(Recv, "ncclRecv");
            PAPOF_SYM(GetErrorString, "ncclGetErrorString");
#undef PAPOF_SYM
            if (!(a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv &&
                  a.GetErrorString))
                a.lib = nullptr;
        }
        it = apis.emplace(key, a).first;
    }
    return it->second.lib ? &it->second : nullptr;
}

#define PAPOF_NCCL(api, expr)                                                                    \
    do {                                                                                         \
        ncclResult_t _r = (expr);                                                                \
        if (_r != ncclSuccess) {                                                                 \
            set_last_error_text(std::string(#expr " failed: ") + (api)->GetErrorString(_r));     \
            return PAPOF_EDEVICE;                                                                \
        }                                                                                        \
    } while (0)

struct RcclTransport : Transport {
    const RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    ~RcclTransport() override {
        if (comm) api->CommDestroy(comm);
    }
    int seen(int* n, int* r) const override {
        if (!api->CommCount || !api->CommUserRank || !comm) return PAPOF_EDEVICE;
        PAPOF_NCCL(api, api->CommCount(comm, n));
        PAPOF_NCCL(api, api->CommUserRank(comm, r));
        return PAPOF_OK;
    }
    int barrier(papof_handle*) override { return PAPOF_OK; }  // every rank writes only its own memory: nothing to order
    // A receive whose peer never sends does not end by itself: the stream is polled against a deadline (PAPOF_TILES_TIMEOUT_S,
    // default 120 s per call), and a rank that runs into it ABORTS its communicator (ncclCommAbort: the library's kernels in
    // flight exit), drains what is left and reports PAPOF_ETIMEOUT.  The communicator is unusable afterwards, as RCCL defines.
    bool aborted = false;
    int drain(papof_handle* h) override {
        const char* e = std::getenv("PAPOF_TILES_TIMEOUT_S");
        const double limit = e && std::atof(e) > 0 ? std::atof(e) : 120.0;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; spins++) {
            const hipError_t q = hipStreamQuery(h->stream);
            if (q == hipSuccess) return aborted ? PAPOF_ETIMEOUT : PAPOF_OK;
            if (q != hipErrorNotReady) {
                set_last_error("hipStreamQuery", q, __FILE__, __LINE__);
                return PAPOF_EDEVICE;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) break;
            if (spins < 2000)
                std::this_thread::yield();
            else
                std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        aborted = true;
        set_last_error_text("a peer of the tile group did not answer within PAPOF_TILES_TIMEOUT_S: communicator aborted");
        if (api->CommAbort) api->CommAbort(comm);
        comm = nullptr;  // (ncclCommAbort frees it)
        (void)hipStreamSynchronize(h->stream);  // without an abort entry point this waits for the library's own watchdog
        return PAPOF_ETIMEOUT;
    }
    int exchange(papof_handle* h, const std::vector<Msg>& sends, const std::vector<Msg>& recvs) override {
        if (sends.empty() && recvs.empty()) return PAPOF_OK;
        if (aborted || !comm) return PAPOF_ETIMEOUT;
        PAPOF_NCCL(api, api->GroupStart());
        for (const Msg& m : sends) PAPOF_NCCL(api, api->Send(m.buf, m.count, ncclDouble, m.peer, comm, h->stream));
        for (const Msg& m : recvs) PAPOF_NCCL(api, api->Recv(m.buf, m.count, ncclDouble, m.peer, comm, h->stream));
        PAPOF_NCCL(api, api->GroupEnd());
        return PAPOF_OK;
    }
};

// All ranks are threads of one process (tests on a one-GPU box): a send is posted in a shared table, the receiver copies
// device-to-device.  Two barriers per exchange (posted / consumed); a rank that fails releases the others with an error.
struct LocalGroup {
    int n;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    bool failed = false;
    std::vector<std::vector<Msg>> posted;
    std::vector<void*> arena_base, sync_base;  // bases(): what every rank posted
    std::vector<long> launched;                // solver kernels of the current call each rank has enqueued (turn_wait)
    explicit LocalGroup(int n_) : n(n_), posted(n_), arena_base(n_, nullptr), sync_base(n_, nullptr), launched(n_, 0) {}
    bool barrier() {  // false: the group has failed (a rank bailed out, or a rank never arrived)
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return false;
        const long gen = generation;
        if (++arrived == n) {
            arrived = 0;
            generation++;
            cv.notify_all();
            return true;
        }
        if (!cv.wait_for(lk, std::chrono::seconds(30), [&] { return generation != gen || failed; })) failed = true;
        if (failed) cv.notify_all();
        return !failed;
    }
    void fail() {
        std::lock_guard<std::mutex> lk(mu);
        failed = true;
        cv.notify_all();
    }
};

struct LocalTransport : Transport {
    std::shared_ptr<LocalGroup> g;
    // PAPOF_BANDS_STAGED=1: behave like a transport without peer addressing, so that the staged protocol the RCCL transport
    // runs (bands_flow) is exercised on the one-GPU box
    bool addresses_peers() const override { return !(std::getenv("PAPOF_BANDS_STAGED") && std::atoi(std::getenv("PAPOF_BANDS_STAGED"))); }
    int barrier(papof_handle* h) override {
        if (hipStreamSynchronize(h->stream) != hipSuccess) {
            g->fail();
            return PAPOF_EDEVICE;
        }
        return g->barrier() ? PAPOF_OK : PAPOF_ETIMEOUT;
    }
    int bases(papof_handle* h, void* arena, void* sync, std::vector<void*>& arenas, std::vector<void*>& syncs) override {
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->arena_base[rank] = arena;
            g->sync_base[rank] = sync;
            g->launched[rank] = 0;
        }
        PAPOF_TRY(barrier(h));  // all posted (same process, same device: the pointers are usable as they are)
        arenas = g->arena_base;
        syncs = g->sync_base;
        return g->barrier() ? PAPOF_OK : PAPOF_ETIMEOUT;  // nobody re-posts before everybody has read
    }
    int turn_wait(int above, long solve) override {
        std::unique_lock<std::mutex> lk(g->mu);
        if (!g->cv.wait_for(lk, std::chrono::seconds(30), [&] { return g->launched[above] > solve || g->failed; }))
            g->failed = true;
        if (g->failed) g->cv.notify_all();
        return g->failed ? PAPOF_ETIMEOUT : PAPOF_OK;
    }
    void turn_done(long solve) override {
        std::lock_guard<std::mutex> lk(g->mu);
        g->launched[rank] = solve + 1;
        g->cv.notify_all();
    }
    int exchange(papof_handle* h, const std::vector<Msg>& sends, const std::vector<Msg>& recvs) override {
        if (hipStreamSynchronize(h->stream) != hipSuccess) {
            g->fail();
            return PAPOF_EDEVICE;
        }
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->posted[rank] = sends;
        }
        if (!g->barrier()) return PAPOF_ETIMEOUT;
        int rc = PAPOF_OK;
        for (const Msg& m : recvs) {
            const Msg* src = nullptr;
            for (const Msg& s : g->posted[m.peer])
                if (s.peer == rank) src = &s;
            if (!src || src->count != m.count) {  // the two ends disagree about the plan
                rc = PAPOF_EINVAL;
                break;
            }
            if (hipMemcpyAsync(m.buf, src->buf, m.count * sizeof(double), hipMemcpyDeviceToDevice, h->stream) !=
                hipSuccess)
                rc = PAPOF_EDEVICE;
        }
        if (rc == PAPOF_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = PAPOF_EDEVICE;
        if (rc != PAPOF_OK) {
            g->fail();
            return rc;
        }
        if (!g->barrier()) return PAPOF_ETIMEOUT;  // nobody repacks its send buffer before every reader is done
        return PAPOF_OK;
    }
};

// ---------------------------------------------------------------------------------------------------------
// owner -> needer exchange of rectangles of full-size planes
// ---------------------------------------------------------------------------------------------------------
constexpr int kMaxMsgs = 24, kMaxPlanes = 2;
struct MsgTable {
    int n, nplanes, W;
    size_t plane_stride;
    double* planes[kMaxPlanes];
    Rect r[kMaxMsgs];
    unsigned long long off[kMaxMsgs];  // doubles from the start of the staging buffer
};

template <bool PACK>
__global__ void k_rects(MsgTable t, double* __restrict__ buf) {
    const int m = blockIdx.z / t.nplanes, p = blockIdx.z - m * t.nplanes;
    const Rect r = t.r[m];
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= r.w() || y >= r.h()) return;
    double* cell = t.planes[p] + (size_t)(r.y0 + y) * t.W + (r.x0 + x);
    double* slot = buf + t.off[m] + ((size_t)p * r.h() + y) * r.w() + x;
    if (PACK)
        *slot = *cell;
    else
        *cell = *slot;
}

// The cells that cross a cut of the exact-order band split, as one message (staged protocol of bands_flow): for sweep k and
// position p the 16-byte cell k >> 1 of block `band` at parity k & 1 of banded (du, dv) planes (sor.hip: ExactArgs) <->
// msg[k * npos + p].  PACK reads the producer's outbox planes, else the cells go into the consumer's planes.
template <bool PACK>
__global__ void k_cut_cells(double2* __restrict__ planes, double2* __restrict__ msg, int nb, int npos, int band, int k0, int K) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, k = k0 + blockIdx.y;  // sweeps k0 .. K-1 of the solve (one chunk)
    if (p >= npos || k >= K) return;
    const size_t cell = (size_t)(k & 1) * npos * nb * kLanes + ((size_t)p * nb + band) * kLanes + (size_t)(k >> 1);
    if (PACK)
        msg[(size_t)k * npos + p] = planes[cell];
    else
        planes[cell] = msg[(size_t)k * npos + p];
}
// ... and the producer's progress counters as the consumer then finds them: every sweep of the band above complete
__global__ void k_cut_counters(unsigned* __restrict__ prog, int nb, int band, int k0, int K, unsigned steps) {
    const int k = k0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) prog[((size_t)k * nb + band) * 32] = steps;
}

}  // namespace

}  // namespace papof

using namespace papof;

struct papof_tiles {
    papof_handle* h = nullptr;
    TileGrid grid{1, 1};
    int halo = 10;  // S: ghost-zone depth in half-sweeps
    std::unique_ptr<Transport> tp;
    double *send_stage = nullptr, *recv_stage = nullptr;  // inside the handle's arena, carved per call (stage_carve)
    size_t stage_cap = 0;  // doubles, each
    long exchanges = 0;    // statistics of the last call
    size_t exchanged_bytes = 0;
};

namespace papof {
namespace {

// The packed messages of an exchange live in two buffers carved from the handle's ARENA at the start of every call
// (stage_carve): nothing is allocated or freed while a call is in flight.  [Round 4: they used to be grown with hipFree /
// hipMalloc in the middle of a call.  hipFree waits for the whole device; harmless with one rank per GPU but for the stall --
// and a deadlock as soon as ranks share a device and a peer's receive waits on this rank's next launch, which is how the
// RCCL stand-in of tests/fake_rccl found it.]
size_t stage_doubles(int H, int W, int C) { return (size_t)H * W * (size_t)std::max(C, 2) + 4096; }  // the final gather is the largest message
int stage_carve(papof_tiles& t, Arena& A, int H, int W, int C) {
    t.stage_cap = stage_doubles(H, W, C);
    t.send_stage = A.f64(t.stage_cap);
    t.recv_stage = A.f64(t.stage_cap);
    if (!t.send_stage || !t.recv_stage) {
        t.stage_cap = 0;
        return PAPOF_ENOMEM;
    }
    return PAPOF_OK;
}
int ensure_stage(papof_tiles& t, size_t doubles) {
    if (doubles <= t.stage_cap) return PAPOF_OK;
    set_last_error_text("tile exchange: a message plan exceeds the staging buffers of the call");
    return PAPOF_EINVAL;
}

using RectOf = std::function<Rect(int)>;

// Every rank r OWNS own(r) (disjoint) and NEEDS need(r) of the given planes (row length W, `plane_stride` doubles
// apart... each plane is its own pointer).  After the call need(me) is valid on this rank.
int exchange_planes(papof_tiles& t, double* const* planes, int nplanes, int W, const RectOf& own, const RectOf& need) {
    const int me = t.tp->rank, n = t.tp->nranks;
    if (n == 1) return PAPOF_OK;
    if (nplanes > kMaxPlanes) return PAPOF_EINVAL;
    MsgTable st{}, rt{};
    st.nplanes = rt.nplanes = nplanes;
    st.W = rt.W = W;
    for (int p = 0; p < nplanes; p++) st.planes[p] = rt.planes[p] = planes[p];
    std::vector<Msg> sends, recvs;
    size_t soff = 0, roff = 0;
    const Rect mine = own(me), want = need(me);
    for (int r = 0; r < n; r++) {
        if (r == me) continue;
        const Rect out = intersect(mine, need(r)), in = intersect(own(r), want);
        if (!out.empty()) {
            if (st.n == kMaxMsgs) return PAPOF_EINVAL;
            st.r[st.n] = out;
            st.off[st.n++] = soff;
            const size_t c = (size_t)out.w() * out.h() * nplanes;
            sends.push_back(Msg{r, nullptr, c});
            soff += c;
        }
        if (!in.empty()) {
            if (rt.n == kMaxMsgs) return PAPOF_EINVAL;
            rt.r[rt.n] = in;
            rt.off[rt.n++] = roff;
            const size_t c = (size_t)in.w() * in.h() * nplanes;
            recvs.push_back(Msg{r, nullptr, c});
            roff += c;
        }
    }
    PAPOF_TRY(ensure_stage(t, std::max(soff, roff)));
    for (size_t i = 0; i < sends.size(); i++) sends[i].buf = t.send_stage + st.off[i];
    for (size_t i = 0; i < recvs.size(); i++) recvs[i].buf = t.recv_stage + rt.off[i];
    const auto launch_dims = [](const MsgTable& m) {
        int mw = 1, mh = 1;
        for (int i = 0; i < m.n; i++) {
            mw = std::max(mw, m.r[i].w());
            mh = std::max(mh, m.r[i].h());
        }
        return dim3((mw + 63) / 64, (mh + 3) / 4, m.n * m.nplanes);
    };
    hipStream_t s = t.h->stream;
    if (st.n) hipLaunchKernelGGL(k_rects<true>, launch_dims(st), dim3(64, 4), 0, s, st, t.send_stage);
    PAPOF_HIP(hipGetLastError());
    PAPOF_TRY(t.tp->exchange(t.h, sends, recvs));
    if (rt.n) hipLaunchKernelGGL(k_rects<false>, launch_dims(rt), dim3(64, 4), 0, s, rt, t.recv_stage);
    PAPOF_HIP(hipGetLastError());
    t.exchanges++;
    t.exchanged_bytes += (soff + roff) * sizeof(double);
    return PAPOF_OK;
}

// The call, from one rank's point of view.  Collective: every rank of the group calls it with the same arguments
// (each with its own device-resident copy of the two frames); rank 0 receives the assembled results.
int tiles_flow(papof_tiles& t, const double* d_im1, const double* d_im2, int H, int W, int C, int levels,
               const papof_params& P, double* d_vx, double* d_vy, double* d_warp, double* timing) {
    papof_handle* h = t.h;
    PAPOF_TRY(check_params(P, levels));
    if (P.sor_mode != PAPOF_SOR_REDBLACK || P.n_inner != 1) {
        set_last_error_text("tiled solves use the red-black sweep order (sor_mode = PAPOF_SOR_REDBLACK) with n_inner = 1");
        return PAPOF_EINVAL;
    }
    double ratio = P.ratio;
    if (ratio > 0.98 || ratio < 0.4) ratio = 0.75;
    std::vector<Level> L;
    std::vector<PyrPlan> plan;
    PAPOF_TRY(pyramid_plan(H, W, P.ratio, levels, L, plan));
    const int n_sor_max = P.n_sor + (levels - 1) * P.n_sor_per_level;
    h->seq.valid = false;
    PAPOF_TRY(ensure_arena(h, arena_bytes_for(H, W, C, levels, n_sor_max, P.ratio) + (size_t)H * W * C * sizeof(double) +
                              2 * (stage_doubles(H, W, C) + 64) * sizeof(double)));
    Arena& A = h->arena;
    A.off = 0;
    A.overflow = false;
    PAPOF_TRY(stage_carve(t, A, H, W, C));
    h->events_used = 0;
    t.exchanges = 0;
    t.exchanged_bytes = 0;
    const size_t np0 = (size_t)H * W;
    const int fc = feature_channels(C);
    const int me = t.tp->rank, S = t.halo, HU = S + 3;
    double tm[PAPOF_N_TIMERS + 1];  // + kTimerFused (flow_internal.h: PhaseClock::collect)
    std::memset(tm, 0, sizeof tm);
    PhaseClock total{h, true}, sorclk{h, true};
    total.phase(PAPOF_T_TOTAL);

    // replicated, read-only: both pyramids (GaussianPyramid::ConstructPyramidLevels, src/GaussianPyramid.cpp:79-108)
    for (int i = 0; i < levels; i++) {
        L[i].p1 = A.f64((size_t)L[i].w * L[i].h * C);
        L[i].p2 = A.f64((size_t)L[i].w * L[i].h * C);
    }
    double* tmp_a = A.f64(np0 * C);
    double* tmp_b = A.f64(np0 * C);
    if (A.overflow) return PAPOF_ENOMEM;
    // Replicated, flow-independent work -- both pyramids and the features of every level (src/OpticalFlow.cpp:757-758,
    // :797-798) -- goes to the handle's PREPARATION stream, coarsest level first, beside the coarse levels' solves on the
    // main stream (as flow_device does): on N GPUs the per-rank solver work shrinks with N, this part does not (rocprof,
    // one rank: 1.1 of 6.3 ms), so it must not sit on the critical path.  One event per level orders the two streams.
    std::vector<double*> F1(levels), F2(levels);
    for (int k = 0; k < levels; k++) {
        const size_t n = (size_t)L[k].w * L[k].h * fc;
        F1[k] = A.f64(n);
        F2[k] = A.f64(n);
    }
    if (A.overflow) return PAPOF_ENOMEM;
    const bool overlap = h->overlap_prep && h->prep_stream != nullptr;
    while (h->sync_events.size() < (size_t)levels + 2) {
        hipEvent_t e;
        PAPOF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->sync_events.push_back(e);
    }
    {
        hipStream_t const main_stream = h->stream;
        if (overlap) {  // after whatever the caller queued on the main stream (the frame uploads)
            PAPOF_HIP(hipEventRecord(h->sync_events[levels + 1], main_stream));
            PAPOF_HIP(hipStreamWaitEvent(h->prep_stream, h->sync_events[levels + 1], 0));
            h->stream = h->prep_stream;  // the launch wrappers enqueue on h->stream
        }
        int rc = hwc_to_planar(h, d_im1, L[0].p1, H, W, C);
        if (rc == PAPOF_OK) rc = hwc_to_planar(h, d_im2, L[0].p2, H, W, C);
        if (rc == PAPOF_OK) rc = build_pyramid(h, L, plan, C, false, tmp_a, tmp_b);
        if (rc == PAPOF_OK) rc = build_pyramid(h, L, plan, C, true, tmp_a, tmp_b);
        for (int k = levels - 1; k >= 0 && rc == PAPOF_OK; k--) {
            rc = im2feature(h, L[k].p1, F1[k], L[k].h, L[k].w, C);
            if (rc == PAPOF_OK) rc = im2feature(h, L[k].p2, F2[k], L[k].h, L[k].w, C);
            if (rc == PAPOF_OK && overlap && hipEventRecord(h->sync_events[k], h->stream) != hipSuccess) rc = PAPOF_EDEVICE;
        }
        h->stream = main_stream;
        if (rc != PAPOF_OK) {
            if (overlap) hipStreamSynchronize(h->prep_stream);
            return rc;
        }
    }

    double* warp = A.f64(np0 * fc);
    double* u = A.f64(np0);
    double* v = A.f64(np0);
    double* u2 = A.f64(np0);
    double* v2 = A.f64(np0);
    double* im1s = A.f64(np0 * fc);
    double* tmp = A.f64(np0 * fc);
    double* blend = A.f64(np0 * fc);
    double* imdt = A.f64(np0 * fc);
    double* phi = A.f64(np0);
    double* warp_hwc = A.f64(np0 * C);  // this rank's tile of warpI2, in place in a full-size interleaved image
    SorPlanes sp{};
    PAPOF_TRY(sor_alloc_planes(A, H, W, PAPOF_SOR_REDBLACK, n_sor_max, sp));
    if (A.overflow) return PAPOF_ENOMEM;
    // the solver reads the weights of the LEFT and UPPER neighbours too: one ring beyond the cells it updates, i.e.
    // beyond the region the assembly wrote its copy of phi for -- so it gets the phi plane itself (valid one ring wider)
    SorPlanes spt = sp;
    spt.phi = phi;
    const Taps g = smooth5_taps();

    int pw = 0, ph = 0;
    for (int k = levels - 1; k >= 0; k--) {
        const int lw = L[k].w, lh = L[k].h;
        const size_t np = (size_t)lw * lh;
        const RectOf own = [&](int r) { return tile_rect(t.grid, r, lw, lh); };
        const Rect T = own(me);
        const Rect Tu = grow(T, HU, lw, lh);  // where this rank keeps (u, v) and the warped features valid
        const double *f1 = F1[k], *f2 = F2[k];  // replicated, prepared on the other stream
        if (overlap) PAPOF_HIP(hipStreamWaitEvent(h->stream, h->sync_events[k], 0));
        if (k == levels - 1) {  // :801-806
            PAPOF_HIP(hipMemsetAsync(u, 0, np * sizeof(double), h->stream));
            PAPOF_HIP(hipMemsetAsync(v, 0, np * sizeof(double), h->stream));
            PAPOF_HIP(hipMemcpyAsync(warp, f2, np * fc * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        } else {  // :809-814: up-sample the coarser flow onto the grown tile, then warp
            const double xr = (double)lw / pw, yr = (double)lh / ph, inv = 1 / ratio;
            const int cw = pw, chh = ph;
            const RectOf own_c = [&](int r) { return tile_rect(t.grid, r, cw, chh); };
            const RectOf need_c = [&](int r) {
                return resize_source(grow(tile_rect(t.grid, r, lw, lh), HU, lw, lh), xr, yr, cw, chh);
            };
            double* uv[2] = {u, v};
            PAPOF_TRY(exchange_planes(t, uv, 2, cw, own_c, need_c));
            PAPOF_TRY(resize(h, u, u2, ph, pw, 1, lh, lw, xr, yr, true, inv, &Tu));
            PAPOF_TRY(resize(h, v, v2, ph, pw, 1, lh, lw, xr, yr, true, inv, &Tu));
            std::swap(u, u2);
            std::swap(v, v2);
            PAPOF_TRY(warp_bilinear(h, f1, f2, u, v, warp, lh, lw, fc, &Tu));
        }
        // OpticalFlow::SmoothFlowSOR (src/OpticalFlow.cpp:238-536) on the tile
        const int n_outer = P.n_outer + k * P.n_outer_per_level, n_sor = P.n_sor + k * P.n_sor_per_level;
        const Rect Rim = grow(T, S + 1, lw, lh);                    // smoothed / blended features
        const Rect Rh{Rim.x0, Tu.y0, Rim.x1, Tu.y1};                // horizontal pass: two more rows for the vertical one
        const Rect Rphi = grow(T, S, lw, lh), Rsys = grow(T, S - 1, lw, lh);
        const RectOf need_d = [&](int r) { return grow(tile_rect(t.grid, r, lw, lh), S, lw, lh); };
        const RectOf need_u = [&](int r) { return grow(tile_rect(t.grid, r, lw, lh), HU, lw, lh); };
        PAPOF_TRY(filter_h(h, f1, tmp, lh, lw, fc, g, &Rh));
        PAPOF_TRY(filter_v(h, tmp, im1s, lh, lw, fc, g, &Rim));
        for (int count = 0; count < n_outer; count++) {
            PAPOF_TRY(filter_h(h, warp, tmp, lh, lw, fc, g, &Rh));
            PAPOF_TRY(smooth_v_blend(h, tmp, im1s, blend, imdt, lh, lw, fc, &Rim));
            PAPOF_TRY(compute_phi(h, u, v, nullptr, phi, lh, lw, &Rphi));
            PAPOF_TRY(assemble_system(h, blend, imdt, phi, u, v, lh, lw, fc, P.alpha, P.omega, sp, nullptr, nullptr,
                                      nullptr, &Rsys));
            sorclk.phase(PAPOF_T_PHASE5_SOR);
            // The solve, from du = dv = 0 (src/OpticalFlow.cpp:452-453): periods of S half-sweeps between exchanges.  A
            // period of s half-sweeps is one or more launches of the LDS-tiled, temporally blocked kernel (sor.hip): a
            // launch of depth g that must deliver region R reads R grown by g, so the launches of a period deliver the
            // tile grown by s - (half-sweeps done) -- the shrinking frame of the ghost-zone scheme, g rings at a time.
            // Launches alternate between the two pairs of (du, dv) planes; an exchange fills the ghost ring of the pair
            // the next launch reads.  The first period reads no unknowns at all.
            const int n_half = 2 * n_sor;
            const int gmax = std::max(2, sor_blocked_depth(h, PAPOF_SOR_REDBLACK, T.h(), T.w()));
            const double *su = nullptr, *sv = nullptr;
            for (int hs = 0; hs < n_half;) {
                const int s = std::min(S, n_half - hs);
                for (int done = 0; done < s;) {
                    const int g = std::min(gmax, s - done);
                    const Rect R = grow(T, s - done - g, lw, lh);
                    double *du = su == sp.du ? sp.du2 : sp.du, *dv = sv == sp.dv ? sp.dv2 : sp.dv;
                    PAPOF_TRY(sor_blocked_launch(h, spt, lh, lw, P.alpha, P.omega, PAPOF_SOR_REDBLACK, g, hs + done, R, su, sv,
                                                 du, dv));
                    su = du;
                    sv = dv;
                    done += g;
                }
                hs += s;
                if (hs < n_half) {
                    double* dd[2] = {const_cast<double*>(su), const_cast<double*>(sv)};
                    PAPOF_TRY(exchange_planes(t, dd, 2, lw, own, need_d));
                }
            }
            sorclk.phase(-1);
            SorPlanes spc = sp;  // the pair of planes that holds the increments now
            spc.du = const_cast<double*>(su);
            spc.dv = const_cast<double*>(sv);
            PAPOF_TRY(update_flow(h, spc, u, v, lh, lw, T));  // :513-514
            double* uv[2] = {u, v};
            PAPOF_TRY(exchange_planes(t, uv, 2, lw, own, need_u));
            PAPOF_TRY(warp_bilinear(h, f1, f2, u, v, warp, lh, lw, fc, &Tu));  // :516
        }
        pw = lw;
        ph = lh;
    }

    // src/OpticalFlow.cpp:841-842: bicubic warp of the ORIGINAL frame 2 on the tile, then everything to rank 0
    {
        double* gx = A.f64(np0 * C);
        double* gy = A.f64(np0 * C);
        double* gxy = A.f64(np0 * C);
        if (A.overflow) return PAPOF_ENOMEM;
        const Taps c3 = central3_taps();
        PAPOF_TRY(filter_h(h, L[0].p2, gx, H, W, C, c3));
        PAPOF_TRY(filter_v(h, L[0].p2, gy, H, W, C, c3));
        PAPOF_TRY(filter_v(h, gx, gxy, H, W, C, c3));
        const Rect T = tile_rect(t.grid, me, W, H);
        PAPOF_TRY(bicubic_warp(h, L[0].p1, L[0].p2, gx, gy, gxy, u, v, warp_hwc, H, W, C, &T));
        const RectOf own = [&](int r) { return tile_rect(t.grid, r, W, H); };
        const RectOf need = [&](int r) { return r == 0 ? Rect{0, 0, W, H} : Rect{0, 0, 0, 0}; };
        double* uv[2] = {u, v};
        PAPOF_TRY(exchange_planes(t, uv, 2, W, own, need));
        const RectOf own_c = [&](int r) {
            const Rect q = tile_rect(t.grid, r, W, H);
            return Rect{q.x0 * C, q.y0, q.x1 * C, q.y1};
        };
        const RectOf need_c = [&](int r) { return r == 0 ? Rect{0, 0, W * C, H} : Rect{0, 0, 0, 0}; };
        double* wp[1] = {warp_hwc};
        PAPOF_TRY(exchange_planes(t, wp, 1, W * C, own_c, need_c));
        if (me == 0) {
            if (!d_vx || !d_vy || !d_warp) return PAPOF_EINVAL;
            PAPOF_HIP(hipMemcpyAsync(d_vx, u, np0 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            PAPOF_HIP(hipMemcpyAsync(d_vy, v, np0 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            PAPOF_HIP(hipMemcpyAsync(d_warp, warp_hwc, np0 * C * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        }
    }
    total.phase(-1);
    PAPOF_TRY(t.tp->drain(h));
    if (overlap) PAPOF_HIP(hipStreamSynchronize(h->prep_stream));
    if (total.err != PAPOF_OK || sorclk.err != PAPOF_OK) return PAPOF_EDEVICE;
    sorclk.collect(tm);
    total.collect(tm);
    if (timing) std::memcpy(timing, tm, PAPOF_N_TIMERS * sizeof(double));
    return PAPOF_OK;
}


// =================================================================================================
// EXACT-ORDER split of one frame pair over the ranks: horizontal ranges of SOLVER BANDS (BASELINE.json configs[4] with the
// reference's own sweep order, src/OpticalFlow.cpp:458-505 -- the only way the multi-rank result can meet the 1e-4 parity
// bar, since any red-black result is 1e-2 away: SURVEY.md F1).  Results are bit-identical to papof_flow_device().
//
//   * The solver's bands are time-skewed (band b owns rows 62b - k .. 62b - k + 61 at sweep k, sor.hip), so the cut
//     between two ranks climbs one row per sweep and a band depends on its upper neighbour only.  Rank g runs the bands
//     [B_g, B_g+1) of every solve with the plain one-sweep-per-wave kernel in its SPLIT form (sor.hip: ExactArgs): the one
//     edge that crosses a cut -- lane 62 of band B_g+1 - 1, one 16-byte cell per step -- is stored by the producer straight
//     into per-sweep inbox cells inside the planes of the rank below, and its progress counter is published there too.
//     Nothing is reused within a solve, so no write-after-read edge leads back across the cut; the consumer polls and
//     loads LOCAL memory only.  The solver kernels of all ranks are in flight at the same time and pipeline exactly as
//     the bands of one launch do.
//   * Everything else is per-pixel work on ROW RANGES of full-size planes (the kernels of kernels.hip take a Rect or a row
//     range, as in the 2-D tiles above): rank g assembles the coefficient rows its bands touch over all sweeps,
//     [62 B_g - K, 62 B_g+1), from blend / imdt rows +-2, warped rows +-4, phi one row more above; replicated read-only
//     inputs (pyramids, features) are built on every rank.
//   * After a solve rank g holds the FINAL increments of the rows [62 B_g - (K-1), 62 B_g+1 - (K-1)) -- a partition of
//     the plane -- adds them to (u, v) there, and the ranks exchange the rows of (u, v) the next iteration's stages read
//     beyond their own (K + 3 rows from below, 5 from above): ONE exchange per outer iteration (the red-black tiles need 6),
//     21 per 1080p config-4 pair.
// Needs directly addressable peer memory (Transport::bases): the LOCAL transport.  [A multi-GPU transport for it must map
// the neighbour's planes and counters (hipIpc / peer access, fine-grained memory, system-scope stores): DESIGN.md 7.]
// =================================================================================================
struct BandSplit {
    int n, nb, K, lh;
    int B(int g) const { return nb >= n ? (int)((long long)g * nb / n) : std::min(g, nb); }
    static int clampi(int x, int hi) { return x < 0 ? 0 : (x > hi ? hi : x); }
    bool has(int g) const { return B(g + 1) > B(g); }
    // rows whose final (du, dv) of a solve rank g holds: a partition of [0, lh)
    void final_rows(int g, int& y0, int& y1) const {
        if (!has(g)) {
            y0 = y1 = 0;
            return;
        }
        y0 = B(g) == 0 ? 0 : clampi(kBandRows * B(g) - (K - 1), lh);
        y1 = B(g + 1) == nb ? lh : clampi(kBandRows * B(g + 1) - (K - 1), lh);
    }
    // coefficient rows the tasks of rank g touch over all sweeps (lane 0 of the first band at the last sweep .. lane 62 of
    // the last band at sweep 0)
    void coef_rows(int g, int& y0, int& y1) const {
        if (!has(g)) {
            y0 = y1 = 0;
            return;
        }
        y0 = clampi(kBandRows * B(g) - K, lh);
        y1 = clampi(kBandRows * B(g + 1), lh);
    }
};

int bands_flow(papof_tiles& t, const double* d_im1, const double* d_im2, int H, int W, int C, int levels,
               const papof_params& P, double* d_vx, double* d_vy, double* d_warp, double* timing) {
    papof_handle* h = t.h;
    PAPOF_TRY(check_params(P, levels));
    const int n_sor_max = P.n_sor + (levels - 1) * P.n_sor_per_level;
    if (P.sor_mode != PAPOF_SOR_EXACT || P.n_inner != 1 || P.interpolation != PAPOF_INTERP_BILINEAR ||
        P.noise_model != PAPOF_NOISE_LAPLACIAN || n_sor_max > 128 || !h->use_dpp) {
        set_last_error_text("the exact-order band split takes the default branches with n_inner = 1 and at most 128 sweeps");
        return PAPOF_EINVAL;
    }
    double ratio = P.ratio;
    if (ratio > 0.98 || ratio < 0.4) ratio = 0.75;
    std::vector<Level> L;
    std::vector<PyrPlan> plan;
    PAPOF_TRY(pyramid_plan(H, W, P.ratio, levels, L, plan));
    h->seq.valid = false;
    PAPOF_TRY(ensure_arena(h, arena_bytes_for(H, W, C, levels, n_sor_max, P.ratio) + (size_t)H * W * C * sizeof(double) +
                              (size_t)(t.tp->nranks + 2) * kLapMaxSlots * 8 * sizeof(unsigned) +  // + the guard's flags of all ranks
                              2 * (stage_doubles(H, W, C) + 64) * sizeof(double)));
    Arena& A = h->arena;
    A.off = 0;
    A.overflow = false;
    PAPOF_TRY(stage_carve(t, A, H, W, C));
    h->events_used = 0;
    h->sor_log.clear();
    t.exchanges = 0;
    t.exchanged_bytes = 0;
    const size_t np0 = (size_t)H * W;
    const int fc = feature_channels(C);
    const int me = t.tp->rank, n = t.tp->nranks;
    double tm[PAPOF_N_TIMERS + 1];  // + kTimerFused (flow_internal.h: PhaseClock::collect)
    std::memset(tm, 0, sizeof tm);
    PhaseClock total{h, true}, sorclk{h, true};
    total.phase(PAPOF_T_TOTAL);

    // ---- identical allocation sequence on every rank: a buffer sits at the same arena offset everywhere, so a peer's
    // pointer is its arena base + my offset
    for (int i = 0; i < levels; i++) {
        L[i].p1 = A.f64((size_t)L[i].w * L[i].h * C);
        L[i].p2 = A.f64((size_t)L[i].w * L[i].h * C);
    }
    double* tmp_a = A.f64(np0 * C);
    double* tmp_b = A.f64(np0 * C);
    // ---- the Laplacian-noise guard (api.hip: LapGuard; src/OpticalFlow.cpp:399-400).  This path runs the OPTIMISTIC pass only,
    // so it must PROVE that the guard cannot have tripped, or refuse: behind every update each rank checks every pixel of the
    // rows it owns -- a witness (|Im1 - warpIm2| >= 2e-20 x pixels), and whether there is a valid sample at all -- and at the end
    // of the call the flags of all ranks are gathered: an estimate is proven when SOME rank has a witness or NO rank has a valid
    // sample (the constant 0.001).  A consulted estimate without a proof ends the call with PAPOF_EINVAL on every rank: such a
    // pair (values scaled to ~1e-21) belongs on the one-GPU call, which has the exact pass.
    long lap_slots = 0;
    for (int k = 0; k < levels; k++) lap_slots += P.n_outer + (long)k * P.n_outer_per_level;
    const bool lap_on = h->lap_guard;
    if (lap_on && (fc > 8 || lap_slots > kLapMaxSlots / 2)) {  // it proves or refuses: never a call without the check
        set_last_error_text("the band split cannot prove the reference's Laplacian-noise guard (src/OpticalFlow.cpp:399-400) for "
                            "this channel count / number of outer iterations: run the pair on one GPU (papof_flow*)");
        return PAPOF_EINVAL;
    }
    const size_t lap_words = (size_t)std::max<long>(lap_slots, 1) * 16;       // per rank: 8 witness + 8 valid words per slot
    unsigned* lap_mine = lap_on ? reinterpret_cast<unsigned*>(A.f64(lap_words / 2)) : nullptr;  // [slot][0..7] witness, [slot][8..15] valid
    unsigned* lap_all = lap_on ? reinterpret_cast<unsigned*>(A.f64(lap_words / 2 * (size_t)n)) : nullptr;  // every rank's, by rank
    constexpr unsigned kLapSet = 1u;  // (the flags are cleared by every call: a set flag needs no pass number here)
    if (lap_on) PAPOF_HIP(hipMemsetAsync(lap_mine, 0, lap_words * sizeof(unsigned), h->stream));
    long lap_slot = 0;
    // the default branches' one-kernel form of warp .. linear system (kernels.hip: k_flow_system), on each rank's rows
    const char* const fused_sw = std::getenv("PAPOF_FUSED_SYSTEM");
    const bool fused = !(fused_sw && fused_sw[0] == '0') && (fc == 5 || fc == 3);
    std::vector<double*> F1(levels), F2(levels);
    for (int k = 0; k < levels; k++) {
        const size_t nn = (size_t)L[k].w * L[k].h * fc;
        F1[k] = A.f64(nn);
        F2[k] = A.f64(nn);
    }
    double* warp = A.f64(np0 * fc);
    double* u = A.f64(np0);
    double* v = A.f64(np0);
    double* u2 = A.f64(np0);
    double* v2 = A.f64(np0);
    double* im1s = A.f64(np0 * fc);
    double* tmp = A.f64(np0 * fc);
    double* blend = A.f64(np0 * fc);
    double* imdt = A.f64(np0 * fc);
    double* phi = A.f64(np0);
    double* warp_hwc = A.f64(np0 * C);
    double* gx = A.f64(np0 * C);
    double* gy = A.f64(np0 * C);
    double* gxy = A.f64(np0 * C);
    SorPlanes sp{};
    PAPOF_TRY(sor_alloc_planes(A, H, W, PAPOF_SOR_EXACT, n_sor_max, sp));
    if (A.overflow) return PAPOF_ENOMEM;

    // ---- progress counters of every solve of the call (plain layout), cleared once; then the peers' bases
    struct LevelCounters {
        size_t off, per;
    };
    std::vector<LevelCounters> LP(levels);
    size_t prog_total = 0;
    for (int k = 0; k < levels; k++) {
        const int Kk = P.n_sor + k * P.n_sor_per_level;
        LP[k].per = (size_t)skew_dims(L[k].h, L[k].w, Kk, 1, 1).nb * Kk * 32;
        LP[k].off = prog_total;
        prog_total += LP[k].per * (size_t)(P.n_outer + k * P.n_outer_per_level);
        if ((long long)skew_dims(L[k].h, L[k].w, Kk, 1, 1).nb * Kk > 2048) {  // all ranks' tasks of a solve are in flight together
            set_last_error_text("solve too large for the co-resident band split");
            return PAPOF_EINVAL;
        }
    }
    PAPOF_TRY(sor_counters_ensure(h, prog_total));
    if (!sor_counters_clear(h, 0, prog_total)) return PAPOF_EDEVICE;
    // DIRECT: the producer's kernel stores the cut cells and its progress into the consumer's memory while both run (needs
    // peer addressing: the LOCAL transport).  STAGED (any transport, RCCL included): the same SPLIT kernels, but a rank's
    // kernel writes the cut cells into an OUTBOX of its own, and after it has finished they travel as one message per solve
    // and cut (n_sor x positions cells, ~1 MB at 1080p) to the rank below, which unpacks them into its inbox cells, marks the
    // band above complete in its counters and only then launches its own bands: ranks take turns within a solve -- correct
    // by construction with nothing but sends and receives at kernel boundaries, and necessarily slower than one GPU for the
    // solver's share (the exact-order solve is one dependency chain); everything else still splits over the ranks.
    const bool direct = t.tp->addresses_peers();
    // staged protocol: launches (= messages per cut) per solve; 1 = the ranks take turns (see the staged branch below)
    const int bands_chunks = std::getenv("PAPOF_BANDS_CHUNKS") ? std::max(1, std::atoi(std::getenv("PAPOF_BANDS_CHUNKS"))) : 1;
    std::vector<void*> arenas(n, nullptr), syncs(n, nullptr);
    double* outbox = nullptr;       // STAGED: banded planes (same layout) that receive the SPLIT kernel's peer stores
    unsigned* outprog = nullptr;    // ... and its peer publications (discarded)
    double* cutmsg = nullptr;       // the packed message (send side and receive side use it in turn)
    if (direct) {
        PAPOF_TRY(t.tp->bases(h, A.base, h->sync_words, arenas, syncs));  // (a barrier: every rank's counters are clear)
    } else {
        size_t cells = 0, cells_d = 0;
        skew_capacity(H, W, n_sor_max, cells, cells_d);
        outbox = A.f64(2 * (cells_d + kLanes));
        const int npos_max = skew_dims(H, W, n_sor_max, 1, 1).npos_d;
        cutmsg = A.f64((size_t)2 * n_sor_max * npos_max);
        size_t per_max = 0;
        for (int k = 0; k < levels; k++) per_max = std::max(per_max, LP[k].per);
        outprog = reinterpret_cast<unsigned*>(A.f64(per_max / 2 + 64));
        if (A.overflow) return PAPOF_ENOMEM;
    }
    const auto peer_ptr = [&](int r, const void* mine, const void* my_base, void* peer_base) -> void* {
        (void)r;
        return (char*)peer_base + ((const char*)mine - (const char*)my_base);
    };

    // ---- replicated, flow-independent work on the preparation stream (as tiles_flow)
    const bool overlap = h->overlap_prep && h->prep_stream != nullptr;
    while (h->sync_events.size() < (size_t)levels + 2) {
        hipEvent_t e;
        PAPOF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->sync_events.push_back(e);
    }
    {
        hipStream_t const main_stream = h->stream;
        if (overlap) {
            PAPOF_HIP(hipEventRecord(h->sync_events[levels + 1], main_stream));
            PAPOF_HIP(hipStreamWaitEvent(h->prep_stream, h->sync_events[levels + 1], 0));
            h->stream = h->prep_stream;
        }
        int rc = hwc_to_planar(h, d_im1, L[0].p1, H, W, C);
        if (rc == PAPOF_OK) rc = hwc_to_planar(h, d_im2, L[0].p2, H, W, C);
        if (rc == PAPOF_OK) rc = build_pyramid(h, L, plan, C, false, tmp_a, tmp_b);
        if (rc == PAPOF_OK) rc = build_pyramid(h, L, plan, C, true, tmp_a, tmp_b);
        for (int k = levels - 1; k >= 0 && rc == PAPOF_OK; k--) {
            rc = im2feature(h, L[k].p1, F1[k], L[k].h, L[k].w, C);
            if (rc == PAPOF_OK) rc = im2feature(h, L[k].p2, F2[k], L[k].h, L[k].w, C);
            if (rc == PAPOF_OK && overlap && hipEventRecord(h->sync_events[k], h->stream) != hipSuccess) rc = PAPOF_EDEVICE;
        }
        if (rc == PAPOF_OK) rc = central3_planes(h, L[0].p2, gx, gy, gxy, H, W, C);
        if (rc == PAPOF_OK && overlap && hipEventRecord(h->sync_events[levels], h->stream) != hipSuccess) rc = PAPOF_EDEVICE;
        h->stream = main_stream;
        if (rc != PAPOF_OK) {
            if (overlap) hipStreamSynchronize(h->prep_stream);
            return rc;
        }
    }

    const Taps g5 = smooth5_taps();
    const char* const silent_env = std::getenv("PAPOF_BANDS_SILENT_RANK");  // fault injection (tests): this rank
    const bool silent = silent_env && std::atoi(silent_env) == me;                 // never launches its solver kernels
    int pw = 0, ph = 0;
    long solve_no = 0;  // solves of this call so far (the same number on every rank)
    BandSplit prev{n, 0, 1, 0};
    for (int k = levels - 1; k >= 0; k--) {
        const int lw = L[k].w, lh = L[k].h;
        const size_t np = (size_t)lw * lh;
        const int K = P.n_sor + k * P.n_sor_per_level, n_outer = P.n_outer + k * P.n_outer_per_level;
        PAPOF_TRY(sor_bind_plain(h, sp, lh, lw, K));
        const BandSplit bs{n, sp.sd.nb, K, lh};
        const int B0 = bs.B(me), B1 = bs.B(me + 1);
        const bool mine = B1 > B0;
        int ra0, ra1;
        bs.coef_rows(me, ra0, ra1);
        const auto rows = [&](int y0, int y1) { return Rect{0, std::max(0, y0), lw, std::min(lh, y1)}; };
        const Rect Rsys = rows(ra0, ra1), Rsm = mine ? rows(ra0 - 2, ra1 + 2) : Rect{0, 0, 0, 0};
        const Rect Rw = mine ? rows(ra0 - 4, ra1 + 4) : Rect{0, 0, 0, 0}, Rphi = mine ? rows(ra0 - 1, ra1) : Rect{0, 0, 0, 0};
        const RectOf own = [&](int r) {
            int y0, y1;
            bs.final_rows(r, y0, y1);
            return y1 > y0 ? Rect{0, y0, lw, y1} : Rect{0, 0, 0, 0};
        };
        const RectOf need_u = [&](int r) {
            int y0, y1;
            bs.coef_rows(r, y0, y1);
            return y1 > y0 ? Rect{0, std::max(0, y0 - 4), lw, std::min(lh, y1 + 4)} : Rect{0, 0, 0, 0};
        };
        const double *f1 = F1[k], *f2 = F2[k];
        if (overlap) PAPOF_HIP(hipStreamWaitEvent(h->stream, h->sync_events[k], 0));
        // every non-cell of the coefficient planes is 0.0, and every cell of the (du, dv) planes -- the inbox cells the rank
        // above writes included -- is finite before anybody reads it (ghost lanes read positions nobody writes)
        PAPOF_TRY(sor_reset_planes(h, sp));
        PAPOF_HIP(hipMemsetAsync(sp.du, 0, (sp.sd.nd + sp.sd.nh) * 16, h->stream));
        if (outbox) PAPOF_HIP(hipMemsetAsync(outbox, 0, (sp.sd.nd + sp.sd.nh) * 16, h->stream));  // (packed whole: finite everywhere)
        if (k == levels - 1) {  // src/OpticalFlow.cpp:801-806
            PAPOF_HIP(hipMemsetAsync(u, 0, np * sizeof(double), h->stream));
            PAPOF_HIP(hipMemsetAsync(v, 0, np * sizeof(double), h->stream));
            PAPOF_HIP(hipMemcpyAsync(warp, f2, np * fc * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            PAPOF_TRY(t.tp->barrier(h));  // the planes are clear before a peer's solver kernel may write its inbox cells
        } else {  // :809-814: the coarser level's final rows -> the rows of it the up-sampling of my rows reads
            const double xr = (double)lw / pw, yr = (double)lh / ph, inv = 1 / ratio;
            const int cw = pw, chh = ph;
            const BandSplit pb = prev;
            const RectOf own_c = [&](int r) {
                int y0, y1;
                pb.final_rows(r, y0, y1);
                return y1 > y0 ? Rect{0, y0, cw, y1} : Rect{0, 0, 0, 0};
            };
            const RectOf need_c = [&](int r) { return resize_source(need_u(r), xr, yr, cw, chh); };
            double* uv[2] = {u, v};
            PAPOF_TRY(exchange_planes(t, uv, 2, cw, own_c, need_c));  // (its barrier also orders the clears above)
            PAPOF_TRY(resize(h, u, u2, ph, pw, 1, lh, lw, xr, yr, true, inv, &Rw));
            PAPOF_TRY(resize(h, v, v2, ph, pw, 1, lh, lw, xr, yr, true, inv, &Rw));
            std::swap(u, u2);
            std::swap(v, v2);
            if (!fused) PAPOF_TRY(warp_bilinear(h, f1, f2, u, v, warp, lh, lw, fc, &Rw));
        }
        // smoothed features of frame 1 on the rows the blend is needed on (getDxs, src/OpticalFlow.cpp:84-90)
        PAPOF_TRY(filter_h(h, f1, tmp, lh, lw, fc, g5, &Rw));
        PAPOF_TRY(filter_v(h, tmp, im1s, lh, lw, fc, g5, &Rsm));
        // the rank that runs band B1 (the next one with bands: ranks without bands sit at the end)
        const int below = me + 1 < n && bs.has(me + 1) ? me + 1 : -1;
        for (int count = 0; count < n_outer; count++) {
            if (mine) {
                if (fused) {  // kernels.hip: k_flow_system on this rank's rows -- the warped frame is never materialised
                    PAPOF_TRY(flow_system(h, f1, f2, u, v, im1s, lh, lw, fc, P.alpha, P.omega, sp, nullptr, Rsys.y0, Rsys.y1));
                } else {
                    PAPOF_TRY(smooth_hv_blend(h, warp, im1s, blend, imdt, lh, lw, fc, Rsm.y0, Rsm.y1));
                    PAPOF_TRY(compute_phi(h, u, v, nullptr, phi, lh, lw, &Rphi));
                    PAPOF_TRY(assemble_system(h, blend, imdt, phi, u, v, lh, lw, fc, P.alpha, P.omega, sp, nullptr, nullptr,
                                              nullptr, &Rsys));
                }
                unsigned* const prog = h->sync_words + 32 + LP[k].off + (size_t)count * LP[k].per;
                SorSplit cut{nullptr, nullptr, B0 > 0, below >= 0};
                if (below >= 0 && direct) {
                    cut.peer_du = (double*)peer_ptr(below, sp.du, A.base, arenas[below]);
                    cut.peer_prog = (unsigned*)peer_ptr(below, prog, h->sync_words, syncs[below]);
                } else if (below >= 0) {
                    cut.peer_du = outbox;
                    cut.peer_prog = outprog;
                }
                if (direct) {
                    if (B0 > 0) PAPOF_TRY(t.tp->turn_wait(me - 1, solve_no));  // (ranks with bands are 0 .. m: the one above is me - 1)
                    sorclk.phase(PAPOF_T_PHASE5_SOR);
                    if (!silent) PAPOF_TRY(sor_solve_bands(h, sp, lh, lw, P.alpha, P.omega, K, prog, B0, B1, &cut));
                    sorclk.phase(-1);
                    t.tp->turn_done(solve_no);
                }
            }
            if (!direct) {
                // STAGED.  The solve of every rank is issued as `chunks` launches over ranges of sweeps [k0, k1); the cut cells of a
                // range travel to the rank below as soon as its launch has ended, so rank g + 1 runs sweeps [k0, k1) while rank g
                // runs [k1, k2).  chunks = 1 (the default): the ranks take turns within a solve.  In stream order on rank g, per
                // range: receive the range's cut cells from g - 1, unpack them into the inbox cells and mark those sweeps of the
                // band above complete, launch the own bands for the range, pack and send the own last band's cut cells to g + 1.
                // What it buys is modelled in DESIGN.md 7 (every range pays a full row traversal: little) and can be measured with
                // PAPOF_BANDS_CHUNKS on a multi-GPU node; results are the same bits for every chunk count (tests/test_gpu_bands.py).
                const int m_ranks = std::min(n, sp.sd.nb);  // ranks with bands: 0 .. m_ranks - 1
                const int npos = sp.sd.npos_d;
                unsigned* const prog = h->sync_words + 32 + LP[k].off + (size_t)count * LP[k].per;
                SorSplit cut{below >= 0 ? outbox : nullptr, below >= 0 ? outprog : nullptr, B0 > 0, below >= 0};
                const int chunks = std::max(1, std::min(bands_chunks, K));
                const auto launch_range = [&](int k0, int k1) -> int {
                    sorclk.phase(PAPOF_T_PHASE5_SOR);
                    if (!silent) PAPOF_TRY(sor_solve_bands(h, sp, lh, lw, P.alpha, P.omega, K, prog, B0, B1, &cut, k0, k1));
                    sorclk.phase(-1);
                    return PAPOF_OK;
                };
                for (int c = 0; c < chunks; c++) {
                    const int k0 = (int)((long long)K * c / chunks), k1 = (int)((long long)K * (c + 1) / chunks);
                    if (k1 <= k0) continue;
                    const size_t msg_doubles = (size_t)2 * (k1 - k0) * npos;
                    double* const msg = cutmsg + (size_t)2 * k0 * npos;  // (k_cut_cells indexes the message by absolute sweep)
                    const dim3 cgrid((npos + 255) / 256, k1 - k0), cblock(256);
                    if (mine && me == 0) PAPOF_TRY(launch_range(k0, k1));
                    // round (c, r): rank r hands the range's cut cells to rank r + 1.  EVERY rank walks every round (a transport
                    // whose exchanges are collective -- LOCAL -- needs that; on RCCL a round without a message costs nothing), so
                    // on rank g the stream holds, per range: receive (round g - 1), unpack, launch, pack, send (round g).
                    for (int r = 0; r + 1 < m_ranks; r++) {
                        std::vector<Msg> sends, recvs;
                        if (me == r && !silent) {  // (fault injection: a silent rank never sends its cut cells either)
                            hipLaunchKernelGGL(k_cut_cells<true>, cgrid, cblock, 0, h->stream, (double2*)outbox, (double2*)cutmsg,
                                               sp.sd.nb, npos, B1 - 1, k0, k1);
                            PAPOF_HIP(hipGetLastError());
                            sends.push_back(Msg{r + 1, msg, msg_doubles});
                        } else if (me == r + 1) {
                            recvs.push_back(Msg{r, msg, msg_doubles});
                        }
                        PAPOF_TRY(t.tp->exchange(h, sends, recvs));
                        t.exchanges++;
                        t.exchanged_bytes += (sends.size() + recvs.size()) * msg_doubles * sizeof(double);
                        if (me == r + 1) {
                            hipLaunchKernelGGL(k_cut_cells<false>, cgrid, cblock, 0, h->stream, (double2*)sp.du, (double2*)cutmsg,
                                               sp.sd.nb, npos, B0 - 1, k0, k1);
                            hipLaunchKernelGGL(k_cut_counters, dim3((k1 - k0 + 63) / 64), dim3(64), 0, h->stream, prog, sp.sd.nb,
                                               B0 - 1, k0, k1, (unsigned)sp.sd.ns);
                            PAPOF_HIP(hipGetLastError());
                            PAPOF_TRY(launch_range(k0, k1));
                        }
                    }
                }
            }
            if (mine) {
                Rect Rf = own(me);
                // :513-514 on the rows whose final increments this rank holds, into the other pair of planes
                PAPOF_TRY(update_warp_phi(h, sp, u, v, u2, v2, f1, f2, warp, nullptr, lh, lw, fc, false, Rf.y0, Rf.y1));
            }
            solve_no++;
            std::swap(u, u2);
            std::swap(v, v2);
            double* uv[2] = {u, v};
            PAPOF_TRY(exchange_planes(t, uv, 2, lw, own, need_u));
            if (lap_on) {  // the noise estimate behind this update (:530), on the rows whose final flow this rank holds
                const Rect Rf = own(me);
                PAPOF_TRY(lap_rows_check(h, f1, f2, u, v, lh, lw, fc, Rf.y0, Rf.y1, lap_mine + lap_slot * 16,
                                         lap_mine + lap_slot * 16 + 8, kLapSet));
                lap_slot++;
            }
            if (mine && !fused && count + 1 < n_outer) PAPOF_TRY(warp_bilinear(h, f1, f2, u, v, warp, lh, lw, fc, &Rw));  // :516
        }
        pw = lw;
        ph = lh;
        prev = bs;
    }

    // src/OpticalFlow.cpp:841-842: bicubic warp of the ORIGINAL frame 2 on the rows this rank owns, then everything to rank 0
    {
        if (overlap) PAPOF_HIP(hipStreamWaitEvent(h->stream, h->sync_events[levels], 0));
        const int lw = W, lh = H;
        const BandSplit pb = prev;
        const RectOf own = [&](int r) {
            int y0, y1;
            pb.final_rows(r, y0, y1);
            return y1 > y0 ? Rect{0, y0, lw, y1} : Rect{0, 0, 0, 0};
        };
        const Rect T = own(me);
        PAPOF_TRY(bicubic_warp(h, L[0].p1, L[0].p2, gx, gy, gxy, u, v, warp_hwc, H, W, C, &T));
        const RectOf need = [&](int r) { return r == 0 ? Rect{0, 0, lw, lh} : Rect{0, 0, 0, 0}; };
        double* uv[2] = {u, v};
        PAPOF_TRY(exchange_planes(t, uv, 2, W, own, need));
        const RectOf own_c = [&](int r) {
            const Rect q = own(r);
            return Rect{q.x0 * C, q.y0, q.x1 * C, q.y1};
        };
        const RectOf need_c = [&](int r) { return r == 0 ? Rect{0, 0, W * C, H} : Rect{0, 0, 0, 0}; };
        double* wp[1] = {warp_hwc};
        PAPOF_TRY(exchange_planes(t, wp, 1, W * C, own_c, need_c));
        if (me == 0) {
            if (!d_vx || !d_vy || !d_warp) return PAPOF_EINVAL;
            PAPOF_HIP(hipMemcpyAsync(d_vx, u, np0 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            PAPOF_HIP(hipMemcpyAsync(d_vy, v, np0 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            PAPOF_HIP(hipMemcpyAsync(d_warp, warp_hwc, np0 * C * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        }
    }
    std::vector<unsigned> lap_host;
    if (lap_on) {  // every rank's flags to every rank: all of them reach the same verdict
        const size_t msg = lap_words / 2;  // doubles
        PAPOF_HIP(hipMemcpyAsync(lap_all + (size_t)me * lap_words, lap_mine, lap_words * sizeof(unsigned), hipMemcpyDeviceToDevice,
                                 h->stream));
        std::vector<Msg> sends, recvs;
        for (int r = 0; r < n; r++)
            if (r != me) {
                sends.push_back(Msg{r, reinterpret_cast<double*>(lap_mine), msg});
                recvs.push_back(Msg{r, reinterpret_cast<double*>(lap_all + (size_t)r * lap_words), msg});
            }
        if (n > 1) PAPOF_TRY(t.tp->exchange(h, sends, recvs));
    }
    total.phase(-1);
    {   // the wait that a silent peer cannot hold for ever (Transport::drain) comes BEFORE anything that blocks the host on the
        // stream -- a copy into pageable memory does, whatever its name says
        const int rc_drain = t.tp->drain(h);
        if (overlap) PAPOF_HIP(hipStreamSynchronize(h->prep_stream));
        PAPOF_TRY(rc_drain);
    }
    if (lap_on) {
        lap_host.resize(lap_words * (size_t)n);
        PAPOF_HIP(hipMemcpy(lap_host.data(), lap_all, lap_host.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    }
    PAPOF_TRY(sor_check(h));  // PAPOF_ETIMEOUT when a bounded wait of this rank's tasks expired (a peer that never published)
    if (lap_on) {
        for (long s = 0; s + 1 < lap_slot; s++)  // the estimate behind the call's last update is never consulted
            for (int c = 0; c < fc; c++) {
                bool witness = false, valid = false;
                for (int r = 0; r < n; r++) {
                    witness = witness || lap_host[(size_t)r * lap_words + s * 16 + c] == kLapSet;
                    valid = valid || lap_host[(size_t)r * lap_words + s * 16 + 8 + c] == kLapSet;
                }
                if (!witness && valid) {
                    set_last_error_text("the reference's Laplacian-noise guard (LapPara < 1E-20, src/OpticalFlow.cpp:399-400) may "
                                        "trip on this pair; the band split has no exact pass: run it on one GPU (papof_flow*)");
                    return PAPOF_EINVAL;
                }
            }
    }
    if (total.err != PAPOF_OK || sorclk.err != PAPOF_OK) return PAPOF_EDEVICE;
    sorclk.collect(tm);
    total.collect(tm);
    if (timing) std::memcpy(timing, tm, PAPOF_N_TIMERS * sizeof(double));
    return PAPOF_OK;
}

}  // namespace
}  // namespace papof

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int papof_tiles_grid(int nranks, int* rows, int* cols) {
    if (nranks < 1 || !rows || !cols) return PAPOF_EINVAL;
    // widest-first split: 8 -> 2 x 4 (tiles of 480 x 540 at 1920 x 1080, the shorter perimeter: SURVEY.md §8e),
    // 4 -> 2 x 2, 2 -> 1 x 2; any other count: the most square factorisation with cols >= rows
    int r = (int)std::sqrt((double)nranks);
    while (r > 1 && nranks % r) r--;
    *rows = r;
    *cols = nranks / r;
    return PAPOF_OK;
}

int papof_tiles_rect(int width, int height, int rows, int cols, int rank, int rect[4]) {
    if (width < 1 || height < 1 || rows < 1 || cols < 1 || rank < 0 || rank >= rows * cols || !rect) return PAPOF_EINVAL;
    const Rect r = tile_rect(TileGrid{rows, cols}, rank, width, height);
    rect[0] = r.x0;
    rect[1] = r.y0;
    rect[2] = r.x1;
    rect[3] = r.y1;
    return PAPOF_OK;
}

int papof_tiles_halo_message(int width, int height, int rows, int cols, int halo, int src, int dst, int rect[4]) {
    if (width < 1 || height < 1 || rows < 1 || cols < 1 || halo < 0 || !rect) return PAPOF_EINVAL;
    const int n = rows * cols;
    if (src < 0 || src >= n || dst < 0 || dst >= n) return PAPOF_EINVAL;
    const TileGrid g{rows, cols};
    Rect r{0, 0, 0, 0};
    if (src != dst) r = intersect(tile_rect(g, src, width, height), grow(tile_rect(g, dst, width, height), halo, width, height));
    rect[0] = r.x0;
    rect[1] = r.y0;
    rect[2] = r.x1;
    rect[3] = r.y1;
    return PAPOF_OK;
}

int papof_bands_plan(int height, int width, int n_sor, int nranks, int rank, int out[6]) {
    if (height < 1 || width < 1 || n_sor < 1 || nranks < 1 || rank < 0 || rank >= nranks || !out) return PAPOF_EINVAL;
    const BandSplit bs{nranks, skew_dims(height, width, n_sor, 1, 1).nb, n_sor, height};
    out[0] = bs.B(rank);
    out[1] = bs.B(rank + 1);
    bs.coef_rows(rank, out[2], out[3]);
    bs.final_rows(rank, out[4], out[5]);
    return PAPOF_OK;
}

int papof_tiles_unique_id(unsigned char id[PAPOF_TILES_ID_BYTES]) {
    if (!id) return PAPOF_EINVAL;
    const RcclApi* api = rccl();
    if (!api) {
        set_last_error_text("librccl could not be loaded");
        return PAPOF_ENODEVICE;
    }
    static_assert(sizeof(ncclUniqueId) <= PAPOF_TILES_ID_BYTES, "unique id does not fit");
    ncclUniqueId uid;
    PAPOF_NCCL(api, api->GetUniqueId(&uid));
    std::memset(id, 0, PAPOF_TILES_ID_BYTES);
    std::memcpy(id, &uid, sizeof uid);
    return PAPOF_OK;
}

static int tiles_common(papof_handle* h, int nranks, int rows, int cols, int halo, papof_tiles** out) {
    if (!h || !out || nranks < 1 || rows < 1 || cols < 1 || rows * cols != nranks || halo < 0) return PAPOF_EINVAL;
    *out = new papof_tiles();
    (*out)->h = h;
    (*out)->grid = TileGrid{rows, cols};
    (*out)->halo = halo == 0 ? 10 : halo;
    return PAPOF_OK;
}

int papof_tiles_create(papof_handle* h, const unsigned char id[PAPOF_TILES_ID_BYTES], int rank, int nranks, int rows,
                       int cols, int halo, papof_tiles** out) {
    if (!id || rank < 0 || rank >= nranks) return PAPOF_EINVAL;
    const RcclApi* api = rccl();
    if (!api) {
        set_last_error_text("librccl could not be loaded");
        return PAPOF_ENODEVICE;
    }
    PAPOF_TRY(tiles_common(h, nranks, rows, cols, halo, out));
    PAPOF_HIP(hipSetDevice(h->device));
    auto tp = std::make_unique<RcclTransport>();
    tp->api = api;
    tp->rank = rank;
    tp->nranks = nranks;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclResult_t r = api->CommInitRank(&tp->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        set_last_error_text(std::string("ncclCommInitRank failed: ") + api->GetErrorString(r));
        tp->comm = nullptr;
        delete *out;
        *out = nullptr;
        return PAPOF_EDEVICE;
    }
    (*out)->tp = std::move(tp);
    return PAPOF_OK;
}

int papof_tiles_create_local(papof_handle* const* handles, int nranks, int rows, int cols, int halo,
                             papof_tiles** out) {
    if (!handles || !out || nranks < 1) return PAPOF_EINVAL;
    auto group = std::make_shared<LocalGroup>(nranks);
    for (int r = 0; r < nranks; r++) out[r] = nullptr;
    for (int r = 0; r < nranks; r++) {
        int rc = tiles_common(handles[r], nranks, rows, cols, halo, &out[r]);
        if (rc != PAPOF_OK) {
            for (int q = 0; q < r; q++) {
                delete out[q];
                out[q] = nullptr;
            }
            return rc;
        }
        auto tp = std::make_unique<LocalTransport>();
        tp->rank = r;
        tp->nranks = nranks;
        tp->g = group;
        out[r]->tp = std::move(tp);
    }
    return PAPOF_OK;
}

int papof_tiles_flow_device(papof_tiles* t, const double* d_im1, const double* d_im2, int height, int width, int c,
                            int pyramid_levels, const papof_params* params, double* d_vx, double* d_vy,
                            double* d_warpI2, double timing_sec[PAPOF_N_TIMERS]) {
    if (!t || !t->h || !t->tp || !d_im1 || !d_im2 || height < 1 || width < 1 || c < 1) return PAPOF_EINVAL;
    papof_params P;
    if (params)
        P = *params;
    else {
        papof_default_params(&P);
        P.sor_mode = PAPOF_SOR_REDBLACK;
    }
    PAPOF_HIP(hipSetDevice(t->h->device));
    // PAPOF_SOR_EXACT: the exact-order split into horizontal ranges of solver bands over ALL ranks (the rows x cols grid is
    // the red-black tiles'); anything else: the 2-D tiles
    int rc = P.sor_mode == PAPOF_SOR_EXACT
                 ? bands_flow(*t, d_im1, d_im2, height, width, c, pyramid_levels, P, d_vx, d_vy, d_warpI2, timing_sec)
                 : tiles_flow(*t, d_im1, d_im2, height, width, c, pyramid_levels, P, d_vx, d_vy, d_warpI2, timing_sec);
    if (rc != PAPOF_OK)
        if (auto* lt = dynamic_cast<LocalTransport*>(t->tp.get())) lt->g->fail();  // release the sibling threads
    return rc;
}

int papof_tiles_stats(const papof_tiles* t, long* exchanges, size_t* bytes) {
    if (!t) return PAPOF_EINVAL;
    if (exchanges) *exchanges = t->exchanges;
    if (bytes) *bytes = t->exchanged_bytes;
    return PAPOF_OK;
}

int papof_tiles_comm_info(const papof_tiles* t, int* nranks_seen, int* rank_seen, int* rows, int* cols, int* halo) {
    if (!t || !t->tp) return PAPOF_EINVAL;
    int n = 0, r = 0;
    PAPOF_TRY(t->tp->seen(&n, &r));
    if (nranks_seen) *nranks_seen = n;
    if (rank_seen) *rank_seen = r;
    if (rows) *rows = t->grid.rows;
    if (cols) *cols = t->grid.cols;
    if (halo) *halo = t->halo;
    return PAPOF_OK;
}

void papof_tiles_destroy(papof_tiles* t) { delete t; }

}  // extern "C"
