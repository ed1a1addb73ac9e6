// papteam_opticalflow_amd/csrc/common.h -- shared host/device declarations of libpapof (gfx950 only).
//
// Device data layout (everything IEEE fp64, as the reference: typedef Image<double> DImage,
// src/Image.h:469):
//   * images / features are PLANAR: plane k of an H x W x C image is a dense row-major H*W array at
//     base + k*H*W (the reference's interleaved HWC layout is converted once on entry and once on exit);
//   * the eight SOR operands of the exact-order solver use the per-band SKEWED, PAIRED layout documented in
//     sor.hip (62-row bands + 2 ghost lanes, cell (lane l, column j) of band b at ((b*NSP + j + l)*64 + l)).
//
// All device code is compiled with -ffp-contract=off: the reference is built without FMA
// (Code/Serial/setup.py:24-25, plain x86-64 gcc), and matching its rounding step for step is what
// makes the GPU results bit-compatible instead of merely close.
#pragma once
#include <cstdlib>

#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/papof.h"

namespace papof {

constexpr int kLanes = 64;     // CDNA wavefront
constexpr int kMaxFsize = 8;   // Gaussian half-width supported by the filter kernels

// Half-open pixel rectangle [x0, x1) x [y0, y1) in the coordinates of a full plane.  Every per-pixel kernel takes
// one: the whole plane on one GPU, a tile (plus the halo a stage needs) when a frame is sharded across GPUs (tiles.hip).
struct Rect {
    int x0, y0, x1, y1;
    __host__ __device__ bool empty() const { return x1 <= x0 || y1 <= y0; }
    __host__ __device__ int w() const { return x1 - x0; }
    __host__ __device__ int h() const { return y1 - y0; }
};

// BATCHED launches (api.hip: flow_batch -- B frame pairs of one shape per launch chain, the reference's TestSuite walking a
// collection of small frames, Code/Serial/TestSuite.py:69-81): every buffer of the call is an ARRAY over the pairs (or over the
// frames) of the batch, and a kernel's blockIdx.y (the layout converters: blockIdx.z) is the item it works on -- each of its
// pointers advanced by that item's stride, in elements of the pointee.  All zero = the ordinary single call.
struct BatchK {
    size_t im;    // per-frame planes (pyramid levels, features, smoothed features, derivative planes) between one pair's frame
                  // and the next pair's: frame stride x 1 (a sequence: pair p = frames p, p + 1) or x 2 (independent pairs)
    size_t uv;    // the flow planes of a pair
    size_t sp;    // the solver's coefficient planes
    size_t d;     // ... and its (du, dv) planes
    size_t wit;   // witness words of the Laplacian-noise guard (unsigned)
    size_t out;   // the interleaved result image
};

struct Taps {                  // 1-D correlation taps, passed by value as a kernel argument
    double t[2 * kMaxFsize + 1];
    int fsize;
};

// thread-local text of the last failing HIP call
void set_last_error(const char* what, hipError_t e, const char* file, int line);
const char* last_error();

#define PAPOF_HIP(expr)                                              \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) {                                      \
            ::papof::set_last_error(#expr, _e, __FILE__, __LINE__);  \
            return PAPOF_EDEVICE;                                    \
        }                                                            \
    } while (0)

#define PAPOF_TRY(expr)              \
    do {                             \
        int _rc = (expr);            \
        if (_rc != PAPOF_OK) return _rc; \
    } while (0)

// Bump arena over one hipMalloc'd block.  Stack discipline: mark()/release().
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    bool overflow = false;
    void* alloc(size_t bytes) {
        size_t o = (off + 255) & ~size_t(255);
        if (o + bytes > cap) {
            overflow = true;
            return nullptr;
        }
        off = o + bytes;
        if (off > peak) peak = off;
        return base + o;
    }
    double* f64(size_t n) { return static_cast<double*>(alloc(n * sizeof(double))); }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
};

constexpr int kLapNzWords = 32 * 8, kLapMaxSlots = 1024;  // flags of the Laplacian-noise guard (papof_handle::lap_flags_dev)
constexpr size_t kLapFlagWords = (size_t)kLapMaxSlots * 8 + kLapNzWords;
inline size_t lap_wit_word(int slot) { return (size_t)(kLapMaxSlots - 1 - slot) * 8; }  // slots grow DOWNWARDS towards ...
inline size_t lap_nz_word(int level) { return (size_t)kLapMaxSlots * 8 + (size_t)level * 8; }  // ... the non-zero flags
constexpr unsigned kLapNone = 0x80000000u;  // witness word = pass number ^ kLapNone: a block that saw EVERY pixel found no valid sample
constexpr int kSorMaxDepth = 32;  // largest software-pipeline depth (steps) of the exact-order SOR kernel
constexpr int kBandRows = kLanes - 2;  // real rows per task: lanes 1..62; lanes 0 / 63 stand for the rows above / below

// Layout of the exact-order solver's operand planes ("globally skewed", paired 16-byte cells):
//   cell (row i, column j)  ->  index (i + j + qt) * hp + (i + rt)
// i.e. one POSITION per anti-diagonal, the rows of a position contiguous.  A task (band b, sweep k) owns the rows
// 62b - k .. 62b - k + 61 (bands climb one row per sweep, see sor.hip), so at step s its 64 lanes touch the 64
// consecutive cells (position 62b - k - 1 + qt + s, rows 62b - k - 1 + rt ...) -- one contiguous 1-KiB access.
// Every cell that is not a real (row, column) of the image holds 0.0.
//
// The unknowns (du, dv) use a different, BANDED PING-PONG layout: D[k & 1][position][band][64 cells] -- task (b, k)
// writes its 64 lanes at step s to the aligned 1-KiB block D[k & 1][s + 1][b] that no other task of the sweep
// touches (see sor.hip).  After the last sweep K-1, row i / column j is found at
//   parity (K-1) & 1, band (i + K-1) / 62, cell c = 1 + (i + K-1) % 62, position j + c + 1.
struct SkewDims {
    int nb;      // bands = tasks per sweep
    int ns;      // steps per task = W + 63
    int rt, qt;  // padding rows above row 0 / positions before position 0 (n_sor + 1: room for the climbing bands)
    int hp;      // storage rows per position (multiple of 8 => 128-byte aligned positions)
    int npos;    // positions
    size_t n;    // cells per paired coefficient plane
    int n_sor;   // sweeps this layout was made for
    int npos_d;  // positions of the (du, dv) planes
    size_t nd;   // cells of the two (du, dv) planes (ping-pong)
    int group;   // sweeps per workgroup of the solver: 1 = one wave per workgroup; M > 1 = M consecutive sweeps of a band
                 // in one workgroup, handed from wave to wave through LDS (sor.hip); the planes then alternate per GROUP
    int dpar;    // which plane holds the values of the last sweep
    size_t nh;   // grouped solver: cells of the halo rows H[sweep][band][position] that follow the two planes
    int fuse;    // sweeps per WAVE: 1, or 2 = k_sor_fused (two consecutive sweeps of a band in one wavefront, the second
                 // two steps behind the first, fed from registers); bands then have 61 rows and climb two rows per pair
    int band_rows, koff, poff;  // where (du, dv)(i, j) sits after the last sweep: t = i + koff, band t / band_rows,
                                // cell c = 1 + t % band_rows, position j + c + poff (kernels.hip: dudv_cell)
};
constexpr int kFusedRows = kLanes - 3;  // fused pairs: lanes 2..62 real in the first sweep, lanes 1..61 in the second
inline SkewDims skew_dims(int h, int w, int n_sor, int group = 1, int fuse = 1) {
    SkewDims d;
    d.fuse = (fuse == 2 && group <= 1) ? 2 : 1;
    d.group = d.fuse == 2 ? 1 : (group < 1 ? 1 : group);
    d.ns = w + kLanes - 1;
    d.n_sor = n_sor;
    d.npos_d = d.ns + 2 * kSorMaxDepth + 72;  // every position a task can touch, prefetch beyond the last step included
    if (d.fuse == 2) {
        const int last = (n_sor + 1) / 2 - 1;  // index of the last pair of sweeps
        d.band_rows = kFusedRows;
        d.nb = (h + 2 * last + 1 + kFusedRows - 1) / kFusedRows;
        d.qt = n_sor + 3;
        d.rt = (d.qt + 7) / 8 * 8;  // row 0 on a 128-byte boundary of its position: 16-row chunks of the assembly kernel are whole lines
        d.hp = (d.rt + kFusedRows * d.nb + 4 + 7) / 8 * 8;
        d.npos = d.qt + d.ns + 2 * kSorMaxDepth + 2 + kFusedRows * (d.nb - 1) + 2;
        d.dpar = last & 1;
        d.koff = 2 * last + 1;
        d.poff = 3;
        d.nh = 0;
    } else {
        d.band_rows = kBandRows;
        d.nb = (h + n_sor - 1 + kBandRows - 1) / kBandRows;
        d.qt = n_sor + 1;
        d.rt = (d.qt + 7) / 8 * 8;
        d.hp = (d.rt + kBandRows * d.nb + 2 + 7) / 8 * 8;
        d.npos = d.qt + d.ns + 2 * kSorMaxDepth + 2 + kBandRows * (d.nb - 1) + 2;
        d.dpar = d.group == 1 ? ((n_sor - 1) & 1) : (((n_sor + d.group - 1) / d.group - 1) & 1);
        d.koff = n_sor - 1;
        d.poff = 1;
        d.nh = d.group == 1 ? 0 : (size_t)n_sor * d.nb * d.npos_d;
    }
    d.n = (size_t)d.npos * d.hp;
    d.nd = (size_t)2 * d.npos_d * d.nb * kLanes;
    return d;
}
// capacity of one level's operand planes: the larger of the layouts the solver may bind (sor_bind)
inline void skew_capacity(int h, int w, int n_sor, size_t& cells, size_t& cells_d) {
    const SkewDims a = skew_dims(h, w, n_sor, 2), b = skew_dims(h, w, n_sor, 1, 2);
    cells = a.n > b.n ? a.n : b.n;
    cells_d = (a.nd + a.nh) > (b.nd + b.nh) ? (a.nd + a.nh) : (b.nd + b.nh);
}

// SOR operands of one solve.  `skew` selects the layout of all eight planes:
//   row-major modes: eight dense H*W planes;
//   skew (exact-order) mode: four PAIRED planes of 16-byte cells -- (phi,xy) (a1,a2) (b1,b2) (du,dv) -- so that
//   every access of the solver is one 16-byte-per-lane, 1-KiB-per-wave instruction.  The eight pointers then
//   alias the pairs with element stride 2: xy = phi + 1, a2 = a1 + 1, b2 = b1 + 1, dv = du + 1.
struct SorPlanes {
    double *phi, *xy, *a1, *a2, *b1, *b2;  // weights, imdxy, omega/diag_u, omega/diag_v, rhs_u, rhs_v
    double *du, *dv;                       // unknowns (written from zero; no initialisation needed)
    double *du2, *dv2;                     // second pair of unknown planes (row-major modes only): launches alternate
    bool skew;
    SkewDims sd;                           // skew mode: layout bound by sor_bind() for (H, W, n_sor) of this solve
    size_t cap_cells, cap_cells_d;         // skew mode: cells the coefficient planes / the (du, dv) allocation can hold
};

}  // namespace papof

namespace papof {
// A few persistent host threads per handle for the pageable <-> pinned copies of the host entry points (spawning
// threads per 8-MiB chunk cost more than the copies; first-touch page faults of the caller's fresh result arrays are
// spread over the workers).
class CopyPool {
public:
    explicit CopyPool(int workers);
    ~CopyPool();
    void copy(char* dst, const char* src, size_t n);  // returns when all bytes are in place
    int workers() const { return (int)threads_.size(); }

private:
    struct Job {
        char* dst;
        const char* src;
        size_t n;
    };
    void loop();
    std::vector<std::thread> threads_;
    std::vector<Job> jobs_;
    std::mutex mu_;
    std::condition_variable wake_, done_;
    int pending_ = 0;
    bool stop_ = false;
};
}  // namespace papof

namespace papof {
// One captured call (hipGraph): everything flow_device enqueues for a given problem -- both streams, ~200 nodes at
// 1080p, ~700 on the reference schedule -- replayed with a single launch.  Valid for exactly these arguments (the arena
// layout is a pure function of them; pointers are part of the key).
struct GraphKey {
    int H, W, C, levels, op, slot1, u8;
    papof_params P;
    const void *fa, *fb;
    void *vx, *vy, *warp;
    const void* arena_base;
    const void* sync_base;  // the progress counters / abort word (h->sync_words): captured memset nodes and kernel arguments hold
                            // their addresses, and a later call that needs more counters reallocates them
};
struct GraphEntry {
    GraphKey key;
    int seen = 0;  // eager calls with this key so far (the first one sizes every lazily grown buffer)
    hipGraphExec_t exec = nullptr;
};
}  // namespace papof

struct papof_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    papof::Arena arena;
    unsigned* sync_words = nullptr;  // progress counters + abort word of the exact-order SOR
    size_t sync_cap = 0;             // in unsigneds
    std::vector<hipEvent_t> events;
    size_t events_used = 0;
    int cu_count = 0;
    // host path (papof_flow): persistent device staging block and pinned bounce buffers
    double* stage_dev = nullptr;
    size_t stage_dev_bytes = 0;
    char* pin = nullptr;
    size_t pin_bytes = 0;
    int host_copy = 1;               // PAPOF_HOST_COPY: 1 = hipMemcpyAsync straight from / to the caller's memory (the runtime's
                                     // pageable path runs at 55 GB/s here; pinned result arrays are direct DMA), 0 = our pinned
                                     // bounce pipeline of round 1 (0.4-0.6 ms slower per 1080p call, same-box A/B)
    int host_threads = 6;            // threads used to move pageable user buffers to / from the pinned buffers
    papof::CopyPool* pool = nullptr; // created by the first host-buffer call
    // hipGraph replay of whole calls (PAPOF_GRAPH=1 / papof_set_graph_mode): for small frames a call is hundreds of
    // launches of a few microseconds of work each, and with several calls in flight the host-side launch path is the
    // limit; captured once (on the second call with the same arguments) a call becomes one hipGraphLaunch
    bool use_graph = false;
    std::vector<papof::GraphEntry> graphs;
    bool use_dpp = false;            // wave_shr/wave_shl DPP moves verified on this device (else ds_bpermute)
    int sor_depth = 0;               // software-pipeline depth R (steps) of the exact-order SOR kernel; 0 = by level size
    unsigned long long* sor_dbg = nullptr;  // device buffer for per-task wait statistics (PAPOF_SOR_DBG)
    // measurement hook: sor_solve() calls it with 1 right before and 0 right after its solver kernel(s), i.e. behind the
    // memset nodes that prepare a solve -- so that the HIP-event time of the roofline is the kernel's own (flow_device)
    void (*sor_mark)(void*, int) = nullptr;
    void* sor_mark_ctx = nullptr;
    int sor_fuse = 0;                // sweeps per wave of the exact-order solver: 1 or 2; 0 = by problem size
    int sor_group = 0;               // consecutive sweeps of a band per workgroup: 1, 2 or 4; 0 = by problem size
    int rb_depth = 0;                // blocked red-black / Jacobi solver: half-sweeps per launch; 0 = by region shape
    int rb_shape = 0;                // ... region shape 1..4 (sor.hip: blocked_shape); 0 = by plane size
    int rb_naive = 0;                // 1: one launch per half-sweep on the planes (cross-check)
    int sor_resident = 0;            // tasks per launch of the exact-order kernels; 0 = 8 per CU (sor.hip: resident_tasks)
    int sor_skip_dead = 1;           // k_sor_exact: lanes whose row lies outside the image neither load nor store (PAPOF_SOR_DEAD=0: A/B)
    unsigned* sor_prog_next = nullptr;  // cleared progress counters for the NEXT sor_solve() (else it clears its own)
    int sor_launches = 0;            // exact-order solver kernels launched by the current / last call (measurement: bench.py)
    // one entry per sor_solve() of the current / last call, in stream order (measurement: bench.py's roofline.by_level);
    // kind: 0 k_sor_exact, 1 k_sor_fused, 2 k_sor_group, 3 k_sor_blocked red-black, 4 k_sor_blocked Jacobi, 5 naive kernels;
    // sec: the solver kernels' own HIP-event time (filled when the call's timers are collected, else 0)
    struct SorSolveLog {
        int H, W, n_sor, kind, depth, launches;
        double sec;
    };
    std::vector<SorSolveLog> sor_log;
    double sor_upper_sec = 0.0;      // ... of which: event time of the launches on the strip streams (added to Phase5_SOR)
    // strips (api.hip: smooth_flow_strips): a level's plane as S horizontal strips of solver bands, each on its own
    // stream, so that a strip's non-solver kernels run in the shadow of the other strips' solves
    int strips = 0;                  // PAPOF_STRIPS: 0 / 1 = off (default), 2..4 = strips per level where the layout allows
    std::vector<hipStream_t> strip_streams;  // S - 1 extra streams (the last strip runs on the main stream)
    std::vector<hipEvent_t> strip_events;    // untimed events ordering the strips, reused from call to call
    size_t strip_events_used = 0;
    // second stream for everything that does not depend on the flow (pyramids, features, smoothed frame 1 of every
    // level, derivative planes of the final bicubic warp): runs beside the coarse levels' latency-bound solves
    hipStream_t prep_stream = nullptr;
    // Laplacian-noise guard (api.hip: LapGuard; src/OpticalFlow.cpp:399-400, :594-639).  Flags of a call, device + page-locked
    // host copy, ONE block shared with the phase stamps so that one copy per call fetches both: 8 witness words per outer
    // iteration (slot s at word (kLapMaxSlots - 1 - s) * 8: the used slots end where the next part begins), kLapNzWords words
    // of per-level non-zero flags of the feature channels, then the stamps.  lap_dev: LapPara of the exact pass (8 doubles),
    // its initial value 0.02 (:773-775) as a device constant, the reduction's scratch.
    unsigned* lap_flags_dev = nullptr;
    unsigned* lap_flags_host = nullptr;
    double* lap_dev = nullptr;
    double* lap_init_dev = nullptr;
    double* lap_scratch_dev = nullptr;
    bool lap_guard = true;     // PAPOF_LAP_GUARD=0: no guard at all (the optimistic pass without witnesses: A/B of their cost)
    bool lap_exact = false;    // the next call starts in the exact pass (its predecessor ended with a channel lacking a proof)
    unsigned lap_epoch = 0;    // a flag is SET when it holds the number of the pass that wrote it: nothing is ever cleared
    int lap_reruns = 0, lap_exact_calls = 0;  // statistics (papof_lap_guard_stats)
    double host_enqueue_sec = 0.0, host_wait_sec = 0.0;  // host wall time of the last call: enqueueing / waiting (papof_last_host_times)
    // Host buffers handed to the call itself (flow_host -> flow_device): the call then issues the PCIe copies where they
    // overlap device work -- frame 2 uploads while frame 1's share of the preparation runs, (vx, vy) go back beside the
    // final bicubic warp, warpI2 in row chunks behind its kernel's chunks -- on a stream of their own.
    struct HostIO {
        bool active = false;
        const void *im1 = nullptr, *im2 = nullptr;  // null: that frame is already on the device (sequence mode)
        size_t nb_in = 0;                           // bytes per input frame
        double *vx = nullptr, *vy = nullptr, *warp = nullptr;
        size_t nb_flow = 0, nb_img = 0;
        bool early_out = false;                     // the result arrays are page-locked: their copies may be queued early
        bool out_issued = false;                    // set by the call when it has queued the result copies itself
    } hostio;
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> copy_events;
    std::vector<hipEvent_t> sync_events;  // untimed events ordering the two streams
    bool overlap_prep = true;
    // phase stamps (flow_internal.h: PhaseClock): slots the kernels write the 100 MHz clock into
    unsigned long long* stamps_dev = nullptr;  // device slots
    unsigned long long* stamps = nullptr;      // pinned host copy, fetched once per call
    int stamps_cap = 0, stamps_used = 0, stamps_fetched = 0;
    unsigned long long* next_stamp = nullptr;  // taken (and cleared) by the next kernel launch that supports stamps
    hipStream_t stamp_stream = nullptr;        // ... on THIS stream only (the main chain): a launch on a strip / preparation
                                               // stream must not consume a stamp of the main stream's phase clock
    bool phase_events = true;        // PAPOF_PHASE_EVENTS=0: record only the total and the solver kernels' events (A/B of the events' cost)
    int sor_xcd_affine = 1;          // 0: off; 1: when there are at most 8 bands; 2: always (see sor.hip)      // all sweeps of a band on one XCD (block index -> task mapping, speed only)
    // sequence mode (papof_seq_*): the pyramid of the last pushed frame stays in the arena and becomes "frame 1" of
    // the next pair.  Valid only while the arena block, the frame shape and the pyramid plan stay the same.
    struct Seq {
        bool valid = false;
        int h = 0, w = 0, c = 0, levels = 0, slot = 0;  // slot: which of the two pyramid slots holds the last frame
        double ratio = 0;
        const char* arena_base = nullptr;
    } seq;
};

namespace papof {

inline unsigned long long* take_stamp(papof_handle* h) {
    if (h->stamp_stream && h->stream != h->stamp_stream) return nullptr;
    unsigned long long* s = h->next_stamp;
    h->next_stamp = nullptr;
    return s;
}
int stamp_only(papof_handle* h);  // kernels.hip

// ---- kernels.hip: launch wrappers (all asynchronous on h->stream) ----
int hwc_to_planar(papof_handle* h, const double* hwc, double* planar, int H, int W, int C, int frames = 1);
int hwc_u8_to_planar(papof_handle* h, const unsigned char* hwc, double* planar, int H, int W, int C, int frames = 1);
int planar_to_hwc(papof_handle* h, const double* planar, double* hwc, int H, int W, int C);
int filter_h(papof_handle* h, const double* src, double* dst, int H, int W, int planes, const Taps& f,
             const Rect* rc = nullptr);
int filter_v(papof_handle* h, const double* src, double* dst, int H, int W, int planes, const Taps& f,
             const Rect* rc = nullptr);
int filter_hv(papof_handle* h, const double* src, double* dst, double* tmp, int H, int W, int planes, const Taps& fh,
              const Taps& fv);  // both passes in one launch (same bits); `tmp` only for half-widths beyond the fused kernel's
int resize(papof_handle* h, const double* src, double* dst, int sh, int sw, int planes, int dh, int dw, double xr,
           double yr, bool use_post, double post, const Rect* rc = nullptr);
int im2feature(papof_handle* h, const double* im, double* feat, int H, int W, int C, unsigned* nz = nullptr, int frames = 1,
               size_t nz_stride = 0);  // frames > 1: contiguous frames of a batch (common.h: BatchK)
int central3_planes(papof_handle* h, const double* src, double* gx, double* gy, double* gxy, int H, int W, int planes);
int warp_bilinear(papof_handle* h, const double* im1, const double* im2, const double* vx, const double* vy,
                  double* out, int H, int W, int planes, const Rect* rc = nullptr);
int smooth_v_blend(papof_handle* h, const double* tmp, const double* im1s, double* blend, double* imdt, int H,
                   int W, int planes, const Rect* rc = nullptr);
int smooth_hv_blend(papof_handle* h, const double* warp, const double* im1s, double* blend, double* imdt, int H,
                    int W, int planes, int row0 = 0, int row1 = -1);  // rows row0 .. row1-1 (-1: to the last row)
int warp_smooth_blend(papof_handle* h, const double* im1, const double* im2, const double* u, const double* v,
                      const double* im1s, double* blend, double* imdt, int H, int W, int planes, unsigned* wit = nullptr,
                      double* phi_out = nullptr);  // phi_out: also phi of (u, v) (compute_phi without an increment)  // warp folded in
int compute_phi(papof_handle* h, const double* u, const double* v, const SorPlanes* prev, double* phi, int H, int W,
                const Rect* rc = nullptr);
int assemble_system(papof_handle* h, const double* blend, const double* imdt, const double* phi, const double* u,
                    const double* v, int H, int W, int planes, double alpha, double omega, const SorPlanes& out,
                    double* opt_imdx2, double* opt_imdy2, const SorPlanes* prev, const Rect* rc = nullptr,
                    const double* gm = nullptr, const double* lap = nullptr);
int flow_system(papof_handle* h, const double* im1, const double* im2, const double* u, const double* v, const double* im1s,
                int H, int W, int planes, double alpha, double omega, const SorPlanes& out, unsigned* wit = nullptr,
                int row0 = 0, int row1 = -1,  // rows row0 .. row1-1 (a rank's range of rows: tiles.hip)
                int batch = 1, const BatchK* bk = nullptr);  // pairs of a batch (blockIdx.y), each pointer advanced by its stride
int lap_scratch_doubles();
int lap_rows_check(papof_handle* h, const double* im1, const double* im2, const double* u, const double* v, int H, int W,
                   int C, int r0, int r1, unsigned* wit, unsigned* val, unsigned mark);  // rows r0 .. r1-1, every pixel
bool lap_one_block_level(int H, int W);  // k_warp_smooth_blend runs one block per channel: exhaustive check of the guard
int lap_small_check(papof_handle* h, const double* im1, const double* im2, const double* u, const double* v, int H, int W,
                    int C, unsigned* wit, int batch = 1, const BatchK* bk = nullptr);  // the same check (EVERY pixel: a witness, or
                                                                                     // "no valid sample at all") for a given flow
int est_laplacian_noise(papof_handle* h, const double* im1, const double* im2, const double* u, const double* v, int H,
                        int W, int C, double* lap, double* scratch);  // the exact pass of the Laplacian-noise guard
int gm_scratch_doubles();
int est_gaussian_mixture(papof_handle* h, const double* im1, const double* im2, int H, int W, int C, double* gm,
                         double* scratch);
int laplacian(papof_handle* h, const double* in, const double* weight, double* out, int H, int W);
int update_and_warp(papof_handle* h, const SorPlanes& sp, double* u, double* v, const double* im1,
                    const double* im2, double* warp, int H, int W, int planes, bool do_warp = true);
int bicubic_warp(papof_handle* h, const double* im1, const double* im2, const double* gx, const double* gy,
                 const double* gxy, const double* vx, const double* vy, double* out_hwc, int H, int W, int C,
                 const Rect* rc = nullptr, bool planar_out = false, bool clamp = true, int batch = 1, const BatchK* bk = nullptr);
int flow_quantize16(papof_handle* h, const double* vx, const double* vy, unsigned short* q, size_t n);
int flow_dequantize16(papof_handle* h, const unsigned short* q, double* vx, double* vy, size_t n);
int flow_to_bgr(papof_handle* h, const double* vx, const double* vy, size_t n, double* partial, unsigned char* bgr);
int update_warp_phi(papof_handle* h, const SorPlanes& sp, const double* u, const double* v, double* u_out, double* v_out,
                    const double* im1, const double* im2, double* warp, double* phi_out, int H, int W, int planes,
                    bool do_warp = true, int row0 = 0, int row1 = -1, unsigned* wit = nullptr, int batch = 1,
                    const BatchK* bk = nullptr);
int update_flow(papof_handle* h, const SorPlanes& sp, double* u, double* v, int H, int W, const Rect& r);
int sor_prep(papof_handle* h, const double* phi, const double* imdxy, const double* imdx2, const double* imdy2,
             const double* rhs1, const double* rhs2, int H, int W, double alpha, double omega, const SorPlanes& out);
int sor_unpack(papof_handle* h, const SorPlanes& sp, double* du, double* dv, int H, int W);
Taps gaussian_taps(double sigma, int fsize);
Taps smooth5_taps();
Taps deriv5_taps();
Taps central3_taps();

// ---- sor.hip ----
// bt: B solves of one shape in one launch (api.hip: flow_batch) -- the operands of pair p are sp's advanced by these strides
// (doubles; counters: unsigneds; `tiny`: the row-major planes of k_sor_tiny's levels); exact order only, the plain one-sweep-per-wave
// kernel or k_sor_tiny, counters cleared ahead by the caller (papof_handle::sor_prog_next)
struct SorBatch {
    int n;
    size_t coef, d, prog, tiny;
};
int sor_solve(papof_handle* h, const SorPlanes& sp, int H, int W, double alpha, double omega, int n_sor, int mode,
              const SorBatch* bt = nullptr);
int sor_redblack_halfsweep(papof_handle* h, const SorPlanes& sp, int H, int W, double alpha, double omega, int colour,
                           const Rect& r);
int sor_blocked_depth(const papof_handle* h, int mode, int H, int W);  // half-sweeps per launch the blocked solver uses
int sor_blocked_launch(papof_handle* h, const SorPlanes& sp, int H, int W, double alpha, double omega, int mode, int g,
                       int hs0, const Rect& out, const double* su, const double* sv, double* du, double* dv);
int sor_plan(const papof_handle* h, int H, int W, int n_sor, int mode, int* launches, int* depth);
int sor_check(papof_handle* h);  // after a stream sync: PAPOF_ETIMEOUT if a device-side wait expired
// one solve as strips of bands on several streams (sor.hip): can this bound layout be solved in strips; clear the
// counters of all `n_solves` solves of a level (before the streams fork); launch bands b0 .. b1-1 of solve `solve_idx`
bool sor_strips_supported(const papof_handle* h, const SorPlanes& sp, int n_sor);
int sor_strips_begin(papof_handle* h, const SorPlanes& sp, int n_sor, int n_solves);
// `split`: the ranges of bands of one solve live in different handles (tiles.hip: bands_flow; sor.hip: ExactArgs)
struct SorSplit {
    double* peer_du;      // (du, dv) planes of the handle that runs band b1 (same bound layout), null: none
    unsigned* peer_prog;  // its counters of this solve
    bool top_cut, bot_cut;
};
int sor_solve_bands(papof_handle* h, const SorPlanes& sp, int H, int W, double alpha, double omega, int n_sor,
                    unsigned* prog, int b0, int b1, const SorSplit* split = nullptr, int k0 = 0, int k1 = -1);  // sweeps k0 .. k1-1 (split only)
int sor_bind_plain(papof_handle* h, SorPlanes& sp, int H, int W, int n_sor);
bool sor_tiny_fits(const papof_handle* h, int H, int W, int n_sor);
constexpr size_t kTinyMaxCells = 8192;  // upper bound of what sor_tiny_fits() accepts (registers of one workgroup)
int sor_alloc_tiny_planes(Arena& A, size_t cells, SorPlanes& sp);  // exact order inside one workgroup (k_sor_tiny; row-major operands)  // as sor_bind, one sweep per wave / workgroup
// progress counters cleared ahead of the solves that use them (flow_device: all of a call's, on the preparation stream)
size_t sor_counters_words(int H, int W, int n_sor);
int sor_counters_ensure(papof_handle* h, size_t words);
unsigned* sor_counters_clear(papof_handle* h, size_t offset_words, size_t words);
int sor_alloc_planes(Arena& A, int H, int W, int mode, int n_sor_cap, SorPlanes& sp);  // carve the operand planes
int sor_bind(papof_handle* h, SorPlanes& sp, int H, int W, int n_sor);  // choose the layout of the next solves (skew mode)
int sor_group_size(const papof_handle* h, int H, int W, int n_sor);   // sweeps per workgroup the solver will use
int sor_reset_planes(papof_handle* h, const SorPlanes& sp);    // zero all padding of the bound layout
int sor_reset_planes_batch(papof_handle* h, const SorPlanes& sp, int batch, size_t stride);  // ... of every pair of a batch
int sor_probe_dpp(papof_handle* h);  // sets h->use_dpp after checking the cross-lane DPP semantics on the device

}  // namespace papof
