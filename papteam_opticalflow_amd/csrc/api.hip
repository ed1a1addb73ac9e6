// papteam_opticalflow_amd/csrc/api.hip -- C ABI (include/papof.h) and the device-resident orchestrator
// that replaces OpticalFlow::Coarse2FineFlow (/root/reference/Code/Serial/src/OpticalFlow.cpp:735-903)
// and Coarse2FineFlowWrapper (src/Coarse2FineFlowWrapper.cpp:14-51).
//
// One call = one H2D of the two frames, every pyramid level / outer iteration / sweep on the device (a main stream and a
// preparation stream for the flow-independent work, no host round trip in between), one D2H of vx, vy, warpI2.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>

#include "common.h"
#include "flow_internal.h"

namespace papof {

namespace {
thread_local std::string g_last_error;
}

void set_last_error(const char* what, hipError_t e, const char* file, int line) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_last_error = buf;
}
const char* last_error() { return g_last_error.c_str(); }
void set_last_error_text(const std::string& text) { g_last_error = text; }

CopyPool::CopyPool(int workers) {
    for (int i = 0; i < workers; i++) threads_.emplace_back([this] { loop(); });
}

CopyPool::~CopyPool() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    wake_.notify_all();
    for (auto& t : threads_) t.join();
}

void CopyPool::loop() {
    for (;;) {
        Job j;
        {
            std::unique_lock<std::mutex> lk(mu_);
            wake_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
            if (jobs_.empty()) return;  // stop
            j = jobs_.back();
            jobs_.pop_back();
        }
        std::memcpy(j.dst, j.src, j.n);
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
}

void CopyPool::copy(char* dst, const char* src, size_t n) {
    const int parts = (int)threads_.size() + 1;
    const size_t part = (n / parts + 4095) & ~size_t(4095);
    int queued = 0;
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (int t = 1; t < parts; t++) {
            const size_t off = (size_t)t * part;
            if (off >= n) break;
            jobs_.push_back(Job{dst + off, src + off, std::min(part, n - off)});
            queued++;
        }
        pending_ += queued;
    }
    if (queued) wake_.notify_all();
    std::memcpy(dst, src, std::min(part, n));
    if (queued) {
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
    }
}

static double wall() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// generous upper bound of the arena one call needs: every buffer is at most level-0 sized
size_t arena_bytes_for(int H, int W, int C, int levels, int n_sor_max, double ratio) {
    const size_t np = (size_t)H * W;
    const int fc = (C == 3) ? 5 : (C == 1 ? 3 : C);
    size_t per_level = 0;  // features of both frames and the smoothed features of frame 1, kept for EVERY level
    {
        std::vector<Level> L;
        std::vector<PyrPlan> plan;
        if (levels >= 1 && pyramid_plan(H, W, ratio, levels, L, plan) == PAPOF_OK)
            for (const Level& l : L) per_level += ((size_t)l.w * l.h * fc * sizeof(double) + 256) * 3;
    }
    size_t sor_cells = 0, sor_cells_d = 0;
    skew_capacity(H, W, n_sor_max, sor_cells, sor_cells_d);
    size_t planes = 0;
    planes += (size_t)2 * C * 5;           // two pyramids: sum of ratio^(2i) < 2.3 for ratio<=.75; 5 is safe to .98
    if (levels > 8) planes += (size_t)2 * C * levels;  // (ratio .98 decays slowly: bound by level count)
    planes += (size_t)2 * C + 2 * C;       // staging of the interleaved inputs + pyramid temporaries
    planes += (size_t)7 * fc;              // F1, F2, warp, F1 smoothed, h-pass temp, blend, imdt
    planes += 8;                           // u, v, resized u, v, phi + slack
    planes += (size_t)3 * C + C;           // bicubic derivative planes + interleaved output
    planes += (size_t)3 * fc;              // in-loop bicubic warping: derivative planes of the frame-2 features
    size_t bytes = planes * np * sizeof(double);
    bytes += 3 * (sor_cells + 2 * kLanes) * 16 + (sor_cells_d + 2 * kLanes) * 16 + 10 * np * sizeof(double);  // SOR operands
    bytes += per_level + np * fc * sizeof(double);  // + the preparation stream's own filter temporary
    bytes += 8 * (std::min<size_t>(kTinyMaxCells, np) * sizeof(double) + 256);  // the row-major operands of k_sor_tiny's levels
    bytes += (size_t)64 * 4096;            // alignment slack
    return bytes;
}

static size_t sor_scratch_bytes(int H, int W, int n_sor) {  // the four paired operand planes of one solve
    size_t cells = 0, cells_d = 0;
    skew_capacity(H, W, n_sor, cells, cells_d);
    return 4 * (cells + 128) * 16 + (cells_d + 128) * 16;
}

int ensure_arena(papof_handle* h, size_t bytes) {
    if (bytes <= h->arena.cap) return PAPOF_OK;
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    if (h->arena.base) PAPOF_HIP(hipFree(h->arena.base));
    h->arena.base = nullptr;
    h->arena.cap = 0;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        set_last_error("hipMalloc(arena)", e, __FILE__, __LINE__);
        return PAPOF_ENOMEM;
    }
    h->arena.base = static_cast<char*>(p);
    h->arena.cap = bytes;
    return PAPOF_OK;
}

int check_params(const papof_params& P, int levels) {
    if (levels < 1) return PAPOF_EINVAL;
    if (P.n_inner < 1) return PAPOF_EINVAL;
    if (P.n_outer + 0 < 1 || P.n_sor < 1 || P.n_outer_per_level < 0 || P.n_sor_per_level < 0) return PAPOF_EINVAL;
    if (P.sor_mode < PAPOF_SOR_EXACT || P.sor_mode > PAPOF_SOR_JACOBI) return PAPOF_EINVAL;
    if (!(P.alpha > 0) || !(P.omega > 0)) return PAPOF_EINVAL;
    if (P.interpolation != PAPOF_INTERP_BILINEAR && P.interpolation != PAPOF_INTERP_BICUBIC) return PAPOF_EINVAL;
    if (P.noise_model != PAPOF_NOISE_LAPLACIAN && P.noise_model != PAPOF_NOISE_GMIXTURE) return PAPOF_EINVAL;
    return PAPOF_OK;
}

int pyramid_plan(int H, int W, double ratio, int nlev, std::vector<Level>& L, std::vector<PyrPlan>& plan) {
    if (ratio > 0.98 || ratio < 0.4) ratio = 0.75;  // :82-83
    const double base_sigma = 1 / ratio - 1;        // :89
    const int n = (int)(std::log(0.25) / std::log(ratio));  // :90
    const double n_sigma = base_sigma * n;          // :91
    L.assign(nlev, Level{});
    plan.assign(nlev, PyrPlan{});
    L[0].w = W;
    L[0].h = H;
    for (int i = 1; i < nlev; i++) {
        PyrPlan p;
        if (i <= n) {  // :95-100
            p.src_level = 0;
            p.sigma = base_sigma * i;
            p.fsize = (int)(p.sigma * 3);
            p.rate = std::pow(ratio, i);
        } else {  // :101-106
            p.src_level = i - n;
            p.sigma = n_sigma;
            p.fsize = (int)(n_sigma * 3);
            p.rate = (double)std::pow(ratio, i) * W / L[i - n].w;
        }
        p.sw = L[p.src_level].w;
        p.sh = L[p.src_level].h;
        if (p.fsize > kMaxFsize) return PAPOF_EINVAL;
        L[i].w = (int)((double)p.sw * p.rate);  // src/Image.h:755-756
        L[i].h = (int)((double)p.sh * p.rate);
        if (L[i].w < 1 || L[i].h < 1) return PAPOF_EINVAL;
        plan[i] = p;
    }
    return PAPOF_OK;
}

// One pyramid level from its source level: GaussianSmoothing + imresize (src/GaussianPyramid.cpp:97-99, :103-105).  Both
// filter passes are one launch; a half-width of 0 (level 1 at ratio 0.75: sigma * 3 truncates to 0) makes the
// smoothing the identity -- a single tap of exactly 1.0 accumulated from 0.0 -- and the resize reads the source directly
// (the resize accumulates from +0.0 as well, so not even the sign of a zero can differ).
int smooth_and_resize(papof_handle* h, const double* src, double* dst, double* tmp_a, double* tmp_b, const PyrPlan& p, int C,
                      int dh, int dw) {
    const Taps g = gaussian_taps(p.sigma, p.fsize);
    if (g.fsize == 0 && g.t[0] == 1.0) return resize(h, src, dst, p.sh, p.sw, C, dh, dw, p.rate, p.rate, false, 0.0);
    PAPOF_TRY(filter_hv(h, src, tmp_b, tmp_a, p.sh, p.sw, C, g, g));
    return resize(h, tmp_b, dst, p.sh, p.sw, C, dh, dw, p.rate, p.rate, false, 0.0);
}

int build_pyramid(papof_handle* h, const std::vector<Level>& L, const std::vector<PyrPlan>& plan, int C, bool second,
                  double* tmp_a, double* tmp_b) {
    for (size_t i = 1; i < L.size(); i++) {
        const PyrPlan& p = plan[i];
        const double* src = second ? L[p.src_level].p2 : L[p.src_level].p1;
        double* dst = second ? L[i].p2 : L[i].p1;
        PAPOF_TRY(smooth_and_resize(h, src, dst, tmp_a, tmp_b, p, C, L[i].h, L[i].w));
    }
    return PAPOF_OK;
}

struct SolveBuffers {
    double *im1s, *tmp, *blend, *imdt, *phi;
    SorPlanes sp;
    SorPlanes sp_tiny{};  // exact order: row-major operands of the levels small enough for k_sor_tiny (sor.hip), else unused
    // the reference's non-default branches (unreachable from its Python entry point; papof_params::interpolation /
    // noise_model): derivative planes of the frame-2 features for in-loop bicubic warping (src/OpticalFlow.cpp:517-521),
    // and the Gaussian-mixture parameters + reduction scratch (:359-367, :539-591)
    double *bgx = nullptr, *bgy = nullptr, *bgxy = nullptr;
    double *gm = nullptr, *gm_scratch = nullptr;
};

int alloc_solve_buffers(Arena& A, int H, int W, int fc, int mode, int n_sor_cap, SolveBuffers& B) {
    const size_t np = (size_t)H * W;
    B.im1s = A.f64(np * fc);
    B.tmp = A.f64(np * fc);
    B.blend = A.f64(np * fc);
    B.imdt = A.f64(np * fc);
    B.phi = A.f64(np);
    PAPOF_TRY(sor_alloc_planes(A, H, W, mode, n_sor_cap, B.sp));
    // (the largest level k_sor_tiny can get is the frame itself: a frame of a few pixels must not ask for 8192 cells)
    if (mode == PAPOF_SOR_EXACT)
        PAPOF_TRY(sor_alloc_tiny_planes(A, std::min<size_t>(kTinyMaxCells, (size_t)H * W), B.sp_tiny));
    return A.overflow ? PAPOF_ENOMEM : PAPOF_OK;
}

// buffers of the non-default branches; gm is (re)set to GaussianMixture::reset()'s values (src/NoiseModel.h:97-107)
int alloc_branch_buffers(papof_handle* h, Arena& A, int H, int W, int fc, int interpolation, int noise_model,
                         SolveBuffers& B) {
    const size_t np = (size_t)H * W;
    if (interpolation == PAPOF_INTERP_BICUBIC) {
        B.bgx = A.f64(np * fc);
        B.bgy = A.f64(np * fc);
        B.bgxy = A.f64(np * fc);
    }
    if (noise_model == PAPOF_NOISE_GMIXTURE) {
        B.gm = A.f64(5 * 8);
        B.gm_scratch = A.f64((size_t)gm_scratch_doubles());
        if (A.overflow || fc > 8) return fc > 8 ? PAPOF_EINVAL : PAPOF_ENOMEM;
        double init[40];
        for (int k = 0; k < fc; k++) {
            init[k] = 0.95;
            init[fc + k] = 0.05;
            init[2 * fc + k] = 0.5;
            init[3 * fc + k] = 0.05 * 0.05;
            init[4 * fc + k] = 0.5 * 0.5;
        }
        PAPOF_HIP(hipMemcpyAsync(B.gm, init, sizeof(double) * 5 * fc, hipMemcpyHostToDevice, h->stream));
        PAPOF_HIP(hipStreamSynchronize(h->stream));  // `init` is a stack array
    }
    return A.overflow ? PAPOF_ENOMEM : PAPOF_OK;
}

// derivative planes {-.5, 0, .5} of the frame-2 features of a level (Image::warpImageBicubicRef, src/Image.h:2587-2595)
int bicubic_planes(papof_handle* h, const double* f2, int H, int W, int fc, SolveBuffers& B) {
    PAPOF_TRY(central3_planes(h, f2, B.bgx, B.bgy, B.bgxy, H, W, fc));
    return PAPOF_OK;
}

// The Laplacian-noise guard.  After every outer iteration the reference estimates, per channel, the mean of |Im1 - warpIm2|
// over its samples in (0, 1e6) (estLaplacianNoise, src/OpticalFlow.cpp:594-639; 0.001 when there is none) and, in the next
// linear system, leaves psi of a channel at 0 while that mean is below 1E-20 (:399-400 on the Psi_1st reset at :333).  No
// 8-bit frame gets there, and the warped image is never materialised on this path, so the reduction would be a gather pass
// per iteration for nothing.  Instead a call runs OPTIMISTICALLY -- no guard -- while it collects proofs that the guard
// could not have tripped: a WITNESS per channel and iteration (one sampled pixel per block of the update kernel with
// |d| >= 2e-20 x pixels: a sum of positives is at least its largest term), or a channel that is all zero in both frames of the
// level (no valid sample: LapPara = 0.001).  A call that ends with a channel lacking both is run AGAIN in the EXACT pass --
// the estimate after every update (kernels.hip: est_laplacian_noise), the guard in the assembly -- and the handle stays in
// the exact pass while its inputs stay that way (duplicate frames, flat synthetic images).  Results: the reference's, either way
// (golden `stage_lapguard`); cost on ordinary frames: one sampled warp per block of a kernel that runs anyway.
struct LapGuard {
    bool on = false;        // Laplacian noise model, at most 8 feature channels, the handle has its flag block
    bool exact = false;     // the exact pass
    bool collect = false;   // witnesses are collected (always, except inside a hipGraph: nobody could act on them)
    bool nz_known = false;  // the non-zero flags of the feature channels are collected too (im2feature ran for this call)
    unsigned* flags = nullptr;  // device: papof_handle::lap_flags_dev
    double* lap = nullptr;      // device: LapPara (exact pass)
    double* scratch = nullptr;
    unsigned epoch = 0;  // what a set flag of this pass holds (papof_handle::lap_epoch)
    int slot = 0;  // outer iterations so far
    std::vector<int> slot_level, slot_channels;
    unsigned* wit(int s) const { return on && collect && s >= 0 && s < kLapMaxSlots ? flags + lap_wit_word(s) : nullptr; }
    unsigned* nz(int level) const { return on && collect && nz_known ? flags + lap_nz_word(level) : nullptr; }
    const double* guard() const { return on && exact ? lap : nullptr; }
    // after the stream has drained and the flags are on the host: estimates that were consulted without a proof
    int unknown(const unsigned* host_flags) const {
        if (!on || !collect) return 0;
        if (slot > kLapMaxSlots) return 1;  // more outer iterations than witness slots: take the exact pass
        int n = 0;
        for (int s = 0; s + 1 < slot; s++)  // the estimate behind the last iteration of a call is never consulted
            for (int c = 0; c < slot_channels[s]; c++) {
                const unsigned w = host_flags[lap_wit_word(s) + c];
                const bool witness = w == epoch || w == (epoch ^ kLapNone);  // a sample, or an exhaustive "no valid sample at all"
                const bool all_zero = nz_known && host_flags[lap_nz_word(slot_level[s]) + c] != epoch;
                n += !(witness || all_zero);
                if (!(witness || all_zero) && std::getenv("PAPOF_LAP_TRACE"))
                    std::fprintf(stderr, "[lap guard] no proof: slot %d of %d, level %d, channel %d\n", s, slot, slot_level[s], c);
            }
        return n;
    }
};

// OpticalFlow::SmoothFlowSOR (src/OpticalFlow.cpp:238-536) for one level, everything on the device.
// genInImageMask (:278) does not influence the results (the mask is never read) and is not executed;
// estLaplacianNoise (:530) only feeds a `< 1e-20` guard: see LapGuard above.
// (u, v) and (ua, va) are two pairs of planes: every outer iteration writes the updated flow into the other pair (its
// update kernel also reads the neighbours' old values, for phi) and the references are swapped -- on return `u`, `v` name
// the planes that hold the result.
int smooth_flow(papof_handle* h, const double* f1, const double* f2, double* warp, double*& u, double*& v, double*& ua,
                double*& va, int H, int W, int fc, double alpha, int n_outer, int n_inner, int n_sor, double omega,
                int mode, SolveBuffers& B, PhaseClock& clk, const double* im1s_ready = nullptr,
                bool final_warp = true, unsigned* prog = nullptr, size_t prog_per_solve = 0, double* out_u = nullptr,
                double* out_v = nullptr, bool fold_warp = false, LapGuard* lg = nullptr, int level = 0) {
    // fold_warp: nobody but getDxs' smoothing reads the warped frame 2 (default branches, caller = flow_device): it is
    // evaluated inside the smoothing kernel from (u, v) and never written -- `warp` is then neither read nor written here
    // prog: cleared progress counters of this level's solves (solve i: prog + i * prog_per_solve), or null;
    // out_u / out_v: where the LAST update writes the flow instead of the other pair of planes (the caller's result buffers)
    const Taps g = smooth5_taps();
    int solve_idx = 0;
    // exact order on a plane small enough to be solved inside one workgroup: row-major operands + k_sor_tiny (sor.hip)
    const bool tiny = mode == PAPOF_SOR_EXACT && B.sp_tiny.phi && (size_t)H * W <= kTinyMaxCells && sor_tiny_fits(h, H, W, n_sor);
    SorPlanes& SP = tiny ? B.sp_tiny : B.sp;
    const double* im1s = im1s_ready;  // smoothed frame 1: constant within the level (prepared ahead by flow_device)
    if (!im1s) {
        clk.phase(PAPOF_T_PHASE1_GENERATE);
        PAPOF_TRY(filter_hv(h, f1, B.im1s, B.tmp, H, W, fc, g, g));
        im1s = B.im1s;
    }
    // ONE kernel from the flow to the linear system (k_flow_system) where it applies: default branches, exact-order layout
    const char* const fused_sw = std::getenv("PAPOF_FUSED_SYSTEM");  // A/B switch, and how the tests reach the pair of kernels
    const bool fused_env = !(fused_sw && fused_sw[0] == '0');
    // (a one-block level keeps the pair of kernels while the guard collects proofs: their exhaustive check needs ONE block)
    const bool fused_system = fused_env && fold_warp && n_inner == 1 && !B.gm && (fc == 5 || fc == 3) &&
                              !(lg && lg->guard()) && !(lg && lg->on && lap_one_block_level(H, W));
    for (int count = 0; count < n_outer; count++) {
        clk.phase(fused_system ? kTimerFused : (int)PAPOF_T_PHASE1_GENERATE);
        if (fused_system)
            PAPOF_TRY(flow_system(h, f1, f2, u, v, im1s, H, W, fc, alpha, omega, SP,
                                  lg && count > 0 ? lg->wit(lg->slot - 1) : nullptr));
        else if (fold_warp)
            // warp + both passes + blend + imdt; the warp at the flow the previous iteration left is what that iteration's
            // noise estimate is about: its witnesses are taken here (LapGuard)
            PAPOF_TRY(warp_smooth_blend(h, f1, f2, u, v, im1s, B.blend, B.imdt, H, W, fc,
                                        lg && count > 0 && !B.gm ? lg->wit(lg->slot - 1) : nullptr,
                                        B.phi));  // ... and phi of the flow as it stands (Phase2, :295-331), by the way
        else
            PAPOF_TRY(smooth_hv_blend(h, warp, im1s, B.blend, B.imdt, H, W, fc));  // both passes + blend + imdt, fused
        // inner fixed-point iterations (src/OpticalFlow.cpp:290-506): after the first one, phi is taken at u + du and
        // psi at imdt + imdx*du + imdy*dv with the increment of the previous solve; the solve itself restarts at 0
        for (int hh = 0; hh < n_inner; hh++) {
            const SorPlanes* prev = hh == 0 ? nullptr : &SP;
            // phi (Phase2): of the level's initial flow, and inside further inner iterations (at u + du), by its own kernel;
            // for every later outer iteration the previous iteration's update kernel has written it already
            if ((count == 0 && !fold_warp) || hh > 0) {  // (fold_warp: warp_smooth_blend has written phi of (u, v))
                clk.phase(PAPOF_T_PHASE2_DERIVATIVES);
                PAPOF_TRY(compute_phi(h, u, v, prev, B.phi, H, W));
            }
            // psi (Phase3, src/OpticalFlow.cpp:377-406) and the linear system (Phase4, :414-448) are ONE kernel here: its
            // time is recorded under Phase4 and apportioned between the two timers when they are collected (kPsiShare)
            if (!fused_system) clk.phase(PAPOF_T_PHASE4_LINEARSYSTEM);
            if (!fused_system)
                PAPOF_TRY(assemble_system(h, B.blend, B.imdt, B.phi, u, v, H, W, fc, alpha, omega, SP, nullptr, nullptr,
                                          prev, nullptr, B.gm, lg ? lg->guard() : nullptr));
            // Phase5_SOR is the solver kernels' own duration (the roofline of the dominant kernel is priced on it): the
            // events are recorded by sor_solve() right around its kernel(s), BEHIND the memset nodes that prepare a solve,
            // which therefore still count as Phase4
            h->sor_mark = [](void* c, int on) {
                static_cast<PhaseClock*>(c)->phase(on ? PAPOF_T_PHASE5_SOR : PAPOF_T_PHASE6_UPDATE);
            };
            h->sor_mark_ctx = &clk;
            h->sor_prog_next = prog ? prog + (size_t)solve_idx * prog_per_solve : nullptr;
            solve_idx++;
            const int rc_solve = sor_solve(h, SP, H, W, alpha, omega, n_sor, mode);
            h->sor_prog_next = nullptr;
            h->sor_mark = nullptr;
            h->sor_mark_ctx = nullptr;
            PAPOF_TRY(rc_solve);
        }
        // Phase6 (opened by the solver's end mark): u += du, v += dv and the re-warp of frame 2 (:513-521)
        // the next outer iteration's phi, fused in -- unless its warp_smooth_blend writes it (fold_warp: there it costs six
        // row-contiguous loads per pixel in a fifth of the blocks; here, three scattered reads of the increment instead of one)
        double* const phi_next = count + 1 < n_outer && !fold_warp ? B.phi : nullptr;
        // the re-warp after the LAST outer iteration of a level (:516) is read by nobody when the caller is flow_device (the
        // next level warps anew, the result is the bicubic warp of the originals): final_warp = false skips it
        const bool rewarp = !fold_warp && !B.bgx && (final_warp || count + 1 < n_outer || B.gm);
        // witnesses of the Laplacian-noise guard (LapGuard); the bicubic branch has no sampled warp: it runs the exact pass
        // -- and taken by the next iteration's warp_smooth_blend where there is one on this level
        const bool wit_later = fold_warp && count + 1 < n_outer;
        // (a one-block level gets the exhaustive check of lap_small_check() behind its last update instead of samples)
        const bool wit_small = fold_warp && !wit_later && lap_one_block_level(H, W);
        unsigned* const wit = lg && !B.gm && !B.bgx && !wit_later && !wit_small ? lg->wit(lg->slot) : nullptr;
        if (out_u && out_v && count + 1 == n_outer) {  // the level's result goes straight to the caller's buffers
            PAPOF_TRY(update_warp_phi(h, SP, u, v, out_u, out_v, f1, f2, warp, phi_next, H, W, fc, rewarp, 0, -1, wit));
            u = out_u;
            v = out_v;
        } else {
            PAPOF_TRY(update_warp_phi(h, SP, u, v, ua, va, f1, f2, warp, phi_next, H, W, fc, rewarp, 0, -1, wit));
            std::swap(u, ua);
            std::swap(v, va);
        }
        if (B.bgx)  // interpolation == Bicubic: warpImageBicubicRef + threshold() on the feature planes
            PAPOF_TRY(bicubic_warp(h, f1, f2, B.bgx, B.bgy, B.bgxy, u, v, warp, H, W, fc, nullptr, true, true));
        if (lg && lg->on && !B.gm && wit_small && lg->wit(lg->slot))
            PAPOF_TRY(lap_small_check(h, f1, f2, u, v, H, W, fc, lg->wit(lg->slot)));
        if (lg && lg->on && !B.gm) {
            lg->slot_level.push_back(level);
            lg->slot_channels.push_back(fc);
            lg->slot++;
            if (lg->exact) {  // :530, on the flow just updated
                if (B.bgx)
                    PAPOF_TRY(est_laplacian_noise(h, f1, warp, nullptr, nullptr, H, W, fc, lg->lap, lg->scratch));
                else
                    PAPOF_TRY(est_laplacian_noise(h, f1, f2, u, v, H, W, fc, lg->lap, lg->scratch));
            }
        }
        if (B.gm) PAPOF_TRY(est_gaussian_mixture(h, f1, warp, H, W, fc, B.gm, B.gm_scratch));  // :524-528
    }
    clk.phase(-1);
    return PAPOF_OK;
}

// =================================================================================================
// STRIPS: one level of the exact-order path as S horizontal strips, each on its own stream.
//
// The exact-order solve is a wave front: band b starts one band hop (63 steps + a hand-off) after band b-1 and finishes
// as much later, so the top of the plane is final long before the bottom, and the bottom bands of the NEXT solve could
// not start before the top ones anyway.  A level is therefore cut into strips of solver bands; per outer iteration a
// strip's chain is [update + warp + phi] -> [smoothing + blend] -> [assembly] -> [its bands of the solve], enqueued on
// the strip's own stream.  The non-solver kernels of the upper strips run while the lower strips still solve, the
// non-solver kernels of the lowest strip while the upper strips' next solve climbs its ramp: the same kernels, the
// same operations per pixel, only the row ranges and the order in time differ (results are bit-identical).
//
// Row ranges.  A stage whose stencil reaches h rows further down than its output can only be run on rows whose inputs
// are final, so the boundary between strip s-1 and strip s moves UP from stage to stage: with F = the first row whose
// final (du, dv) lies in the first band of strip s, the update kernel of strip s-1 ends at F - 1 (phi of row i reads
// row i + 1), the smoothing at F - 3 (5 x 5), the assembly at F - 5 (5-point derivatives of the blend), and the next
// solve's boundary is the last band boundary at or above that row: boundaries move up by one band per solve (two when the
// sweep count exceeds 57).  The strip below starts each stage where the strip above ended; what it reads across the
// boundary is ordered by ONE event per iteration (recorded behind the upper strip's assembly).  Inside a solve the
// strips meet through the solver's progress counters (sor.hip: sor_solve_bands).  Dependencies between stages point
// from upper to lower strips only -- except the solver's write-after-read guard, which keeps an upper strip within two
// sweeps per band of the strip below.
// =================================================================================================
struct LevelInit {  // how the level's initial flow and warp come about (src/OpticalFlow.cpp:801-816)
    bool coarsest;          // u = v = 0 and warp = frame 2's features: already enqueued by the caller
    const double *pu, *pv;  // else: the coarser level's flow, up-sampled and scaled per strip, then warped
    int ph, pw;
    double xr, yr, inv;
};

struct StripSchedule {
    int S = 1, n_solves = 0;
    // per solve n = 0 .. n_solves (n_solves = the final update) and boundary s = 0 .. S: first band of strip s in solve n
    // (beta), first row of strip s in the update / warp stage (rU), phi (rP), smoothing (rS), assembly (rA)
    std::vector<int> beta, rU, rP, rS, rA;
    int idx(int n, int s) const { return n * (S + 1) + s; }
};

bool plan_strips(const papof_handle* h, const SorPlanes& sp, int H, int n_sor, int n_solves, int want,
                 StripSchedule& out) {
    if (want < 2 || n_solves < 1 || !sor_strips_supported(h, sp, n_sor)) return false;
    const int BR = sp.sd.band_rows, koff = sp.sd.koff, nb = sp.sd.nb;
    const int d = (koff + 5 + BR - 1) / BR;  // bands a boundary moves up per solve
    for (int S = std::min(want, 4); S >= 2; --S) {
        StripSchedule q;
        q.S = S;
        q.n_solves = n_solves;
        const size_t cells = (size_t)(n_solves + 1) * (S + 1);
        q.beta.assign(cells, 0);
        q.rU.assign(cells, 0);
        q.rP.assign(cells, 0);
        q.rS.assign(cells, 0);
        q.rA.assign(cells, 0);
        // boundaries of the first solve: even shares of the bands, moved down by half of what they will climb
        std::vector<int> b0(S + 1, 0);
        b0[S] = nb;
        int top = std::min(nb - 1, (H - 6) / BR);  // the last boundary's first stage (row BR * beta + 4) lies inside the plane
        for (int s = S - 1; s >= 1; --s) {
            b0[s] = std::min(top, (s * nb + S / 2) / S + d * (n_solves - 1) / 2);
            top = b0[s] - 1;
        }
        bool ok = true;
        for (int n = 0; n <= n_solves && ok; n++) {
            for (int s = 0; s <= S; s++) {
                const int i = q.idx(n, s);
                if (s == 0 || s == S) {
                    q.beta[i] = s == 0 ? 0 : nb;
                    q.rU[i] = q.rP[i] = q.rS[i] = q.rA[i] = s == 0 ? 0 : H;
                    continue;
                }
                if (n == 0) {
                    q.beta[i] = b0[s];
                    q.rU[i] = BR * b0[s] + 4;
                    q.rP[i] = q.rU[i] - 1;
                } else {
                    const int fin = BR * q.beta[q.idx(n - 1, s)] - koff;  // first row finalised by the strip's first band
                    q.rU[i] = fin - 1;
                    q.rP[i] = q.rU[i];
                }
                q.rS[i] = q.rU[i] - 2;
                q.rA[i] = q.rU[i] - 4;
                if (n > 0) q.beta[i] = q.rA[i] >= 0 ? q.rA[i] / BR : 0;
                if (n == n_solves) q.beta[i] = q.beta[q.idx(n - 1, s)];  // no solve follows the final update
            }
            for (int s = 1; s <= S && ok; s++) {
                const int i = q.idx(n, s), j = q.idx(n, s - 1);
                ok = q.beta[i] > q.beta[j] && q.rU[i] > q.rU[j] && q.rA[i] > q.rA[j] && q.rS[i] > q.rS[j] &&
                     q.rP[i] > q.rP[j] && q.rA[i] >= 0 && q.rU[i] <= H;
            }
        }
        if (ok) {
            out = q;
            return true;
        }
    }
    return false;
}

hipEvent_t strip_event(papof_handle* h) {
    if (h->strip_events_used == h->strip_events.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        h->strip_events.push_back(e);
    }
    return h->strip_events[h->strip_events_used++];
}

int smooth_flow_strips(papof_handle* h, const StripSchedule& q, const LevelInit& li, const double* f1, const double* f2,
                       double* warp, double*& u, double*& v, double*& ua, double*& va, int H, int W, int fc, double alpha,
                       int n_sor, double omega, SolveBuffers& B, PhaseClock& clk, PhaseClock& uclk, const double* im1s,
                       unsigned* prog, size_t prog_per_solve, double* out_u, double* out_v) {
    const int S = q.S, n_outer = q.n_solves;
    if (S < 2 || (int)h->strip_streams.size() < S - 1 || !im1s || !prog) return PAPOF_EINVAL;
    hipStream_t const main_stream = h->stream;
    const auto stream_of = [&](int s) { return s == S - 1 ? main_stream : h->strip_streams[s]; };
    struct StreamSwap {
        papof_handle* h;
        hipStream_t saved;
        StreamSwap(papof_handle* hh, hipStream_t s) : h(hh), saved(hh->stream) { h->stream = s; }
        ~StreamSwap() { h->stream = saved; }
    };
    clk.phase(PAPOF_T_ALLOCATION);
    hipEvent_t const fork = strip_event(h);  // (the counters of every solve were cleared before the streams fork: `prog`)
    if (!fork) return PAPOF_EDEVICE;
    PAPOF_HIP(hipEventRecord(fork, main_stream));
    PAPOF_HIP(hipStreamWaitEvent(stream_of(0), fork, 0));  // strip s > 0 waits for strip s - 1 in every iteration
    std::vector<hipEvent_t> done(S, nullptr);
    const auto sor_on_main = [](void* c, int on) {
        static_cast<PhaseClock*>(c)->phase(on ? PAPOF_T_PHASE5_SOR : PAPOF_T_PHASE6_UPDATE);
    };
    const auto sor_on_strip = [](void* c, int on) { static_cast<PhaseClock*>(c)->phase(on ? PAPOF_T_PHASE5_SOR : -1); };
    for (int n = 0; n <= n_outer; n++) {
        const bool last = n == n_outer;
        for (int s = 0; s < S; s++) {
            StreamSwap on_strip(h, stream_of(s));
            const bool crit = s == S - 1;  // the lowest strip's chain (main stream) carries the stamped phase timers
            if (s > 0) PAPOF_HIP(hipStreamWaitEvent(h->stream, done[s - 1], 0));
            const int i0 = q.idx(n, s), i1 = q.idx(n, s + 1);
            const Rect ru{0, q.rU[i0], W, q.rU[i1]}, rp{0, q.rP[i0], W, q.rP[i1]}, ra{0, q.rA[i0], W, q.rA[i1]};
            const double *un = u, *vn = v;  // the flow the assembly reads
            if (n == 0) {
                if (!li.coarsest) {  // src/OpticalFlow.cpp:809-814
                    if (crit) clk.phase(PAPOF_T_ALLOCATION);
                    PAPOF_TRY(resize(h, li.pu, u, li.ph, li.pw, 1, H, W, li.xr, li.yr, true, li.inv, &ru));
                    PAPOF_TRY(resize(h, li.pv, v, li.ph, li.pw, 1, H, W, li.xr, li.yr, true, li.inv, &ru));
                    PAPOF_TRY(warp_bilinear(h, f1, f2, u, v, warp, H, W, fc, &ru));
                }
                if (crit) clk.phase(PAPOF_T_PHASE2_DERIVATIVES);
                PAPOF_TRY(compute_phi(h, u, v, nullptr, B.phi, H, W, &rp));
            } else {  // Phase6 of iteration n - 1 with the phi of iteration n folded in (the main chain's Phase6 was
                      // opened by its solver's end mark)
                double* const wu = last && out_u && out_v ? out_u : ua;
                double* const wv = last && out_u && out_v ? out_v : va;
                PAPOF_TRY(update_warp_phi(h, B.sp, u, v, wu, wv, f1, f2, warp, last ? nullptr : B.phi, H, W, fc, !last,
                                          ru.y0, ru.y1));
                un = wu;
                vn = wv;
            }
            if (!last) {
                if (crit) clk.phase(PAPOF_T_PHASE1_GENERATE);
                PAPOF_TRY(smooth_hv_blend(h, warp, im1s, B.blend, B.imdt, H, W, fc, q.rS[i0], q.rS[i1]));
                if (crit) clk.phase(PAPOF_T_PHASE4_LINEARSYSTEM);
                PAPOF_TRY(assemble_system(h, B.blend, B.imdt, B.phi, un, vn, H, W, fc, alpha, omega, B.sp, nullptr, nullptr,
                                          nullptr, &ra, nullptr));
            }
            if (!crit) {
                done[s] = strip_event(h);
                if (!done[s]) return PAPOF_EDEVICE;
                PAPOF_HIP(hipEventRecord(done[s], h->stream));
            }
            if (!last) {
                h->sor_mark = crit ? +sor_on_main : +sor_on_strip;
                h->sor_mark_ctx = crit ? &clk : &uclk;
                const int rc = sor_solve_bands(h, B.sp, H, W, alpha, omega, n_sor, prog + (size_t)n * prog_per_solve,
                                               q.beta[i0], q.beta[i1]);
                h->sor_mark = nullptr;
                h->sor_mark_ctx = nullptr;
                PAPOF_TRY(rc);
            }
        }
        if (n == n_outer && out_u && out_v) {
            u = out_u;
            v = out_v;
        } else if (n > 0) {
            std::swap(u, ua);
            std::swap(v, va);
        }
    }
    // join: the final update of strip S - 2 waited for strip S - 3's, and so on -- and the main stream for S - 2's
    clk.phase(-1);
    return PAPOF_OK;
}

int feature_channels(int C) { return C == 3 ? 5 : (C == 1 ? 3 : C); }

void ensure_strip_streams(papof_handle* h) {  // without them levels are simply not cut
    while (h->strip_streams.size() < 3) {
        hipStream_t ss = nullptr;
        if (hipStreamCreateWithFlags(&ss, hipStreamNonBlocking) != hipSuccess) break;
        h->strip_streams.push_back(ss);
    }
}

// An interleaved HWC frame resident on the device: fp64 in [0,1] (the reference's buffers) or the decoded uint8
// samples, which are scaled by 1/255 while they are planarised (OpticalFlowCalculation.py:69-70).
struct FrameIn {
    const void* d;
    bool u8;
};
int load_frame(papof_handle* h, const FrameIn& f, double* planar, int H, int W, int C) {
    if (f.u8) return hwc_u8_to_planar(h, static_cast<const unsigned char*>(f.d), planar, H, W, C);
    return hwc_to_planar(h, static_cast<const double*>(f.d), planar, H, W, C);
}

// What a call does with the two pyramid slots at the head of the arena (a collection is a
// 102-frame video walked as 101 overlapping pairs, TestSuite.py:69-81):
//   kPair     both frames are given; nothing is kept;
//   kSeqPrime only `a` is given: build its pyramid and keep it (no flow);
//   kSeqNext  only `b` is given: frame 1 is the pyramid kept by the previous kSeqPrime / kSeqNext call, the pyramid of
//             `b` is built into the other slot and kept for the next push.
enum SeqOp { kPair, kSeqPrime, kSeqNext };

bool seq_matches(const papof_handle* h, int H, int W, int C, int levels, double ratio, size_t need) {
    const papof_handle::Seq& q = h->seq;
    return q.valid && q.h == H && q.w == W && q.c == C && q.levels == levels && q.ratio == ratio &&
           q.arena_base == h->arena.base && need <= h->arena.cap;
}

// The whole call on device-resident buffers: ONE pass (flow_device below runs it once, or twice: LapGuard).
int flow_pass(papof_handle* h, const FrameIn& fa, const FrameIn& fb, SeqOp op, int H, int W, int C, int levels,
              const papof_params& P, double* d_vx, double* d_vy, double* d_warp, double* timing, LapGuard* lg) {
    const double t_entry = wall();
    PAPOF_TRY(check_params(P, levels));
    double ratio = P.ratio;
    if (ratio > 0.98 || ratio < 0.4) ratio = 0.75;
    std::vector<Level> L;
    std::vector<PyrPlan> plan;
    PAPOF_TRY(pyramid_plan(H, W, P.ratio, levels, L, plan));
    const int n_sor_max = P.n_sor + (levels - 1) * P.n_sor_per_level;
    const size_t need = arena_bytes_for(H, W, C, levels, n_sor_max, P.ratio);
    if (op == kSeqNext && !seq_matches(h, H, W, C, levels, ratio, need)) {
        g_last_error = "sequence push does not continue the primed sequence (shape, levels, ratio or arena changed)";
        return PAPOF_EINVAL;
    }
    const int slot1 = op == kSeqNext ? h->seq.slot : 0;  // pyramid slot of frame 1
    h->seq.valid = false;  // re-established below once the kept pyramid is complete
    PAPOF_TRY(ensure_arena(h, need));
    Arena& A = h->arena;
    A.off = 0;
    A.overflow = false;
    h->events_used = 0;
    const size_t np0 = (size_t)H * W;
    const int fc = feature_channels(C);
    double tm[PAPOF_N_TIMERS + 1];  // + kTimerFused (flow_internal.h)
    std::memset(tm, 0, sizeof tm);
    const size_t tm_bytes = PAPOF_N_TIMERS * sizeof(double);  // what the caller gets

    // ---- hipGraph mode: eager on the first call with these arguments, captured on the second, replayed afterwards
    enum { kEager, kCapture, kReplay } gmode = kEager;
    GraphEntry* ge = nullptr;
    if (h->use_graph && op != kSeqPrime && P.phase_timing == 0 && P.noise_model == PAPOF_NOISE_LAPLACIAN) {
        GraphKey key;
        std::memset(&key, 0, sizeof key);
        key.H = H;
        key.W = W;
        key.C = C;
        key.levels = levels;
        key.op = (int)op;
        key.slot1 = slot1;
        key.u8 = fb.u8 ? 1 : 0;
        key.P = P;
        key.fa = op == kSeqNext ? nullptr : fa.d;
        key.fb = fb.d;
        key.vx = d_vx;
        key.vy = d_vy;
        key.warp = d_warp;
        key.arena_base = A.base;
        key.sync_base = h->sync_words;
        for (GraphEntry& e : h->graphs)
            if (std::memcmp(&e.key, &key, sizeof key) == 0) ge = &e;
        if (!ge) {
            if (h->graphs.size() >= 8) {  // a handful of shapes per handle is the use case; start over beyond that
                for (GraphEntry& e : h->graphs)
                    if (e.exec) hipGraphExecDestroy(e.exec);
                h->graphs.clear();
            }
            h->graphs.push_back(GraphEntry{});
            ge = &h->graphs.back();
            ge->key = key;
        }
        gmode = ge->exec ? kReplay : (ge->seen >= 1 ? kCapture : kEager);
        ge->seen++;
    }
    const auto keep_seq = [&](int slot) {
        h->seq.valid = true;
        h->seq.h = H;
        h->seq.w = W;
        h->seq.c = C;
        h->seq.levels = levels;
        h->seq.ratio = ratio;
        h->seq.slot = slot;
        h->seq.arena_base = A.base;
    };
    const auto launch_graph = [&]() -> int {  // replay + the only two timers a graph call has: total, and nothing else
        hipEvent_t e0, e1;
        PAPOF_HIP(hipEventCreate(&e0));
        PAPOF_HIP(hipEventCreate(&e1));
        hipEventRecord(e0, h->stream);
        hipError_t ge_rc = hipGraphLaunch(ge->exec, h->stream);
        hipEventRecord(e1, h->stream);
        hipError_t sy = hipStreamSynchronize(h->stream);
        float ms = 0;
        if (ge_rc == hipSuccess && sy == hipSuccess) hipEventElapsedTime(&ms, e0, e1);
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        if (ge_rc != hipSuccess || sy != hipSuccess) {
            set_last_error("hipGraphLaunch", ge_rc != hipSuccess ? ge_rc : sy, __FILE__, __LINE__);
            return PAPOF_EDEVICE;
        }
        if (P.sor_mode == PAPOF_SOR_EXACT) PAPOF_TRY(sor_check(h));
        tm[PAPOF_T_TOTAL] = ms * 1e-3;
        if (timing) std::memcpy(timing, tm, tm_bytes);
        if (op == kSeqNext) keep_seq(slot1 ^ 1);
        return PAPOF_OK;
    };
    if (gmode == kReplay) return launch_graph();
    const bool capturing = gmode == kCapture;
    if (capturing) {
        hipError_t ce = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal);
        if (ce != hipSuccess) {  // no graphs on this runtime: stay eager
            h->use_graph = false;
            gmode = kEager;
        }
    }
    const bool in_capture = gmode == kCapture;
    // All ten reference timers (src/OpticalFlow.cpp:850-860) are ALWAYS measured, with HIP events recorded on the streams
    // (no synchronisation): `clk` on the main stream, `pclk` on the preparation stream.  With phase_timing == 0 the two
    // streams overlap, so Construction / Allocation / PostProcessing (preparation stream) run beside the solver phases
    // and the ten values add up to more than the total; phase_timing == 1 runs everything on one stream.
    PhaseClock clk{h, !in_capture};
    PhaseClock pclk{h, !in_capture};
    PhaseClock total{h, !in_capture};
    clk.only_sor = pclk.only_sor = (!h->phase_events && P.phase_timing == 0) || P.phase_timing == 2;
    PhaseClock uclk{h, !in_capture};  // solver launches of the upper strips (smooth_flow_strips): events on their streams
    uclk.only_sor = true;
    h->strip_events_used = 0;
    h->sor_launches = 0;
    h->sor_log.clear();
    h->sor_upper_sec = 0.0;
    clk.stamps = true;  // the main stream's phase boundaries are in-kernel stamps, not events (flow_internal.h)
    h->stamps_used = 0;
    h->next_stamp = nullptr;
    h->stamp_stream = h->stream;  // only launches on the main stream take phase stamps (take_stamp)

    total.phase(PAPOF_T_TOTAL);
    for (int i = 0; i < levels; i++) {  // the two pyramid slots: always the first allocations, at fixed offsets
        double* slot[2];
        slot[0] = A.f64((size_t)L[i].w * L[i].h * C);
        slot[1] = A.f64((size_t)L[i].w * L[i].h * C);
        L[i].p1 = slot[slot1];
        L[i].p2 = slot[slot1 ^ 1];
    }
    double* tmp_a = A.f64(np0 * C);
    double* tmp_b = A.f64(np0 * C);
    const auto abandon_capture = [&]() {  // an error return must not leave a stream capture open on this handle
        if (!in_capture) return;
        hipGraph_t graph = nullptr;
        hipStreamEndCapture(h->stream, &graph);
        if (graph) hipGraphDestroy(graph);
        h->use_graph = false;
    };
    if (A.overflow) {
        abandon_capture();
        return PAPOF_ENOMEM;
    }
    const auto keep = [&](int slot) {
        h->seq.valid = true;
        h->seq.h = H;
        h->seq.w = W;
        h->seq.c = C;
        h->seq.levels = levels;
        h->seq.ratio = ratio;
        h->seq.slot = slot;
        h->seq.arena_base = A.base;
    };
    if (op == kSeqPrime) {
        PAPOF_TRY(load_frame(h, fa, L[0].p1, H, W, C));
        PAPOF_TRY(build_pyramid(h, L, plan, C, false, tmp_a, tmp_b));
        PAPOF_HIP(hipStreamSynchronize(h->stream));
        keep(slot1);
        if (timing) std::memset(timing, 0, tm_bytes);
        return PAPOF_OK;
    }

    // Everything that does not depend on the flow -- the pyramids, and per level the features of both frames
    // (src/OpticalFlow.cpp:797-798) and the smoothed features of frame 1 (getDxs, :84-90) -- is PREPARED on a second
    // stream, coarsest level first, while the main stream already solves the coarse levels: their SOR solves are
    // dependency-latency bound and leave most of the chip idle.  One event per level orders the two streams.  With all
    // ten reference timers requested the call runs on one stream so that the phases do not overlap.
    std::vector<double*> F1(levels), F2(levels), S1(levels);
    for (int k = 0; k < levels; k++) {
        const size_t n = (size_t)L[k].w * L[k].h * fc;
        F1[k] = A.f64(n);
        F2[k] = A.f64(n);
        S1[k] = A.f64(n);
    }
    double* prep_tmp = A.f64(np0 * fc);
    double* gx = A.f64(np0 * C);  // central differences of frame 2 for the final bicubic warp (src/Image.h:2590-2594)
    double* gy = A.f64(np0 * C);
    double* gxy = A.f64(np0 * C);
    double* warp = A.f64(np0 * fc);
    // the flow lives in two PAIRS of planes (the update of an outer iteration writes the other pair); within a pair v follows
    // u at the pitch of the level in work, so that the up-sampling of (u, v) is one launch and their first clear one fill
    // [a handle that cuts levels into strips (PAPOF_STRIPS, an A/B switch) keeps the full-resolution pitch on every level: its
    // strips read the previous level's pair while they write this level's, row range by row range]
    const bool paired = h->strips <= 1;
    double* u = A.f64(2 * np0);
    double* v = u ? u + np0 : nullptr;
    double* u2 = A.f64(2 * np0);
    double* v2 = u2 ? u2 + np0 : nullptr;
    SolveBuffers B;
    {
        const int rc_alloc = alloc_solve_buffers(A, H, W, fc, P.sor_mode, n_sor_max, B);
        if (rc_alloc != PAPOF_OK || A.overflow) {
            abandon_capture();
            return rc_alloc != PAPOF_OK ? rc_alloc : PAPOF_ENOMEM;
        }
    }

    {
        const int rc_br = alloc_branch_buffers(h, A, H, W, fc, P.interpolation, P.noise_model, B);
        if (rc_br != PAPOF_OK) {
            abandon_capture();
            return rc_br;
        }
    }

    // Exact-order path: the progress counters of EVERY solve of the call are cleared at once on the preparation stream (one
    // fill instead of one per solve in front of each solver launch on the main stream).  [Also tried: one set of coefficient
    // planes per level, zeroed ahead on the preparation stream instead of sor_reset_planes() on the main stream at the start
    // of each level -- 0.2 ms SLOWER per 1080p pair (same-box A/B 11.83 vs 11.62 ms, all of it in the solver kernels): the
    // fills in front of a level leave the planes in the Infinity Cache, exactly as the (du, dv) clear in front of a solve.]
    struct LevelCounters {
        size_t prog_off = 0, prog_per = 0;  // offset (unsigneds) of the level's first solve, stride per solve
    };
    std::vector<LevelCounters> LP(levels);
    size_t prog_total = 0;
    const bool exact = P.sor_mode == PAPOF_SOR_EXACT;
    if (exact) {
        for (int k = 0; k < levels; k++) {
            LP[k].prog_per = sor_counters_words(L[k].h, L[k].w, P.n_sor + k * P.n_sor_per_level);
            LP[k].prog_off = prog_total;
            prog_total += LP[k].prog_per * (size_t)((P.n_outer + k * P.n_outer_per_level) * P.n_inner);
        }
        if (!in_capture) {
            const int rc_c = sor_counters_ensure(h, prog_total);
            if (rc_c != PAPOF_OK) return rc_c;
        } else if (prog_total + 64 > h->sync_cap) {  // the eager call before the capture sized them
            abandon_capture();
            return PAPOF_EDEVICE;
        }
    }

    const bool overlap = h->overlap_prep && P.phase_timing != 1 && h->prep_stream != nullptr;
    hipStream_t const main_stream = h->stream, prep = overlap ? h->prep_stream : h->stream;
    while (h->sync_events.size() < (size_t)levels + 2) {
        hipEvent_t e;
        PAPOF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->sync_events.push_back(e);
    }
    struct StreamSwap {  // the launch wrappers enqueue on h->stream
        papof_handle* h;
        hipStream_t saved;
        StreamSwap(papof_handle* hh, hipStream_t s) : h(hh), saved(hh->stream) { h->stream = s; }
        ~StreamSwap() { h->stream = saved; }
    };
    const auto prepare = [&]() -> int {
        StreamSwap on_prep(h, prep);
        pclk.phase(PAPOF_T_ALLOCATION);  // the reference's buffers are zero-filled when they are allocated (src/Image.h:518-532)
        if (exact && !sor_counters_clear(h, 0, prog_total)) return PAPOF_EDEVICE;
        pclk.phase(PAPOF_T_CONSTRUCTION);  // src/OpticalFlow.cpp:757-758 (and the wrapper's copies, Coarse2FineFlowWrapper.cpp:23-28)
        if (op != kSeqNext) PAPOF_TRY(load_frame(h, fa, L[0].p1, H, W, C));
        PAPOF_TRY(load_frame(h, fb, L[0].p2, H, W, C));
        // pyramid levels coarsest first when every level is derived from level 0 (<= 5 levels at ratio 0.75,
        // src/GaussianPyramid.cpp:95-100); deeper pyramids chain through finer levels (:101-106): build those in order
        bool from_level0 = true;
        for (int i = 1; i < levels; i++) from_level0 = from_level0 && plan[i].src_level == 0;
        const Taps g5 = smooth5_taps();
        const auto build_level = [&](int i) -> int {
            if (i == 0) return PAPOF_OK;
            const PyrPlan& q = plan[i];
            for (int second = (op == kSeqNext ? 1 : 0); second < 2; second++) {
                const double* src = second ? L[q.src_level].p2 : L[q.src_level].p1;
                double* dst = second ? L[i].p2 : L[i].p1;
                PAPOF_TRY(smooth_and_resize(h, src, dst, tmp_a, tmp_b, q, C, L[i].h, L[i].w));
            }
            return PAPOF_OK;
        };
        if (!from_level0)
            for (int i = 1; i < levels; i++) PAPOF_TRY(build_level(i));
        for (int k = levels - 1; k >= 0; k--) {
            if (from_level0) {
                pclk.phase(PAPOF_T_CONSTRUCTION);
                PAPOF_TRY(build_level(k));
            }
            pclk.phase(PAPOF_T_ALLOCATION);  // im2feature is inside the reference's Allocation timer (:797-798)
            PAPOF_TRY(im2feature(h, L[k].p1, F1[k], L[k].h, L[k].w, C, lg ? lg->nz(k) : nullptr));
            PAPOF_TRY(im2feature(h, L[k].p2, F2[k], L[k].h, L[k].w, C, lg ? lg->nz(k) : nullptr));
            pclk.phase(PAPOF_T_PHASE1_GENERATE);  // smoothing of frame 1: first half of getDxs (:84-90)
            PAPOF_TRY(filter_hv(h, F1[k], S1[k], prep_tmp, L[k].h, L[k].w, fc, g5, g5));
            pclk.phase(-1);
            if (overlap) PAPOF_HIP(hipEventRecord(h->sync_events[k], prep));
        }
        pclk.phase(PAPOF_T_POSTPROCESSING);  // derivative planes of the final bicubic warp (src/Image.h:2590-2594)
        PAPOF_TRY(central3_planes(h, L[0].p2, gx, gy, gxy, H, W, C));
        pclk.phase(-1);
        if (overlap) PAPOF_HIP(hipEventRecord(h->sync_events[levels], prep));
        return PAPOF_OK;
    };
    // Host buffers (flow_host): the uploads are queued from HERE, on the copy stream, in an order that lets frame 1's whole
    // share of the preparation -- planarise, its pyramid, its features and their smoothing on every level -- run while frame
    // 2 is still crossing PCIe (0.9 ms per 1080p float64 frame); only frame 2's share (coarsest level first, one event per
    // level) is left when the upload ends, and less streaming work runs beside the coarse levels' solves.  Same kernels, same
    // inputs: same bits.  [The device-resident call keeps the level-interleaved order above: there the main stream starts
    // the coarsest level as early as possible.]
    papof_handle::HostIO& io = h->hostio;
    const bool hostio = io.active && overlap && !in_capture && h->copy_stream && h->copy_events.size() >= 8;
    const auto prepare_hostio = [&]() -> int {
        StreamSwap on_prep(h, prep);
        hipStream_t const cs = h->copy_stream;
        pclk.phase(PAPOF_T_ALLOCATION);
        if (exact && !sor_counters_clear(h, 0, prog_total)) return PAPOF_EDEVICE;
        bool from_level0 = true;
        for (int i = 1; i < levels; i++) from_level0 = from_level0 && plan[i].src_level == 0;
        const Taps g5 = smooth5_taps();
        const auto build_level = [&](int i, int second) -> int {
            if (i == 0) return PAPOF_OK;
            const PyrPlan& q = plan[i];
            const double* src = second ? L[q.src_level].p2 : L[q.src_level].p1;
            double* dst = second ? L[i].p2 : L[i].p1;
            return smooth_and_resize(h, src, dst, tmp_a, tmp_b, q, C, L[i].h, L[i].w);
        };
        // ---- frame 1: upload (this call may hold the host until the bytes are staged), then its whole share
        if (io.im1) {
            PAPOF_HIP(hipMemcpyAsync(const_cast<void*>(fa.d), io.im1, io.nb_in, hipMemcpyHostToDevice, cs));
            PAPOF_HIP(hipEventRecord(h->copy_events[0], cs));
            PAPOF_HIP(hipStreamWaitEvent(prep, h->copy_events[0], 0));
        }
        pclk.phase(PAPOF_T_CONSTRUCTION);
        if (op != kSeqNext) {
            PAPOF_TRY(load_frame(h, fa, L[0].p1, H, W, C));
            for (int i = 1; i < levels; i++) PAPOF_TRY(build_level(i, 0));
        }
        for (int k = levels - 1; k >= 0; k--) {
            pclk.phase(PAPOF_T_ALLOCATION);
            PAPOF_TRY(im2feature(h, L[k].p1, F1[k], L[k].h, L[k].w, C, lg ? lg->nz(k) : nullptr));
            pclk.phase(PAPOF_T_PHASE1_GENERATE);
            PAPOF_TRY(filter_hv(h, F1[k], S1[k], prep_tmp, L[k].h, L[k].w, fc, g5, g5));
        }
        pclk.phase(-1);
        // ---- frame 2: upload beside the kernels just queued, then its share, coarsest level first
        if (io.im2) {
            PAPOF_HIP(hipMemcpyAsync(const_cast<void*>(fb.d), io.im2, io.nb_in, hipMemcpyHostToDevice, cs));
            PAPOF_HIP(hipEventRecord(h->copy_events[1], cs));
            PAPOF_HIP(hipStreamWaitEvent(prep, h->copy_events[1], 0));
        }
        pclk.phase(PAPOF_T_CONSTRUCTION);
        PAPOF_TRY(load_frame(h, fb, L[0].p2, H, W, C));
        if (!from_level0)
            for (int i = 1; i < levels; i++) PAPOF_TRY(build_level(i, 1));
        for (int k = levels - 1; k >= 0; k--) {
            if (from_level0) {
                pclk.phase(PAPOF_T_CONSTRUCTION);
                PAPOF_TRY(build_level(k, 1));
            }
            pclk.phase(PAPOF_T_ALLOCATION);
            PAPOF_TRY(im2feature(h, L[k].p2, F2[k], L[k].h, L[k].w, C, lg ? lg->nz(k) : nullptr));
            pclk.phase(-1);
            PAPOF_HIP(hipEventRecord(h->sync_events[k], prep));
        }
        pclk.phase(PAPOF_T_POSTPROCESSING);
        PAPOF_TRY(central3_planes(h, L[0].p2, gx, gy, gxy, H, W, C));
        pclk.phase(-1);
        PAPOF_HIP(hipEventRecord(h->sync_events[levels], prep));
        return PAPOF_OK;
    };
    if (io.active && !hostio) {  // the caller left the uploads to this call, which cannot overlap them: plain copies, in order
        if (io.im1) PAPOF_HIP(hipMemcpyAsync(const_cast<void*>(fa.d), io.im1, io.nb_in, hipMemcpyHostToDevice, main_stream));
        if (io.im2) PAPOF_HIP(hipMemcpyAsync(const_cast<void*>(fb.d), io.im2, io.nb_in, hipMemcpyHostToDevice, main_stream));
    }
    if (overlap) {  // the preparation starts after whatever the caller queued on the main stream (the frame uploads)
        PAPOF_HIP(hipEventRecord(h->sync_events[levels + 1], main_stream));
        PAPOF_HIP(hipStreamWaitEvent(prep, h->sync_events[levels + 1], 0));
    }
    {
        const int rc = hostio ? prepare_hostio() : prepare();
        if (rc != PAPOF_OK) {
            if (in_capture) {
                hipGraph_t graph = nullptr;
                hipStreamEndCapture(main_stream, &graph);
                if (graph) hipGraphDestroy(graph);
                h->use_graph = false;
            } else if (overlap) {
                hipStreamSynchronize(prep);
            }
            return rc;
        }
    }

    int pw = 0, ph = 0;
    int rc_main = PAPOF_OK;
    const auto solve_levels = [&]() -> int {
        if (lg && lg->guard())  // LapPara starts every call at 0.02 (src/OpticalFlow.cpp:773-775)
            PAPOF_HIP(hipMemcpyAsync(lg->lap, h->lap_init_dev, 8 * sizeof(double), hipMemcpyDeviceToDevice, main_stream));
        for (int k = levels - 1; k >= 0; k--) {
            const int lw = L[k].w, lh = L[k].h;
            const size_t np = (size_t)lw * lh;
            clk.phase(-1);  // waiting for the preparation stream is nobody's phase
            if (overlap) PAPOF_HIP(hipStreamWaitEvent(main_stream, h->sync_events[k], 0));
            clk.phase(PAPOF_T_ALLOCATION);  // flow up-sampling and first warp of the level (:801-814)
            const double *f1 = F1[k], *f2 = F2[k];
            const int n_sor_k = P.n_sor + k * P.n_sor_per_level, n_outer_k = P.n_outer + k * P.n_outer_per_level;
            unsigned* prog_k = nullptr;
            if (exact) prog_k = h->sync_words + 32 + LP[k].prog_off;  // this level's counters, cleared by the preparation stream
            const bool tiny_k = exact && (size_t)lw * lh <= kTinyMaxCells && sor_tiny_fits(h, lh, lw, n_sor_k);  // k_sor_tiny level
            if (!tiny_k) PAPOF_TRY(sor_bind(h, B.sp, lh, lw, n_sor_k));
            // the level's result goes straight to the caller's buffers on the finest level
            double* const out_u = k == 0 ? d_vx : nullptr;
            double* const out_v = k == 0 ? d_vy : nullptr;
            // strips (smooth_flow_strips): big levels of the exact-order path, default branches, overlapping streams allowed
            StripSchedule sch;
            // OFF by default (PAPOF_STRIPS=2..4 enables): measured slower on every level it applies to
            const int want = h->strips > 1 ? h->strips : 1;
            const bool strips = overlap && (int)h->strip_streams.size() >= 3 && P.sor_mode == PAPOF_SOR_EXACT &&
                                P.n_inner == 1 && !B.bgx && !B.gm && want >= 2 && !tiny_k &&
                                plan_strips(h, B.sp, lh, n_sor_k, n_outer_k, want, sch);
            LevelInit li{k == levels - 1, nullptr, nullptr, ph, pw, 0.0, 0.0, 1 / ratio};
            const bool fold_warp = !strips && !B.bgx && !B.gm;  // the warped frame 2 lives only inside the smoothing kernel
            if (k == levels - 1) {  // src/OpticalFlow.cpp:801-806
                if (paired) {
                    v = u + np;
                    v2 = u2 + np;
                    PAPOF_HIP(hipMemsetAsync(u, 0, 2 * np * sizeof(double), h->stream));
                } else {
                    PAPOF_HIP(hipMemsetAsync(u, 0, np * sizeof(double), h->stream));
                    PAPOF_HIP(hipMemsetAsync(v, 0, np * sizeof(double), h->stream));
                }
                if (!fold_warp)
                    PAPOF_HIP(hipMemcpyAsync(warp, f2, np * fc * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
                if (B.bgx) PAPOF_TRY(bicubic_planes(h, f2, lh, lw, fc, B));
            } else {  // :809-816
                const double xr = (double)lw / pw, yr = (double)lh / ph, inv = 1 / ratio;
                if (paired) v2 = u2 + np;  // the pair that receives this level's flow, at this level's pitch
                if (strips) {  // up-sampling and warp happen per strip
                    li.pu = u;
                    li.pv = v;
                    li.xr = xr;
                    li.yr = yr;
                } else if (paired) {
                    PAPOF_TRY(resize(h, u, u2, ph, pw, 2, lh, lw, xr, yr, true, inv));  // u and v: two planes of one launch
                } else {
                    PAPOF_TRY(resize(h, u, u2, ph, pw, 1, lh, lw, xr, yr, true, inv));
                    PAPOF_TRY(resize(h, v, v2, ph, pw, 1, lh, lw, xr, yr, true, inv));
                }
                std::swap(u, u2);
                std::swap(v, v2);
                if (paired) v2 = u2 + np;  // the free pair, at this level's pitch too
                if (strips || fold_warp) {
                } else if (!B.bgx) {
                    PAPOF_TRY(warp_bilinear(h, f1, f2, u, v, warp, lh, lw, fc));
                } else {  // interpolation == Bicubic (:816): warpImageBicubicRef, no threshold here
                    PAPOF_TRY(bicubic_planes(h, f2, lh, lw, fc, B));
                    PAPOF_TRY(bicubic_warp(h, f1, f2, B.bgx, B.bgy, B.bgxy, u, v, warp, lh, lw, fc, nullptr, true, false));
                }
            }
            if (!tiny_k) PAPOF_TRY(sor_reset_planes(h, B.sp));
            if (strips)
                PAPOF_TRY(smooth_flow_strips(h, sch, li, f1, f2, warp, u, v, u2, v2, lh, lw, fc, P.alpha, n_sor_k, P.omega, B,
                                             clk, uclk, S1[k], prog_k, LP[k].prog_per, out_u, out_v));
            else
                PAPOF_TRY(smooth_flow(h, f1, f2, warp, u, v, u2, v2, lh, lw, fc, P.alpha, n_outer_k, P.n_inner, n_sor_k,
                                      P.omega, P.sor_mode, B, clk, S1[k], false, prog_k, exact ? LP[k].prog_per : 0, out_u,
                                      out_v, fold_warp, lg, k));
            pw = lw;
            ph = lh;
        }
        clk.phase(-1);
        if (overlap) PAPOF_HIP(hipStreamWaitEvent(main_stream, h->sync_events[levels], 0));
        clk.phase(PAPOF_T_POSTPROCESSING);  // src/OpticalFlow.cpp:841-842
        if (hostio && io.early_out && u == d_vx && v == d_vy) {
            // page-locked result arrays: (vx, vy) are final -- their copies run beside the bicubic warp -- and warpI2 follows
            // its kernel in row chunks (the kernel takes a Rect; rows of the interleaved image are contiguous)
            hipStream_t const cs = h->copy_stream;
            PAPOF_HIP(hipEventRecord(h->copy_events[2], main_stream));
            PAPOF_HIP(hipStreamWaitEvent(cs, h->copy_events[2], 0));
            PAPOF_HIP(hipMemcpyAsync(io.vx, d_vx, io.nb_flow, hipMemcpyDeviceToHost, cs));
            PAPOF_HIP(hipMemcpyAsync(io.vy, d_vy, io.nb_flow, hipMemcpyDeviceToHost, cs));
            constexpr int kChunks = 4;
            for (int q = 0; q < kChunks; q++) {
                const Rect rc{0, (int)((long long)H * q / kChunks), W, (int)((long long)H * (q + 1) / kChunks)};
                if (rc.empty()) continue;
                PAPOF_TRY(bicubic_warp(h, L[0].p1, L[0].p2, gx, gy, gxy, u, v, d_warp, H, W, C, &rc));
                PAPOF_HIP(hipEventRecord(h->copy_events[3 + q], main_stream));
                PAPOF_HIP(hipStreamWaitEvent(cs, h->copy_events[3 + q], 0));
                const size_t off = (size_t)rc.y0 * W * C, cnt = (size_t)rc.h() * W * C;
                PAPOF_HIP(hipMemcpyAsync(io.warp + off, d_warp + off, cnt * sizeof(double), hipMemcpyDeviceToHost, cs));
            }
            io.out_issued = true;
            return PAPOF_OK;
        }
        PAPOF_TRY(bicubic_warp(h, L[0].p1, L[0].p2, gx, gy, gxy, u, v, d_warp, H, W, C));
        if (u != d_vx) PAPOF_HIP(hipMemcpyAsync(d_vx, u, np0 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        if (v != d_vy) PAPOF_HIP(hipMemcpyAsync(d_vy, v, np0 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        return PAPOF_OK;
    };
    rc_main = solve_levels();
    if (in_capture) {  // close the capture whatever happened; a failed capture falls back to eager calls for good
        hipGraph_t graph = nullptr;
        hipError_t ee = hipStreamEndCapture(main_stream, &graph);
        if (rc_main == PAPOF_OK && ee == hipSuccess && graph &&
            hipGraphInstantiate(&ge->exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            hipGraphDestroy(graph);
            return launch_graph();
        }
        if (graph) hipGraphDestroy(graph);
        ge->exec = nullptr;
        h->use_graph = false;
        g_last_error = "hipGraph capture failed; graph mode switched off for this handle";
        return rc_main != PAPOF_OK ? rc_main : PAPOF_EDEVICE;
    }
    if (rc_main != PAPOF_OK) {  // never leave work of this call running on any stream
        for (hipStream_t ss : h->strip_streams) hipStreamSynchronize(ss);
        hipStreamSynchronize(main_stream);
        if (overlap) hipStreamSynchronize(prep);
        return rc_main;
    }
    clk.phase(-1);
    PAPOF_TRY(stamp_only(h));  // the closing stamp of the last phase
    total.phase(-1);
    h->stamps_fetched = 0;
    {   // behind the total's end event, not part of any timer: the stamps and, in front of them in the same block, the guard's
        // flags of this call (the used witness slots end where the non-zero flags begin, those where the stamps begin)
        const bool flags = lg && lg->on && lg->collect;
        const size_t first = flags ? lap_wit_word(std::min(lg->slot, kLapMaxSlots) - 1) : kLapFlagWords;
        const size_t bytes = (kLapFlagWords - first) * sizeof(unsigned) + (size_t)h->stamps_used * sizeof(unsigned long long);
        if (bytes > 0)
            PAPOF_HIP(hipMemcpyAsync(h->lap_flags_host + first, h->lap_flags_dev + first, bytes, hipMemcpyDeviceToHost, h->stream));
        h->stamps_fetched = h->stamps_used;
    }
    const double t_enqueued = wall();
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    h->host_enqueue_sec = t_enqueued - t_entry;
    h->host_wait_sec = wall() - t_enqueued;
    h->next_stamp = nullptr;
    if (P.sor_mode == PAPOF_SOR_EXACT) PAPOF_TRY(sor_check(h));
    if (clk.err != PAPOF_OK || pclk.err != PAPOF_OK || total.err != PAPOF_OK) return PAPOF_EDEVICE;
    clk.collect(tm);
    pclk.collect(tm);
    total.collect(tm);
    if (clk.sor_span_sec.size() == h->sor_log.size())  // one span per solve (not when a level ran as strips)
        for (size_t i = 0; i < h->sor_log.size(); i++) h->sor_log[i].sec = clk.sor_span_sec[i];
    if (uclk.err != PAPOF_OK) return PAPOF_EDEVICE;
    {   // Phase5_SOR = the solver kernels' own time, all strips (on the main chain's time line only the lowest strip's)
        double tu[PAPOF_N_TIMERS + 1];
        std::memset(tu, 0, sizeof tu);
        uclk.collect(tu);
        h->sor_upper_sec = tu[PAPOF_T_PHASE5_SOR];
        tm[PAPOF_T_PHASE5_SOR] += tu[PAPOF_T_PHASE5_SOR];
    }
    {   // the fused assembly kernel was recorded under Phase4: psi's share of it is Phase3_PsiData.  kPsiShare = the
        // kernel's arithmetic that belongs to :377-406 (per channel: t*t, + eps, sqrt, 2*, 1/) over all of it, counted in
        // the kernel's ISA (fp64 sqrt and division are ~25 instructions each); a fixed apportioning, not a measurement.
        constexpr double kPsiShare = 0.3;
        const double fused = tm[PAPOF_T_PHASE4_LINEARSYSTEM];
        tm[PAPOF_T_PHASE3_PSIDATA] += kPsiShare * fused;
        tm[PAPOF_T_PHASE4_LINEARSYSTEM] = fused - kPsiShare * fused;
    }
    {   // phi (Phase2_Derivatives, :295-331) has no kernel of its own on the default path any more: the warp-and-smooth kernel of
        // every outer iteration (recorded under Phase1) writes it by the way.  kPhiShare = phi's part of that kernel -- its
        // loads, the square root and the division in one block of five -- as a fixed apportioning of Phase1, not a
        // measurement; a Phase2 measured by compute_phi() (inner iterations, the non-default branches) adds to it.
        constexpr double kPhiShare = 0.04;
        const double fused = tm[PAPOF_T_PHASE1_GENERATE];
        tm[PAPOF_T_PHASE2_DERIVATIVES] += kPhiShare * fused;
        tm[PAPOF_T_PHASE1_GENERATE] = fused - kPhiShare * fused;
    }
    {   // k_flow_system did the work of four timers in one launch (the levels it applies to): apportioned by the times of the
        // kernels it replaces at level 0 -- warp / smoothing / derivatives 122 us of which phi 4 %, psi + system 109 us at 30 : 70
        const double f = tm[kTimerFused];
        tm[PAPOF_T_PHASE1_GENERATE] += 0.51 * f;
        tm[PAPOF_T_PHASE2_DERIVATIVES] += 0.02 * f;
        tm[PAPOF_T_PHASE3_PSIDATA] += 0.14 * f;
        tm[PAPOF_T_PHASE4_LINEARSYSTEM] += f - 0.51 * f - 0.02 * f - 0.14 * f;
    }
    if (timing) std::memcpy(timing, tm, tm_bytes);
    if (op == kSeqNext) keep(slot1 ^ 1);  // the frame just solved against becomes frame 1 of the next push
    return PAPOF_OK;
}

// The whole call on device-resident buffers, with the Laplacian-noise guard (LapGuard): the optimistic pass, and the exact
// pass behind it when a consulted estimate is left without a proof.
int flow_device(papof_handle* h, const FrameIn& fa, const FrameIn& fb, SeqOp op, int H, int W, int C, int levels,
                const papof_params& P, double* d_vx, double* d_vy, double* d_warp, double* timing) {
    LapGuard lg;
    const int fc = feature_channels(C);
    // [strips (PAPOF_STRIPS, off by default, an A/B switch) update the flow per strip and carry no witnesses]
    lg.on = h->lap_guard && op != kSeqPrime && P.noise_model == PAPOF_NOISE_LAPLACIAN && fc <= 8 && levels >= 1 &&
            h->lap_flags_dev != nullptr && h->strips <= 1;
    if (!lg.on) return flow_pass(h, fa, fb, op, H, W, C, levels, P, d_vx, d_vy, d_warp, timing, nullptr);
    lg.flags = h->lap_flags_dev;
    lg.lap = h->lap_dev;
    lg.scratch = h->lap_scratch_dev;
    // the non-zero flags of the feature channels are written by im2feature's 1- and 3-channel branches only (any other channel
    // count is passed through untouched, src/OpticalFlow.cpp:956-960): without them no channel may count as "all zero"
    lg.nz_known = (C == 1 || C == 3) && (size_t)levels * 8 <= (size_t)kLapNzWords;
    // inside a hipGraph nobody could act on the flags, and the bicubic branch has no sampled warp: always the exact pass
    const bool always_exact = h->use_graph || P.interpolation == PAPOF_INTERP_BICUBIC;
    lg.collect = !always_exact;
    lg.exact = always_exact || h->lap_exact;
    const papof_handle::Seq seq0 = h->seq;
    double first_pass_total = 0.0;
    for (int pass = 0; pass < 2; pass++) {
        // a flag is set when it holds the number of the pass that wrote it: no clearing, nothing stale can be taken for a proof
        if (++h->lap_epoch >= kLapNone) {  // (2^31 passes later: start over on cleared flags)
            PAPOF_HIP(hipMemsetAsync(h->lap_flags_dev, 0, kLapFlagWords * sizeof(unsigned), h->stream));
            h->lap_epoch = 1u;
        }
        lg.epoch = h->lap_epoch;
        lg.slot = 0;
        lg.slot_level.clear();
        lg.slot_channels.clear();
        PAPOF_TRY(flow_pass(h, fa, fb, op, H, W, C, levels, P, d_vx, d_vy, d_warp, timing, &lg));
        if (pass == 1 && timing) timing[PAPOF_T_TOTAL] += first_pass_total;  // the call cost both passes (phases: the exact pass's)
        if (!lg.collect) return PAPOF_OK;
        const int unknown = lg.unknown(h->lap_flags_host);
        if (lg.exact) {  // the exact pass is right whatever the flags say; they decide how the NEXT call starts
            if (pass == 0) h->lap_exact_calls++;
            h->lap_exact = unknown > 0;
            return PAPOF_OK;
        }
        if (unknown == 0) return PAPOF_OK;
        // ---- a consulted estimate without a proof: the same call again, in the exact pass.  The frames are on the device
        // already (host uploads of the first pass), the kept pyramid of a sequence is where it was.
        h->lap_reruns++;
        lg.exact = true;
        if (timing) first_pass_total = timing[PAPOF_T_TOTAL];
        // the first pass may have queued early result copies on the copy stream (HostIO): they must have left the device
        // buffers before the second pass rewrites them
        if (h->copy_stream) PAPOF_HIP(hipStreamSynchronize(h->copy_stream));
        h->seq = seq0;
        h->hostio.im1 = nullptr;
        h->hostio.im2 = nullptr;
    }
    return PAPOF_OK;
}

}  // namespace papof

// =================================================================================================
// C ABI
// =================================================================================================
using namespace papof;

extern "C" {

int papof_version(void) { return PAPOF_VERSION; }

void papof_default_params(papof_params* p) {
    if (!p) return;
    p->alpha = 0.012;
    p->ratio = 0.75;
    p->n_outer = 7;
    p->n_outer_per_level = 1;
    p->n_inner = 1;
    p->n_sor = 30;
    p->n_sor_per_level = 3;
    p->omega = 1.8;
    p->sor_mode = PAPOF_SOR_EXACT;
    p->phase_timing = 0;
    p->interpolation = PAPOF_INTERP_BILINEAR;
    p->noise_model = PAPOF_NOISE_LAPLACIAN;
}

const char* papof_strerror(int code) {
    switch (code) {
        case PAPOF_OK: return "ok";
        case PAPOF_EINVAL: return "invalid argument";
        case PAPOF_ENODEVICE: return "no usable gfx950 device";
        case PAPOF_ENOMEM: return "out of memory";
        case PAPOF_EDEVICE: return "HIP runtime error";
        case PAPOF_ETIMEOUT: return "device-side wait timed out";
    }
    return "unknown error";
}

const char* papof_last_error(void) { return last_error(); }

const char* papof_timing_key(int i) {
    static const char* keys[PAPOF_N_TIMERS] = {"Allocation",         "Construction",        "Phase1_Generate",
                                               "Phase2_Derivatives", "Phase3_PsiData",      "Phase4_LinearSystem",
                                               "Phase5_SOR",         "Phase6_Update",       "PostProcessing",
                                               "Total C++ Execution"};
    return (i >= 0 && i < PAPOF_N_TIMERS) ? keys[i] : "";
}

int papof_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int papof_create(int device, papof_handle** out) {
    if (!out) return PAPOF_EINVAL;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0 || device < 0 || device >= n) {
        set_last_error("hipGetDeviceCount", e, __FILE__, __LINE__);
        return PAPOF_ENODEVICE;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
        set_last_error("hipGetDeviceProperties", e, __FILE__, __LINE__);
        return PAPOF_ENODEVICE;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {  // the code object is gfx950-only by design
        g_last_error = std::string("device is ") + prop.gcnArchName + ", libpapof is built for gfx950 only";
        return PAPOF_ENODEVICE;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        set_last_error("hipSetDevice", e, __FILE__, __LINE__);
        return PAPOF_ENODEVICE;
    }
    papof_handle* h = new papof_handle();
    h->device = device;
    h->cu_count = prop.multiProcessorCount;
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) {
        set_last_error("hipStreamCreate", e, __FILE__, __LINE__);
        delete h;
        return PAPOF_ENODEVICE;
    }
    // The preparation stream may be confined to a share of the CUs (PAPOF_PREP_CUS = number of CUs; the mask's bits are
    // dealt round-robin over the XCDs): its streaming kernels then take longer, still hidden behind the coarse levels'
    // solves, but load the memory system less -- those latency-bound solves run 0.6 ms per 1080p pair slower beside an
    // unconfined preparation stream than alone (same-box A/B).
    int prep_cus = 0;
    if (const char* cs = std::getenv("PAPOF_PREP_CUS")) prep_cus = std::atoi(cs);
    if (prep_cus > 0 && prep_cus < h->cu_count) {
        uint32_t mask[16] = {0};
        for (int i = 0; i < prep_cus && i < 512; i++) mask[i / 32] |= 1u << (i % 32);
        e = hipExtStreamCreateWithCUMask(&h->prep_stream, (uint32_t)((h->cu_count + 31) / 32), mask);
        if (e != hipSuccess) h->prep_stream = nullptr;  // fall back to an ordinary stream
    }
    if (!h->prep_stream && (e = hipStreamCreateWithFlags(&h->prep_stream, hipStreamNonBlocking)) != hipSuccess) {
        set_last_error("hipStreamCreate", e, __FILE__, __LINE__);
        papof_destroy(h);
        return PAPOF_ENODEVICE;
    }
    if (const char* cs = std::getenv("PAPOF_STRIPS")) h->strips = std::max(0, std::atoi(cs));
    // strip streams (smooth_flow_strips; opt-in) are created only where they are used: a handle is two streams otherwise, and
    // the runtime multiplexes streams onto a limited number of hardware queues (eight tile ranks on one device = 16 streams)
    if (h->strips > 1) ensure_strip_streams(h);
    if (const char* cs = std::getenv("PAPOF_GRAPH")) h->use_graph = std::atoi(cs) != 0;
    if (const char* cs = std::getenv("PAPOF_OVERLAP")) h->overlap_prep = std::atoi(cs) != 0;
    if (const char* cs = std::getenv("PAPOF_PHASE_EVENTS")) h->phase_events = std::atoi(cs) != 0;
    if (const char* cs = std::getenv("PAPOF_HOST_COPY")) h->host_copy = std::atoi(cs);
    if (const char* cs = std::getenv("PAPOF_HOST_THREADS")) h->host_threads = std::max(1, std::atoi(cs));
    if (const char* cs = std::getenv("PAPOF_SOR_DEPTH")) h->sor_depth = std::max(4, std::atoi(cs));
    if (const char* cs = std::getenv("PAPOF_SOR_FUSE")) h->sor_fuse = std::max(1, std::atoi(cs));
    if (const char* cs = std::getenv("PAPOF_SOR_GROUP")) h->sor_group = std::max(1, std::atoi(cs));
    if (const char* cs = std::getenv("PAPOF_SOR_XCD")) h->sor_xcd_affine = std::atoi(cs);
    if (const char* cs = std::getenv("PAPOF_RB_DEPTH")) h->rb_depth = std::max(0, std::atoi(cs));
    if (const char* cs = std::getenv("PAPOF_RB_SHAPE")) h->rb_shape = std::max(0, std::atoi(cs));
    if (const char* cs = std::getenv("PAPOF_RB_NAIVE")) h->rb_naive = std::atoi(cs) != 0;
    if (const char* cs = std::getenv("PAPOF_SOR_RESIDENT")) h->sor_resident = std::max(0, std::atoi(cs));
    if (const char* cs = std::getenv("PAPOF_SOR_DEAD")) h->sor_skip_dead = std::atoi(cs) != 0 ? 1 : 0;
    int rc = sor_probe_dpp(h);
    if (rc != PAPOF_OK) {
        papof_destroy(h);
        return rc;
    }
    // flags of the Laplacian-noise guard + slots for the phase stamps: one block of device memory and one pinned host copy,
    // fetched with ONE copy per call (common.h: lap_flags_dev); LapPara, its initial value and the reduction's scratch
    {
        const size_t flag_bytes = kLapFlagWords * sizeof(unsigned), bytes = flag_bytes + 4096 * sizeof(unsigned long long);
        const size_t lap_doubles = 16 + (size_t)lap_scratch_doubles();
        std::vector<double> init(8, 0.02);  // src/OpticalFlow.cpp:773-775
        if (hipMalloc((void**)&h->lap_flags_dev, bytes) == hipSuccess &&
            hipHostMalloc((void**)&h->lap_flags_host, bytes, hipHostMallocDefault) == hipSuccess &&
            hipMalloc((void**)&h->lap_dev, lap_doubles * sizeof(double)) == hipSuccess &&
            hipMemset(h->lap_flags_dev, 0, flag_bytes) == hipSuccess &&
            hipMemcpy(h->lap_dev + 8, init.data(), 8 * sizeof(double), hipMemcpyHostToDevice) == hipSuccess) {
            h->stamps_dev = reinterpret_cast<unsigned long long*>(h->lap_flags_dev + kLapFlagWords);
            h->stamps = reinterpret_cast<unsigned long long*>(h->lap_flags_host + kLapFlagWords);
            h->stamps_cap = 4096;
            h->lap_init_dev = h->lap_dev + 8;
            h->lap_scratch_dev = h->lap_dev + 16;
            std::memset(h->lap_flags_host, 0, bytes);
        } else {  // the guard cannot be left out: its absence would change results on some inputs
            g_last_error = "cannot allocate the flag / stamp block of the handle";
            papof_destroy(h);
            return PAPOF_ENOMEM;
        }
        const char* lgv = std::getenv("PAPOF_LAP_GUARD");
        h->lap_guard = !(lgv && lgv[0] == '0');
    }
    *out = h;
    return PAPOF_OK;
}

void papof_destroy(papof_handle* h) {
    if (!h) return;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    if (h->prep_stream) hipStreamSynchronize(h->prep_stream);
    for (hipStream_t ss : h->strip_streams) {
        hipStreamSynchronize(ss);
        hipStreamDestroy(ss);
    }
    for (hipEvent_t e : h->events) hipEventDestroy(e);
    for (hipEvent_t e : h->sync_events) hipEventDestroy(e);
    for (hipEvent_t e : h->strip_events) hipEventDestroy(e);
    for (papof::GraphEntry& e : h->graphs)
        if (e.exec) hipGraphExecDestroy(e.exec);
    if (h->prep_stream) hipStreamDestroy(h->prep_stream);
    if (h->copy_stream) {
        hipStreamSynchronize(h->copy_stream);
        hipStreamDestroy(h->copy_stream);
    }
    for (hipEvent_t e : h->copy_events) hipEventDestroy(e);
    if (h->arena.base) hipFree(h->arena.base);
    if (h->sync_words) hipFree(h->sync_words);
    if (h->stage_dev) hipFree(h->stage_dev);
    if (h->pin) hipHostFree(h->pin);
    if (h->lap_flags_host) hipHostFree(h->lap_flags_host);  // the stamps live inside these two blocks
    if (h->lap_flags_dev) hipFree(h->lap_flags_dev);
    if (h->lap_dev) hipFree(h->lap_dev);
    delete h->pool;
    hipStreamDestroy(h->stream);
    delete h;
}

void* papof_stream(papof_handle* h) { return h ? (void*)h->stream : nullptr; }

int papof_host_alloc(size_t bytes, void** out) {
    if (!out) return PAPOF_EINVAL;
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_last_error("hipHostMalloc", e, __FILE__, __LINE__);
        *out = nullptr;
        return e == hipErrorNoDevice || e == hipErrorInvalidDevice ? PAPOF_ENODEVICE : PAPOF_ENOMEM;
    }
    return PAPOF_OK;
}

int papof_host_free(void* p) {
    if (!p) return PAPOF_OK;
    PAPOF_HIP(hipHostFree(p));
    return PAPOF_OK;
}

int papof_dev_alloc(papof_handle* h, size_t bytes, void** out) {
    if (!h || !out) return PAPOF_EINVAL;
    PAPOF_HIP(hipSetDevice(h->device));
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) {
        set_last_error("hipMalloc", e, __FILE__, __LINE__);
        return PAPOF_ENOMEM;
    }
    return PAPOF_OK;
}

int papof_dev_free(papof_handle* h, void* p) {
    if (!h) return PAPOF_EINVAL;
    PAPOF_HIP(hipSetDevice(h->device));
    PAPOF_HIP(hipFree(p));
    return PAPOF_OK;
}

int papof_dev_upload(papof_handle* h, void* dst, const void* src, size_t bytes) {
    if (!h || !dst || !src) return PAPOF_EINVAL;
    PAPOF_HIP(hipSetDevice(h->device));
    PAPOF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    return PAPOF_OK;
}

int papof_dev_download(papof_handle* h, void* dst, const void* src, size_t bytes) {
    if (!h || !dst || !src) return PAPOF_EINVAL;
    PAPOF_HIP(hipSetDevice(h->device));
    PAPOF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    return PAPOF_OK;
}

namespace {

int device_call(papof_handle* h, const FrameIn& fa, const FrameIn& fb, SeqOp op, int height, int width, int c,
                int pyramid_levels, const papof_params* params, double* d_vx, double* d_vy, double* d_warpI2,
                double* timing_sec) {
    if (!h || height < 1 || width < 1 || c < 1) return PAPOF_EINVAL;
    if (op != kSeqNext && !fa.d) return PAPOF_EINVAL;
    if (op != kSeqPrime && (!fb.d || !d_vx || !d_vy || !d_warpI2)) return PAPOF_EINVAL;
    papof_params P;
    if (params)
        P = *params;
    else
        papof_default_params(&P);
    PAPOF_HIP(hipSetDevice(h->device));
    return flow_device(h, fa, fb, op, height, width, c, pyramid_levels, P, d_vx, d_vy, d_warpI2, timing_sec);
}

// prime or continue?  (a push whose shape / plan differs from the kept frame starts a new sequence)
SeqOp seq_op_for(papof_handle* h, int H, int W, int C, int levels, const papof_params* params) {
    papof_params P;
    if (params)
        P = *params;
    else
        papof_default_params(&P);
    double ratio = P.ratio;
    if (ratio > 0.98 || ratio < 0.4) ratio = 0.75;
    if (levels < 1) return kSeqPrime;
    const size_t need = arena_bytes_for(H, W, C, levels, P.n_sor + (levels - 1) * P.n_sor_per_level, P.ratio);
    return seq_matches(h, H, W, C, levels, ratio, need) ? kSeqNext : kSeqPrime;
}

}  // namespace

int papof_flow_device(papof_handle* h, const double* d_im1, const double* d_im2, int height, int width, int c,
                      int pyramid_levels, const papof_params* params, double* d_vx, double* d_vy, double* d_warpI2,
                      double timing_sec[PAPOF_N_TIMERS]) {
    return device_call(h, FrameIn{d_im1, false}, FrameIn{d_im2, false}, kPair, height, width, c, pyramid_levels,
                       params, d_vx, d_vy, d_warpI2, timing_sec);
}

int papof_flow_device_u8(papof_handle* h, const unsigned char* d_im1, const unsigned char* d_im2, int height,
                         int width, int c, int pyramid_levels, const papof_params* params, double* d_vx,
                         double* d_vy, double* d_warpI2, double timing_sec[PAPOF_N_TIMERS]) {
    return device_call(h, FrameIn{d_im1, true}, FrameIn{d_im2, true}, kPair, height, width, c, pyramid_levels, params,
                       d_vx, d_vy, d_warpI2, timing_sec);
}

int papof_set_graph_mode(papof_handle* h, int on) {
    if (!h) return PAPOF_EINVAL;
    h->use_graph = on != 0;
    return PAPOF_OK;
}

int papof_set_stream_overlap(papof_handle* h, int on) {
    if (!h) return PAPOF_EINVAL;
    h->overlap_prep = on != 0;
    return PAPOF_OK;
}

int papof_seq_reset(papof_handle* h) {
    if (!h) return PAPOF_EINVAL;
    h->seq.valid = false;
    return PAPOF_OK;
}

int papof_seq_push_device(papof_handle* h, const void* d_frame, int is_u8, int height, int width, int c,
                          int pyramid_levels, const papof_params* params, double* d_vx, double* d_vy,
                          double* d_warpI2, double timing_sec[PAPOF_N_TIMERS], int* have_flow) {
    if (!h || !d_frame || !have_flow) return PAPOF_EINVAL;
    *have_flow = 0;
    const SeqOp op = seq_op_for(h, height, width, c, pyramid_levels, params);
    const FrameIn f{d_frame, is_u8 != 0};
    PAPOF_TRY(device_call(h, f, f, op, height, width, c, pyramid_levels, params, d_vx, d_vy, d_warpI2, timing_sec));
    *have_flow = op == kSeqNext ? 1 : 0;
    return PAPOF_OK;
}

namespace {

// Pageable user memory <-> pinned bounce buffer through the handle's persistent copy pool (one memcpy thread moves
// ~10 GB/s; PCIe Gen5 wants ~50).
void parallel_copy(papof_handle* h, char* dst, const char* src, size_t n) {
    if (h->host_threads <= 1 || n < (size_t(1) << 20)) {
        std::memcpy(dst, src, n);
        return;
    }
    if (!h->pool) h->pool = new CopyPool(h->host_threads - 1);
    h->pool->copy(dst, src, n);
}

int ensure_host_stage(papof_handle* h, size_t dev_bytes, size_t pin_bytes) {
    if (dev_bytes > h->stage_dev_bytes) {
        PAPOF_HIP(hipStreamSynchronize(h->stream));
        if (h->stage_dev) PAPOF_HIP(hipFree(h->stage_dev));
        h->stage_dev = nullptr;
        h->stage_dev_bytes = 0;
        hipError_t e = hipMalloc((void**)&h->stage_dev, dev_bytes);
        if (e != hipSuccess) {
            set_last_error("hipMalloc(stage)", e, __FILE__, __LINE__);
            return PAPOF_ENOMEM;
        }
        h->stage_dev_bytes = dev_bytes;
    }
    if (pin_bytes > h->pin_bytes) {
        PAPOF_HIP(hipStreamSynchronize(h->stream));
        if (h->pin) PAPOF_HIP(hipHostFree(h->pin));
        h->pin = nullptr;
        h->pin_bytes = 0;
        hipError_t e = hipHostMalloc((void**)&h->pin, pin_bytes, hipHostMallocDefault);
        if (e != hipSuccess) {
            set_last_error("hipHostMalloc(pinned staging)", e, __FILE__, __LINE__);
            return PAPOF_ENOMEM;
        }
        h->pin_bytes = pin_bytes;
    }
    return PAPOF_OK;
}

constexpr size_t kChunk = size_t(8) << 20;  // bounce granularity: the DMA of one chunk overlaps the memcpy of the next

// user (pageable) -> pinned -> device, chunk by chunk on the handle's stream
int upload_chunked(papof_handle* h, char* dev, const char* user, char* pin, size_t n) {
    for (size_t off = 0; off < n; off += kChunk) {
        const size_t m = std::min(kChunk, n - off);
        parallel_copy(h, pin + off, user + off, m);
        PAPOF_HIP(hipMemcpyAsync(dev + off, pin + off, m, hipMemcpyHostToDevice, h->stream));
    }
    return PAPOF_OK;
}

// Host buffers in, host buffers out.  `u8`: the frames are uint8 samples (1 byte per sample over PCIe instead of 8).
// kPair uploads both frames, kSeqPrime only im1 (and returns no flow), kSeqNext only im2.
int flow_host(papof_handle* h, const void* im1, const void* im2, bool u8, SeqOp op, int height, int width, int c,
              int pyramid_levels, const papof_params* params, double* vx, double* vy, double* warpI2,
              double* timing_sec) {
    if (!h || height < 1 || width < 1 || c < 1 || pyramid_levels < 1) return PAPOF_EINVAL;
    if (op != kSeqNext && !im1) return PAPOF_EINVAL;
    if (op != kSeqPrime && (!im2 || !vx || !vy || !warpI2)) return PAPOF_EINVAL;
    const double t0 = wall();
    PAPOF_HIP(hipSetDevice(h->device));
    const size_t np = (size_t)height * width, nb_img = np * c * sizeof(double), nb_flow = np * sizeof(double);
    const size_t nb_in = np * c * (u8 ? 1 : sizeof(double));
    // device staging for the interleaved frames and results (separate from the arena, which flow_device resets) and a
    // pinned bounce buffer: inputs [im1 | im2], then reused for the outputs [warpI2 | vx | vy]
    const size_t out_bytes = nb_img + 2 * nb_flow;
    PAPOF_TRY(ensure_host_stage(h, 3 * nb_img + 2 * nb_flow, std::max(2 * nb_img, out_bytes)));
    double* d1 = h->stage_dev;
    double* d2 = d1 + np * c;
    double* dw = d2 + np * c;
    double* dx = dw + np * c;
    double* dy = dx + np;
    double tm[PAPOF_N_TIMERS];
    std::memset(tm, 0, sizeof tm);
    // host_copy = 1: the runtime's own pageable path (measured 55 GB/s both ways on this platform: hipMemcpyAsync from / to
    // pageable memory); 0: our pinned bounce pipeline (pageable -> pinned by a thread pool, DMA per 8-MiB chunk)
    const bool plain = h->host_copy == 1;
    // PAPOF_HOSTIO=0: the copies around the call as in round 2 (A/B); default: the call issues them itself where they overlap
    // device work (common.h: HostIO).  Not in graph mode (a captured call replays fixed pointers) and not for a priming push.
    static const bool hostio_env = !(std::getenv("PAPOF_HOSTIO") && std::atoi(std::getenv("PAPOF_HOSTIO")) == 0);
    bool use_io = plain && hostio_env && !h->use_graph && op != kSeqPrime && h->overlap_prep && h->prep_stream;
    if (use_io && !h->copy_stream) {
        if (hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess) h->copy_stream = nullptr;
        while (h->copy_stream && h->copy_events.size() < 8) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) break;
            h->copy_events.push_back(e);
        }
    }
    use_io = use_io && h->copy_stream && h->copy_events.size() >= 8;
    if (use_io) {
        papof_params Pq;
        if (params)
            Pq = *params;
        else
            papof_default_params(&Pq);
        use_io = Pq.phase_timing != 1;  // one-stream timing mode: nothing overlaps by definition
    }
    if (use_io) {
        h->hostio = papof_handle::HostIO{};
        h->hostio.active = true;
        h->hostio.im1 = op != kSeqNext ? im1 : nullptr;
        h->hostio.im2 = im2;
        h->hostio.nb_in = nb_in;
    } else if (plain) {
        if (op != kSeqNext) PAPOF_HIP(hipMemcpyAsync(d1, im1, nb_in, hipMemcpyHostToDevice, h->stream));
        if (op != kSeqPrime) PAPOF_HIP(hipMemcpyAsync(d2, im2, nb_in, hipMemcpyHostToDevice, h->stream));
    } else {
        if (op != kSeqNext) PAPOF_TRY(upload_chunked(h, (char*)d1, (const char*)im1, h->pin, nb_in));
        if (op != kSeqPrime) PAPOF_TRY(upload_chunked(h, (char*)d2, (const char*)im2, h->pin + nb_img, nb_in));
    }
    // The caller's result arrays are usually fresh allocations (pyflow.pyx: np.zeros per call): their first-touch page
    // faults (~20k pages at 1080p, ~4 ms) are taken by a helper thread WHILE the GPU computes, not while copying back.
    const auto is_pinned = [](const void* p) {  // page-locked (papof_host_alloc / hipHostRegister): resident, DMA-able
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) {
            (void)hipGetLastError();  // plain pageable memory is "invalid value" to the query: not an error of this call
            return false;
        }
        return a.type == hipMemoryTypeHost;
    };
    std::thread prefault;
    const bool out_pinned = op != kSeqPrime && is_pinned(warpI2) && is_pinned(vx) && is_pinned(vy);
    if (use_io) {
        h->hostio.vx = vx;
        h->hostio.vy = vy;
        h->hostio.warp = warpI2;
        h->hostio.nb_flow = nb_flow;
        h->hostio.nb_img = nb_img;
        h->hostio.early_out = out_pinned;
    }
    if (op != kSeqPrime && np * sizeof(double) >= (size_t(1) << 20) && !out_pinned)
        prefault = std::thread([=] {
            const auto touch = [](double* p, size_t bytes) {
                volatile char* q = reinterpret_cast<volatile char*>(p);
                for (size_t o = 0; o < bytes; o += 4096) q[o] = 0;
                if (bytes) q[bytes - 1] = 0;
            };
            touch(warpI2, nb_img);
            touch(vx, nb_flow);
            touch(vy, nb_flow);
        });
    const int rc_dev = device_call(h, FrameIn{d1, u8}, FrameIn{d2, u8}, op, height, width, c, pyramid_levels, params, dx,
                                   dy, dw, tm);
    const bool out_issued = use_io && h->hostio.out_issued;
    h->hostio.active = false;
    if (use_io && h->copy_stream && hipStreamSynchronize(h->copy_stream) != hipSuccess && rc_dev == PAPOF_OK) {
        if (prefault.joinable()) prefault.join();
        return PAPOF_EDEVICE;
    }
    if (prefault.joinable()) prefault.join();
    PAPOF_TRY(rc_dev);
    if (out_issued) {  // the result copies were queued by the call itself and have landed
        if (timing_sec) {
            tm[PAPOF_T_TOTAL] = wall() - t0;
            std::memcpy(timing_sec, tm, sizeof tm);
        }
        return PAPOF_OK;
    }
    if (op == kSeqPrime) {
        if (timing_sec) std::memcpy(timing_sec, tm, sizeof tm);
        return PAPOF_OK;
    }
    if (plain) {
        PAPOF_HIP(hipMemcpyAsync(warpI2, dw, nb_img, hipMemcpyDeviceToHost, h->stream));
        PAPOF_HIP(hipMemcpyAsync(vx, dx, nb_flow, hipMemcpyDeviceToHost, h->stream));
        PAPOF_HIP(hipMemcpyAsync(vy, dy, nb_flow, hipMemcpyDeviceToHost, h->stream));
        PAPOF_HIP(hipStreamSynchronize(h->stream));
    } else
    // device -> pinned in chunks (dw, dx, dy are contiguous), each chunk handed to the user as soon as it has landed
    {
        const size_t n_chunks = (out_bytes + kChunk - 1) / kChunk;
        std::vector<hipEvent_t> done(n_chunks);
        for (size_t i = 0; i < n_chunks; i++) {
            const size_t off = i * kChunk, m = std::min(kChunk, out_bytes - off);
            PAPOF_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
            PAPOF_HIP(hipMemcpyAsync(h->pin + off, (const char*)dw + off, m, hipMemcpyDeviceToHost, h->stream));
            PAPOF_HIP(hipEventRecord(done[i], h->stream));
        }
        auto scatter = [&](size_t off, size_t m) {  // [off, off+m) of [warpI2 | vx | vy] -> the three user buffers
            const size_t bounds[4] = {0, nb_img, nb_img + nb_flow, nb_img + 2 * nb_flow};
            char* dst[3] = {(char*)warpI2, (char*)vx, (char*)vy};
            for (int k = 0; k < 3; k++) {
                const size_t lo = std::max(off, bounds[k]), hi = std::min(off + m, bounds[k + 1]);
                if (lo < hi) parallel_copy(h, dst[k] + (lo - bounds[k]), h->pin + lo, hi - lo);
            }
        };
        int rc = PAPOF_OK;
        for (size_t i = 0; i < n_chunks; i++) {
            const size_t off = i * kChunk, m = std::min(kChunk, out_bytes - off);
            if (rc == PAPOF_OK && hipEventSynchronize(done[i]) != hipSuccess) rc = PAPOF_EDEVICE;
            if (rc == PAPOF_OK) scatter(off, m);
            hipEventDestroy(done[i]);
        }
        PAPOF_TRY(rc);
    }
    if (timing_sec) {
        tm[PAPOF_T_TOTAL] = wall() - t0;  // the caller-visible total includes both PCIe transfers
        std::memcpy(timing_sec, tm, sizeof tm);
    }
    return PAPOF_OK;
}

}  // namespace

int papof_flow(papof_handle* h, const double* im1, const double* im2, int height, int width, int c,
               int pyramid_levels, const papof_params* params, double* vx, double* vy, double* warpI2,
               double timing_sec[PAPOF_N_TIMERS]) {
    return flow_host(h, im1, im2, false, kPair, height, width, c, pyramid_levels, params, vx, vy, warpI2, timing_sec);
}

int papof_flow_u8(papof_handle* h, const unsigned char* im1, const unsigned char* im2, int height, int width, int c,
                  int pyramid_levels, const papof_params* params, double* vx, double* vy, double* warpI2,
                  double timing_sec[PAPOF_N_TIMERS]) {
    return flow_host(h, im1, im2, true, kPair, height, width, c, pyramid_levels, params, vx, vy, warpI2, timing_sec);
}

namespace {
int seq_push_host(papof_handle* h, const void* frame, bool u8, int height, int width, int c, int pyramid_levels,
                  const papof_params* params, double* vx, double* vy, double* warpI2, double* timing_sec,
                  int* have_flow) {
    if (!h || !frame || !have_flow) return PAPOF_EINVAL;
    *have_flow = 0;
    const SeqOp op = seq_op_for(h, height, width, c, pyramid_levels, params);
    PAPOF_TRY(flow_host(h, frame, frame, u8, op, height, width, c, pyramid_levels, params, vx, vy, warpI2, timing_sec));
    *have_flow = op == kSeqNext ? 1 : 0;
    return PAPOF_OK;
}
}  // namespace

int papof_seq_push(papof_handle* h, const double* frame, int height, int width, int c, int pyramid_levels,
                   const papof_params* params, double* vx, double* vy, double* warpI2,
                   double timing_sec[PAPOF_N_TIMERS], int* have_flow) {
    return seq_push_host(h, frame, false, height, width, c, pyramid_levels, params, vx, vy, warpI2, timing_sec,
                         have_flow);
}

int papof_seq_push_u8(papof_handle* h, const unsigned char* frame, int height, int width, int c, int pyramid_levels,
                      const papof_params* params, double* vx, double* vy, double* warpI2,
                      double timing_sec[PAPOF_N_TIMERS], int* have_flow) {
    return seq_push_host(h, frame, true, height, width, c, pyramid_levels, params, vx, vy, warpI2, timing_sec,
                         have_flow);
}

static std::mutex g_default_mu;
static papof_handle* g_default = nullptr;

int papof_coarse2fine_flow(const double* im1, const double* im2, int h, int w, int c, int pyramid_levels,
                           const papof_params* params, double* vx, double* vy, double* warpI2,
                           double timing_sec[PAPOF_N_TIMERS]) {
    std::lock_guard<std::mutex> lock(g_default_mu);
    if (!g_default) {
        int dev = 0;
        if (const char* s = std::getenv("PAPOF_DEVICE")) dev = std::atoi(s);
        PAPOF_TRY(papof_create(dev, &g_default));
    }
    return papof_flow(g_default, im1, im2, h, w, c, pyramid_levels, params, vx, vy, warpI2, timing_sec);
}

// -------------------------------------------------------------------------------------------------
// stage entry points (host buffers, reference HWC layout)
// -------------------------------------------------------------------------------------------------
namespace {

struct Scope {  // arena scope + device selection for one stage call
    papof_handle* h;
    int rc;
    Scope(papof_handle* hh, size_t bytes) : h(hh), rc(PAPOF_OK) {
        if (!h) {
            rc = PAPOF_EINVAL;
            return;
        }
        if (hipSetDevice(h->device) != hipSuccess) {
            rc = PAPOF_EDEVICE;
            return;
        }
        h->seq.valid = false;  // stage calls reuse the arena from offset 0
        rc = ensure_arena(h, bytes + (1 << 20));
        h->arena.off = 0;
        h->arena.overflow = false;
    }
    // upload an HWC host image as planar device planes
    double* up_planar(const double* host, int H, int W, int C) {
        const size_t n = (size_t)H * W * C;
        double* raw = h->arena.f64(n);
        double* pl = h->arena.f64(n);
        if (!raw || !pl) {
            rc = PAPOF_ENOMEM;
            return nullptr;
        }
        if (hipMemcpyAsync(raw, host, n * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess) {
            rc = PAPOF_EDEVICE;
            return nullptr;
        }
        if (C == 1) return raw;
        int r = hwc_to_planar(h, raw, pl, H, W, C);
        if (r != PAPOF_OK) rc = r;
        return pl;
    }
    double* dev(size_t n) {
        double* p = h->arena.f64(n);
        if (!p) rc = PAPOF_ENOMEM;
        return p;
    }
    // download planar device planes into an HWC host image
    int down_planar(const double* planar, double* host, int H, int W, int C) {
        const size_t n = (size_t)H * W * C;
        const double* src = planar;
        if (C != 1) {
            double* raw = h->arena.f64(n);
            if (!raw) return PAPOF_ENOMEM;
            PAPOF_TRY(planar_to_hwc(h, planar, raw, H, W, C));
            src = raw;
        }
        PAPOF_HIP(hipMemcpyAsync(host, src, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        PAPOF_HIP(hipStreamSynchronize(h->stream));
        return PAPOF_OK;
    }
};

size_t img_bytes(int H, int W, int C, int copies) { return (size_t)H * W * C * sizeof(double) * copies; }

}  // namespace

int papof_stage_pyramid(papof_handle* h, const double* im, int height, int width, int c, double ratio, int levels,
                        int* dims, double* data, long* n_elems) {
    if (!h || !dims || height < 1 || width < 1 || c < 1 || levels < 1) return PAPOF_EINVAL;
    std::vector<Level> L;
    std::vector<PyrPlan> plan;
    PAPOF_TRY(pyramid_plan(height, width, ratio, levels, L, plan));
    long total = 0;
    for (int i = 0; i < levels; i++) {
        dims[2 * i] = L[i].w;
        dims[2 * i + 1] = L[i].h;
        total += (long)L[i].w * L[i].h * c;
    }
    if (n_elems) *n_elems = total;
    if (!data) return PAPOF_OK;
    if (!im) return PAPOF_EINVAL;
    Scope S(h, img_bytes(height, width, c, 6) + (size_t)total * sizeof(double) * 2);
    PAPOF_TRY(S.rc);
    L[0].p1 = S.up_planar(im, height, width, c);
    for (int i = 1; i < levels; i++) L[i].p1 = S.dev((size_t)L[i].w * L[i].h * c);
    double* ta = S.dev((size_t)height * width * c);
    double* tb = S.dev((size_t)height * width * c);
    PAPOF_TRY(S.rc);
    PAPOF_TRY(build_pyramid(h, L, plan, c, false, ta, tb));
    long off = 0;
    for (int i = 0; i < levels; i++) {
        PAPOF_TRY(S.down_planar(L[i].p1, data + off, L[i].h, L[i].w, c));
        off += (long)L[i].w * L[i].h * c;
    }
    return PAPOF_OK;
}

int papof_stage_gaussian(papof_handle* h, const double* im, int height, int width, int c, double sigma, int fsize,
                         double* out) {
    if (!h || !im || !out || height < 1 || width < 1 || c < 1 || fsize < 0 || fsize > kMaxFsize)
        return PAPOF_EINVAL;
    Scope S(h, img_bytes(height, width, c, 6));
    PAPOF_TRY(S.rc);
    double* src = S.up_planar(im, height, width, c);
    double* ta = S.dev((size_t)height * width * c);
    double* tb = S.dev((size_t)height * width * c);
    PAPOF_TRY(S.rc);
    const Taps g = gaussian_taps(sigma, fsize);
    PAPOF_TRY(filter_h(h, src, ta, height, width, c, g));
    PAPOF_TRY(filter_v(h, ta, tb, height, width, c, g));
    return S.down_planar(tb, out, height, width, c);
}

int papof_stage_resize_ratio(papof_handle* h, const double* im, int height, int width, int c, double ratio,
                             double* out) {
    if (!h || !im || !out || height < 1 || width < 1 || c < 1 || !(ratio > 0)) return PAPOF_EINVAL;
    const int dw = (int)((double)width * ratio), dh = (int)((double)height * ratio);
    if (dw < 1 || dh < 1) return PAPOF_EINVAL;
    Scope S(h, img_bytes(height, width, c, 3) + img_bytes(dh, dw, c, 3));
    PAPOF_TRY(S.rc);
    double* src = S.up_planar(im, height, width, c);
    double* dst = S.dev((size_t)dw * dh * c);
    PAPOF_TRY(S.rc);
    PAPOF_TRY(resize(h, src, dst, height, width, c, dh, dw, ratio, ratio, false, 0.0));
    return S.down_planar(dst, out, dh, dw, c);
}

int papof_stage_resize_wh(papof_handle* h, const double* im, int height, int width, int c, int dst_w, int dst_h,
                          double* out) {
    if (!h || !im || !out || height < 1 || width < 1 || c < 1 || dst_w < 1 || dst_h < 1) return PAPOF_EINVAL;
    Scope S(h, img_bytes(height, width, c, 3) + img_bytes(dst_h, dst_w, c, 3));
    PAPOF_TRY(S.rc);
    double* src = S.up_planar(im, height, width, c);
    double* dst = S.dev((size_t)dst_w * dst_h * c);
    PAPOF_TRY(S.rc);
    PAPOF_TRY(resize(h, src, dst, height, width, c, dst_h, dst_w, (double)dst_w / width, (double)dst_h / height,
                     false, 0.0));
    return S.down_planar(dst, out, dst_h, dst_w, c);
}

int papof_stage_im2feature(papof_handle* h, const double* im, int height, int width, int c, double* out, int* fc) {
    if (!h || height < 1 || width < 1 || c < 1) return PAPOF_EINVAL;
    const int f = feature_channels(c);
    if (fc) *fc = f;
    if (!out) return PAPOF_OK;
    if (!im) return PAPOF_EINVAL;
    Scope S(h, img_bytes(height, width, c, 3) + img_bytes(height, width, f, 3));
    PAPOF_TRY(S.rc);
    double* src = S.up_planar(im, height, width, c);
    double* dst = S.dev((size_t)height * width * f);
    PAPOF_TRY(S.rc);
    PAPOF_TRY(im2feature(h, src, dst, height, width, c));
    return S.down_planar(dst, out, height, width, f);
}

int papof_stage_warpFL(papof_handle* h, const double* im1, const double* im2, const double* vx, const double* vy,
                       int height, int width, int c, double* out) {
    if (!h || !im1 || !im2 || !vx || !vy || !out || height < 1 || width < 1 || c < 1) return PAPOF_EINVAL;
    Scope S(h, img_bytes(height, width, c, 8) + img_bytes(height, width, 1, 4));
    PAPOF_TRY(S.rc);
    double* a = S.up_planar(im1, height, width, c);
    double* b = S.up_planar(im2, height, width, c);
    double* fx = S.up_planar(vx, height, width, 1);
    double* fy = S.up_planar(vy, height, width, 1);
    double* dst = S.dev((size_t)height * width * c);
    PAPOF_TRY(S.rc);
    PAPOF_TRY(warp_bilinear(h, a, b, fx, fy, dst, height, width, c));
    return S.down_planar(dst, out, height, width, c);
}

int papof_stage_getDxs(papof_handle* h, const double* im1, const double* im2, int height, int width, int c,
                       double* imdx, double* imdy, double* imdt) {
    if (!h || !im1 || !im2 || !imdx || !imdy || !imdt || height < 1 || width < 1 || c < 1) return PAPOF_EINVAL;
    Scope S(h, img_bytes(height, width, c, 14));
    PAPOF_TRY(S.rc);
    const size_t n = (size_t)height * width * c;
    double* a = S.up_planar(im1, height, width, c);
    double* b = S.up_planar(im2, height, width, c);
    double *tmp = S.dev(n), *im1s = S.dev(n), *blend = S.dev(n), *dt = S.dev(n), *gx = S.dev(n), *gy = S.dev(n);
    PAPOF_TRY(S.rc);
    const Taps g = smooth5_taps(), d = deriv5_taps();
    PAPOF_TRY(filter_h(h, a, tmp, height, width, c, g));
    PAPOF_TRY(filter_v(h, tmp, im1s, height, width, c, g));
    PAPOF_TRY(filter_h(h, b, tmp, height, width, c, g));
    PAPOF_TRY(smooth_v_blend(h, tmp, im1s, blend, dt, height, width, c));
    PAPOF_TRY(filter_h(h, blend, gx, height, width, c, d));
    PAPOF_TRY(filter_v(h, blend, gy, height, width, c, d));
    PAPOF_TRY(S.down_planar(gx, imdx, height, width, c));
    PAPOF_TRY(S.down_planar(gy, imdy, height, width, c));
    return S.down_planar(dt, imdt, height, width, c);
}

int papof_stage_linear_system(papof_handle* h, const double* im1, const double* warp, const double* u,
                              const double* v, int height, int width, int c, double alpha, double* phi,
                              double* imdxy, double* imdx2, double* imdy2, double* rhs1, double* rhs2) {
    if (!h || !im1 || !warp || !u || !v || !phi || !imdxy || !imdx2 || !imdy2 || !rhs1 || !rhs2 || height < 1 ||
        width < 1 || c < 1)
        return PAPOF_EINVAL;
    Scope S(h, img_bytes(height, width, c, 10) + img_bytes(height, width, 1, 20));
    PAPOF_TRY(S.rc);
    const size_t np = (size_t)height * width, n = np * c;
    double* a = S.up_planar(im1, height, width, c);
    double* b = S.up_planar(warp, height, width, c);
    double* du = S.up_planar(u, height, width, 1);
    double* dv = S.up_planar(v, height, width, 1);
    double *tmp = S.dev(n), *im1s = S.dev(n), *blend = S.dev(n), *dt = S.dev(n), *dphi = S.dev(np);
    SorPlanes sp{};
    sp.skew = false;
    sp.phi = S.dev(np);
    sp.xy = S.dev(np);
    sp.a1 = S.dev(np);
    sp.a2 = S.dev(np);
    sp.b1 = S.dev(np);
    sp.b2 = S.dev(np);
    double *x2 = S.dev(np), *y2 = S.dev(np);
    PAPOF_TRY(S.rc);
    const Taps g = smooth5_taps();
    PAPOF_TRY(filter_h(h, a, tmp, height, width, c, g));
    PAPOF_TRY(filter_v(h, tmp, im1s, height, width, c, g));
    PAPOF_TRY(filter_h(h, b, tmp, height, width, c, g));
    PAPOF_TRY(smooth_v_blend(h, tmp, im1s, blend, dt, height, width, c));
    PAPOF_TRY(compute_phi(h, du, dv, nullptr, dphi, height, width));
    PAPOF_TRY(assemble_system(h, blend, dt, dphi, du, dv, height, width, c, alpha, 1.8, sp, x2, y2, nullptr));
    PAPOF_TRY(S.down_planar(dphi, phi, height, width, 1));
    PAPOF_TRY(S.down_planar(sp.xy, imdxy, height, width, 1));
    PAPOF_TRY(S.down_planar(x2, imdx2, height, width, 1));
    PAPOF_TRY(S.down_planar(y2, imdy2, height, width, 1));
    PAPOF_TRY(S.down_planar(sp.b1, rhs1, height, width, 1));
    return S.down_planar(sp.b2, rhs2, height, width, 1);
}

int papof_stage_laplacian(papof_handle* h, const double* in, const double* weight, int height, int width,
                          double* out) {
    if (!h || !in || !weight || !out || height < 1 || width < 1) return PAPOF_EINVAL;
    Scope S(h, img_bytes(height, width, 1, 6));
    PAPOF_TRY(S.rc);
    double* a = S.up_planar(in, height, width, 1);
    double* w = S.up_planar(weight, height, width, 1);
    double* o = S.dev((size_t)height * width);
    PAPOF_TRY(S.rc);
    PAPOF_TRY(laplacian(h, a, w, o, height, width));
    return S.down_planar(o, out, height, width, 1);
}

namespace {
int alloc_sor_planes(Scope& S, int H, int W, int mode, int n_sor, SorPlanes& sp) {
    if (mode == PAPOF_SOR_EXACT && (size_t)H * W <= kTinyMaxCells && sor_tiny_fits(S.h, H, W, n_sor)) {  // k_sor_tiny
        const int rc_t = sor_alloc_tiny_planes(S.h->arena, (size_t)H * W, sp);
        if (rc_t != PAPOF_OK) S.rc = rc_t;
        return S.rc;
    }
    int rc = sor_alloc_planes(S.h->arena, H, W, mode, n_sor, sp);
    if (rc == PAPOF_OK) rc = sor_bind(S.h, sp, H, W, n_sor);
    if (rc != PAPOF_OK) S.rc = rc;
    return S.rc;
}
}  // namespace

int papof_stage_sor(papof_handle* h, const double* phi, const double* imdxy, const double* imdx2,
                    const double* imdy2, const double* rhs1, const double* rhs2, int height, int width,
                    double alpha, double omega, int n_sor, int sor_mode, double* du, double* dv) {
    if (!h || !phi || !imdxy || !imdx2 || !imdy2 || !rhs1 || !rhs2 || !du || !dv || height < 1 || width < 1 ||
        n_sor < 1 || sor_mode < PAPOF_SOR_EXACT || sor_mode > PAPOF_SOR_JACOBI)
        return PAPOF_EINVAL;
    const size_t np = (size_t)height * width;
    Scope S(h, img_bytes(height, width, 1, 16) + sor_scratch_bytes(height, width, n_sor) +
                   12 * (size_t)height * width * sizeof(double));
    PAPOF_TRY(S.rc);
    double* p = S.up_planar(phi, height, width, 1);
    double* xy = S.up_planar(imdxy, height, width, 1);
    double* x2 = S.up_planar(imdx2, height, width, 1);
    double* y2 = S.up_planar(imdy2, height, width, 1);
    double* r1 = S.up_planar(rhs1, height, width, 1);
    double* r2 = S.up_planar(rhs2, height, width, 1);
    double *ou = S.dev(np), *ov = S.dev(np);
    SorPlanes sp{};
    PAPOF_TRY(alloc_sor_planes(S, height, width, sor_mode, n_sor, sp));
    PAPOF_TRY(sor_reset_planes(h, sp));
    PAPOF_TRY(sor_prep(h, p, xy, x2, y2, r1, r2, height, width, alpha, omega, sp));
    PAPOF_TRY(sor_solve(h, sp, height, width, alpha, omega, n_sor, sor_mode));
    PAPOF_TRY(sor_unpack(h, sp, ou, ov, height, width));
    PAPOF_TRY(S.down_planar(ou, du, height, width, 1));
    PAPOF_TRY(S.down_planar(ov, dv, height, width, 1));
    if (sor_mode == PAPOF_SOR_EXACT) PAPOF_TRY(sor_check(h));
    return PAPOF_OK;
}

int papof_stage_smoothflow(papof_handle* h, const double* im1, const double* im2, double* warp, double* u,
                           double* v, int height, int width, int c, double alpha, int n_outer, int n_inner,
                           int n_sor, double omega, int sor_mode) {
    return papof_stage_smoothflow_ex(h, im1, im2, warp, u, v, height, width, c, alpha, n_outer, n_inner, n_sor, omega,
                                     sor_mode, PAPOF_INTERP_BILINEAR, PAPOF_NOISE_LAPLACIAN, nullptr);
}

int papof_stage_smoothflow_ex(papof_handle* h, const double* im1, const double* im2, double* warp, double* u,
                              double* v, int height, int width, int c, double alpha, int n_outer, int n_inner,
                              int n_sor, double omega, int sor_mode, int interpolation, int noise_model, double* gm) {
    if (!h || !im1 || !im2 || !warp || !u || !v || height < 1 || width < 1 || c < 1 || n_outer < 1 || n_sor < 1 ||
        sor_mode < PAPOF_SOR_EXACT || sor_mode > PAPOF_SOR_JACOBI)
        return PAPOF_EINVAL;
    if (n_inner < 1) return PAPOF_EINVAL;
    if (interpolation < 0 || interpolation > 1 || noise_model < 0 || noise_model > 1) return PAPOF_EINVAL;
    Scope S(h, img_bytes(height, width, c, 16) + img_bytes(height, width, 1, 8) +
                   sor_scratch_bytes(height, width, n_sor) +
                   12 * (size_t)height * width * sizeof(double) + (1 << 20));
    PAPOF_TRY(S.rc);
    double* f1 = S.up_planar(im1, height, width, c);
    double* f2 = S.up_planar(im2, height, width, c);
    double* w = S.up_planar(warp, height, width, c);
    double* du = S.up_planar(u, height, width, 1);
    double* dv = S.up_planar(v, height, width, 1);
    double* du_alt = S.dev((size_t)height * width);
    double* dv_alt = S.dev((size_t)height * width);
    PAPOF_TRY(S.rc);
    SolveBuffers B;
    PAPOF_TRY(alloc_solve_buffers(h->arena, height, width, c, sor_mode, n_sor, B));
    PAPOF_TRY(alloc_branch_buffers(h, h->arena, height, width, c, interpolation, noise_model, B));
    if (B.gm && gm) PAPOF_HIP(hipMemcpyAsync(B.gm, gm, sizeof(double) * 5 * c, hipMemcpyHostToDevice, h->stream));
    if (B.bgx) PAPOF_TRY(bicubic_planes(h, f2, height, width, c, B));
    PAPOF_TRY(sor_bind(h, B.sp, height, width, n_sor));
    PAPOF_TRY(sor_reset_planes(h, B.sp));
    PhaseClock clk{h, false};
    LapGuard lg;  // a stage call always takes the exact pass of the Laplacian-noise guard, LapPara starting at 0.02 (:773-775)
    lg.on = h->lap_guard && noise_model == PAPOF_NOISE_LAPLACIAN && c <= 8 && h->lap_dev != nullptr;
    lg.exact = true;
    lg.lap = h->lap_dev;
    lg.scratch = h->lap_scratch_dev;
    if (lg.on) PAPOF_HIP(hipMemcpyAsync(lg.lap, h->lap_init_dev, 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    PAPOF_TRY(smooth_flow(h, f1, f2, w, du, dv, du_alt, dv_alt, height, width, c, alpha, n_outer, n_inner, n_sor, omega,
                          sor_mode, B, clk, nullptr, true, nullptr, 0, nullptr, nullptr, false, &lg, 0));
    PAPOF_TRY(S.down_planar(w, warp, height, width, c));
    PAPOF_TRY(S.down_planar(du, u, height, width, 1));
    PAPOF_TRY(S.down_planar(dv, v, height, width, 1));
    if (B.gm && gm) {
        PAPOF_HIP(hipMemcpyAsync(gm, B.gm, sizeof(double) * 5 * c, hipMemcpyDeviceToHost, h->stream));
        PAPOF_HIP(hipStreamSynchronize(h->stream));
    }
    if (sor_mode == PAPOF_SOR_EXACT) PAPOF_TRY(sor_check(h));
    return PAPOF_OK;
}

int papof_stage_est_gaussian_mixture(papof_handle* h, const double* im1, const double* im2, int height, int width,
                                     int c, double* gm) {
    if (!h || !im1 || !im2 || !gm || height < 1 || width < 1 || c < 1 || c > 8) return PAPOF_EINVAL;
    Scope S(h, img_bytes(height, width, c, 6) + (1 << 20));
    PAPOF_TRY(S.rc);
    double* a = S.up_planar(im1, height, width, c);
    double* b = S.up_planar(im2, height, width, c);
    double* g = S.dev(40);
    double* scratch = S.dev((size_t)gm_scratch_doubles());
    PAPOF_TRY(S.rc);
    PAPOF_HIP(hipMemcpyAsync(g, gm, sizeof(double) * 5 * c, hipMemcpyHostToDevice, h->stream));
    PAPOF_TRY(est_gaussian_mixture(h, a, b, height, width, c, g, scratch));
    PAPOF_HIP(hipMemcpyAsync(gm, g, sizeof(double) * 5 * c, hipMemcpyDeviceToHost, h->stream));
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    return PAPOF_OK;
}

int papof_pyramid_levels_for_min_width(int width, double ratio, int min_width, int* levels) {
    if (width < 1 || min_width < 1 || !levels) return PAPOF_EINVAL;
    if (ratio > 0.98 || ratio < 0.4) ratio = 0.75;  // src/GaussianPyramid.cpp:50-51
    *levels = (int)(std::log((double)min_width / width) / std::log(ratio));  // :53
    return PAPOF_OK;
}

int papof_stage_bicubic_warp(papof_handle* h, const double* im1, const double* im2, const double* vx,
                             const double* vy, int height, int width, int c, double* out) {
    return papof_stage_bicubic_warp_ex(h, im1, im2, vx, vy, height, width, c, 1, out);
}

int papof_stage_bicubic_warp_ex(papof_handle* h, const double* im1, const double* im2, const double* vx,
                                const double* vy, int height, int width, int c, int clamp, double* out) {
    if (!h || !im1 || !im2 || !vx || !vy || !out || height < 1 || width < 1 || c < 1) return PAPOF_EINVAL;
    Scope S(h, img_bytes(height, width, c, 10) + img_bytes(height, width, 1, 4));
    PAPOF_TRY(S.rc);
    const size_t n = (size_t)height * width * c;
    double* a = S.up_planar(im1, height, width, c);
    double* b = S.up_planar(im2, height, width, c);
    double* fx = S.up_planar(vx, height, width, 1);
    double* fy = S.up_planar(vy, height, width, 1);
    double *gx = S.dev(n), *gy = S.dev(n), *gxy = S.dev(n), *o = S.dev(n);
    PAPOF_TRY(S.rc);
    PAPOF_TRY(central3_planes(h, b, gx, gy, gxy, height, width, c));
    PAPOF_TRY(bicubic_warp(h, a, b, gx, gy, gxy, fx, fy, o, height, width, c, nullptr, false, clamp != 0));
    PAPOF_HIP(hipMemcpyAsync(out, o, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    return PAPOF_OK;
}

int papof_flow_quantize16(papof_handle* h, const double* vx, const double* vy, int height, int width,
                          unsigned short* out) {
    if (!h || !vx || !vy || !out || height < 1 || width < 1) return PAPOF_EINVAL;
    const size_t n = (size_t)height * width;
    Scope S(h, n * (2 * sizeof(double) + 2 * sizeof(unsigned short)) + 4096);
    PAPOF_TRY(S.rc);
    double* a = S.up_planar(vx, height, width, 1);
    double* b = S.up_planar(vy, height, width, 1);
    unsigned short* q = reinterpret_cast<unsigned short*>(S.dev((n * 2 * sizeof(unsigned short) + 7) / 8));
    PAPOF_TRY(S.rc);
    PAPOF_TRY(flow_quantize16(h, a, b, q, n));
    PAPOF_HIP(hipMemcpyAsync(out, q, n * 2 * sizeof(unsigned short), hipMemcpyDeviceToHost, h->stream));
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    return PAPOF_OK;
}

int papof_flow_dequantize16(papof_handle* h, const unsigned short* q, int height, int width, double* vx,
                            double* vy) {
    if (!h || !q || !vx || !vy || height < 1 || width < 1) return PAPOF_EINVAL;
    const size_t n = (size_t)height * width;
    Scope S(h, n * (2 * sizeof(double) + 2 * sizeof(unsigned short)) + 4096);
    PAPOF_TRY(S.rc);
    unsigned short* dq = reinterpret_cast<unsigned short*>(S.dev((n * 2 * sizeof(unsigned short) + 7) / 8));
    double* a = S.dev(n);
    double* b = S.dev(n);
    PAPOF_TRY(S.rc);
    PAPOF_HIP(hipMemcpyAsync(dq, q, n * 2 * sizeof(unsigned short), hipMemcpyHostToDevice, h->stream));
    PAPOF_TRY(flow_dequantize16(h, dq, a, b, n));
    PAPOF_HIP(hipMemcpyAsync(vx, a, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    PAPOF_HIP(hipMemcpyAsync(vy, b, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    return PAPOF_OK;
}

int papof_flow_to_bgr(papof_handle* h, const double* vx, const double* vy, int height, int width,
                      unsigned char* bgr) {
    if (!h || !vx || !vy || !bgr || height < 1 || width < 1) return PAPOF_EINVAL;
    const size_t n = (size_t)height * width;
    Scope S(h, n * (2 * sizeof(double) + 3) + 16384);
    PAPOF_TRY(S.rc);
    double* a = S.up_planar(vx, height, width, 1);
    double* b = S.up_planar(vy, height, width, 1);
    double* partial = S.dev(512);
    unsigned char* out = reinterpret_cast<unsigned char*>(S.dev((n * 3 + 7) / 8));
    PAPOF_TRY(S.rc);
    PAPOF_TRY(flow_to_bgr(h, a, b, n, partial, out));
    PAPOF_HIP(hipMemcpyAsync(bgr, out, n * 3, hipMemcpyDeviceToHost, h->stream));
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    return PAPOF_OK;
}

int papof_sor_plan(papof_handle* h, int height, int width, int n_sor, int sor_mode, int* launches, int* depth) {
    return sor_plan(h, height, width, n_sor, sor_mode, launches, depth);
}

// SOR micro-benchmark on synthetic planes resident in HBM: phi~U(0.5,50), imdx2/imdy2~U(0,.05),
// imdxy~U(-.02,.02), rhs~U(-.01,.01); alpha .012, omega 1.8.
int papof_last_sor_stats(papof_handle* h, int* launches, double* strip_streams_sec) {
    if (!h) return PAPOF_EINVAL;
    if (launches) *launches = h->sor_launches;
    if (strip_streams_sec) *strip_streams_sec = h->sor_upper_sec;
    return PAPOF_OK;
}

int papof_lap_guard_stats(papof_handle* h, int out[4]) {
    if (!h || !out) return PAPOF_EINVAL;
    out[0] = h->lap_reruns;
    out[1] = h->lap_exact_calls;
    out[2] = h->lap_exact ? 1 : 0;
    out[3] = h->lap_guard ? 1 : 0;
    return PAPOF_OK;
}

int papof_last_host_times(papof_handle* h, double out[3]) {
    if (!h || !out) return PAPOF_EINVAL;
    out[0] = h->host_enqueue_sec;
    out[1] = h->host_wait_sec;
    out[2] = 0.0;
    return PAPOF_OK;
}

int papof_last_sor_solves(papof_handle* h, int cap, int* n, int* info, double* sec) {
    if (!h || !n) return PAPOF_EINVAL;
    *n = (int)h->sor_log.size();
    for (int i = 0; i < *n && i < cap; i++) {
        const papof_handle::SorSolveLog& e = h->sor_log[i];
        if (info) {
            int* o = info + (size_t)i * 6;
            o[0] = e.H;
            o[1] = e.W;
            o[2] = e.n_sor;
            o[3] = e.kind;
            o[4] = e.depth;
            o[5] = e.launches;
        }
        if (sec) sec[i] = e.sec;
    }
    return PAPOF_OK;
}

int papof_strip_plan(papof_handle* h, int height, int width, int n_sor, int n_outer, int want_strips, int* strips,
                     int* out, int cap, int* band_rows, int* koff, int* bands) {
    if (!h || !strips || height < 1 || width < 1 || n_sor < 1 || n_outer < 1) return PAPOF_EINVAL;
    SorPlanes sp{};
    sp.skew = true;
    sp.cap_cells = sp.cap_cells_d = ~size_t(0);
    PAPOF_TRY(sor_bind(h, sp, height, width, n_sor));
    if (band_rows) *band_rows = sp.sd.band_rows;
    if (koff) *koff = sp.sd.koff;
    if (bands) *bands = sp.sd.nb;
    StripSchedule q;
    const int want = want_strips > 0 ? want_strips : (h->strips > 1 ? h->strips : 1);
    if (!plan_strips(h, sp, height, n_sor, n_outer, want, q)) {
        *strips = 1;
        return PAPOF_OK;
    }
    *strips = q.S;
    if (out) {
        const int need = (n_outer + 1) * (q.S + 1) * 5;
        if (cap < need) return PAPOF_EINVAL;
        for (int n = 0; n <= n_outer; n++)
            for (int s = 0; s <= q.S; s++) {
                int* o = out + (size_t)q.idx(n, s) * 5;
                o[0] = q.beta[q.idx(n, s)];
                o[1] = q.rU[q.idx(n, s)];
                o[2] = q.rP[q.idx(n, s)];
                o[3] = q.rS[q.idx(n, s)];
                o[4] = q.rA[q.idx(n, s)];
            }
    }
    return PAPOF_OK;
}

// Test aid (papof.h): one exact-order solve on synthetic planes, whole and as two strips of bands on two streams
// (sor_solve_bands), `reps` times; *mismatches = 16-byte cells of the (du, dv) planes, both parities, that differ.
int papof_test_sor_strips(papof_handle* h, int height, int width, int n_sor, int split_band, int reps, int delay_us,
                          long long* mismatches, int* bands) {
    if (!h || !mismatches) return PAPOF_EINVAL;
    ensure_strip_streams(h);
    if (h->strip_streams.empty()) return PAPOF_EINVAL;
    const size_t np = (size_t)height * width;
    Scope S(h, img_bytes(height, width, 1, 16) + sor_scratch_bytes(height, width, n_sor) +
                   12 * (size_t)height * width * sizeof(double));
    PAPOF_TRY(S.rc);
    std::vector<double> host(np * 6);
    std::mt19937_64 rng(2);
    auto fill = [&](size_t k, double lo, double hi) {
        std::uniform_real_distribution<double> d(lo, hi);
        for (size_t i = 0; i < np; i++) host[k * np + i] = d(rng);
    };
    fill(0, 0.5, 50.0);
    fill(1, -0.02, 0.02);
    fill(2, 0.0, 0.05);
    fill(3, 0.0, 0.05);
    fill(4, -0.01, 0.01);
    fill(5, -0.01, 0.01);
    double* planes[6];
    for (int k = 0; k < 6; k++) planes[k] = S.up_planar(host.data() + k * np, height, width, 1);
    SorPlanes sp{};
    PAPOF_TRY(alloc_sor_planes(S, height, width, PAPOF_SOR_EXACT, n_sor, sp));
    PAPOF_TRY(sor_reset_planes(h, sp));
    PAPOF_TRY(sor_prep(h, planes[0], planes[1], planes[2], planes[3], planes[4], planes[5], height, width, 0.012, 1.8, sp));
    if (bands) *bands = sp.sd.nb;
    const size_t cells = sp.sd.nd;
    std::vector<double> want(cells * 2), got(cells * 2);
    PAPOF_TRY(sor_solve(h, sp, height, width, 0.012, 1.8, n_sor, PAPOF_SOR_EXACT));
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    PAPOF_TRY(sor_check(h));
    PAPOF_HIP(hipMemcpy(want.data(), sp.du, cells * 16, hipMemcpyDeviceToHost));
    *mismatches = 0;
    if (!sor_strips_supported(h, sp, n_sor) || split_band < 1 || split_band >= sp.sd.nb) return PAPOF_EINVAL;
    hipStream_t const main_stream = h->stream, top = h->strip_streams[0];
    hipEvent_t ev;
    PAPOF_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int r = 0; r < reps; r++) {
        PAPOF_TRY(sor_strips_begin(h, sp, n_sor, 1));
        PAPOF_HIP(hipEventRecord(ev, main_stream));
        PAPOF_HIP(hipStreamWaitEvent(top, ev, 0));
        h->stream = top;
        int rc = sor_solve_bands(h, sp, height, width, 0.012, 1.8, n_sor, h->sync_words + 32, 0, split_band);
        h->stream = main_stream;
        PAPOF_TRY(rc);
        if (delay_us > 0) {  // the lower strip starts later, as behind its own assembly kernels
            hipStreamSynchronize(main_stream);
            const double t0 = wall();
            while ((wall() - t0) * 1e6 < delay_us) {}
        }
        PAPOF_TRY(sor_solve_bands(h, sp, height, width, 0.012, 1.8, n_sor, h->sync_words + 32, split_band, sp.sd.nb));
        PAPOF_HIP(hipStreamSynchronize(top));
        PAPOF_HIP(hipStreamSynchronize(main_stream));
        PAPOF_TRY(sor_check(h));
        PAPOF_HIP(hipMemcpy(got.data(), sp.du, cells * 16, hipMemcpyDeviceToHost));
        long long mm = 0;
        std::vector<long long> by_band(sp.sd.nb, 0), by_quad(16, 0);
        long long by_par[2] = {0, 0};
        int pmin = 1 << 30, pmax = -1;
        const size_t per_par = (size_t)sp.sd.npos_d * sp.sd.nb * kLanes;
        for (size_t i = 0; i < cells; i++)
            if (got[2 * i] != want[2 * i] || got[2 * i + 1] != want[2 * i + 1]) {
                ++mm;
                const size_t par = i / per_par, rest = i % per_par;
                const int pos = (int)(rest / ((size_t)sp.sd.nb * kLanes)), b = (int)((rest / kLanes) % sp.sd.nb), c = (int)(rest % kLanes);
                by_par[par]++;
                by_band[b]++;
                by_quad[c / 4]++;
                pmin = std::min(pmin, pos);
                pmax = std::max(pmax, pos);
                if (mm <= 6 && std::getenv("PAPOF_SOR_DBG")) std::fprintf(stderr, "  [split dbg] rep %d mismatch parity %zu pos %d band %d cell %d: got %.6g want %.6g\n", r, par, pos, b, c, got[2 * i], want[2 * i]);
            }
        if (mm && std::getenv("PAPOF_SOR_DBG")) {
            std::fprintf(stderr, "  [split dbg] rep %d: %lld cells, parity0 %lld parity1 %lld, pos %d..%d, by band:", r, mm, by_par[0], by_par[1], pmin, pmax);
            for (int b = 0; b < sp.sd.nb; b++) std::fprintf(stderr, " %lld", by_band[b]);
            std::fprintf(stderr, " | by lane quad:");
            for (int k = 0; k < 16; k++) std::fprintf(stderr, " %lld", by_quad[k]);
            std::fprintf(stderr, "\n");
        }
        *mismatches += mm;
    }
    hipEventDestroy(ev);
    return PAPOF_OK;
}

int papof_bench_sor(papof_handle* h, int height, int width, int n_sor, int sor_mode, int reps, unsigned seed,
                    double* ms_per_solve) {
    if (!h || !ms_per_solve || height < 1 || width < 1 || n_sor < 1 || reps < 1 || sor_mode < PAPOF_SOR_EXACT ||
        sor_mode > PAPOF_SOR_JACOBI)
        return PAPOF_EINVAL;
    const size_t np = (size_t)height * width;
    Scope S(h, img_bytes(height, width, 1, 16) + sor_scratch_bytes(height, width, n_sor) +
                   12 * (size_t)height * width * sizeof(double));
    PAPOF_TRY(S.rc);
    std::vector<double> host(np * 6);
    std::mt19937_64 rng(seed);
    auto fill = [&](size_t k, double lo, double hi) {
        std::uniform_real_distribution<double> d(lo, hi);
        for (size_t i = 0; i < np; i++) host[k * np + i] = d(rng);
    };
    fill(0, 0.5, 50.0);
    fill(1, -0.02, 0.02);
    fill(2, 0.0, 0.05);
    fill(3, 0.0, 0.05);
    fill(4, -0.01, 0.01);
    fill(5, -0.01, 0.01);
    double* planes[6];
    for (int k = 0; k < 6; k++) planes[k] = S.up_planar(host.data() + k * np, height, width, 1);
    SorPlanes sp{};
    PAPOF_TRY(alloc_sor_planes(S, height, width, sor_mode, n_sor, sp));
    PAPOF_TRY(sor_reset_planes(h, sp));
    PAPOF_TRY(sor_prep(h, planes[0], planes[1], planes[2], planes[3], planes[4], planes[5], height, width, 0.012, 1.8,
                       sp));
    PAPOF_TRY(sor_solve(h, sp, height, width, 0.012, 1.8, n_sor, sor_mode));  // warm-up
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    hipEvent_t e0, e1;
    PAPOF_HIP(hipEventCreate(&e0));
    PAPOF_HIP(hipEventCreate(&e1));
    PAPOF_HIP(hipEventRecord(e0, h->stream));
    for (int r = 0; r < reps; r++) PAPOF_TRY(sor_solve(h, sp, height, width, 0.012, 1.8, n_sor, sor_mode));
    PAPOF_HIP(hipEventRecord(e1, h->stream));
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    float ms = 0;
    PAPOF_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (sor_mode == PAPOF_SOR_EXACT) PAPOF_TRY(sor_check(h));
    *ms_per_solve = (double)ms / reps;
    if (std::getenv("PAPOF_SOR_DBG") && sor_mode == PAPOF_SOR_EXACT && sp.sd.fuse == 1) {  // per-task statistics
        const size_t ntask = (size_t)sp.sd.nb * n_sor;
        PAPOF_HIP(hipMalloc((void**)&h->sor_dbg, ntask * 8 * sizeof(unsigned long long)));
        PAPOF_HIP(hipMemset(h->sor_dbg, 0, ntask * 8 * sizeof(unsigned long long)));
        PAPOF_TRY(sor_solve(h, sp, height, width, 0.012, 1.8, n_sor, sor_mode));
        PAPOF_HIP(hipStreamSynchronize(h->stream));
        std::vector<unsigned long long> st(ntask * 8);
        PAPOF_HIP(hipMemcpy(st.data(), h->sor_dbg, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        hipFree(h->sor_dbg);
        h->sor_dbg = nullptr;
        if (sp.sd.group > 1) {
            for (int k = 0; k < n_sor && k < 8; k++)
                for (int b = 0; b < sp.sd.nb && b < 3; b++) {
                    const unsigned long long* o = &st[((size_t)k * sp.sd.nb + b) * 4];
                    std::fprintf(stderr, "[sor dbg] sweep %2d band %2d: total %8.1f us  wait_covered %7.1f  lds_in %7.1f  lds_out %7.1f\n",
                                 k, b, o[0] / 2400.0, o[1] / 2400.0, o[2] / 2400.0, o[3] / 2400.0);
                }
        } else {  // k_sor_exact: time line of every task, us since the first task entered (s_memrealtime = 100 MHz)
            unsigned long long t0 = ~0ull;
            for (size_t t = 0; t < ntask; t++)
                if (st[t * 8]) t0 = std::min(t0, st[t * 8]);
            for (int k = 0; k < n_sor && k < 10; k++)
                for (int b = 0; b < sp.sd.nb && b < 3; b++) {
                    const unsigned long long* o = &st[((size_t)k * sp.sd.nb + b) * 8];
                    std::fprintf(stderr, "[sor dbg] sweep %2d band %2d: entry %7.2f  covered %7.2f  iter0 %7.2f  iter1 %7.2f  iter2 %7.2f  iter3 %7.2f  end %8.2f\n",
                                 k, b, (o[0] - t0) * 0.01, (o[1] - t0) * 0.01, (o[2] - t0) * 0.01, (o[3] - t0) * 0.01,
                                 (o[4] - t0) * 0.01, (o[5] - t0) * 0.01, (o[6] - t0) * 0.01);
                }
        }
    }
    return PAPOF_OK;
}

}  // extern "C"
