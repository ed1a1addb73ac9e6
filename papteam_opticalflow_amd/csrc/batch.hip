// papteam_opticalflow_amd/csrc/batch.hip -- B frame pairs of ONE shape in ONE launch chain (papof_flow_batch*).
//
// Why.  The reference's own benchmark walks collections of SMALL frames (Code/Serial/TestSuite.py:69-81, :91: 101 pairs per
// collection, 240x135 ... 1920x1080, pyramidLevels 2 / 4 / 8 / 15).  One 240x135 pair on the reference schedule is ~210 kernel
// launches of a few microseconds of work each and 45 hand-off-bound solves on a chip it cannot fill.  Running many pairs side
// by side as independent calls (flow_collection: one handle, stream and host thread per sequence) plateaus at ~2 ms per pair
// with 16 in flight, and the limiter is the DEVICE's dispatch of ~84 k small dependent kernels per second over 16 queues -- the
// host spends 1.0 of 24 ms per call enqueueing, while every little kernel is stretched to 50-65 us
// (profiles/r04_collection_trace_240.txt, r04_collection_concurrency_240x16.txt).  Here the pairs of a batch share every
// launch: each buffer of the call is an ARRAY over the pairs (or frames), the per-pixel kernels take the pair from blockIdx.y
// (common.h: BatchK; plane-parallel kernels simply see C x frames planes), and ONE solver launch holds the tasks of all pairs
// (sor.hip: ExactArgs::bs_*, TinyArgs::bstride).  Same kernels, same operations per pair: every pair's results are bit-identical
// to the single call's (tests/test_gpu_batch.py).
//
// Scope: the default branches in the reference's own sweep order (exact order, n_inner = 1, bilinear warping, Laplacian noise
// model, 1 or 3 input channels, fewer than 16 solver bands per level); anything else -- and a batch of one -- runs as
// consecutive single calls through the same entry point.  A consecutive-pairs batch (`sequence`: pair i = frames i, i + 1, a
// video) builds every frame's pyramid and features once.
//
// The Laplacian-noise guard (api.hip: LapGuard; src/OpticalFlow.cpp:399-400): the batch runs the OPTIMISTIC pass with one block
// of witness flags per pair and non-zero flags per frame (few-pixel levels: an exhaustive check behind every update); a pair that
// ends without a proof for a consulted estimate (duplicate frames) is run again through the single call, which has the exact
// pass.  Results: the reference's, either way.
#include <algorithm>
#include <chrono>
#include <cstring>

#include "common.h"
#include "flow_internal.h"

namespace papof {

namespace {

double now_sec() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct BatchOut {
    const double* uv = nullptr;    // [pair][2][H * W]: the final flow
    const double* warp = nullptr;  // [pair][H * W * C]
    std::vector<int> unproven;     // pairs whose Laplacian-noise guard could not be proven open: run them through the single call
};

bool batch_applies(const papof_handle* h, int B, int H, int W, int C, int levels, const papof_params& P) {
    if (B < 2 || !h->use_dpp || h->use_graph || h->strips > 1) return false;
    if (P.sor_mode != PAPOF_SOR_EXACT || P.n_inner != 1 || P.interpolation != PAPOF_INTERP_BILINEAR ||
        P.noise_model != PAPOF_NOISE_LAPLACIAN)
        return false;
    if (C != 1 && C != 3) return false;
    const int n_sor_max = P.n_sor + (levels - 1) * P.n_sor_per_level;
    if (skew_dims(H, W, n_sor_max, 1, 1).nb >= 16) return false;  // big frames: the two-sweeps-per-wave kernel; they fill the chip alone
    long slots = 0;
    for (int k = 0; k < levels; k++) slots += P.n_outer + (long)k * P.n_outer_per_level;
    return slots <= kLapMaxSlots;
}

// Everything on h->stream, one stream: B pairs fill the chip by themselves.
int flow_batch_device(papof_handle* h, int B, int sequence, const void* const* frames, bool u8, int H, int W, int C, int levels,
                      const papof_params& P, double* timing, BatchOut& out) {
    PAPOF_TRY(check_params(P, levels));
    double ratio = P.ratio;
    if (ratio > 0.98 || ratio < 0.4) ratio = 0.75;
    std::vector<Level> L;
    std::vector<PyrPlan> plan;
    PAPOF_TRY(pyramid_plan(H, W, P.ratio, levels, L, plan));
    const int fstep = sequence ? 1 : 2, nF = sequence ? B + 1 : 2 * B;
    const int fc = feature_channels(C);
    const size_t np0 = (size_t)H * W;
    const int n_sor_max = P.n_sor + (levels - 1) * P.n_sor_per_level;
    const bool guard = h->lap_guard && h->lap_flags_dev != nullptr;
    // ---- arena: one block, arrays over frames / pairs
    size_t pyr_px = 0;
    for (const Level& l : L) pyr_px += (size_t)l.w * l.h;
    size_t cells = 0, cells_d = 0;
    skew_capacity(H, W, n_sor_max, cells, cells_d);
    const size_t tiny_cells = std::min<size_t>(kTinyMaxCells, np0);
    size_t bytes = 0;
    bytes += (size_t)nF * np0 * C * (u8 ? 1 : 8) + 4096;                            // the uploaded frames
    bytes += (size_t)nF * pyr_px * (C + 2 * fc) * 8 + (size_t)levels * 3 * 4096;     // pyramids, features, smoothed features
    bytes += (size_t)nF * np0 * C * 8 * 5 + 5 * 4096;                               // two filter temporaries, three derivative planes
    bytes += (size_t)B * (4 * np0 * 8 + np0 * C * 8) + 3 * 4096;                    // two flow pairs, the warped image
    bytes += (size_t)B * ((3 * 2 * ((cells + kLanes + 63) / 64 * 64) + 2 * (cells_d + kLanes)) * 8 + 2 * 4096);  // solver operands
    bytes += (size_t)B * (8 * (tiny_cells * 8 + 256));                              // row-major operands of k_sor_tiny's levels
    bytes += (size_t)(B + nF) * kLapFlagWords * 4 + 8192;                           // guard flags
    bytes += (size_t)1 << 20;
    h->seq.valid = false;  // (the arena is laid out anew: a kept sequence pyramid does not survive a batch)
    PAPOF_TRY(ensure_arena(h, bytes));
    Arena& A = h->arena;
    A.off = 0;
    A.overflow = false;
    h->events_used = 0;
    h->sor_log.clear();
    h->sor_launches = 0;
    hipStream_t const st = h->stream;

    unsigned char* stage = static_cast<unsigned char*>(A.alloc((size_t)nF * np0 * C * (u8 ? 1 : 8)));
    std::vector<double*> Lp(levels), F(levels), S(levels);
    for (int k = 0; k < levels; k++) {
        const size_t n = (size_t)L[k].w * L[k].h;
        Lp[k] = A.f64((size_t)nF * n * C);
        F[k] = A.f64((size_t)nF * n * fc);
        S[k] = A.f64((size_t)nF * n * fc);
    }
    double* tmp_a = A.f64((size_t)nF * np0 * C);
    double* tmp_b = A.f64((size_t)nF * np0 * C);
    double* gx = A.f64((size_t)nF * np0 * C);
    double* gy = A.f64((size_t)nF * np0 * C);
    double* gxy = A.f64((size_t)nF * np0 * C);
    double* uvA = A.f64((size_t)B * 2 * np0);
    double* uvB = A.f64((size_t)B * 2 * np0);
    double* warp = A.f64((size_t)B * np0 * C);
    unsigned* wit = reinterpret_cast<unsigned*>(A.alloc((size_t)B * kLapFlagWords * sizeof(unsigned)));
    unsigned* nzf = reinterpret_cast<unsigned*>(A.alloc((size_t)nF * kLapNzWords * sizeof(unsigned)));
    if (A.overflow) return PAPOF_ENOMEM;
    // the pairs' solver operands: identical allocation sequences, hence a constant stride from pair to pair
    std::vector<SorPlanes> SP(B), ST(B);
    for (int p = 0; p < B; p++) PAPOF_TRY(sor_alloc_planes(A, H, W, PAPOF_SOR_EXACT, n_sor_max, SP[p]));
    for (int p = 0; p < B; p++) PAPOF_TRY(sor_alloc_tiny_planes(A, tiny_cells, ST[p]));
    if (A.overflow) return PAPOF_ENOMEM;
    const size_t sp_stride = (size_t)(SP[1].phi - SP[0].phi), spd_stride = (size_t)(SP[1].du - SP[0].du);
    const size_t st_stride = (size_t)(ST[1].phi - ST[0].phi);
    for (int p = 1; p < B; p++)
        if ((size_t)(SP[p].phi - SP[0].phi) != p * sp_stride || (size_t)(SP[p].du - SP[0].du) != p * spd_stride ||
            (size_t)(ST[p].phi - ST[0].phi) != p * st_stride || (size_t)(ST[p].du - ST[0].du) != p * st_stride)
            return PAPOF_EDEVICE;

    // ---- counters of every solve of every pair, cleared once
    struct LevelCounters {
        size_t off, per;
        bool tiny;
    };
    std::vector<LevelCounters> LC(levels);
    size_t prog_total = 0;
    for (int k = 0; k < levels; k++) {
        const int Kk = P.n_sor + k * P.n_sor_per_level, n_outer = P.n_outer + k * P.n_outer_per_level;
        LC[k].tiny = (size_t)L[k].w * L[k].h <= kTinyMaxCells && sor_tiny_fits(h, L[k].h, L[k].w, Kk);
        LC[k].per = LC[k].tiny ? 0 : (size_t)skew_dims(L[k].h, L[k].w, Kk, 1, 1).nb * Kk * 32;
        LC[k].off = prog_total;
        prog_total += LC[k].per * (size_t)n_outer * B;
    }
    PAPOF_TRY(sor_counters_ensure(h, prog_total));

    // ---- timers: the total, and the solver kernels' own time (events around their launches)
    PhaseClock total{h, true}, sorclk{h, true};
    sorclk.only_sor = true;
    double tm[PAPOF_N_TIMERS + 1];
    std::memset(tm, 0, sizeof tm);
    total.phase(PAPOF_T_TOTAL);

    // ---- uploads, layout conversion, pyramids, features (frames are planes x frames for the plane-parallel kernels)
    const size_t frame_bytes = np0 * C * (u8 ? 1 : 8);
    bool flat_in = true;  // frames stacked in one host block (the Python binding's): one copy
    for (int f = 1; f < nF && flat_in; f++)
        flat_in = (const unsigned char*)frames[f] == (const unsigned char*)frames[0] + (size_t)f * frame_bytes;
    if (flat_in) {
        PAPOF_HIP(hipMemcpyAsync(stage, frames[0], (size_t)nF * frame_bytes, hipMemcpyHostToDevice, st));
    } else {
        for (int f = 0; f < nF; f++)
            PAPOF_HIP(hipMemcpyAsync(stage + (size_t)f * frame_bytes, frames[f], frame_bytes, hipMemcpyHostToDevice, st));
    }
    if (prog_total && !sor_counters_clear(h, 0, prog_total)) return PAPOF_EDEVICE;
    if (++h->lap_epoch >= kLapNone) {  // (2^31 passes later: start over on cleared flags, as flow_device does -- the handle's own
        // flag block holds pass numbers that are never cleared otherwise, and `epoch ^ kLapNone` must not be 0)
        if (h->lap_flags_dev) PAPOF_HIP(hipMemsetAsync(h->lap_flags_dev, 0, kLapFlagWords * sizeof(unsigned), st));
        h->lap_epoch = 1u;
    }
    const unsigned epoch = h->lap_epoch;  // a flag is set when it holds this call's number
    if (guard) {
        PAPOF_HIP(hipMemsetAsync(wit, 0, (size_t)B * kLapFlagWords * sizeof(unsigned), st));
        PAPOF_HIP(hipMemsetAsync(nzf, 0, (size_t)nF * kLapNzWords * sizeof(unsigned), st));
    }
    if (u8)
        PAPOF_TRY(hwc_u8_to_planar(h, stage, Lp[0], H, W, C, nF));
    else
        PAPOF_TRY(hwc_to_planar(h, reinterpret_cast<const double*>(stage), Lp[0], H, W, C, nF));
    for (int i = 1; i < levels; i++)
        PAPOF_TRY(smooth_and_resize(h, Lp[plan[i].src_level], Lp[i], tmp_a, tmp_b, plan[i], C * nF, L[i].h, L[i].w));
    const Taps g5 = smooth5_taps();
    const bool nz_known = guard && (size_t)levels * 8 <= (size_t)kLapNzWords;
    for (int k = 0; k < levels; k++) {
        PAPOF_TRY(im2feature(h, Lp[k], F[k], L[k].h, L[k].w, C, nz_known ? nzf + (size_t)k * 8 : nullptr, nF, kLapNzWords));
        PAPOF_TRY(filter_hv(h, F[k], S[k], nullptr, L[k].h, L[k].w, fc * nF, g5, g5));
    }
    PAPOF_TRY(central3_planes(h, Lp[0], gx, gy, gxy, H, W, C * nF));

    // ---- levels, coarse to fine (src/OpticalFlow.cpp:784-823); (u, v) of pair p: planes 2p, 2p + 1 at the level's pitch
    double *uv = uvA, *uv2 = uvB;
    int pw = 0, ph = 0, slot = 0;
    std::vector<int> slot_level;
    h->sor_mark = [](void* c, int on) { static_cast<PhaseClock*>(c)->phase(on ? PAPOF_T_PHASE5_SOR : -1); };
    h->sor_mark_ctx = &sorclk;
    const auto levels_loop = [&]() -> int {
        for (int k = levels - 1; k >= 0; k--) {
            const int lw = L[k].w, lh = L[k].h;
            const size_t np = (size_t)lw * lh;
            const int K = P.n_sor + k * P.n_sor_per_level, n_outer = P.n_outer + k * P.n_outer_per_level;
            if (k == levels - 1) {
                PAPOF_HIP(hipMemsetAsync(uv, 0, (size_t)B * 2 * np * sizeof(double), st));  // :801-806 (the warp is folded in)
            } else {  // :809-812: bilinear up-sampling times 1 / ratio, all pairs' u and v in one launch
                PAPOF_TRY(resize(h, uv, uv2, ph, pw, 2 * B, lh, lw, (double)lw / pw, (double)lh / ph, true, 1 / ratio));
                std::swap(uv, uv2);
            }
            const bool tiny = LC[k].tiny;
            SorPlanes sp = tiny ? ST[0] : SP[0];
            if (!tiny) {
                PAPOF_TRY(sor_bind_plain(h, sp, lh, lw, K));
                PAPOF_TRY(sor_reset_planes_batch(h, sp, B, sp_stride));
            }
            BatchK bk{};
            bk.im = (size_t)fstep * np * fc;
            bk.uv = 2 * np;
            bk.sp = tiny ? st_stride : sp_stride;
            bk.d = tiny ? st_stride : spd_stride;
            bk.wit = kLapFlagWords;
            const SorBatch bt{B, sp_stride, spd_stride, LC[k].per, st_stride};
            const double *f1 = F[k], *f2 = F[k] + np * fc, *s1 = S[k];
            // Few-pixel levels (the deep end of an 8- or 15-level pyramid): the flow leaves the image altogether in some iterations,
            // every warped value is then frame 1's and there is NO valid sample -- which only a look at every pixel can tell from
            // "no witness among the samples".  There the estimate behind EVERY update is checked exhaustively (one small block per
            // pair and channel): a witness, or the constant 0.001 (kLapNone).
            const bool exhaustive = guard && np <= 4096;
            for (int count = 0; count < n_outer; count++) {
                unsigned* const w_prev = guard && !exhaustive && count > 0 ? wit + lap_wit_word(slot - 1) : nullptr;
                PAPOF_TRY(flow_system(h, f1, f2, uv, uv + np, s1, lh, lw, fc, P.alpha, P.omega, sp, w_prev, 0, -1, B, &bk));
                h->sor_prog_next = tiny ? nullptr : h->sync_words + 32 + LC[k].off + (size_t)count * LC[k].per * B;
                const int rc = sor_solve(h, sp, lh, lw, P.alpha, P.omega, K, PAPOF_SOR_EXACT, &bt);
                h->sor_prog_next = nullptr;
                PAPOF_TRY(rc);
                // the estimate behind the level's last update is witnessed by the update kernel itself (nobody evaluates that warp)
                unsigned* const w_now = guard && !exhaustive && count + 1 == n_outer ? wit + lap_wit_word(slot) : nullptr;
                PAPOF_TRY(update_warp_phi(h, sp, uv, uv + np, uv2, uv2 + np, f1, f2, nullptr, nullptr, lh, lw, fc, false, 0, -1,
                                          w_now, B, &bk));
                std::swap(uv, uv2);
                if (exhaustive) PAPOF_TRY(lap_small_check(h, f1, f2, uv, uv + np, lh, lw, fc, wit + lap_wit_word(slot), B, &bk));
                slot_level.push_back(k);
                slot++;
            }
            pw = lw;
            ph = lh;
        }
        // ---- :841-842: bicubic warp of the ORIGINAL frame 2 of every pair, clamped to [0, 1]
        BatchK bk{};
        bk.im = (size_t)fstep * np0 * C;
        bk.uv = 2 * np0;
        bk.out = np0 * C;
        PAPOF_TRY(bicubic_warp(h, Lp[0], Lp[0] + np0 * C, gx + np0 * C, gy + np0 * C, gxy + np0 * C, uv, uv + np0, warp, H, W, C,
                               nullptr, false, true, B, &bk));
        return PAPOF_OK;
    };
    const int rc_levels = levels_loop();
    h->sor_mark = nullptr;
    h->sor_mark_ctx = nullptr;
    total.phase(-1);
    if (rc_levels != PAPOF_OK) {
        hipStreamSynchronize(st);
        return rc_levels;
    }
    std::vector<unsigned> hw, hn;
    PAPOF_HIP(hipStreamSynchronize(st));
    PAPOF_TRY(sor_check(h));
    if (guard) {
        hw.resize((size_t)B * kLapFlagWords);
        hn.resize((size_t)nF * kLapNzWords);
        PAPOF_HIP(hipMemcpy(hw.data(), wit, hw.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        PAPOF_HIP(hipMemcpy(hn.data(), nzf, hn.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        for (int p = 0; p < B; p++) {
            bool open = true;
            const int fa = p * fstep, fb = fa + 1;
            for (int s = 0; s + 1 < slot && open; s++)  // the estimate behind the call's last update is never consulted
                for (int c = 0; c < fc; c++) {
                    const unsigned w = hw[(size_t)p * kLapFlagWords + lap_wit_word(s) + c];
                    const bool witness = w == epoch || w == (epoch ^ kLapNone);  // a sample, or an exhaustive "no valid sample at all"
                    const size_t nzw = (size_t)slot_level[s] * 8 + c;
                    const bool all_zero = nz_known && hn[(size_t)fa * kLapNzWords + nzw] != epoch &&
                                          hn[(size_t)fb * kLapNzWords + nzw] != epoch;
                    if (!(witness || all_zero)) open = false;
                }
            if (!open) out.unproven.push_back(p);
        }
    }
    if (total.err != PAPOF_OK || sorclk.err != PAPOF_OK) return PAPOF_EDEVICE;
    sorclk.collect(tm);
    total.collect(tm);
    if (timing) std::memcpy(timing, tm, PAPOF_N_TIMERS * sizeof(double));
    out.uv = uv;
    out.warp = warp;
    return PAPOF_OK;
}

}  // namespace

int flow_batch_host(papof_handle* h, int n_pairs, int sequence, const void* const* frames, bool u8, int H, int W, int C, int levels,
                    const papof_params* params, double* const* vx, double* const* vy, double* const* warpI2, double* timing_sec) {
    if (!h || n_pairs < 1 || !frames || !vx || !vy || !warpI2 || H < 1 || W < 1 || C < 1 || levels < 1) return PAPOF_EINVAL;
    const int nF = sequence ? n_pairs + 1 : 2 * n_pairs;
    for (int f = 0; f < nF; f++)
        if (!frames[f]) return PAPOF_EINVAL;
    for (int p = 0; p < n_pairs; p++)
        if (!vx[p] || !vy[p] || !warpI2[p]) return PAPOF_EINVAL;
    papof_params P;
    if (params)
        P = *params;
    else
        papof_default_params(&P);
    PAPOF_TRY(check_params(P, levels));
    PAPOF_HIP(hipSetDevice(h->device));
    const double t0 = now_sec();
    double tm[PAPOF_N_TIMERS];
    std::memset(tm, 0, sizeof tm);
    const auto single = [&](int p) -> int {  // the ordinary call on pair p (it has every branch, and the guard's exact pass)
        const void* a = frames[sequence ? p : 2 * p];
        const void* b = frames[sequence ? p + 1 : 2 * p + 1];
        double t1[PAPOF_N_TIMERS];
        const int rc = u8 ? papof_flow_u8(h, (const unsigned char*)a, (const unsigned char*)b, H, W, C, levels, &P, vx[p], vy[p],
                                          warpI2[p], t1)
                          : papof_flow(h, (const double*)a, (const double*)b, H, W, C, levels, &P, vx[p], vy[p], warpI2[p], t1);
        if (rc == PAPOF_OK)
            for (int i = 0; i < PAPOF_N_TIMERS; i++) tm[i] += t1[i];
        return rc;
    };
    if (!batch_applies(h, n_pairs, H, W, C, levels, P)) {
        for (int p = 0; p < n_pairs; p++) PAPOF_TRY(single(p));
        if (timing_sec) std::memcpy(timing_sec, tm, sizeof tm);
        return PAPOF_OK;
    }
    // One launch of the solver holds the tasks of ALL pairs, and all of them must be resident (sor.hip: resident_tasks): a
    // collection larger than that -- or than a few GB of arena -- goes through in sub-batches of the same shape.
    {
        const int nb0 = skew_dims(H, W, P.n_sor + (levels - 1) * P.n_sor_per_level, 1, 1).nb;
        const size_t per_pair = arena_bytes_for(H, W, C, levels, P.n_sor + (levels - 1) * P.n_sor_per_level, P.ratio);
        int max_b = (int)std::max<size_t>(2, std::min<size_t>((size_t)1024 / (size_t)nb0, ((size_t)24 << 30) / per_pair));
        if (const char* e = std::getenv("PAPOF_BATCH_MAX")) max_b = std::max(2, std::min(max_b, std::atoi(e)));  // (and the tests')
        if (n_pairs > max_b) {
            const int step = sequence ? 1 : 2;
            for (int p0 = 0; p0 < n_pairs;) {
                int nb_ = std::min(max_b, n_pairs - p0);
                if (n_pairs - (p0 + nb_) == 1 && nb_ > 2) nb_ -= 1;  // leave two pairs for the last sub-batch rather than one
                double t1[PAPOF_N_TIMERS];
                PAPOF_TRY(flow_batch_host(h, nb_, sequence, frames + (size_t)p0 * step, u8, H, W, C, levels, &P, vx + p0, vy + p0,
                                          warpI2 + p0, t1));
                for (int i = 0; i < PAPOF_N_TIMERS; i++) tm[i] += t1[i];
                p0 += nb_;
            }
            if (timing_sec) std::memcpy(timing_sec, tm, sizeof tm);
            return PAPOF_OK;
        }
    }
    BatchOut out;
    {
        const int rc = flow_batch_device(h, n_pairs, sequence, frames, u8, H, W, C, levels, P, tm, out);
        if (rc == PAPOF_ENOMEM) {  // no room for the arrays of a batch on this device: the pairs one after the other
            std::memset(tm, 0, sizeof tm);
            for (int p = 0; p < n_pairs; p++) PAPOF_TRY(single(p));
            if (timing_sec) std::memcpy(timing_sec, tm, sizeof tm);
            return PAPOF_OK;
        }
        PAPOF_TRY(rc);
    }
    const size_t np0 = (size_t)H * W;
    // Result arrays laid out as the device's ([pair][vx, vy][H x W] and [pair][H x W x c] in one block each -- what the Python
    // binding allocates, page-locked) come back as TWO copies; anything else as three per pair.
    bool flat = true;
    for (int p = 0; p < n_pairs && flat; p++)
        flat = vx[p] == vx[0] + (size_t)p * 2 * np0 && vy[p] == vx[p] + np0 && warpI2[p] == warpI2[0] + (size_t)p * np0 * C;
    if (flat) {
        PAPOF_HIP(hipMemcpyAsync(vx[0], out.uv, (size_t)n_pairs * 2 * np0 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        PAPOF_HIP(hipMemcpyAsync(warpI2[0], out.warp, (size_t)n_pairs * np0 * C * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    } else {
        for (int p = 0; p < n_pairs; p++) {
            PAPOF_HIP(hipMemcpyAsync(vx[p], out.uv + (size_t)p * 2 * np0, np0 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            PAPOF_HIP(hipMemcpyAsync(vy[p], out.uv + (size_t)p * 2 * np0 + np0, np0 * sizeof(double), hipMemcpyDeviceToHost,
                                     h->stream));
            PAPOF_HIP(hipMemcpyAsync(warpI2[p], out.warp + (size_t)p * np0 * C, np0 * C * sizeof(double), hipMemcpyDeviceToHost,
                                     h->stream));
        }
    }
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    for (int p : out.unproven) {
        h->lap_reruns++;
        PAPOF_TRY(single(p));
    }
    tm[PAPOF_T_TOTAL] = now_sec() - t0;  // the caller's view: uploads, the launch chain, downloads, re-runs
    if (timing_sec) std::memcpy(timing_sec, tm, sizeof tm);
    return PAPOF_OK;
}

}  // namespace papof

using namespace papof;

extern "C" {

int papof_flow_batch(papof_handle* h, int n_pairs, int sequence, const double* const* frames, int height, int width, int c,
                     int pyramid_levels, const papof_params* params, double* const* vx, double* const* vy, double* const* warpI2,
                     double timing_sec[PAPOF_N_TIMERS]) {
    return flow_batch_host(h, n_pairs, sequence, reinterpret_cast<const void* const*>(frames), false, height, width, c,
                           pyramid_levels, params, vx, vy, warpI2, timing_sec);
}

int papof_flow_batch_u8(papof_handle* h, int n_pairs, int sequence, const unsigned char* const* frames, int height, int width,
                        int c, int pyramid_levels, const papof_params* params, double* const* vx, double* const* vy,
                        double* const* warpI2, double timing_sec[PAPOF_N_TIMERS]) {
    return flow_batch_host(h, n_pairs, sequence, reinterpret_cast<const void* const*>(frames), true, height, width, c,
                           pyramid_levels, params, vx, vy, warpI2, timing_sec);
}

}  // extern "C"
