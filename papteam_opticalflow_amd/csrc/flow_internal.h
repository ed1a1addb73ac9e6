// papteam_opticalflow_amd/csrc/flow_internal.h -- pieces of the orchestrator (api.hip) shared with the tiled
// multi-GPU orchestrator (tiles.hip).  Internal; nothing here is part of the C ABI.
#pragma once

#include <string>
#include <utility>
#include <vector>

#include "common.h"

namespace papof {

void set_last_error_text(const std::string& text);

// ---- phase timers ----
// Event mode (default): every phase boundary is a HIP event recorded on the handle's current stream (no synchronisation).
// Stamp mode (`stamps`, the main-stream clock of flow_device): a phase boundary costs NOTHING on the stream -- the first
// kernel launched after phase() writes the GPU's constant 100 MHz clock into a slot when it starts
// (kernels.hip: stamp_now; device memory, copied back once at the end of the call: a kernel that writes host-mapped
// memory pays for it when it ends), and a phase lasts from its stamp to the next one.  A HIP event between two kernels costs
// ~2.6 us of stream time, and a call has ~100 phase boundaries: measured 11.8 vs 11.5 ms per 1080p pair.  Only
// Phase5_SOR keeps a HIP event pair as well (the roofline of the dominant kernel is priced on those events).
// Internal eleventh timer (arrays handed to PhaseClock::collect hold PAPOF_N_TIMERS + 1 values): the kernel that does the work
// of Phase1 ... Phase4 in one launch (kernels.hip: k_flow_system); api.hip apportions it to those four when it reports.
constexpr int kTimerFused = PAPOF_N_TIMERS;
struct PhaseClock {
    papof_handle* h;
    bool on;
    std::vector<std::pair<int, std::pair<size_t, size_t>>> spans;  // (timer index, (event a, event b))
    size_t open = 0;
    int open_idx = -1;
    int err = PAPOF_OK;
    bool only_sor = false;  // measurement aid (PAPOF_PHASE_EVENTS=0): nothing but Phase5_SOR (and the total) is measured
    bool stamps = false;
    std::vector<std::pair<int, int>> marks;  // stamp mode: (slot, timer index) in stream order
    std::vector<double> sor_span_sec;        // collect(): seconds of every Phase5_SOR span, in stream order (one per solve)
    std::vector<int> sor_owner;              // stamp mode: the timer whose span runs across that solve (see phase())
    size_t new_event() {
        if (h->events_used == h->events.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) {
                err = PAPOF_EDEVICE;
                return 0;
            }
            h->events.push_back(e);
        }
        hipEventRecord(h->events[h->events_used], h->stream);
        return h->events_used++;
    }
    // close the running span (if any) and open a new one attributed to timer `idx` (-1: none)
    void phase(int idx) {
        if (!on) return;
        if (stamps) {
            if (open_idx == PAPOF_T_PHASE5_SOR) {
                const size_t e = new_event();
                spans.push_back({PAPOF_T_PHASE5_SOR, {open, e}});
            }
            if (idx == PAPOF_T_PHASE5_SOR) {
                open = new_event();
                sor_owner.push_back(marks.empty() ? PAPOF_T_PHASE4_LINEARSYSTEM : marks.back().second);
            }
            const bool sor_edge = idx == PAPOF_T_PHASE5_SOR;
            open_idx = idx;
            // The solver kernels carry no stamp (a store at the head of the critical task of the exact-order kernels was
            // measured to cost ~12 us per solve): the span that contains a solve keeps running as Phase4 (or kTimerFused) until
            // the update kernel's stamp, and collect() subtracts the solver's own (event-measured) time from it.
            if (only_sor || !h->stamps_dev || sor_edge) return;
            if (h->next_stamp && !marks.empty()) {
                marks.back().second = idx;  // no stamping kernel ran in the previous phase: it is absorbed by its predecessor
            } else if (h->stamps_used < h->stamps_cap) {
                const int slot = h->stamps_used++;
                h->next_stamp = h->stamps_dev + slot;
                marks.push_back({slot, idx});
            }
            return;
        }
        if (only_sor && idx != PAPOF_T_PHASE5_SOR && idx != PAPOF_T_TOTAL) {
            if (open_idx < 0) return;  // nothing running: no event needed
            idx = -1;
        }
        const size_t e = new_event();
        if (open_idx >= 0) spans.push_back({open_idx, {open, e}});
        open = e;
        open_idx = idx;
    }
    void collect(double* t) {  // after the stream has drained
        if (!on) return;
        for (auto& s : spans) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, h->events[s.second.first], h->events[s.second.second]) == hipSuccess)
                t[s.first] += ms * 1e-3;
            if (s.first == PAPOF_T_PHASE5_SOR) sor_span_sec.push_back(ms * 1e-3);
        }
        // h->stamps holds the slots 0 .. stamps_fetched-1 (copied back by fetch_stamps() before the stream was drained)
        for (size_t i = 0; i + 1 < marks.size(); i++) {
            const int idx = marks[i].second;
            if (marks[i + 1].first >= h->stamps_fetched) break;
            const unsigned long long a = h->stamps[marks[i].first], b = h->stamps[marks[i + 1].first];
            if (idx >= 0 && idx != PAPOF_T_PHASE5_SOR && b >= a) t[idx] += (double)(b - a) * 1e-8;  // 100 MHz ticks
        }
        if (stamps && !marks.empty()) {  // the spans that ran across the solves: take the solver kernels' time out
            for (size_t i = 0; i < sor_span_sec.size() && i < sor_owner.size(); i++)
                if (sor_owner[i] >= 0) t[sor_owner[i]] -= sor_span_sec[i];
            for (int own : {(int)PAPOF_T_PHASE4_LINEARSYSTEM, kTimerFused})
                if (t[own] < 0.0) t[own] = 0.0;
        }
    }
};

struct Level {
    int w, h;
    double *p1, *p2;  // planar pyramid levels of frame 1 / frame 2
};


// GaussianPyramid::ConstructPyramidLevels (src/GaussianPyramid.cpp:79-108) for one frame, planar.
// levels[i].p (selected by `which`) must be pre-allocated with the dims computed by pyramid_dims().
struct PyrPlan {
    int sw, sh, src_level, fsize;
    double sigma, rate;
};


size_t arena_bytes_for(int H, int W, int C, int levels, int n_sor_max, double ratio);
int ensure_arena(papof_handle* h, size_t bytes);
int check_params(const papof_params& P, int levels);
int pyramid_plan(int H, int W, double ratio, int nlev, std::vector<Level>& L, std::vector<PyrPlan>& plan);
int build_pyramid(papof_handle* h, const std::vector<Level>& L, const std::vector<PyrPlan>& plan, int C, bool second,
                  double* tmp_a, double* tmp_b);
int smooth_and_resize(papof_handle* h, const double* src, double* dst, double* tmp_a, double* tmp_b, const PyrPlan& p, int C,
                      int dh, int dw);  // one pyramid level from its source level; C may count the planes of SEVERAL contiguous frames
int feature_channels(int C);
// batch.hip: B frame pairs of one shape in one launch chain (host frames in, host results out); falls back to single calls
int flow_batch_host(papof_handle* h, int n_pairs, int sequence, const void* const* frames, bool u8, int H, int W, int C, int levels,
                    const papof_params* params, double* const* vx, double* const* vy, double* const* warpI2, double* timing_sec);
void ensure_strip_streams(papof_handle* h);

}  // namespace papof
