// papteam_opticalflow_amd/csrc/flow_internal.h -- pieces of the orchestrator (api.hip) shared with the tiled
// multi-GPU orchestrator (tiles.hip).  Internal; nothing here is part of the C ABI.
#pragma once

#include <string>
#include <utility>
#include <vector>

#include "common.h"

namespace papof {

void set_last_error_text(const std::string& text);

// ---- phase timers (HIP events on the handle's stream) ----
struct PhaseClock {
    papof_handle* h;
    bool on;
    std::vector<std::pair<int, std::pair<size_t, size_t>>> spans;  // (timer index, (event a, event b))
    size_t open = 0;
    int open_idx = -1;
    int err = PAPOF_OK;
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
int feature_channels(int C);

}  // namespace papof
