// A simulated GPU for gsplat::RoundsController (csrc/rounds_controller.h): frame times follow a cost model of a scene,
// observations arrive `lag` frames after a frame was issued (4 timing slots, like api.hip's ring).
//   usage: sim <dense|sparse|dense_then_sparse> <lag> <frames> [noise amplitude, default 0.004]
// prints one JSON line.
#include "rounds_controller.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>

static double g_noise = 0.004;
static double frame_ms(bool dense, bool two, double f, uint32_t k) {
    // measurement noise, deterministic: a slow wobble plus a pseudo-random part
    const uint32_t h = (k * 2654435761u) >> 8;
    const double wobble = 1.0 + g_noise * (0.5 * std::sin(0.7 * k) + ((h & 0xFFFF) / 65535.0 - 0.5));
    if (dense) return (two ? 0.60 + 1.15 * f + 0.0004 / f : 1.11) * wobble;   // c3d-like: measured 0.89 @0.25, 0.66 @0.04
    return (two ? 0.88 + 0.3 * std::fabs(f - 0.3) : 0.79) * wobble;           // c3-like: two rounds never pay
}

struct Pending { uint32_t ready_at, trial; bool counts; float ms; };

int main(int argc, char **argv) {
    const std::string scene = argc > 1 ? argv[1] : "dense";
    const uint32_t lag = argc > 2 ? (uint32_t)atoi(argv[2]) : 1u, frames = argc > 3 ? (uint32_t)atoi(argv[3]) : 3000u;
    if (argc > 4) g_noise = atof(argv[4]);
    gsplat::RoundsController ctl;
    std::deque<Pending> ring;
    double total = 0, tail_total = 0, tail_best = 0;
    uint32_t tail_frames = 0, tail_two = 0, first_hold_two = 0, switches = 0;
    bool last_two = false;
    for (uint32_t k = 0; k < frames; ++k) {
        const bool dense = scene == "dense" || (scene == "dense_then_sparse" && k < frames / 2);
        while (!ring.empty() && ring.front().ready_at <= k) {
            ctl.observe(ring.front().trial, ring.front().counts, ring.front().ms);
            ring.pop_front();
        }
        bool wants = false, counts = false;
        const bool two = ctl.begin_frame(&wants, &counts);
        const double f = ctl.frac16 / 65536.0;
        const double ms = frame_ms(dense, two, f, k);
        if (wants && ring.size() < 4) ring.push_back({k + lag, ctl.trial, counts, (float)ms});
        total += ms;
        if (two != last_two) ++switches;
        last_two = two;
        if (ctl.phase == gsplat::RoundsController::HOLD && ctl.inc_two && !first_hold_two) first_hold_two = k;
        const bool tail = scene == "dense_then_sparse" ? k >= frames - frames / 4 : k >= 300;
        if (tail) {
            ++tail_frames;
            tail_two += two;
            tail_total += ms;
            tail_best += dense ? 0.60 + 2.0 * std::sqrt(1.15 * 0.0004) : 0.79;
        }
    }
    printf("{\"scene\": \"%s\", \"lag\": %u, \"frames\": %u, \"final_two\": %d, \"final_frac\": %.5f, \"tail_two_share\": %.4f, "
           "\"tail_ms\": %.5f, \"tail_best_ms\": %.5f, \"first_hold_two\": %u, \"switches\": %u, \"avg_ms\": %.5f}\n",
           scene.c_str(), lag, frames, (int)ctl.inc_two, ctl.best_frac16 / 65536.0, tail_frames ? (double)tail_two / tail_frames : 0.0,
           tail_frames ? tail_total / tail_frames : 0.0, tail_frames ? tail_best / tail_frames : 0.0, first_hold_two, switches,
           total / frames);
    return 0;
}
