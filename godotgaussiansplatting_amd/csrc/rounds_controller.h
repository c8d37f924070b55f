// One round or two?  The controller of the two-round frame schedule, free of HIP so that it can be
// driven by a simulated GPU on the CPU (tests/test_rounds_controller.py).
//
// Whether two rounds pay depends on the scene: where every tile saturates early (a dense capture) round B is nearly
// empty and the pair-level work shrinks several-fold; where most tiles never saturate the second round's launches cost
// more than the pairs it saves.  So the context MEASURES.  api.hip times frames with a ring of event pairs that is
// polled, never waited for, and reports each time as it becomes available (observe); begin_frame gives the setting of
// the next frame.  A session starts on one round; every few hundred frames (first after six) the setting held gets a
// re-check: short trials — eight frames at most — of itself and of a few alternatives.  When two rounds first beat
// one, the fraction climbs in steps of x0.8 / x1.25 while the frame time falls and the result meets one round once more
// before it is held.  Any setting gives the same image; the worst a trial can do is cost a few slower frames.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>

namespace gsplat {

struct RoundsController {
    enum Phase { CLIMB = 0, TRY_ONE = 1, HOLD = 2, RECHECK = 3 };
    static constexpr uint32_t HOLD_FRAMES = 400, FIRST_HOLD = 6, RECHECK_MAX_FRAMES = 8;
    static constexpr uint32_t FRAC_MIN = 256, FRAC_MAX = 49152;  // x 1/65536: 0.4 % .. 75 % of the visible splats

    int phase = HOLD;
    bool two = false;            // the setting of the current trial (a session starts by holding one round)
    uint32_t frac16 = 16384;     // size of round A as a fraction of the visible splats, x 65536
    int dir = -1;
    uint32_t reversals = 0, trial = 0, trial_frames = 0, trial_obs = 0, hold_left = FIRST_HOLD;
    float trial_ms = 0.0f, prev_ms = 0.0f, best_two_ms = 0.0f;
    uint32_t best_frac16 = 16384;
    bool inc_two = false;        // the setting being held (the incumbent of the next re-check)
    int cand = 0, cand_best = 0; // re-check: candidate on trial / best so far
    float cand_best_ms = 0.0f;
    bool cand_two = false;
    uint32_t cand_frac16 = 16384;
    bool debug = false;
    const void *tag = nullptr;

    // a frame time has become available (ms of a frame issued under trial `of_trial`; counts = not the trial's first frame)
    void observe(uint32_t of_trial, bool counts, float ms) {
        if (!counts || of_trial != trial) return;
        trial_ms = (trial_obs == 0u || ms < trial_ms) ? ms : trial_ms;
        ++trial_obs;
    }

    // once per eligible frame, after the observations.  Returns true = this frame runs in two rounds (with frac16);
    // *wants_timing: the frame should be timed if a slot is free; *counts: its time will count (observe's argument).
    bool begin_frame(bool *wants_timing, bool *counts) {
        if (phase == HOLD) {
            if (hold_left == 0u || --hold_left == 0u) {  // look again: the scene or the camera may have moved on
                phase = RECHECK; cand_best_ms = 0.0f; cand_best = 0;
                set_candidate(0);
            }
        } else if (trial_obs >= 2u) {
            conclude(true);
        } else if (phase == RECHECK && trial_frames >= RECHECK_MAX_FRAMES) {
            // a re-check never keeps a candidate for long: where the host runs far ahead of the GPU the times arrive too late
            conclude(false);
        }
        *wants_timing = phase != HOLD;
        *counts = trial_frames >= 1u;  // (the first frame of a trial still runs on the previous setting's history)
        ++trial_frames;
        return two;
    }

  private:
    void new_trial() {
        ++trial;
        trial_frames = 0; trial_obs = 0; trial_ms = 0.0f;
    }

    // re-check: the incumbent and a few alternatives get a short trial each; candidate 0 is the incumbent.  Holding one
    // round: two rounds with a quarter and with a twenty-fifth of the splats in round A (a scene that has become dense
    // shows in either).  Holding two rounds: one round, and the fraction's two neighbours.
    int last_candidate() const { return inc_two ? 3 : 2; }

    void set_candidate(int k) {
        cand = k;
        const uint32_t f = best_frac16;
        if (!inc_two) {
            two = k != 0;
            frac16 = k == 2 ? 2621u : 16384u;  // 0.04, 0.25
        } else if (k == 0) { two = true; frac16 = f; }
        else if (k == 1) { two = false; frac16 = f; }
        else if (k == 2) { two = true; frac16 = (uint32_t)std::min<uint64_t>(FRAC_MAX, (uint64_t)f * 5u / 4u); }
        else { two = true; frac16 = (uint32_t)std::max<uint64_t>(FRAC_MIN, (uint64_t)f * 4u / 5u); }
        new_trial();
    }

    void hold(bool hold_two, uint32_t f) {
        two = inc_two = hold_two;
        frac16 = best_frac16 = f;
        phase = HOLD; hold_left = HOLD_FRAMES;
        new_trial();
    }

    void conclude(bool measured) {
        const float ms = trial_ms;
        if (debug)
            fprintf(stderr, "[rounds] ctx %p trial %u phase %d cand %d %s frac %.4f -> %.4f ms (%u obs)\n", tag, trial, phase, cand,
                    two ? "two" : "one", frac16 / 65536.0, ms, trial_obs);
        if (phase == CLIMB) {
            if (best_two_ms == 0.0f || ms < best_two_ms) { best_two_ms = ms; best_frac16 = frac16; }
            if (prev_ms != 0.0f && ms > prev_ms) { dir = -dir; ++reversals; }
            prev_ms = ms;
            uint64_t f = frac16;
            f = dir < 0 ? f * 4u / 5u : f * 5u / 4u;
            if (f < FRAC_MIN) { f = FRAC_MIN; dir = 1; ++reversals; }
            if (f > FRAC_MAX) { f = FRAC_MAX; dir = -1; ++reversals; }
            frac16 = (uint32_t)f;
            if (reversals >= 3u) {  // the minimum is bracketed: now the other candidate, one round
                frac16 = best_frac16;
                phase = TRY_ONE; two = false;
            }
            new_trial();
        } else if (phase == TRY_ONE) {
            hold(best_two_ms < 0.97f * ms, best_frac16);  // (a tie goes to the simpler frame)
        } else if (phase == RECHECK) {
            if (measured && (cand == 0 || cand_best_ms == 0.0f || ms < (cand_best == 0 ? 0.97f : 1.0f) * cand_best_ms)) {
                if (cand == 0 || cand_best_ms != 0.0f) {  // (no incumbent time: nothing to compare with)
                    cand_best_ms = ms; cand_best = cand;
                    cand_two = two; cand_frac16 = frac16;
                }
            }
            if (cand < last_candidate() && (cand > 0 || measured)) {
                set_candidate(cand + 1);
            } else if (cand_best_ms != 0.0f && cand_two && !inc_two) {
                // one round was held and two rounds won: climb from the fraction that won before holding anything
                phase = CLIMB; two = true; frac16 = cand_frac16;
                dir = -1; reversals = 0; prev_ms = 0.0f;
                best_two_ms = 0.0f; best_frac16 = cand_frac16;
                new_trial();
            } else if (cand_best_ms != 0.0f) {
                hold(cand_two, cand_two ? cand_frac16 : best_frac16);
            } else {
                hold(inc_two, best_frac16);
            }
        }
    }
};

}  // namespace gsplat
