// bisbm_pass_policy.hpp -- which depth of pass the next sweep launch of the production kernel runs (host only: no HIP in
// here, so tests/test_pass_policy.py drives it on a CPU with made-up timings).
//
// The production kernel exists once per pass depth (two / four / eight steps of a chain evaluated per pass, DESIGN.md
// section 6); the chain is the same chain whatever runs, so the choice is purely a matter of speed, and the speed of a
// depth depends on the graph, the partition and where the chain is -- it is MEASURED: bisbm_anneal times every launch
// (HIP events), tells this class (depth, updates per ms, accepted fraction) and asks it for the depth of the next one.
//
//   depth 1 = two steps per pass, 2 = four, 3 = eight; max_depth = the deepest the shape allows (0 / 1: nothing to choose).
//
// Rules (round 4; the round-3 selector re-measured every depth twice whenever the accepted fraction had moved by 0.1 and
// flapped between depths that measure within noise):
//   * start-up (after a new partition: init / shuffle / merge / split): the preferred depth -- the deepest on graphs of up
//     to 10^5 nodes, the shallowest on larger ones, where each won in every regime measured -- twice (the first launches
//     of a process run up to 20 % slow: a depth's figure is the BEST of its last two launches), then its neighbour twice
//     (once if it comes out more than 25 % behind), on to the next neighbour only while each step gains;
//   * steady state: the incumbent runs.  One launch in kProbeEvery looks at a neighbouring depth (one side, then the
//     other); sooner -- but never within kProbeGap launches of the last look -- when the accepted fraction has moved by
//     more than kRegime since that neighbour was last measured in the direction that favours it: deep passes pay where
//     few steps move, so a falling accepted fraction sends the look to the deeper neighbour, a rising one to the
//     shallower;
//   * hysteresis: the incumbent is replaced only by a look that came out more than kSwitchGain ahead of the incumbent's
//     figure (best of its last two launches).  Outliers are to the slow side, so a slow look changes nothing and a
//     fast one is real.
#pragma once

#include <cmath>
#include <cstdint>

namespace bisbm {

class PassDepthPolicy {
public:
    static constexpr double kSwitchGain = 1.03;  // a look must be this much faster than the incumbent to replace it
    static constexpr uint32_t kProbeEvery = 16;  // steady regime: one launch in 16 looks at a neighbour
    static constexpr uint32_t kProbeGap = 4;     // ... and never two looks within 4 launches, whatever the regime does
    static constexpr double kRegime = 0.1;       // accepted fraction: how far is "another regime"
    static constexpr uint32_t kMaxDepth = 3;

    // a new partition was put in place from outside: everything measured so far belongs to another chain state
    void reset() { *this = PassDepthPolicy(); }

    // depth of the next launch.  small_graph: n <= 10^5 (the deepest pass is the preferred one there)
    uint32_t choose(uint32_t max_depth, bool small_graph) {
        probing_ = 0;
        if (max_depth < 2u) return max_depth;
        if (max_depth > kMaxDepth) max_depth = kMaxDepth;
        if (cur_ > max_depth) cur_ = 0;  // (the shape changed under us without a reset: start over)
        if (max_depth != max_depth_) {
            max_depth_ = max_depth;
            if (!settled_) cur_ = 0;
        }
        if (cur_ == 0u) {  // first launch after a reset
            cur_ = small_graph ? max_depth : 1u;
            dir_ = small_graph ? -1 : +1;
            settled_ = false;
        }
        if (!settled_) return startup();
        // steady state
        const uint32_t since = launches_ - last_probe_at_;
        if (since < kProbeGap) return cur_;
        const uint32_t up = cur_ < max_depth_ ? cur_ + 1 : 0u, down = cur_ > 1u ? cur_ - 1 : 0u;
        uint32_t cand = 0;
        // the regime moved in the direction that favours a neighbour?
        if (up && last_acc_ >= 0 && s_[up][0].speed > 0 && s_[up][0].acc - last_acc_ > kRegime) cand = up;
        if (!cand && down && last_acc_ >= 0 && s_[down][0].speed > 0 && last_acc_ - s_[down][0].acc > kRegime) cand = down;
        if (!cand && since >= kProbeEvery) {  // the periodic look: one side, then the other
            probe_up_ = !probe_up_;
            cand = (probe_up_ && up) ? up : (down ? down : up);
        }
        if (!cand) return cur_;
        probing_ = cand;
        return cand;
    }

    // what a launch at `depth` measured: updates per ms and the fraction of its steps that were accepted
    void record(uint32_t depth, double speed, double acc) {
        if (depth < 1u || depth > kMaxDepth || !(speed > 0)) return;
        s_[depth][1] = s_[depth][0];
        s_[depth][0] = Sample{speed, acc};
        n_[depth] += 1;
        last_acc_ = acc;
        launches_ += 1;
        if (settled_ && probing_ == depth && depth != cur_) {
            last_probe_at_ = launches_;
            looks_ += 1;
            if (speed > kSwitchGain * figure(cur_)) {  // hysteresis: only a clear win replaces the incumbent
                probe_up_ = !(depth > cur_);           // (the next periodic look goes on in the same direction: choose() flips first)
                cur_ = depth;
                switches_ += 1;
            }
        }
        probing_ = 0;
    }

    // ---- diagnostics (BISBM_PASS_LOG, tests) ----
    uint32_t current() const { return cur_; }
    bool settled() const { return settled_; }
    uint32_t switches() const { return switches_; }
    uint32_t looks() const { return looks_; }
    uint32_t launches() const { return launches_; }
    // a depth's figure: the best of its last two launches that belong to the present regime (0: none)
    double figure(uint32_t d) const {
        if (d < 1u || d > kMaxDepth) return 0;
        double f = 0;
        for (int i = 0; i < 2; ++i)
            if (s_[d][i].speed > f && in_regime(s_[d][i])) f = s_[d][i].speed;
        return f;
    }

private:
    struct Sample {
        double speed = 0, acc = -1;
    };
    bool in_regime(const Sample& s) const { return s.speed > 0 && !(last_acc_ >= 0 && std::fabs(last_acc_ - s.acc) > kRegime); }
    // measured often enough to be believed: twice, or once if it came out more than 25 % behind a depth measured twice
    bool trusted(uint32_t d) const {
        if (n_[d] >= 2u) return true;
        if (n_[d] == 0u) return false;
        double twice = 0;
        for (uint32_t e = 1; e <= kMaxDepth; ++e)
            if (n_[e] >= 2u && figure(e) > twice) twice = figure(e);
        return s_[d][0].speed < 0.75 * twice;
    }
    uint32_t startup() {
        if (!trusted(cur_)) return cur_;
        // walk from the preferred depth towards its neighbour while each step gains
        for (;;) {
            const int next = (int)cur_ + dir_;
            if (next < 1 || next > (int)max_depth_) break;
            if (!trusted((uint32_t)next)) return (uint32_t)next;
            if (!(best_of_two((uint32_t)next) > kSwitchGain * best_of_two(cur_))) break;
            cur_ = (uint32_t)next;
        }
        settled_ = true;
        last_probe_at_ = launches_;
        probe_up_ = dir_ < 0;  // (the first periodic look goes back towards where the walk came from: choose() flips first)
        return cur_;
    }
    double best_of_two(uint32_t d) const { return s_[d][0].speed > s_[d][1].speed ? s_[d][0].speed : s_[d][1].speed; }

    Sample s_[kMaxDepth + 1][2];
    uint32_t n_[kMaxDepth + 1] = {0, 0, 0, 0};
    uint32_t cur_ = 0, max_depth_ = 0, probing_ = 0;
    int dir_ = +1;
    bool settled_ = false, probe_up_ = false;
    double last_acc_ = -1;
    uint32_t launches_ = 0, last_probe_at_ = 0, switches_ = 0, looks_ = 0;
};

}  // namespace bisbm
