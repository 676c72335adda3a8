// Unit test of sgo_optimize_gn's decision functions (sparse_gslam_amd/csrc/sgo_rules.h) on RECORDED iteration-count sequences:
// pure host code, no GPU, no library.  Every rank of a multi-GPU run feeds these functions the same numbers (counts and sums that
// are bit-identical on all ranks) and must get the same decision -- the functions read nothing else.  The sequences are the ones
// NOTES.md / profiles/ record; the expected decisions were worked out by hand from DESIGN.md section 5's statement of each rule.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sgo_rules.h"

using namespace sgo::rules;

static int fails = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++fails;                                                             \
    }                                                                      \
  } while (0)

// The count rules as optimize_gn drives them over one call: returns the iterations after whose solve a rebuild is decided, feeding
// the counts of a recorded call (a rebuild resets the call's best, as in the driver).
static std::vector<int> rebuild_points(const std::vector<int>& counts, int iters) {
  std::vector<int> at;
  int call_best = 0, rebuilds = 0;
  const int max_rb = max_rebuilds(iters);
  bool pending = false;
  for (int it = 0; it < (int)counts.size(); ++it) {
    if (pending) {
      ++rebuilds;
      call_best = 0;
      pending = false;
    }
    const int eq = counts[it];
    if (call_best == 0 || eq < call_best) call_best = eq;
    if (rebuilds < max_rb && staleness(eq, call_best, iters - it - 1).rebuild()) {
      pending = true;
      at.push_back(it);
    }
  }
  return at;
}

int main() {
  // ---- counts at equal tolerance (pcg_tol_cap: later solves of a call stop at the first one's absolute accuracy)
  CHECK(equal_tolerance_count(20, 1e-8, 1e-6) == 27);    // 20 * 8 / 6
  CHECK(equal_tolerance_count(20, 1e-8, 1e-8) == 20);
  CHECK(equal_tolerance_count(20, 1e-8, 1e-9) == 20);    // a tighter solve is not scaled
  CHECK(equal_tolerance_count(31, 1e-8, 1.0) == 31);     // (tolk = 1: nothing to scale by)
  CHECK(equal_tolerance_count(0, 1e-8, 1e-6) == 0);

  // ---- caps
  CHECK(max_rebuilds(20) == 7 && max_rebuilds(5) == 3 && max_rebuilds(1) == 3 && max_rebuilds(100) == 34);
  CHECK(bail_out_cap(22) == 128);
  CHECK(lag_cap(0) == 0 && lag_cap(25) == 28);
  CHECK(probe_iteration(30) == 10 && probe_iteration(9) == 4);

  // ---- staleness by counts.  C4, bench start (profiles/r05_bench_c4.json): 30 in the first solve, 19-24 afterwards: never stale
  {
    const std::vector<int> c4 = {30, 27, 25, 24, 24, 23, 23, 22, 22, 21, 21, 21, 20, 20, 20, 20, 19, 19, 19, 19};
    CHECK(rebuild_points(c4, 20).empty());
  }
  // C5 re-optimised from its initial poses (NOTES.md section 26): 88 in the first solve, 31-34 at the end of the call BEFORE --
  // the comparison is with the best of THIS call, so the first solve cannot trip the "doubled" rule (88 > 2 * 34 + 10 did)
  {
    const std::vector<int> c5 = {88, 70, 58, 50, 45, 41, 39, 37, 36, 35, 34, 34, 33, 33, 32, 32, 31, 31, 31, 31};
    CHECK(rebuild_points(c5, 20).empty());
    CHECK(staleness(88, 34, 19).doubled);   // (what the carried-over best made of it)
  }
  // C4 from the dead-reckoned start, first solves of profiles/r05_bench_c4_steps20.json: 56, then 121 -- not doubled (121 <= 122)
  // but 65 iterations over the best with 18 iterations left pays; the rebuilt hierarchy's first solve (64) sets a new best,
  // 221 > 2 * 64 + 10 is "doubled"
  {
    const Staleness s1 = staleness(121, 56, 18);
    CHECK(!s1.doubled && s1.pays);
    const Staleness s2 = staleness(221, 64, 16);
    CHECK(s2.doubled);
    const std::vector<int> odom = {56, 121, 64, 221, 71, 86, 181, 62, 74, 88};
    const std::vector<int> at = rebuild_points(odom, 20);
    CHECK(at.size() == 4 && at[0] == 1 && at[1] == 3 && at[2] == 6 && at[3] == 9);   // (88 against 62 with ten iterations left pays too)
  }
  // "pays" needs iterations left: the same excess in the call's last iteration is not worth a set-up
  CHECK(!staleness(121, 56, 0).pays && !staleness(121, 56, 2).pays && staleness(121, 56, 3).pays);
  // a quarter above the best is the threshold: 4 * 70 = 280 = 5 * 56 is not above it
  CHECK(!staleness(70, 56, 19).pays && staleness(71, 56, 19).pays);
  // the cap on rebuilds per call holds whatever the counts do
  {
    std::vector<int> wild;
    for (int k = 0; k < 40; ++k) wild.push_back(k % 2 ? 400 : 20);
    CHECK((int)rebuild_points(wild, 40).size() <= max_rebuilds(40));
  }

  // ---- lagged refresh: the learned sensitivity
  CHECK(lag_allowed(0.006, kLagSlopeStart) == 4.0 / 2700.0);       // the cautious start: 0.15 % of movement
  CHECK(lag_allowed(0.006, 100.0) == 0.006);                        // never more than tau
  CHECK(lag_allowed(0.006, 0.0) == 0.006);
  {
    // C4 (NOTES.md section 20): 0.3 % of movement cost 2 iterations -> slope 667: tau binds (4 / 667 = 0.006)
    double s = lag_slope_after_kept(kLagSlopeStart, false, 2, 0.003);
    CHECK(s > 666.0 && s < 667.0 && lag_allowed(0.006, s) > 0.00599);
    // 50 k / 250 k: 0.2 % cost 14 -> 7000: only 0.057 % may be kept over; a cheaper observation lowers the slope by 20 % at most
    s = lag_slope_after_kept(s, true, 14, 0.002);
    CHECK(s == 7000.0 && lag_allowed(0.006, s) < 0.0006);
    s = lag_slope_after_kept(s, true, 1, 0.002);
    CHECK(s == 5600.0);
    // an interruption at next to no movement must not lock every later solve out: bounded, and fresh solves let it decay
    s = lag_slope_after_interrupt(s, 1e-5);
    CHECK(s == kLagSlopeMax && lag_allowed(0.006, s) == 1e-4);
    int fresh = 0;
    while (s > kLagSlopeStart && fresh < 1000) {
      s = lag_slope_after_fresh(s);
      ++fresh;
    }
    CHECK(fresh == 53 && s == kLagSlopeStart);                      // ln(40000 / 2700) / ln(1 / 0.95) = 52.6
    CHECK(lag_slope_after_fresh(667.0) == 667.0);                   // a LOW slope is not raised by fresh solves
  }
  CHECK(!kept_solve_too_slow(33, 20) && kept_solve_too_slow(34, 20));   // 20 + 8 + 5

  // ---- the aggregation's staleness across calls and the trial (NOTES.md sections 22-23)
  CHECK(moved_far(21.0, 100.0, 0.0, 100000));           // blocks 21 % away (the growth session of tests/test_gpu_lagged_refresh.py)
  CHECK(!moved_far(4.0, 100.0, 900.0, 100000));         // 4 %, 0.9 % of the rows by a quarter: not far
  CHECK(moved_far(4.0, 100.0, 1001.0, 100000));
  CHECK(!moved_far(1.0, 0.0, 0.0, 10));                 // (no blocks: nothing moved)
  CHECK(reaggregate(true, false, 20, 19, 24, 0, 7, false));    // first solve 24 against the aggregation's best 19
  CHECK(!reaggregate(true, false, 20, 19, 20, 0, 7, false));   // within 10 %
  CHECK(!reaggregate(true, true, 20, 19, 24, 0, 7, false));    // a trial was lost on this graph: off
  CHECK(!reaggregate(true, false, 4, 19, 24, 0, 7, false));    // too few iterations left to pay
  CHECK(!reaggregate(true, false, 20, 0, 24, 0, 7, false));    // the aggregation has no record yet
  CHECK(!reaggregate(true, false, 20, 19, 24, 7, 7, false));   // the call's rebuilds are used up
  CHECK(!reaggregate(true, false, 20, 19, 24, 0, 7, true));    // a rebuild is pending anyway
  CHECK(!trial_reverts(22, 33));    // the C4-sized session: 22 / 24 against 33: kept
  CHECK(trial_reverts(38, 27));     // 40 k / 60 k: 42 / 38 against 27: the old one comes back
  CHECK(trial_reverts(24, 27));     // not under 0.85: back as well

  if (fails) {
    std::fprintf(stderr, "%d check(s) failed\n", fails);
    return 1;
  }
  std::printf("rules ok\n");
  return 0;
}
