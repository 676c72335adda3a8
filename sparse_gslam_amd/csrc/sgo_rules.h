// sgo_rules.h -- the decisions sgo_optimize_gn takes about its multigrid hierarchy, as PURE functions of iteration counts and of
// sums every rank holds bit-identically: no clocks, no device state, no environment.  Every rank of a multi-GPU run must take the
// same decision from the same numbers, so the rules live here, apart from the driver that feeds them (optimize_gn, sgo_solve.cpp),
// and are unit-tested on recorded count sequences without a GPU (tests/cpp/rules_unit.cpp, tests/test_rules.py).
// DESIGN.md section 5 says what each rule is for and where its constants were measured.
#pragma once
#include <algorithm>
#include <cmath>

namespace sgo {
namespace rules {

// Iteration counts are compared at EQUAL tolerance: a solve that stopped at the absolute criterion (a looser relative tolerance
// tolk > tol0, pcg_tol_cap) is scaled to what tol0 would have cost -- PCG converges linearly, iterations ~ log(1 / tolerance).
inline int equal_tolerance_count(int iter, double tol0, double tolk) {
  return (tolk > tol0 && tolk < 1.0 && tol0 > 0.0) ? (int)std::lround(iter * std::log(tol0) / std::log(tolk)) : iter;
}

// How many set-ups one call may redo: three for a short call, one per three Gauss-Newton iterations for a long one.
inline int max_rebuilds(int iters) { return std::max(3, (iters + 2) / 3); }

// The bail-out cap of a solve behind a hierarchy whose best count is `best` (0: the hierarchy has not solved anything yet).
inline int bail_out_cap(int best) { return 4 * best + 40; }

// Staleness of the hierarchy by counts (fresh solves only, against the best count of the CURRENT call): rebuild when the count has
// more than doubled, or is > 25 % above the best while the iterations saved over the rest of the call exceed a set-up's worth
// (~150 PCG iterations).
struct Staleness {
  bool doubled, pays;
  bool rebuild() const { return doubled || pays; }
};
constexpr int kRebuildCostIters = 150;   // what a set-up is worth in PCG iterations (swept: NOTES.md sections 27, 30)
inline Staleness staleness(int eq_iter, int call_best, int iterations_left, int cost_iters = kRebuildCostIters) {
  Staleness s;
  s.doubled = eq_iter > 2 * call_best + 10;
  s.pays = 4 * eq_iter > 5 * call_best && (long long)(eq_iter - call_best) * iterations_left > cost_iters;
  return s;
}

// Lagged refresh: the movement of the level-0 diagonal blocks (relative, summed over the rows) up to which a solve keeps the coarse
// operators of the solve before -- what costs this graph's solves four PCG iterations by the learned slope, at most tau.
inline double lag_allowed(double tau, double slope) { return std::min(tau, 4.0 / std::max(slope, 1.0)); }
constexpr double kLagSlopeStart = 2700.0;    // 0.15 % of movement: the cautious start on a graph not seen before
constexpr double kLagSlopeMax = 40000.0;     // no solve is locked out over more than 0.01 % of movement
// ... the slope after a KEPT solve that took `excess` iterations more than the last fresh one over a movement of `moved`:
// the first observation replaces the start value, later ones raise the slope at once and lower it by 20 % per solve.
inline double lag_slope_after_kept(double slope, bool seen, int excess, double moved) {
  const double obs = std::min(kLagSlopeMax, std::max(0.5, (double)excess) / std::max(moved, 1e-5));
  return seen ? std::max(obs, 0.8 * slope) : obs;
}
// ... after a kept solve that had to be INTERRUPTED (progress probe or cap): counted as sixteen iterations over.
inline double lag_slope_after_interrupt(double slope, double moved) {
  return std::min(std::max(slope, 16.0 / std::max(moved, 1e-6)), kLagSlopeMax);
}
// ... after a FRESH solve: a high slope decays by 5 % towards the start value (it is re-examined in time).
inline double lag_slope_after_fresh(double slope) { return slope > kLagSlopeStart ? std::max(kLagSlopeStart, 0.95 * slope) : slope; }
// A kept solve that cost more than a refresh is worth makes the next solve refresh whatever the movement says.
inline bool kept_solve_too_slow(int eq_iter, int fresh_pcg) { return eq_iter > fresh_pcg + 8 + fresh_pcg / 4; }
// The iteration cap of a solve behind kept operators (0: none).
inline int lag_cap(int fresh_pcg) { return fresh_pcg > 0 ? fresh_pcg + 3 : 0; }
// The iteration at which the progress probe of the solves that keep these operators looks.
inline int probe_iteration(int fresh_iter) { return std::max(4, fresh_iter / 3); }

// The aggregation's own staleness, across calls: blocks moved far since the hierarchy was AGGREGATED (sum of ||D - D_ref||_F against
// sum ||D||_F over the rows; rows that moved by a quarter) AND the call's first solve visibly above that aggregation's best count.
inline bool moved_far(double sum_diff, double sum_norm, double rows_quarter, int n) {
  return sum_norm > 0.0 && (sum_diff > 0.05 * sum_norm || rows_quarter > 0.01 * (double)n);
}
inline bool reaggregate(bool far, bool rule_off, int iterations_left_incl, int agg_best, int eq_iter, int rebuilds, int max_rb, bool rebuild_pending) {
  return far && !rule_off && iterations_left_incl >= 5 && agg_best > 0 && 10 * eq_iter > 11 * agg_best && rebuilds < max_rb && !rebuild_pending;
}
// The trial's verdict: the re-made hierarchy's better count of its first two solves against the old one's first solve of the call;
// true = the old hierarchy comes back.
inline bool trial_reverts(int trial_best, int trial_old) { return 100 * trial_best > 85 * trial_old; }

}  // namespace rules
}  // namespace sgo
