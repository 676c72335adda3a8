// sgo_amg.h -- rigid-body smoothed-aggregation multigrid preconditioner for the block-CSR
// Gauss-Newton Hessian.  See sgo_amg.hip for the algorithm and DESIGN.md section 5.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "sgo_internal.h"

namespace sgo {

struct AmgConfig {
  double theta = 0.02;     // strength-of-connection threshold on level 0
  double theta_coarse = 0.02;  // ... on the coarser levels
  double theta_scale = 1.0;    // both thresholds are multiplied by this (the caller halves it when a hierarchy's first solve stalls)
  double omega = 0.8;      // block-Jacobi damping
  int max_levels = 10;
  int nu_coarse = 1;           // smoothing sweeps on the coarser V-cycle levels; amg_create picks 2 for
                               // graphs with >= 6 * 10^5 level-0 blocks, where a coarse sweep is cheap next to level 0
  bool smooth = true;          // smoothed aggregation: P = (I - omega_p D^-1 A) T (env SGO_AMG_SMOOTH=0: tentative P)
  double omega_p = 0.66;       // damping of the prolongator smoothing step
  double theta_filter = 1e-3;       // ... "strong" for that filter: w_ij >= theta_filter sqrt(w_ii w_jj) -- far below the aggregation's
                                    // threshold: only connections that are NEGLIGIBLE are dropped (closures DCS has switched off),
                                    // not merely the ones the aggregation does not follow (env SGO_AMG_THETA_FILTER)
  bool filtered_smoothing = true;   // a level whose transfer smoothed with the whole operator would be too dense tries the operator of
                                    // the strong connections before it falls back to the tentative transfer (SaHost::filtered,
                                    // sgo_amg_host.h; env SGO_AMG_FILTER=0: straight to the tentative one, as before round 5)
  bool fold = true;             // folded V-cycle (sgo_amg.hip): the post- and pre-smoothing sweeps of the smoothed levels folded into
                                // the transfer operator P~ = (I - omega D^-1 A) P; one sweep per level (env SGO_AMG_FOLD=0: the
                                // sweeps as launches of their own, nu_coarse as below)
  int fold0_rows = 60000;       // ... level 0 too on graphs of at most this many rows (single GPU; measured, scripts/fold0_sweep.py:
                                // 50k rows / 500k edges 2.84 against 2.94 ms per GN iteration, C4's 100k rows 4.62 against 4.54)
  bool lists_on_device = true;  // the product lists of A P and P^T A P are made on the device from the host's patterns
                                // (env SGO_AMG_LISTS=host: on the host, the reference the device lists are tested against)
  int coarsest_nodes = 400;  // stop coarsening at or below this many nodes; that level is inverted densely
                             // (blocked Gauss-Jordan over 3x that many unknowns) once per GN iteration
  // A rebuild inside sgo_optimize_gn may KEEP the aggregates of the hierarchy it replaces (the partition of every level's nodes)
  // and re-make everything that depends on the values -- the filter's mask, the patterns of P, A P and P^T A P, the lists --:
  // level l's aggregate of node i (renumbered form) and the visiting order handed to level l + 1.  nullptr: aggregate anew.
  const struct AmgKeptAgg* keep_agg = nullptr;
};
struct AmgKeptAgg {
  std::vector<std::vector<int>> agg, visit_c;   // per coarsened level
  std::vector<int> nc;
};

// profiling hook supplied by the context (brackets a launch with HIP events when enabled)
struct AmgProf {
  void* user = nullptr;
  void (*begin)(void* user, int kid, double bytes) = nullptr;
  void (*end)(void* user) = nullptr;
};

// Multi-GPU, row-owner mode: the partition (device and host image) and a way to have the exchange buffers grown
struct AmgHalo {
  HaloDev* dev = nullptr;            // amg_create fills in pemax / pent (entries of P in the boundary rows)
  int G = 1, bmax = 0;
  const int* bnd_host = nullptr;     // [G][bmax] boundary rows of every rank, -1 padded
  bool (*reserve)(void* user, size_t packet_doubles) = nullptr;
  void* user = nullptr;
  const std::vector<double>* w0 = nullptr;   // strength weights of level 0's logical slots (the caller made them from the edge
                                             // list: a rank holds current blocks for its own rows only)
};

struct ChunkArena;
struct DevArena;   // device memory of the hierarchy: owned by the caller, rewound by it after amg_destroy
struct Amg;  // opaque

// logical structure of a level's operator on the host (slot list sorted by row, diagonal slot first)
struct HostLevel {
  int n = 0, nslot = 0;
  std::vector<int> rowptr, row, col;
  std::vector<int> visit;   // order in which the greedy aggregation visits the nodes (empty: 0, 1, 2, ...).  Level 0
                            // is visited along the trajectory (ascending vertex id) although its rows are numbered
                            // along a Hilbert curve: chains of odometry edges then pair up regularly
};

// Build the hierarchy for the level-0 matrix: S0 is its symmetric storage (the cycle's level-0 products
// run on it), A0 its logical view (set-up kernels) and H0 the same structure on the host.  The values
// (S0.ublk, dblk, dinv) must hold the linearisation at the initial poses (they provide the strength of
// connection).  `d_poses` / `d_free_id` give the positions of the level-0 nodes: pos of row h =
// poses[3*free_id[h] + 0..1].  `scratch` (optional) provides host memory for the set-up's large temporary
// lists; it is rewound here and may be rewound again by the caller once amg_create has returned.
Amg* amg_create(hipStream_t s, const BsrDev& A0, const Sym0Dev& S0, const Tile0Dev& T0, const HostLevel& H0, const double* d_poses,
                const int* d_free_id, const AmgConfig& cfg, const AmgProf& prof, std::string* err,
                ChunkArena* scratch, DevArena* arena, struct AmgHostL0* pre0 = nullptr, const AmgHalo* halo = nullptr);
// The same hierarchy set up entirely ON THE DEVICE (sgo_amg_dev.inc; single GPU): parallel aggregation, patterns by sort / scan /
// compress passes.  What a rebuild inside sgo_optimize_gn uses (build_amg, sgo_solve.cpp).
// aggregate_on_device = false (default of the callers): the aggregation stays the host's greedy walk along the trajectory (its
// aggregates are the better ones, and it is a fifth of the host set-up's time); everything else on the device.
Amg* amg_create_dev(hipStream_t s, const BsrDev& A0, const Sym0Dev& S0, const Tile0Dev& T0, const HostLevel& H0, const double* d_poses,
                    const int* d_free_id, const AmgConfig& cfg, const AmgProf& prof, std::string* err, ChunkArena* scratch, DevArena* arena,
                    DevArena* tmp_arena, bool aggregate_on_device, const struct AmgHostL0* pre0);
// Level 0's host analysis (aggregation, patterns and product lists of the transfer, structure of level 1) made ahead
// of amg_create from the level's logical structure and the strength weights w (Frobenius norms of the slots' blocks
// at the initial poses, logical slot order): amg_host_l0_run may execute on a helper thread while the caller still
// builds the level-0 storage; amg_create(..., pre0) then starts from its result.  `scratch` belongs to the run until
// amg_create has returned.
struct AmgHostL0;
AmgHostL0* amg_host_l0_new();
// agg_only: the aggregation alone (the patterns are then made on the device: amg_create_dev(..., pre0))
void amg_host_l0_run(AmgHostL0* p, const HostLevel& H0, const std::vector<double>& w, const AmgConfig& cfg, ChunkArena* scratch, bool agg_only = false);
bool amg_host_l0_agg_only(const AmgHostL0* p);
void amg_host_l0_free(AmgHostL0* p);
bool amg_host_l0_ready(const AmgHostL0* p);
AmgConfig amg_effective_config(const AmgConfig& cfg_in, int n, int nslot);
void amg_destroy(Amg* m);
// Recompute the coarse operators for the current level-0 values and poses (once per GN iteration).
int amg_update(Amg* m, hipStream_t s, std::string* err);
// z = M^-1 r (one cycle).  If dotvec != nullptr, partials[0..nparts) receive the per-block
// partial sums of dotvec . z; returns nparts (the grid of the last kernel).
// dotvec2 (optional) adds partials[kMaxPartials + ..] = dotvec2 . z.
// xs0_ready: 1 = the producer of r has already left omega Dinv r in amg_xs0() (k_finalize, k_update_xr) --
// the cycle's first level-0 smoothing sweep from zero --, 0 = the cycle computes it first; 2 (multi-GPU row-owner
// mode) = as 1, and the copies of the neighbours' boundary rows are current too (no exchange at the cycle's entry).
int amg_apply(Amg* m, hipStream_t s, const double* r, double* z, const double* dotvec, double* partials,
              const PcgScalars* S, const double* dotvec2 = nullptr, int xs0_ready = 0);
// Multi-GPU: the cycle's two level-0 products evaluate the work units [u0, u1) only (this rank's tiles = the rows
// [row0, row1)); the residual pass stays a per-rank partial that the restriction folds into a partial coarse
// right-hand side, which is all-reduced over `comm` (3 n_c doubles); the post-smoothing pass's result is all-reduced
// as a whole (3 n doubles).  Everything else of the cycle runs replicated.
struct Comm;
void amg_set_shard(Amg* m, Comm* comm, int u0, int u1, int row0, int row1, const HaloDev* slices = nullptr);   // slices: all-gather the product vectors by rank slices
bool amg_comm_failed(const Amg* m);
int amg_debug_coarse_rhs(Amg* m, hipStream_t s, const double* r, double* out_dev, int cap3);   // test hook, see sgo_amg.hip
double* amg_xs0(Amg* m);     // [n][3] on the device; nullptr for a single-level (dense) hierarchy
double amg_omega(const Amg* m);
// true when the last amg_update met a non-positive pivot in the coarsest operator (synchronises `s`)
bool amg_coarsest_not_spd(Amg* m, hipStream_t s);
int amg_num_levels(const Amg* m);
bool amg_has_filtered(const Amg* m);   // some level's transfer is smoothed with the filtered operator (SaHost::filtered)
long long amg_level0_bytes(const Amg* m);   // device bytes of the level-0 transfer (P, A P, product lists) this rank holds
void amg_describe(const Amg* m, std::string* out);
void amg_kept_aggregates(const Amg* m, AmgKeptAgg* out);   // host copies of the hierarchy's aggregates (AmgConfig::keep_agg)

}  // namespace sgo
