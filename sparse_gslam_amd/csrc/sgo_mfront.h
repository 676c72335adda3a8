// sgo_mfront.h -- the mid-size path: optimize(iters) through a MULTIFRONTAL sparse Cholesky factorisation spread over the
// chip, for graphs between the single-launch direct path (sgo_direct.h: <= 60 separator poses, one workgroup) and the
// multigrid PCG (whose iteration is eight or more launch-floor-bound kernels below ~10^5 rows: DESIGN.md section 7,
// "mid-size regime").  This is the regime of the reference's own largest graphs -- a few thousand keyframe poses with a
// closure every few poses (mit-killian: 5 489 poses / 7 629 edges), re-optimised after every accepted closure
// (src/sparse_gslam/src/submap_loop_closer.cpp:286-287) through g2o's sparse Cholesky (src/sparse_gslam/src/graphs.cpp:19).
//
//   * Host, once per sgo_set_graph_se2 (the counterpart of g2o's symbolic analysis): nested dissection of the free poses --
//     rows in Hilbert order of the initial positions (or in id order, whichever gives the cheaper tree), index bisection,
//     separator = a MINIMUM vertex cover of the cut (Koenig's theorem on the bipartite cut graph), leaves of <= 32 poses --,
//     the fronts' row structures, the assembly lists (every edge goes to the front of its first-eliminated endpoint) and the
//     extend-add maps, the fronts of equal height gathered into levels.
//   * Device, per Gauss-Newton iteration: two launches per LEVEL of the tree for the factorisation -- a parent's matrix is
//     gathered tile by tile from its children by many workgroups, the children's Schur complements formed on the fly on the
//     fp64 matrix cores (v_mfma_f64_16x16x4_f64: no update matrix is stored); then one workgroup per front adds the edges'
//     terms and factorises the own columns left-looking in 16-column panels, with the right-hand side as the front's last
//     ROW, so the forward substitution rides along --, one launch per level for the backward substitution, one for the edges
//     (error, Jacobians, 6x6 element and chi2), one for the update.  No atomics on floating-point data; every sum has a fixed
//     order.
// A graph qualifies when the critical path of its tree (the largest front of every level) is short enough and no front
// outgrows a workgroup's LDS panel; otherwise the caller keeps the multigrid PCG (C2's Manhattan world with four edges per
// pose has 580-row fronts and 0.24 Gflop on the critical path: not for one workgroup per front).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "sgo_direct.h"
#include "sgo_internal.h"

namespace sgo {

constexpr int kMfPanel = 16;          // panel width (= the MFMA tile)
constexpr int kMfThreads = 512;       // workgroup of the factor / solve kernels
constexpr int kMfMaxDim = 1023;       // largest front (scalar rows incl. the right-hand side row): the LDS panel is kMfMaxDim + 1 rows x 16 columns
constexpr int kMfElem = 27;           // per-edge terms: D_ii (6) D_jj (6) H_ij (9) b_i (3) b_j (3)

struct MfFront {            // one front; the fronts are numbered in post-order (children before parents)
  int e0 = 0;               // first elimination position of its own poses
  int own = 0, nb = 0;      // own poses, boundary poses
  int bnd_off = 0;          // boundary: elimination positions bnd[bnd_off .. + nb), ascending
  int kid[2] = {-1, -1};
  int map_off[2] = {0, 0};  // child k: cmap[map_off[k] + i] = local pose index (own first, then boundary) of the child's boundary pose i
  int parent = -1;
  int height = 0;
  int tgt0 = 0, tgt1 = 0;   // assembly targets
  int ld = 0;               // leading dimension of the front's matrix (column-major, rows 0 .. 3 (own + nb) incl. the right-hand side row)
  long long off = 0;        // its place in the arena (doubles)
};

struct MfTarget {           // one 3x3 block of a front that original edges contribute to
  int li, lj;               // local pose indices, li >= lj
  int c0, c1;               // contributions contrib[c0 .. c1): edge << 2 | part
};                          // part 0: D_ii + b_i (diagonal target of vertices()[0]); 1: D_jj + b_j; 2: H_ij as stored (row i, column j); 3: its transpose

struct MfPlan {
  int n = 0;
  int order_kind = 0;                 // 0: Hilbert order of the initial positions, 1: vertex id order
  std::vector<int> elim_vertex;       // [n] elimination position -> vertex id
  std::vector<MfFront> fronts;
  std::vector<int> bnd, cmap;
  std::vector<MfTarget> targets;
  std::vector<int> contrib;
  std::vector<int> level_ptr, level_front;   // fronts of height h: level_front[level_ptr[h] .. level_ptr[h + 1]), largest first
  // figures of merit
  int height = 0, max_dim = 0, max_own = 0, max_bnd = 0;
  double flops = 0.0, crit_flops = 0.0;      // multiply-adds x 2 of the partial factorisations: all fronts / the largest front of every level
  int crit_panels = 0;                       // 16-column panels on the critical path
  long long arena_doubles = 0;
};

struct MfLimits {
  int max_rows = 65536;
  int leaf = 32;
  double max_crit_flops = 80e6;
  long long max_arena_bytes = (long long)1 << 30;
  double max_degree = 2.5;           // edges between free poses per free pose: above it the graph is refused unanalysed
  int both_orders_rows = 8192;       // up to this size both row orders are analysed, above it only the Hilbert order
  int only_kind = -1;                // >= 0: analyse this row order alone (the caller knows which one wins on this graph's kind)
};

// Host analysis.  false with *why set: the graph does not qualify.
bool mfront_analyze(int V, int n, const int* free_id, const double* poses, int E, const int* ei, const int* ej, const MfLimits& lim,
                    MfPlan* plan, std::string* why);

struct Mfront;   // opaque: plan + device arrays

struct MfrontInfo {
  int n = 0, fronts = 0, height = 0, max_dim = 0, max_own = 0, max_bnd = 0, order_kind = 0, crit_panels = 0;
  double flops = 0.0, crit_flops = 0.0;
  size_t arena_bytes = 0;
};

// nullptr with *why set: the graph does not qualify; nullptr with *err set: HIP failure.
// Device arrays come out of `arena` (the context's per-graph arena: rewound, not freed, by the next set-up -- the reference
// re-initialises after every closure).  *order_hint (in / out, -1: none): the row order the previous graph of this context
// chose; when the new graph has about as many poses that order is analysed alone.
Mfront* mfront_create(hipStream_t s, DevArena* arena, int V, int n, const int* free_id, const double* poses, int E, const int* ei,
                      const int* ej, int max_rows, int* order_hint, std::string* why, std::string* err);
void mfront_destroy(Mfront* m);
const MfrontInfo& mfront_info(const Mfront* m);
// iters x { edges + chi2, factorise (levels up), substitute (levels down), update } + the closing chi2 on the stream; outputs
// as direct_optimize (d_hist[2 (iters + 1)], DirectResult: done / fail / fail_iter / stamps)
hipError_t mfront_optimize(Mfront* m, hipStream_t s, const EdgeListDev& el, double* d_poses, int iters, double* d_hist,
                           DirectResult* d_res);
double mfront_bytes(const Mfront* m, int E, int iters);

}  // namespace sgo
