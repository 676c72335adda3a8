// sgo_direct.h -- the small-graph path: optimize(iters) of a pose graph of the size the reference itself
// produces (intel-lab: ~1k poses and a few tens of closures, slc.cpp:205-288) as ONE kernel launch of one
// workgroup -- linearisation, a sparse block LDL^T in nested-dissection order, the update and chi2 of every
// Gauss-Newton iteration -- instead of ~150 launches per iteration of the multigrid PCG, whose fixed costs
// (a dense coarsest inverse, 8 launches per PCG iteration) dominate at this size.
#pragma once
#include <string>

#include <hip/hip_runtime.h>

#include "sgo_internal.h"

namespace sgo {

struct Direct;   // opaque: elimination plan + device arrays

struct DirectInfo {
  int n = 0;         // free poses
  int n_chain = 0;   // ... eliminated level by level (cyclic reduction of the trajectory's chain segments)
  int n_sep = 0;     // ... separators (a vertex cover of the non-chain edges), factorised as one dense block in LDS
  int levels = 0;    // sparse elimination levels (height of the elimination tree below the separators)
  int slots = 0;     // stored off-diagonal blocks of the sparse columns (incl. fill and padding)
  int contributions = 0;   // Schur-complement block products per factorisation
  size_t lds_bytes = 0;
};

// Result of one direct_optimize call (pinned host memory, filled by the kernel).
struct DirectResult {
  int done;          // Gauss-Newton updates applied
  int fail;          // 0 ok; 1 a pivot block was not positive definite; 2 non-finite update
  int fail_iter;
  int pad;
  unsigned long long cycles;     // shader clock cycles (s_memtime) between the first and the last stamp
  unsigned long long phase[8];   // wall_clock64 of the LAST iteration: start, edges done, assembled, sparse forward done,
                                 // separators eliminated, separators solved, sparse backward done, poses updated
  unsigned long long stamp[2 * SGO_MAX_ITERS + 4];   // wall_clock64 (100 MHz) at the start of iteration k [2k] and
                                                     // after its assembly [2k + 1]; the closing chi2 pass starts at
                                                     // [2 iters] and ends at [2 iters + 1]; after a failure in
                                                     // iteration k the call ends at [2k + 2]
};

// Host analysis + upload.  nullptr with *why set: the graph does not qualify (too many separators / levels /
// rows; the caller uses the multigrid path); nullptr with *err set: HIP failure.
Direct* direct_create(hipStream_t s, DevArena* arena, int V, int n, const int* free_id, int E, const int* ei, const int* ej,
                      int max_rows, std::string* why, std::string* err);
void direct_destroy(Direct* d);
const DirectInfo& direct_info(const Direct* d);

// iters x { chi2, linearise, factorise, solve, update } + the final chi2 on the stream; d_hist[2 (iters + 1)] gets
// (chi2, robust chi2) at the start of every iteration and at the end; d_res is a device DirectResult.
hipError_t direct_optimize(Direct* d, hipStream_t s, const EdgeListDev& el, double* d_poses, int iters, double* d_hist,
                           DirectResult* d_res);
// algorithmic bytes of one call (profile table)
double direct_bytes(const Direct* d, int E, int iters);

}  // namespace sgo
