/* sgo.h -- C-ABI of the MI355X-native SE(2) pose-graph optimiser (libsgo.so).
 *
 * This is the drop-in boundary under sparse-gslam's g2o call sites.  The reference drives a
 * g2o::SparseOptimizer (src/sparse_gslam/include/graphs.h:31-40) configured as
 *   OptimizationAlgorithmGaussNewton + BlockSolver<BlockSolverTraits<3,3>> + LinearSolverEigen
 * (src/sparse_gslam/src/graphs.cpp:17-23) through
 *   initializeOptimization(); optimize(20); computeActiveErrors();
 * (src/sparse_gslam/src/submap_loop_closer.cpp:286-288, src/sparse_gslam/src/log_runner.cpp:203-204).
 * The C++ mirror of that API (include/g2o/...) marshals the pointer graph into the flat arrays
 * below at initializeOptimization() and calls these entry points; INTEGRATION.md shows the
 * binding a maintainer adds.
 *
 * Conventions
 *   - plain pointers and sizes, caller-owned buffers, no exceptions, no torch types;
 *   - every function returning int returns >= 0 on success and a negative SGO_E* code on
 *     failure; sgo_last_error() gives the text;
 *   - one sgo_ctx is thread-compatible (one thread at a time, like one g2o::SparseOptimizer,
 *     serialised by the caller's boost::shared_mutex, graphs.h:21,32); different contexts are
 *     independent (own HIP stream and device buffers, no global mutable state);
 *   - all arithmetic is fp64 (g2o number_t = double).
 *
 * Array layout at the boundary (host memory)
 *   poses[V][3]  x, y, theta of vertex id v (ids dense from 0: drone.cpp:64,121; slc.cpp:212)
 *   fixed[V]     1 = VertexSE2::setFixed(true) (drone.cpp:66,75) -> excluded from the system
 *   ei[E], ej[E] vertex ids of EdgeSE2::vertices()[0], [1]   (slc.cpp:214-215, 279-280)
 *   meas[E][3]   EdgeSE2::setMeasurement(SE2(x,y,theta))      (slc.cpp:217, 275)
 *   info[E][6]   EdgeSE2::information(), upper triangle o11,o12,o13,o22,o23,o33 (slc.cpp:216, 276)
 *   phi[E]       DCS parameter of the edge's robust kernel (RobustKernelDCS::setDelta,
 *                slc.cpp:57; attached slc.cpp:283); < 0 = no robust kernel (odometry edges)
 */
#ifndef SGO_H_
#define SGO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGO_VERSION 107          /* 0.1.7: sgo_stats.pcg_converged may be 2 (a solve accepted at the floating-point floor of its system), the incremental overlay keeps 64 touched + hub rows, SGO_AMG_SETUP (no signature changed); 0.1.6: sgo_plan_rows takes the measurements (row order of graphs whose poses contradict their closures); 0.1.5: the multifrontal path for mid-size graphs (sgo_mfront_plan; sgo_solver_description names it); 0.1.4: sgo_kernel_profile_samples, sgo_update_graph_se2 (incremental set-up); 0.1.3: row-owner multi-GPU mode (sgo_comm_host_allgather, sgo_debug_level0_bytes); 0.1.2: sgo_comm_init_host; 0.1.1: sgo_opts.direct_rows (took a reserved slot), sgo_solver_description */
#define SGO_MAX_ITERS 256        /* capacity of the per-iteration arrays in sgo_stats */

/* error codes (negative).  -1 mirrors g2o's optimize() "nothing to optimise". */
#define SGO_OK 0
#define SGO_ENOTHING (-1)
#define SGO_EINVAL (-2)
#define SGO_EHIP (-3)
#define SGO_ENOGRAPH (-4)
#define SGO_ECOMM (-5)
#define SGO_ENOMEM (-6)

/* linear solver behind BlockSolver::solve (replaces LinearSolverEigen, graphs.cpp:19) */
#define SGO_SOLVER_PCG_BJ 0      /* block-Jacobi preconditioned CG */
#define SGO_SOLVER_PCG_AMG 1     /* CG preconditioned by a rigid-body smoothed-aggregation multigrid V-cycle
                                    with block-Jacobi smoothing (K-cycle on levels that keep the
                                    tentative prolongator) */

typedef struct sgo_ctx sgo_ctx;

typedef struct sgo_opts {
  int32_t struct_size;     /* = sizeof(sgo_opts); lets the struct grow compatibly */
  int32_t solver;          /* SGO_SOLVER_*; env SGO_SOLVER={pcg,amg} overrides the default */
  double pcg_tol;          /* stop when ||r|| <= pcg_tol * ||b||   (env SGO_PCG_TOL); chain-like graphs
                              (< 4 Hessian blocks per free pose, i.e. ill-conditioned and cheap to
                              iterate on) use pcg_tol / 10.  Inside one sgo_optimize_gn call the later
                              Gauss-Newton iterations keep the ABSOLUTE accuracy of the first solve,
                              ||r|| <= pcg_tol * ||b_first||, once their own right-hand side has shrunk
                              below ||b_first|| -- see pcg_tol_cap */
  int32_t pcg_maxit;       /* cap on PCG iterations per GN iteration (env SGO_PCG_MAXIT) */
  int32_t pcg_chunk;       /* graph mode: pcg_chunk / 16 replays of the 2-iteration hipGraph are kept in
                              flight speculatively between checks of the device-side stop flag;
                              plain mode: iterations launched between two host checks */
  int32_t use_graph;       /* 1: replay PCG iterations (incl. the multigrid cycle) from a hipGraph;
                              0: plain stream launches */
  int32_t profile;         /* 1: every launch carries its own start/stop HIP events
                              (hipExtLaunchKernelGGL; forces use_graph=0); per-kernel totals through
                              sgo_kernel_profile() */
  int32_t verbose;         /* mirrors SparseOptimizer::setVerbose (graphs.cpp:21) */
  int32_t direct_rows;     /* small-graph path (solver PCG_AMG, one GPU): a graph with at most this many free poses
                              whose elimination analysis fits -- trajectory chain + closures covered by <= 60
                              separator poses -- runs sgo_optimize_gn as ONE kernel launch of a sparse block
                              LDL^T (nested dissection of the chain, separators dense in LDS); 0 = never
                              (env SGO_DIRECT_ROWS); the single-step entry points keep using the PCG path.
                              The path itself works (and wins) up to ~100k chain-like poses; the default stops
                              where two backward-stable solvers stop agreeing to 1e-6 in chi2 (kappa ~ n^2).
                              Graphs it refuses are offered to the MID-SIZE path next (same conditions: PCG_AMG, one
                              GPU, direct_rows > 0): a multifrontal sparse Cholesky factorisation in nested-dissection
                              order, one launch per level of the elimination tree -- up to 12 288 free poses (env
                              SGO_MFRONT_ROWS), at most 2.5 edges per pose, no front of more than 1 023 rows and
                              at most 80 Mflop on the tree's critical path (env SGO_MFRONT_CRIT_MFLOP); env
                              SGO_MFRONT=0 switches it off.  What it refuses takes the multigrid PCG */
  double pcg_tol_cap;      /* loosest RELATIVE tolerance that absolute criterion may reach (default 1e-6; 0: every
                              solve uses pcg_tol relative to its own ||b||; env SGO_PCG_TOL_CAP).  As Gauss-Newton
                              converges ||b|| falls by orders of magnitude; solving each step to 1e-8 of ITSELF
                              buys no accuracy of the iterates (chi2 is second order in the step's error) and
                              costs a quarter of the PCG iterations */
  int32_t pcg_warm_start;  /* 1 (default): from the second Gauss-Newton iteration of a call PCG starts from the previous
                              step scaled by the energy-optimal factor (b.x_prev)/(x_prev.H x_prev) instead of from zero:
                              same stopping test, about two iterations fewer per solve (env SGO_PCG_WARM) */
  int32_t reserved[4];
} sgo_opts;

/* Environment variables (the call sites are frozen, so knobs come from the environment).  Read when a
 * context is created or a graph is set, or at the start of sgo_optimize_gn / sgo_solve (sgo_ctx::CallKnobs, read_call_knobs) -- never
 * inside a solve.
 *   mirrors of sgo_opts fields (the environment wins): SGO_SOLVER={pcg,amg}, SGO_PCG_TOL, SGO_PCG_TOL_CAP, SGO_PCG_MAXIT,
 *     SGO_PCG_CHUNK, SGO_PCG_WARM, SGO_USE_GRAPH, SGO_PROFILE, SGO_VERBOSE, SGO_DIRECT_ROWS, SGO_DEVICE
 *   path selection: SGO_MFRONT=0 (no multifrontal path), SGO_MFRONT_ROWS / _CRIT_MFLOP / _DEGREE / _LEAF (its admission limits
 *     and leaf size), SGO_INCREMENTAL=0 (sgo_update_graph_se2 is always a full set-up), SGO_SPMV0={tile,group} (level-0 product
 *     kernel), SGO_PRECOND_F32=0 (fp64 blocks in the preconditioner's level-0 passes)
 *   multigrid set-up: SGO_AMG_THETA (strength threshold), SGO_AMG_THETA_FILTER / SGO_AMG_FILTER=0 (filtered smoothing),
 *     SGO_AMG_SMOOTH=0 (tentative transfers only), SGO_AMG_OMEGA, SGO_AMG_OMEGA_P, SGO_AMG_NU, SGO_AMG_FOLD, SGO_AMG_FOLD0_ROWS,
 *     SGO_AMG_KDEPTH, SGO_AMG_FCG2_DEPTH (cycle shape), SGO_HOST_THREADS (worker pool of the host set-up),
 *     SGO_AMG_SETUP={device,rebuilds,host} (round 6; default device on one GPU: the hierarchy's patterns -- P, A P, P^T A P, the
 *     tentative map -- and lists are made on the device from the host's aggregation, bit-identical to the host set-up's; rebuilds:
 *     only for the rebuilds inside sgo_optimize_gn; host: never), SGO_AMG_AGG=device (opt-in: the aggregation on the device as well --
 *     a parallel independent-set aggregation, weaker hierarchies: DESIGN.md section 5)
 *   inside sgo_optimize_gn: SGO_AMG_LAG=0 (the hierarchy's coarse operators are refreshed before EVERY solve; default: a solve keeps
 *     those of the solve before while the level-0 diagonal blocks have barely moved, and a hierarchy whose aggregation the blocks
 *     have moved far away from is re-made once inside the next call: DESIGN.md section 5), SGO_AMG_LAG_TAU (the largest relative
 *     movement a solve may keep its coarse operators over, 0.006)
 *   multi-GPU: SGO_COMM_MODE={owner,allreduce}, SGO_COMM_GRAPH (see sgo_comm_init), SGO_OWNER_MIN_ROWS, SGO_RCCL_LIB (library path)
 *   test hooks and A/B switches of scripts/ (not for production): SGO_AMG_LISTS=host, SGO_SETUP_PIPELINE, SGO_TILE_LDS,
 *     SGO_FIRST_SOLVE_CAP, SGO_PCG_STALL_WINDOW, SGO_TEST_FAIL_TRIAL_BUILD, SGO_TEST_FAIL_DEVICE_SETUP, SGO_AMG_FORCE_REBUILD, SGO_AMG_KEEP_AGG, SGO_AMG_REBUILD_COST, SGO_MIRROR, SGO_LANCZOS (sgo_debug_lanczos), SGO_MFRONT_DEBUG, SGO_AMG_LAG_FORCE / SGO_AMG_LAG_SLOPE
 *     (scripts/lag_calib.py, tests/test_gpu_lagged_refresh.py)
 * Removed in round 5 (measured, not kept: NOTES.md sections 9-10): SGO_DEFLATE, SGO_OWNER_XS_EXCHANGE, SGO_MFRONT_FUSED_SOLVE.
 * Of the interface SURVEY.md section 8(b) sketched, three items do not exist, on purpose: SGO_NGPU (one process per GPU: the
 * launcher sets the world size, sgo_comm_init takes it), SGO_SOLVER=direct_cpu (the product has no CPU path; the CPU solver is
 * the oracle, test infrastructure), sgo_stats.bytes_moved (algorithmic bytes are per kernel: sgo_kernel_profile). */

/* Defaults (also applied when opts == NULL):
 * solver = PCG_AMG (graphs with <= 400 free poses are preconditioned by an explicit dense inverse,
 * i.e. solved directly; falls back to PCG_BJ only when a larger graph cannot be coarsened),
 * pcg_tol = 1e-8, pcg_maxit = 20000, pcg_chunk = 16, use_graph = 1, direct_rows = 8192, pcg_tol_cap = 1e-6, pcg_warm_start = 1. */
void sgo_default_opts(sgo_opts* o);

typedef struct sgo_stats {
  int32_t iters_requested;
  int32_t iters_done;                        /* GN updates actually applied (the return value is 0 whenever a
                                                linear solve failed, as g2o's optimize(); this still counts the
                                                updates applied before the failure) */
  double chi2[SGO_MAX_ITERS + 1];            /* activeChi2 at the START of iteration k; [iters_done] = final */
  double robust_chi2[SGO_MAX_ITERS + 1];     /* activeRobustChi2, same indexing */
  int32_t pcg_iters[SGO_MAX_ITERS];          /* PCG iterations of GN iteration k (0 on the direct small-graph path) */
  int32_t pcg_converged[SGO_MAX_ITERS];      /* 1 = reached pcg_tol; 2 (round 6) = stopped without reaching it with x at the floating-point
                                              * floor of its system (normwise backward error |r| / (|H| |x| + |b|) <= 1e-12): the step WAS
                                              * applied, as the solution of a backward-stable direct solver (LinearSolverEigen) would be;
                                              * 0 = failed: the step was not applied, sgo_optimize_gn returned 0 */
  double pcg_relres[SGO_MAX_ITERS];          /* final ||r|| / ||b|| (recurrence residual) */
  double seconds[SGO_MAX_ITERS];             /* device time of GN iteration k (HIP events) */
  double seconds_linearize[SGO_MAX_ITERS];   /* ... of which error/Jacobian/assembly */
  double seconds_solve[SGO_MAX_ITERS];       /* ... of which the linear solve */
  double seconds_total;                      /* wall time of the whole call (host clock) */
  double seconds_setup;                      /* host structure build + upload of the last set_graph */
} sgo_stats;

int sgo_version(void);

/* Replaces: new g2o::SparseOptimizer + setup_pose_opt() (graphs.cpp:17-23, graphs.h:38).
 * device = HIP device ordinal (env SGO_DEVICE overrides when device < 0).  NULL on failure
 * (sgo_last_error(NULL) has the reason). */
sgo_ctx* sgo_create(int device, const sgo_opts* opts);
/* Replaces: ~SparseOptimizer.  Never frees caller memory (README.md:22-23). */
void sgo_destroy(sgo_ctx* ctx);

/* Replaces: SparseOptimizer::initializeOptimization() (slc.cpp:286, log_runner.cpp:203): takes
 * the active vertices/edges, builds the hessian index map (non-fixed vertices in ascending id),
 * the block-sparse structure (rows numbered internally along a Hilbert curve through the poses, every
 * off-diagonal block stored once inside a tile of rows) and the device-resident SoA edge arrays.  Edges
 * whose endpoints are both fixed stay active for chi2 only.  With a communicator attached
 * (sgo_comm_init) every rank passes the same full graph. */
int sgo_set_graph_se2(sgo_ctx* ctx, int32_t V, const double* poses, const uint8_t* fixed, int32_t E,
                      const int32_t* ei, const int32_t* ej, const double* meas, const double* info,
                      const double* phi);

/* Incremental form of the same call for the reference's flow: after every accepted loop closure it appends a chain of
 * new poses with their odometry edges (slc.cpp:205-226) and one closure edge (slc.cpp:272-285) to the graph it optimised
 * before, then calls initializeOptimization(); optimize(20) again (slc.cpp:286-287).  Arguments as sgo_set_graph_se2 -- the
 * WHOLE new graph -- plus n_resident_edges: the caller's statement that edges [0, n_resident_edges) and the vertices they
 * use are the graph this context already holds, unchanged (ids, measurement, information, kernel, fixed flags; estimates
 * may differ: `poses` is uploaded in full).  When that is the resident graph and the appended part has the shape above --
 * new free poses whose mutual edges form a chain in id order, any edges between new and resident poses or among resident
 * poses, within the capacities of sgo_overlay.h -- the appended part becomes an overlay beside the resident level-0
 * structure and multigrid hierarchy, which are kept (the Gauss-Newton systems are still solved exactly: the chain is
 * eliminated by a block-tridiagonal factorisation, the resident rows by PCG on the Schur complement).  Otherwise --
 * including n_resident_edges == 0, a multi-GPU context, a graph on the single-launch direct path, an overlay whose PCG
 * iteration counts have drifted -- the call IS sgo_set_graph_se2.  sgo_solver_description says which happened.  While an
 * overlay is resident the single-step entry points (sgo_linearize, sgo_hessian_apply, sgo_precondition, sgo_solve) return
 * SGO_EINVAL; sgo_optimize_gn, sgo_chi2, sgo_edge_chi2, sgo_get_poses, sgo_set_poses, sgo_num_free, sgo_free_ids cover the
 * whole graph. */
int sgo_update_graph_se2(sgo_ctx* ctx, int32_t V, const double* poses, const uint8_t* fixed, int32_t E,
                         const int32_t* ei, const int32_t* ej, const double* meas, const double* info,
                         const double* phi, int32_t n_resident_edges);

/* Replaces: VertexSE2::setEstimate on every vertex (slc.cpp:219) without a structure rebuild. */
int sgo_set_poses(sgo_ctx* ctx, const double* poses);
/* Replaces: reading VertexSE2::estimate() after optimize() (slc.cpp:144-148, log_runner.cpp:258-267). */
int sgo_get_poses(sgo_ctx* ctx, double* poses);

/* Replaces: SparseOptimizer::optimize(iters) with OptimizationAlgorithmGaussNewton (slc.cpp:287,
 * log_runner.cpp:204): iters x { computeActiveErrors; buildSystem; solve; update }, no damping, no
 * convergence test.  Returns iterations done; 0 when a linear solve failed -- PCG breakdown (H not
 * positive definite) or pcg_maxit reached without pcg_tol (on one GPU also: the residual has not reached a new minimum for
 * max(3000, 30 x the previous solve's count) iterations -- a solve that stagnates is not ground on to pcg_maxit): that step
 * is not applied, the estimates stay
 * at the last applied update (out->iters_done of them) and sgo_last_error has the reason, as
 * g2o::SparseOptimizer::optimize returns 0 on OptimizationAlgorithm::Fail; SGO_ENOTHING (-1) when
 * there is no free active vertex; or another negative code.
 * `out` may be NULL. */
int sgo_optimize_gn(sgo_ctx* ctx, int32_t iters, sgo_stats* out);

/* Replaces: computeActiveErrors(); activeChi2(); activeRobustChi2() (slc.cpp:288, drone.cpp:162-165). */
int sgo_chi2(sgo_ctx* ctx, double* plain, double* robust);
/* Replaces: EdgeSE2::computeError(); chi2() per edge (log_runner.cpp:183-184, the 11.345 gate).
 * e2[E] in the edge order given to sgo_set_graph_se2. */
int sgo_edge_chi2(sgo_ctx* ctx, double* e2);

/* Replaces: the covariance producer of a loop closure and its use as the edge information
 * (src/sparse_gslam/src/cartographer_bindings/fast_correlative_scan_matcher_2d.cc:537-561,
 * submap_loop_closer.cpp:276), batched: for match q the scores of the
 * (2 w + 1) x (2 w + 1) x (2 scan_window + 1) window around the best candidate, in the reference's
 * loop order (x offset i outermost, y offset j, scan index k fastest), give
 *     x(i,j,k) = (-j res, -i res, (k - num_angular_perturbations) angular_step)
 *                 (include/cartographer_bindings/correlative_scan_matcher_2d.h:78-82)
 *     K = sum score x x^T, u = sum score x, s = sum score,   cov = K / s - u u^T / s^2,
 * and information = cov^-1 (3x3 row-major, cofactors / determinant as Eigen's fixed-size inverse).
 * A window with s == 0 or a singular covariance yields non-finite entries, as the reference's
 * arithmetic does.  Needs no graph.  scores are addressed from win[q].score_offset. */
typedef struct sgo_match_window {
  int32_t x_index_offset;   /* best candidate (cells) */
  int32_t y_index_offset;
  int32_t scan_index;
  int32_t scan_window;      /* min(w, scan_index, n_scans - 1 - scan_index) at the call site */
  int32_t w_size;           /* 5 in the reference */
  int32_t num_angular_perturbations;
  double resolution;        /* SearchParameters::resolution */
  double angular_step;      /* SearchParameters::angular_perturbation_step_size */
  int64_t score_offset;     /* first score of this window in scores[] */
} sgo_match_window;
int sgo_closure_information(sgo_ctx* ctx, int32_t n, const sgo_match_window* win, const float* scores,
                            int64_t n_scores, double* cov /*[n][9]*/, double* info /*[n][9]*/);

/* ---- single-step entry points (what BlockSolver::buildSystem / solve expose inside g2o);
 *      used by the parity tests to check each kernel against the oracle. ---------------------- */
/* number of free (non-fixed, active) vertices n; hessian index h -> vertex id in free_ids[n] */
int sgo_num_free(sgo_ctx* ctx);
int sgo_free_ids(sgo_ctx* ctx, int32_t* free_ids);
/* buildSystem at the current poses: b[n][3], diag[n][9] (row-major 3x3), chi2 values. */
int sgo_linearize(sgo_ctx* ctx, double* b, double* diag, double* plain, double* robust);
/* y = H x with the Hessian of the last sgo_linearize (x, y: [n][3]). */
int sgo_hessian_apply(sgo_ctx* ctx, const double* x, double* y);
/* solve H x = b of the last sgo_linearize; returns PCG iterations (>= 0). */
int sgo_solve(sgo_ctx* ctx, double* x, double* relres);
/* z = M^-1 r with the configured preconditioner of the last sgo_linearize (r, z: [n][3]). */
int sgo_precondition(sgo_ctx* ctx, const double* r, double* z);

/* ---- profiling (opts.profile = 1) ---------------------------------------------------------- */
/* Per-kernel totals accumulated since the last sgo_profile_reset: for kernel slot k,
 * name (static string), launches, total milliseconds (HIP events on the ctx stream), and the
 * algorithmic bytes those launches were specified to move.
 * Slots are named as rocprofv3 prints the kernel; the launches of the three block-stream kernels on the
 * finest multigrid level are kept in slots of their own, named "<kernel> @level0" (the same kernels'
 * coarse-level launches sit on the launch-latency floor and would blur the bandwidth figure).
 * Returns the number of slots; fills up to cap entries. */
typedef struct sgo_kernel_stat {
  const char* name;
  int64_t launches;
  double ms;
  double bytes;
} sgo_kernel_stat;
int sgo_kernel_profile(sgo_ctx* ctx, sgo_kernel_stat* out, int cap);
/* The single launches behind slot `slot` of sgo_kernel_profile (same order of slots): their durations in milliseconds, in
 * launch order, up to the first 16384 per slot since the last sgo_profile_reset -- for medians and percentiles (a mean hides
 * what a few outliers or early-exit launches do to it).  In profile mode the PCG loop checks the stop flag after every
 * iteration, so no launch past convergence is among them.  Returns the number of samples written (<= cap), < 0 on error. */
int sgo_kernel_profile_samples(sgo_ctx* ctx, int slot, float* out_ms, int cap);
int sgo_profile_reset(sgo_ctx* ctx);
/* Milliseconds an EMPTY event bracket measures on this context's stream: the upper bound of the
 * per-launch bias of sgo_kernel_profile's times relative to rocprofv3 kernel durations. */
double sgo_profile_overhead_ms(sgo_ctx* ctx);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ------------------------------------------
 * Every rank is given the SAME full graph (sgo_set_graph_se2 with identical arguments) and makes the same row plan: the
 * rows of the level-0 Hessian in Hilbert order of the poses, cut into 256 tiles per rank; rank r owns the contiguous range
 * of tiles sgo_shard_range(ntiles, nranks, r).  Two modes, chosen per graph (sgo_solver_description says which):
 *   row-owner mode  (at most a quarter of the rows have an edge into another rank's range -- spatially local closures): a rank
 *     holds the Hessian blocks and edge operands of ITS rows only and linearises, multiplies, smooths, restricts, prolongates and
 *     updates these rows; per PCG iteration it exchanges the boundary rows of two vectors and the partial sums of the dot
 *     products (fixed-size all-gather packets, reduced in rank order: bit-identical scalars on all ranks) and all-reduces the
 *     coarse right-hand side; per Gauss-Newton iteration the boundary rows of the smoothed prolongator and the step (for the
 *     replicated pose update) travel, and the level-1 Galerkin blocks are all-reduced.  The coarse multigrid levels run replicated.
 *   all-reduce mode (random long-range closures: most rows are boundary rows): every rank holds the whole graph, evaluates every
 *     level-0 Hessian product of the solve for the rows of its tiles only, zeros elsewhere, and the product vector is summed over
 *     ranks with ncclAllReduce (one non-zero contributor per row: the sum is exact) -- the per-PCG-step exchange BASELINE.json's
 *     north_star names.  Linearisation, coarse levels and vector recurrences run replicated on bit-identical data.
 * chi2 is summed over edge ranges in both.  Every rank takes the same decisions (stopping, hierarchy rebuilds) from
 * bit-identical scalars. */
/* 128-byte unique id for rendezvous (wraps ncclGetUniqueId); rank 0 creates it, the host layer
 * broadcasts it (torch.distributed / MPI / a file), every rank passes it to sgo_comm_init. */
#define SGO_UNIQUE_ID_BYTES 128
int sgo_comm_unique_id(void* id_out);
int sgo_comm_init(sgo_ctx* ctx, int nranks, int rank, const void* unique_id);
int sgo_comm_size(sgo_ctx* ctx);
/* The same multi-GPU mode over a transport the CALLER brings (MPI, gloo, a socket layer) instead of RCCL: every
 * collective copies its vector to pinned host memory, calls `fn` -- which must replace buf[0..count) by its sum
 * over all ranks, bit-identical on every rank, and return 0 -- and copies the result back.  For nodes without
 * xGMI / RCCL and for multi-process tests on one GPU; slower than sgo_comm_init.  `fn` is called on the thread
 * that called into the library.  Must precede sgo_set_graph_se2. */
typedef int (*sgo_host_allreduce_fn)(double* buf, size_t count, void* user);
int sgo_comm_init_host(sgo_ctx* ctx, int nranks, int rank, sgo_host_allreduce_fn fn, void* user);
/* Optional companion of sgo_comm_init_host for the row-owner mode's packets: recv[r * count .. (r + 1) * count) = rank r's
 * send[0 .. count) (host memory), return 0.  Without it libsgo performs an all-gather as an all-reduce of zero-padded slots
 * through `fn` (exact, nranks times the bytes).  Call after sgo_comm_init_host, before sgo_set_graph_se2. */
typedef int (*sgo_host_allgather_fn)(const double* send, size_t count, double* recv, void* user);
int sgo_comm_host_allgather(sgo_ctx* ctx, sgo_host_allgather_fn fn);
/* The contiguous range [begin, end) of `count` work units (tiles, edges) that rank `rank` of
 * `nranks` evaluates.  Pure function (no GPU needed); exposed so the host layer and the CPU tests
 * can reproduce the partition. */
void sgo_shard_range(int32_t count, int32_t nranks, int32_t rank, int32_t* begin, int32_t* end);
/* The row plan sgo_set_graph_se2 makes for a graph, computed on the host alone (no GPU, no context): the
 * number of free active vertices n, the vertex id of every internal row (row_vertex[n], Hilbert order), the
 * tiles (tile_row_begin[ntiles + 1], capacity tile_cap) and the first row of every rank's tile range
 * (rank_row_begin[nranks + 1]).  Output pointers may be NULL.  Lets multi-process callers and the CPU tests
 * see which rows of a level-0 product each rank contributes.  meas (as sgo_set_graph_se2's; may be NULL): with it the plan is
 * also the one sgo_set_graph_se2 makes for a graph whose initial poses contradict its closures -- a dead-reckoned start --,
 * whose rows are ordered by spanning-tree positions instead of the poses (0.1.6).  Exception: a graph sgo_set_graph_se2 puts
 * on a factorisation path (single-launch direct, multifrontal: <= 12 288 free poses) has no level-0 row plan at set-up; the first
 * single-step entry point (sgo_linearize, ...) makes one from the poses CURRENT at that moment in plain Hilbert order, which this
 * function reproduces when given those poses and meas = NULL.  (The trailing `meas` argument was ADDED in 0.1.6: a caller built
 * against an older header must be rebuilt -- SGO_VERSION / sgo_version() is the check.) */
int sgo_plan_rows(int32_t V, const double* poses, const uint8_t* fixed, int32_t E, const int32_t* ei, const int32_t* ej,
                  int32_t nranks, int32_t* n_free, int32_t* row_vertex, int32_t* ntiles, int32_t* tile_row_begin,
                  int32_t tile_cap, int32_t* rank_row_begin, const double* meas);
/* The elimination plan of the multifrontal path (mid-size graphs: DESIGN.md section 5c), computed on the host alone (no GPU,
 * no context).  stats[12] = { free poses, fronts, levels, largest front (scalar rows), its own poses, its boundary poses,
 * flops per factorisation, flops on the critical path (largest front of every level), 16-column panels on the critical path,
 * bytes of frontal matrices, row order (0 Hilbert, 1 id), assembly targets }; elim_vertex[n] (optional): vertex id at every
 * elimination position; front_of_elim[n] (optional): its front (fronts are numbered children-first).  leaf <= 0 and
 * max_crit_mflop <= 0 take the defaults of sgo_set_graph_se2.  SGO_ENOTHING: the graph does not qualify (sgo_last_error
 * says why; stats[0] is still set). */
int sgo_mfront_plan(int32_t V, const double* poses, const uint8_t* fixed, int32_t E, const int32_t* ei, const int32_t* ej,
                    int32_t leaf, double max_crit_mflop, int64_t* stats, int32_t* elim_vertex, int32_t* front_of_elim);

/* Test hook: make this context evaluate the tile range of rank `rank` of `nranks` WITHOUT a
 * communicator (collectives are skipped), so that the per-rank partial products can be inspected on
 * a single GPU: summing sgo_hessian_apply's result over rank = 0..nranks-1 must reproduce the
 * single-rank product exactly.  Must precede sgo_set_graph_se2. */
int sgo_debug_set_shard(sgo_ctx* ctx, int nranks, int rank);

/* Further test / measurement hooks (no g2o counterpart).
 * sgo_debug_coarse_rhs: the coarse right-hand side the first half of a multigrid cycle makes from r ([n][3], hessian
 *   order): first sweep from zero, level-0 residual pass, restriction; under sgo_debug_set_shard this rank's partial
 *   (the partials of all ranks sum to the single-rank vector: the cycle's small all-reduce).  Returns its length,
 *   0 without a multi-level hierarchy.  Requires sgo_linearize.
 * sgo_debug_spmv0_us: mean microseconds of `reps` back-to-back launches of the level-0 Hessian product kernel on the
 *   resident graph (mode 0 = H x, 1 = residual, 2 = Jacobi sweep); variant 16 selects the wave-group kernel,
 *   variant 32 prints per-phase cycle stamps of the tile kernel. */
int sgo_debug_coarse_rhs(sgo_ctx* ctx, const double* r, double* out, int cap);
/* Device bytes of the level-0 structure THIS rank holds for the resident graph (Hessian blocks, per-slot edge operands,
 * per-slot / per-block index arrays, level-0 transfer blocks and product lists): proportional to 1 / nranks in row-owner mode. */
int64_t sgo_debug_level0_bytes(sgo_ctx* ctx);
double sgo_debug_spmv0_us(sgo_ctx* ctx, int mode, int variant, int reps);
/* Diagnostic (env SGO_LANCZOS=1 when the graph is set): alpha, beta of every PCG iteration of the last solve as pairs in
 * iteration order -- the Lanczos matrix of the preconditioned operator follows from them (scripts/ritz_probe.py).  Returns
 * the iterations written (<= cap pairs), < 0 on error. */
int sgo_debug_lanczos(sgo_ctx* ctx, double* out, int cap);

/* One line naming the solver the resident graph's sgo_optimize_gn runs ("direct_ldlt: ...", "multifrontal_cholesky: ...",
 * "pcg_amg: L0 n=... ", "pcg_block_jacobi ..."): which path a graph took, and for the refusals of the direct and the
 * multifrontal path the reason ("pcg_amg: ...; direct path not used: more separators than the dense block holds;
 * multifrontal path not used: 4.00 edges per free pose > 2.50").  Never NULL. */
const char* sgo_solver_description(sgo_ctx* ctx);

/* Text of the last error on this context (or, with ctx == NULL, of the last failed sgo_create /
 * context-free call on this thread).  Never NULL. */
const char* sgo_last_error(sgo_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* SGO_H_ */
