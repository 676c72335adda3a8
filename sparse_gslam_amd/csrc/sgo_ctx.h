// sgo_ctx.h -- the context behind include/sgo.h's sgo_ctx and the declarations shared by the host sources of libsgo:
//   sgo_plan.cpp       host-only row plan (hessian order, Hilbert row order, compact slots, tiles)
//   sgo_structure.cpp  device-resident graph: edge arrays, level-0 storage, tile view, logical view
//   sgo_solve.cpp      Gauss-Newton driver: chi2, linearise, PCG loop (hipGraph replay), multigrid set-up calls
//   sgo_api.cpp        the C-ABI entry points
// Not part of the public ABI.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "sgo_amg.h"
#include "sgo_comm.h"
#include "sgo_direct.h"
#include "sgo_mfront.h"
#include "sgo_internal.h"
#include "sgo_overlay.h"

using namespace sgo;   // (internal header: the global sgo_ctx of the C-ABI is made of sgo:: types)

// Bump allocator over one host block that only ever grows (uninitialised memory, 64-byte aligned).
struct HostArena {
  std::unique_ptr<char[]> mem;
  size_t cap = 0, used = 0;
  void reserve(size_t bytes) {   // invalidates earlier take()s
    used = 0;
    if (bytes <= cap) return;
    mem.reset();
    cap = bytes + bytes / 4;
    mem.reset(new char[cap + 64]);
  }
  void* take(size_t bytes) {
    char* base = (char*)(((uintptr_t)mem.get() + 63) & ~(uintptr_t)63);
    void* q = base + used;
    used += (bytes + 63) & ~(size_t)63;
    return used <= cap ? q : nullptr;
  }
};

// Host image of the row-owner partition (multi-GPU): what the multigrid set-up needs to list the entries of P in the
// boundary rows.
struct HaloHost {
  int G = 1, me = 0, bmax = 0;
  std::vector<int> rank_row;   // [G + 1]
  std::vector<int> bnd;        // [G][bmax], -1 padded
};

struct sgo_ctx {
  int device = 0;
  HostArena stage;
  ChunkArena amg_scratch;   // host lists of the multigrid set-up, reused across set-ups
  hipStream_t stream = nullptr;
  sgo_opts opts{};
  std::string err;
  Comm comm;
  int shard_u0 = 0, shard_u1 = 0, shard_units = 0;   // multi-GPU: this rank's range of level-0 work units (tiles)
  int shard_row0 = 0, shard_row1 = 0;                //            = these rows
  std::vector<int> unit_row0;                        // first row of every work unit (+ n at the end)
  bool owner = false;            // multi-GPU row-owner mode (HaloDev, sgo_internal.h); false with a communicator: all-reduce mode
  bool gather_slices = false;    // all-reduce mode with a communicator: product vectors travel as an all-gather of the ranks' slices
  bool replicated = false;       // multi-rank context, but this graph has fewer level-0 work units than ranks: every rank runs the
                                 // whole single-GPU computation (bit-identical on all ranks, no collective), see build_rows
  HaloDev halo;
  HaloHost halo_host;
  bool halo_failed = false;
  std::vector<double> order_xy;     // [V][2] positions the row order follows when they are not the initial poses (a graph whose
                                    // poses contradict its closures: plan_order_positions, sgo_plan.cpp); empty otherwise
  // Lagged refresh of the hierarchy's coarse operators (optimize_gn): d_dref = the level-0 diagonal blocks the coarse operators were
  // last made from, h_dchg = k_diag_change's three sums (host-mapped), amg_lag_tau = relative movement up to which a solve keeps them
  bool amg_skip_update = false;     // this solve keeps the coarse operators of the previous one
  bool amg_lag_on = false;          // inside sgo_optimize_gn, after the call's first refresh
  bool amg_force_keep = false;      // calibration hook (SGO_AMG_LAG_FORCE, scripts/lag_calib.py)
  bool amg_lag_expect = false;      // the previous iteration's movement says this one may keep: do_linearize waits for the sums
  bool amg_dchg_pending = false;    // k_diag_change's sums of this iteration have not been read yet
  bool amg_ref_valid = false;       // d_dref holds the blocks of the resident coarse operators
  double amg_lag_tau = 0.0;
  // what a unit of movement costs this graph's solves in PCG iterations, learned from the kept solves (C4: ~1000, a 50k / 250k
  // graph: ~4000-8000); a solve keeps while slope x movement <= 4 iterations (a refresh is worth 5-10).  Carried over a set-up when
  // the graph has about the size of the one before (the reference re-initialises a slowly growing graph before every optimize).
  double amg_lag_slope = 2700.0;
  bool amg_lag_slope_seen = false;
  int amg_lag_n = 0;
  int probe_dev_k = -1;             // what the device's PcgScalars hold (k_set_probe is launched only on a change)
  double probe_dev_max = -1.0;
  int amg_probe_k = 0;              // progress probe of the kept solves (PcgScalars::probe_k / probe_max), from the last fresh solve
  double amg_probe_max = 0.0;
  int amg_lag_cap = 0;              // iteration cap of a solve behind kept operators (then: refresh and solve again)
  double amg_lag_rows = 1.0;        // ... and the share of rows that may have moved by more than a quarter
  double* d_dref = nullptr;
  // ... and the blocks the hierarchy's AGGREGATION was made from (build_amg): a call whose first solve finds them far from the
  // current ones and needs visibly more iterations than the hierarchy's best redoes the set-up once (optimize_gn)
  double* d_dref_agg = nullptr;
  bool agg_ref_valid = false;
  int agg_best = 0;                 // the fewest PCG iterations a fresh solve behind this aggregation has taken
  int agg_grid = 0;                 // k_diag_change's workgroups of that measurement (sums in the second half of h_dchg), 0: none pending
  double* h_dchg = nullptr;         // pinned [2][3][kMaxPartials]: k_diag_change's per-workgroup sums (against d_dref, against d_dref_agg)
  int dchg_grid = 0;                // ... of the launch whose sums are pending
  double* h_dchg_dev = nullptr;     // its device address
  double last_dchg[3] = {0, 0, 0};
  std::string lag_note;             // sgo_solver_description: how many solves of the last call kept their coarse operators
  bool amg_no_filter = false;       // this graph's hierarchy rebuilds keep the tentative transfer where the smoothed one is refused
                                    // (a filtered hierarchy's solve was abandoned: optimize_gn); cleared by the next set-up
  int* d_comm_flag = nullptr;       // one int for the collective decision about the captured PCG graph (run_pcg)
  bool comm_graph_failed = false;   // capturing the RCCL collectives into the PCG hipGraph failed once: plain launches since
  long long level0_bytes = 0;    // device bytes of the level-0 structure this rank holds (blocks, operands, per-slot / per-block indices)
  double *halo_send = nullptr, *halo_recv = nullptr;   // exchange buffers (hipMalloc, grown on demand, kept across graphs)
  size_t halo_cap = 0;
  int halo_ranks = 0;            // world size the receive buffer was sized for

  // graph (host)
  bool has_graph = false;
  int V = 0, E = 0, n = 0;
  std::vector<int> free_id;      // hessian index (g2o order: free active vertices in ascending id) -> vertex id
  std::vector<int> row_of_asc;   // hessian index -> internal row (Hilbert order, build_structure)
  HostLevel H0;                  // logical level-0 structure on the host (multigrid set-up input)
  double setup_seconds = 0.0;

  // device
  DevArena graph_arena;           // device arrays of the resident graph (rewound by the next set_graph)
  DevArena amg_arena;             // ... of the multigrid hierarchy (rewound when the hierarchy is rebuilt)
  // The hierarchy a TRIAL rebuild replaced (optimize_gn's re-aggregation rule): kept with its arena until the trial is decided -- a
  // re-made hierarchy that does not solve visibly faster is dropped for it (revert_amg) -- or until the next rebuild.
  DevArena amg_arena_prev;
  DevArena amg_tmp_arena;         // temporaries of the device set-up (sgo_amg_dev.inc): rewound per level, kept between set-ups
  Amg* amg_prev = nullptr;
  std::string amg_prev_desc;
  bool agg_rule_off = false;      // a trial was lost on this graph: the rule does not fire again before the next set-up
  // The environment knobs a solve depends on, read ONCE per entry-point call (read_call_knobs: sgo_optimize_gn, sgo_solve) and
  // never inside a solve (include/sgo.h's table says so).
  struct CallKnobs {
    int comm_graph = -1;          // SGO_COMM_GRAPH: -1 unset (the default rule decides), 0 / 1
    int stall_window = -1;        // SGO_PCG_STALL_WINDOW: -1 unset
    int first_solve_cap = 600;    // SGO_FIRST_SOLVE_CAP (test hook)
    bool fail_trial_build = false;   // SGO_TEST_FAIL_TRIAL_BUILD (test hook: the re-aggregation trial's set-up "fails")
    bool fail_device_setup = false;  // SGO_TEST_FAIL_DEVICE_SETUP (test hook: the device set-up "fails", the host set-up takes over)
    bool keep_agg = false;           // SGO_AMG_KEEP_AGG: a rebuild inside the call keeps the replaced hierarchy's aggregates
    int setup_mode = 2;              // SGO_AMG_SETUP: host (0: the host's aggregation, patterns and lists for every set-up, as before round 6),
                                     // rebuilds (1: the patterns on the device for the rebuilds inside sgo_optimize_gn only), device (2, the
                                     // default on one GPU: for every set-up; the set-up pipeline's helper thread then makes level 0's
                                     // aggregation alone).  The multi-GPU modes keep the host set-up.
    bool dev_aggregation = false;    // SGO_AMG_AGG=device: the device set-up aggregates on the device too (a parallel independent-set
                                     // aggregation: weaker hierarchies, measured; default: the host's greedy aggregation)
    bool lag_on = true;              // SGO_AMG_LAG=0: the coarse operators are refreshed before every solve
    double lag_tau = 0.006;          // SGO_AMG_LAG_TAU
    double lag_slope = 0.0;          // SGO_AMG_LAG_SLOPE (test hook; 0: unset)
    int lag_force_from = -1;         // SGO_AMG_LAG_FORCE (calibration hook, scripts/lag_calib.py)
    int rebuild_cost = 150;          // SGO_AMG_REBUILD_COST (sweep knob): what the count rules take a set-up to be worth, in PCG iterations
    bool force_rebuild = false;      // SGO_AMG_FORCE_REBUILD (test hook): the hierarchy is re-made before the call's first solve
  } knobs;
  bool test_fail_trial_build = false;
  bool in_optimize = false;       // inside sgo_optimize_gn (build_amg: which set-up a rebuild takes)
  bool floor_seen = false;        // a solve of the current call was accepted at the floating-point floor (run_pcg: a shorter stagnation window)
  AmgKeptAgg kept_agg;            // (what such a rebuild keeps: host copies, taken from the hierarchy before it is destroyed)
  double* d_poses = nullptr;
  int* d_free_id = nullptr;
  EdgeListDev el;
  Sym0Dev S0;                    // level-0 Hessian, symmetric storage (the solve's products run on this)
  Tile0Dev T0;                   // ... its tile view (ntile == 0: no tile view, products use the wave-group kernel)
  BsrDev A;                      // its logical view (multigrid set-up kernels)
  EdgeSlotsDev es;
  int *d_rowptr = nullptr, *d_eidx = nullptr, *d_hrowptr = nullptr;   // compact-slot row pointers, slot -> edge, logical row pointers
                                                                       // (k_row_strength: set-up pipeline and row-owner mode)
  double *d_dgb = nullptr, *d_b = nullptr, *d_x = nullptr, *d_r = nullptr, *d_z = nullptr, *d_p = nullptr,
         *d_q = nullptr, *d_s1 = nullptr, *d_s2 = nullptr, *d_e2 = nullptr;
  double* d_partials = nullptr;   // [3][kMaxPartials]
  double* d_hist = nullptr;       // [SGO_MAX_ITERS + 2][2] chi2 history
  PcgScalars* d_S = nullptr;
  double* d_lanczos = nullptr;    // [kLanczosMax][3] alpha, beta, r.z per PCG iteration of the last solve (env SGO_LANCZOS; lives in the graph arena)
  int pcg_exec_key = 0;           // what the captured PCG iteration contains (overlay term)
  PcgScalars* h_S = nullptr;      // pinned
  double* h_hist = nullptr;       // pinned
  bool linearized = false;

  Amg* amg = nullptr;             // non-null when the AMG preconditioner is active
  bool amg_pending = false;       // the hierarchy is built on first use (graphs that optimize() through `direct`)
  bool rows_pending = false;      // ... and so are the row plan / level-0 structures of the PCG path (build_structure)
  std::vector<uint8_t> lz_fixed;  // what that deferred build needs of the caller's arrays
  std::vector<int32_t> lz_ei, lz_ej;
  Direct* direct = nullptr;       // small-graph path: optimize() is one launch (sgo_direct.h)
  std::string direct_why;         // why the last graph did not qualify for it
  Mfront* mf = nullptr;           // mid-size path: optimize() through the multifrontal factorisation (sgo_mfront.h)
  std::string mf_why;             // why the last graph did not qualify for it
  int mf_order_hint = -1;         // row order | poses << 1 of the last graph that took it (the next set-up of a grown graph analyses that order alone)
  DirectResult* d_dres = nullptr;
  DirectResult* h_dres = nullptr; // pinned
  double* d_zparts = nullptr;     // [2][kMaxPartials] partials of r.z from the cycle's last kernel
  std::string solver_desc;
  std::string solver_text;        // what sgo_solver_description hands out

  hipGraphExec_t pcg_exec = nullptr;
  int pcg_exec_chunk = 0;
  int pcg_pred = 0;               // PCG iterations of the previous solve (prediction for the next)
  bool pcg_stalled = false;       // run_pcg ended the last solve because r.r had not reached a new minimum for pcg_stall_window iterations
  int pcg_stall_window = 0;
  double tol_scale = 1.0;         // < 1 on chain-like graphs (see sgo_set_graph_se2)
  double* d_xprev = nullptr;      // the previous Gauss-Newton step of the running sgo_optimize_gn (PCG warm start)
  bool warm_valid = false;
  double bb_ref = 0.0;            // |b|^2 of the first solve of the running sgo_optimize_gn (0: relative tolerance only)
  double tol_cap = 0.0;           // loosest relative tolerance the absolute criterion may reach (0: off; opts.pcg_tol_cap)
  int pcg_softcap = 0;            // > 0: iteration cap of the next solve (sgo_optimize_gn: stale-hierarchy bail-out)
  double amg_theta_scale = 1.0;   // strength thresholds of the next hierarchy build, as a factor (halved when a hierarchy's first solve stalls)
  // level 0's multigrid host analysis running ahead on a helper thread (build_structure starts it, build_amg joins it)
  AmgHostL0* l0_pre = nullptr;
  std::thread l0_thread;
  std::vector<double> l0_w;
  int amg_best = 0;               // fewest PCG iterations seen with the current hierarchy (0: none yet);
                                  // kept across optimize() calls so that a hierarchy adapted to other poses is noticed
  PcgScalars* h_S2 = nullptr;     // pinned [2]: pipelined read-back of the stop flag (plain launches)
  double* h_pose_stage = nullptr; // pinned staging of the pose uploads (sgo_set_poses / sgo_update_graph_se2): a pageable source makes
  size_t pose_stage_cap = 0;      // the runtime pin and unpin the caller's buffer per call -- 10-20 ms every few calls on fresh buffers
  PcgScalars* h_Sz = nullptr;     // pinned: the scalars as the last k_update_p left them (RecDev::mirror; hipGraph replay)
  PcgScalars* d_Sz = nullptr;     // ... its device address
  hipEvent_t ev_S[2] = {nullptr, nullptr};

  // incremental re-initialisation (sgo_update_graph_se2, sgo_overlay.h): V / E above count the appended part too, n stays
  // the resident structure's rows
  Overlay ov;
  // PCG iterations of the FIRST solve of an optimize() call -- always run at pcg_tol relative to its own right-hand side, hence
  // comparable from call to call (the later solves of a call stop at the absolute accuracy of the first and their counts follow
  // how far the right-hand side has shrunk): of the first call on the resident structure alone, and of the latest call.  An
  // overlay whose first solve costs too many more is dropped for a full set-up (counts only).
  int its_base = 0;
  int its_last = 0;
  std::string update_note;        // what the last sgo_update_graph_se2 did (sgo_solver_description)

  // profiling
  struct Rec { int kid; hipEvent_t a, b; };
  std::vector<hipEvent_t> ev_pool;
  std::vector<hipEvent_t> iter_events;   // time stamps of sgo_optimize_gn, reused across calls
  std::vector<Rec> pending;
  double prof_ms[K_COUNT] = {0};
  int64_t prof_launches[K_COUNT] = {0};
  double prof_bytes[K_COUNT] = {0};
  std::vector<float> prof_samples[K_COUNT];   // single-launch durations (ms), the first kProfSamples per slot
  static constexpr size_t kProfSamples = 16384;
  void* amg_scope = nullptr;  // Scope* of the AMG launch being bracketed
  double prof_null_ms = -1.0; // time of an empty event bracket on this stream (calibration)
};

namespace sgo {

// does this graph's solve run sharded over the ranks of the context's communicator (or of the rank-emulation hook)?
inline bool multi_rank(const sgo_ctx* c) { return (c->comm.nranks > 1 || c->comm.active()) && !c->replicated; }

extern thread_local std::string g_err;   // error text of context-free calls (sgo_last_error(NULL))

inline double wall_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Static-partition parallel loop over [0, n) on the host pool (structure build only).
template <class F>
void parallel_for(int n, F&& fn) {
  host_parallel_for(n, 8192, [&fn](int lo, int hi, int) { fn(lo, hi); });
}

#define HIP_TRY(ctx, expr)                                                                     \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                          \
      return SGO_EHIP;                                                                         \
    }                                                                                          \
  } while (0)

template <class T>
int dalloc(sgo_ctx* c, T** p, size_t count) {
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  void* q = c->graph_arena.take(bytes);
  if (!q) {
    c->err = "out of device memory (" + std::to_string(bytes) + " bytes)";
    return SGO_ENOMEM;
  }
  *p = (T*)q;
  return SGO_OK;
}

template <class T>
int upload(sgo_ctx* c, T** p, const std::vector<T>& v) {
  int rc = dalloc(c, p, v.size());
  if (rc) return rc;
  if (!v.empty()) HIP_TRY(c, hipMemcpyAsync(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
  return SGO_OK;
}

// Host staging buffer WITHOUT value-initialisation: the structure build writes every element it
// later reads, and zero-filling ~300 MB of std::vector storage was a third of its time on C4.
// The memory comes from the context's staging arena, which is kept between sgo_set_graph_se2 calls
// (the reference re-initialises a slowly growing graph before every optimize(20)): no mmap / page
// faults / munmap of ~300 MB per call.
template <class T>
struct HostBuf {
  T* p = nullptr;
  size_t n = 0;
  HostBuf(HostArena& a, size_t count) : p((T*)a.take(count * sizeof(T))), n(count) {}
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  T* data() { return p; }
  size_t size() const { return n; }
};
template <class T>
int upload(sgo_ctx* c, T** p, const HostBuf<T>& v) {
  int rc = dalloc(c, p, v.n);
  if (rc) return rc;
  if (v.n) HIP_TRY(c, hipMemcpyAsync(*p, v.p, v.n * sizeof(T), hipMemcpyHostToDevice, c->stream));
  return SGO_OK;
}

// ---- profiling: HIP events around each launch on the ctx stream (sgo_solve.cpp) ------------
hipEvent_t get_event(sgo_ctx* c);
void prof_calibrate(sgo_ctx* c);
void prof_flush(sgo_ctx* c);
// Brackets ONE kernel launch (default): the launch itself carries the events (SGO_LAUNCH ->
// hipExtLaunchKernelGGL), so the measured time is the kernel's dispatch-to-completion time, as in
// rocprofv3's kernel trace.  multi = true brackets a whole launch sequence with two event records.
struct Scope {
  sgo_ctx* c;
  int kid;
  bool multi;
  hipEvent_t a = nullptr, b = nullptr;
  Scope(sgo_ctx* c_, int kid_, double bytes, bool multi_ = false) : c(c_), kid(kid_), multi(multi_) {
    if (!c->opts.profile) return;
    if (c->prof_null_ms < 0.0) prof_calibrate(c);
    c->prof_launches[kid]++;
    c->prof_bytes[kid] += bytes;
    a = get_event(c);
    if (multi) {
      hipEventRecord(a, c->stream);
    } else {
      b = get_event(c);
      tl_launch_ev.start = a;
      tl_launch_ev.stop = b;
    }
  }
  ~Scope() {
    if (!a) return;
    if (multi) {
      b = get_event(c);
      hipEventRecord(b, c->stream);
    } else if (tl_launch_ev.start == a) {  // no launch consumed the events: drop the sample
      tl_launch_ev = LaunchEvents();
      c->prof_launches[kid]--;
      c->ev_pool.push_back(a);
      c->ev_pool.push_back(b);
      return;
    }
    c->pending.push_back({kid, a, b});
    if (c->pending.size() >= 2048) prof_flush(c);
  }
};

// ---- host-only row plan (sgo_plan.cpp) ------------------------------------------------------
// The hessian index map of g2o -- free active vertices in ascending id -- is what the API speaks
// (c->free_id, sgo_free_ids, sgo_linearize, ...).  Internally the rows are numbered along a Hilbert
// curve through the initial poses (c->row_of_asc maps one to the other), which makes the symmetric
// storage of Sym0Dev work: the endpoints of almost every edge end up a few hundred rows apart.
// Host-only plan of the level-0 rows (no GPU involved; also behind sgo_plan_rows for the multi-process tests):
// g2o's hessian order, the internal Hilbert row order, the compact slot positions of every edge and the tiles.
struct RowPlan {
  int n = 0, ns = 0;
  std::vector<int> free_id;      // hessian index (free active vertices in ascending id) -> vertex id
  std::vector<int> row_of_asc;   // hessian index -> internal row
  std::vector<int> row_vertex;   // internal row -> vertex id
  std::vector<int> hpos;         // vertex id -> internal row (-1: fixed or inactive)
  std::vector<int> rowptr;       // [n + 1] compact slots of row r
  std::vector<int> pos_i, pos_j; // [E] slot of edge e in the row of its first / second endpoint (-1: none)
  std::vector<int> col;          // [ns] column (internal row) of the slot, -1: fixed column
  std::vector<TileDesc> tiles;   // row0 / row1 filled in
  std::vector<int> tile_of_row;
  std::vector<int> chunk_cnt;    // scratch of the slot placement ([chunk][row])
  int tile_lds = 0;
  bool tiles_ok = true;
};

int plan_rows_order(int V, const double* poses, const uint8_t* fixed, int E, const int32_t* ei, const int32_t* ej,
                    std::string* err, RowPlan& P, const std::vector<int>* known_free = nullptr, const double* order_xy = nullptr);
bool plan_order_positions(int V, const double* poses, const uint8_t* fixed, int E, const int32_t* ei, const int32_t* ej,
                          const double* meas, std::vector<double>& xy);
void plan_rows_tiles(int tile_div, RowPlan& P);
int plan_rows(int V, const double* poses, const uint8_t* fixed, int E, const int32_t* ei, const int32_t* ej, int tile_div,
              std::string* err, RowPlan& P, const double* meas = nullptr);
// One tile per CU of an MI355X.  A constant, not the device's CU count: every rank of a multi-GPU run -- and the
// host-only sgo_plan_rows -- must cut the same tiles whatever device it sits on.
constexpr int kTileDiv = 256;
constexpr long long kSmallGraphPairs = 150000;   // below: the wave-group kernel instead of the tile kernel (sgo_plan.cpp)

// ---- device-resident graph (sgo_structure.cpp) ----------------------------------------------
int upload_poses(sgo_ctx* c, const double* poses, int V);   // through the pinned staging buffer; returns after the copy was queued AND the caller's array is no longer needed
void l0_join(sgo_ctx* c, bool keep);
inline void l0_discard(sgo_ctx* c) { l0_join(c, false); }
void free_graph(sgo_ctx* c);
int halo_reserve(sgo_ctx* c, size_t packet_doubles);   // exchange buffers for packets of this many doubles per rank
int build_edges(sgo_ctx* c, int V, const double* poses, const uint8_t* fixed, int E, const int32_t* ei, const int32_t* ej,
                const double* meas, const double* info, const double* phi);
int build_structure(sgo_ctx* c, int V, const double* poses, const uint8_t* fixed, int E, const int32_t* ei,
                    const int32_t* ej);

// ---- Gauss-Newton building blocks (sgo_solve.cpp) -------------------------------------------
int do_chi2(sgo_ctx* c, double* d_out2, double* d_e2);
int start_pcg(sgo_ctx* c, int grid);
int do_linearize(sgo_ctx* c);
int do_spmv(sgo_ctx* c, const double* x, double* y, bool dot, const PcgScalars* S, int* grid_out);
int run_pcg(sgo_ctx* c);
void read_call_knobs(sgo_ctx* c);   // the environment knobs of a solve, once per entry-point call
int build_amg(sgo_ctx* c, bool keep_old = false, bool keep_agg = false);
int revert_amg(sgo_ctx* c);
std::string multi_gpu_description(const sgo_ctx* c);
int build_rows(sgo_ctx* c, const double* poses, const uint8_t* fixed, const int32_t* ei, const int32_t* ej);
int ensure_rows(sgo_ctx* c);
int ensure_amg(sgo_ctx* c);
int vec_to_device(sgo_ctx* c, const double* host_asc, double* dev);
int vec_from_device(sgo_ctx* c, const double* dev, double* host_asc);
int check_graph(sgo_ctx* c);
int optimize_gn(sgo_ctx* c, int32_t iters, sgo_stats* out);   // sgo_optimize_gn behind its argument checks
double debug_spmv0_us(sgo_ctx* c, int mode, int variant, int reps);

}  // namespace sgo

// include/sgo.h promises that no exception crosses the C boundary: host allocations (std::vector, std::string, new)
// inside an entry point are caught here and reported as SGO_ENOMEM / SGO_EINVAL
#define SGO_CATCH(ctx)                                                              \
  catch (const std::bad_alloc&) {                                                   \
    if (ctx) (ctx)->err = "out of host memory";                                     \
    return SGO_ENOMEM;                                                              \
  }                                                                                 \
  catch (const std::exception& e_) {                                                \
    if (ctx) (ctx)->err = std::string("internal error: ") + e_.what();              \
    return SGO_EINVAL;                                                              \
  }                                                                                 \
  catch (...) {                                                                     \
    if (ctx) (ctx)->err = "internal error (unknown exception)";                     \
    return SGO_EINVAL;                                                              \
  }
