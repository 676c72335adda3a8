// sgo_api.cpp -- host side of libsgo: the C-ABI of include/sgo.h over the HIP kernels.
//
// Mirrors the control flow of g2o's SparseOptimizer::optimize() with
// OptimizationAlgorithmGaussNewton as sparse-gslam configures it
// (src/sparse_gslam/src/graphs.cpp:17-23; called at submap_loop_closer.cpp:286-288 and
// log_runner.cpp:203-204):
//     for k in 0..iters:  computeActiveErrors; buildSystem; solve; update
// with the linear solve done by preconditioned CG on the device instead of LinearSolverEigen.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "sgo_amg.h"
#include "sgo_direct.h"
#include "sgo_comm.h"
#include "sgo_internal.h"

using namespace sgo;

namespace {

thread_local std::string g_err;  // for ctx == NULL

double wall_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Static-partition parallel loop over [0, n) on the host pool (structure build only).
template <class F>
void parallel_for(int n, F&& fn) {
  host_parallel_for(n, 8192, [&fn](int lo, int hi, int) { fn(lo, hi); });
}

constexpr double kPi = 3.14159265358979323846;
double normalize_theta_h(double t) {
  if (t >= -kPi && t < kPi) return t;
  double m = std::floor(t / (2 * kPi));
  t = t - m * 2 * kPi;
  if (t >= kPi) t -= 2 * kPi;
  if (t < -kPi) t += 2 * kPi;
  return t;
}

}  // namespace

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
  double* d_poses = nullptr;
  int* d_free_id = nullptr;
  EdgeListDev el;
  Sym0Dev S0;                    // level-0 Hessian, symmetric storage (the solve's products run on this)
  Tile0Dev T0;                   // ... its tile view (ntile == 0: no tile view, products use the wave-group kernel)
  BsrDev A;                      // its logical view (multigrid set-up kernels)
  EdgeSlotsDev es;
  double *d_dgb = nullptr, *d_b = nullptr, *d_x = nullptr, *d_r = nullptr, *d_z = nullptr, *d_p = nullptr,
         *d_q = nullptr, *d_s1 = nullptr, *d_s2 = nullptr, *d_e2 = nullptr;
  double* d_partials = nullptr;   // [3][kMaxPartials]
  double* d_hist = nullptr;       // [SGO_MAX_ITERS + 2][2] chi2 history
  PcgScalars* d_S = nullptr;
  PcgScalars* h_S = nullptr;      // pinned
  double* h_hist = nullptr;       // pinned
  bool linearized = false;

  Amg* amg = nullptr;             // non-null when the AMG preconditioner is active
  bool amg_pending = false;       // the hierarchy is built on first use (graphs that optimize() through `direct`)
  bool rows_pending = false;      // ... and so are the row plan / level-0 structures of the PCG path (build_structure)
  std::vector<uint8_t> lz_fixed;  // what that deferred build needs of the caller's arrays
  std::vector<int32_t> lz_ei, lz_ej;
  int cu_count = 0;               // compute units of the device (tiles per launch)
  Direct* direct = nullptr;       // small-graph path: optimize() is one launch (sgo_direct.h)
  std::string direct_why;         // why the last graph did not qualify for it
  DirectResult* d_dres = nullptr;
  DirectResult* h_dres = nullptr; // pinned
  double* d_zparts = nullptr;     // [2][kMaxPartials] partials of r.z from the cycle's last kernel
  std::string solver_desc;
  std::string solver_text;        // what sgo_solver_description hands out

  hipGraphExec_t pcg_exec = nullptr;
  int pcg_exec_chunk = 0;
  int pcg_pred = 0;               // PCG iterations of the previous solve (prediction for the next)
  double tol_scale = 1.0;         // < 1 on chain-like graphs (see sgo_set_graph_se2)
  double* d_xprev = nullptr;      // the previous Gauss-Newton step of the running sgo_optimize_gn (PCG warm start)
  bool warm_valid = false;
  bool amg_skip_update = false;   // this solve reuses the hierarchy's values of the previous one (sgo_optimize_gn's late iterations)
  double bb_ref = 0.0;            // |b|^2 of the first solve of the running sgo_optimize_gn (0: relative tolerance only)
  double tol_cap = 0.0;           // loosest relative tolerance the absolute criterion may reach (0: off; opts.pcg_tol_cap)
  int pcg_softcap = 0;            // > 0: iteration cap of the next solve (sgo_optimize_gn: stale-hierarchy bail-out)
  // level 0's multigrid host analysis running ahead on a helper thread (build_structure starts it, build_amg joins it)
  AmgHostL0* l0_pre = nullptr;
  std::thread l0_thread;
  std::vector<double> l0_w;
  int amg_best = 0;               // fewest PCG iterations seen with the current hierarchy (0: none yet);
                                  // kept across optimize() calls so that a hierarchy adapted to other poses is noticed
  PcgScalars* h_S2 = nullptr;     // pinned [2]: pipelined read-back of the stop flag
  hipEvent_t ev_S[2] = {nullptr, nullptr};

  // profiling
  struct Rec { int kid; hipEvent_t a, b; };
  std::vector<hipEvent_t> ev_pool;
  std::vector<hipEvent_t> iter_events;   // time stamps of sgo_optimize_gn, reused across calls
  std::vector<Rec> pending;
  double prof_ms[K_COUNT] = {0};
  int64_t prof_launches[K_COUNT] = {0};
  double prof_bytes[K_COUNT] = {0};
  void* amg_scope = nullptr;  // Scope* of the AMG launch being bracketed
  double prof_null_ms = -1.0; // time of an empty event bracket on this stream (calibration)
};

namespace {

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

// Joins the helper thread of the level-0 analysis; `keep`: leave its result for build_amg, otherwise drop it.
void l0_join(sgo_ctx* c, bool keep) {
  if (c->l0_thread.joinable()) c->l0_thread.join();
  if (!keep && c->l0_pre) {
    amg_host_l0_free(c->l0_pre);
    c->l0_pre = nullptr;
  }
  // (c->l0_w keeps its storage: a fresh 17-MB vector per call is 4000 page faults on the set-up's critical path)
}
void l0_discard(sgo_ctx* c) { l0_join(c, false); }

void free_graph(sgo_ctx* c) {
  l0_discard(c);
  if (c->pcg_exec) {
    hipGraphExecDestroy(c->pcg_exec);
    c->pcg_exec = nullptr;
  }
  if (c->amg) {
    amg_destroy(c->amg);
    c->amg = nullptr;
  }
  if (c->direct) {
    direct_destroy(c->direct);
    c->direct = nullptr;
  }
  c->amg_pending = false;
  c->rows_pending = false;
  c->amg_arena.rewind();
  c->graph_arena.rewind();   // the caller has synchronised the stream: nothing in flight reads these arrays
  c->pcg_pred = 0;
  c->A = BsrDev();
  c->S0 = Sym0Dev();
  c->T0 = Tile0Dev();
  c->es = EdgeSlotsDev();
  c->el = EdgeListDev();
  c->d_xprev = nullptr;
  c->warm_valid = false;
  c->has_graph = false;
  c->linearized = false;
}

// ---- profiling: HIP events around each launch on the ctx stream ---------------------------
hipEvent_t get_event(sgo_ctx* c) {
  if (!c->ev_pool.empty()) {
    hipEvent_t e = c->ev_pool.back();
    c->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}
// An event pair with nothing between them still measures a few us on the queue; it is measured
// once (median of 33 empty brackets) and REPORTED (sgo_profile_overhead_ms) as the bias bound of
// the per-kernel averages relative to rocprofv3's kernel durations -- it is not subtracted.
void prof_calibrate(sgo_ctx* c) {
  if (c->prof_null_ms >= 0.0) return;
  std::vector<float> v;
  for (int k = 0; k < 33; ++k) {
    hipEvent_t a = get_event(c), b = get_event(c);
    hipEventRecord(a, c->stream);
    hipEventRecord(b, c->stream);
    hipStreamSynchronize(c->stream);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) == hipSuccess) v.push_back(ms);
    c->ev_pool.push_back(a);
    c->ev_pool.push_back(b);
  }
  std::sort(v.begin(), v.end());
  c->prof_null_ms = v.empty() ? 0.0 : v[v.size() / 2];
}
void prof_flush(sgo_ctx* c) {
  if (c->pending.empty()) return;
  hipStreamSynchronize(c->stream);
  for (auto& r : c->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) c->prof_ms[r.kid] += ms;
    c->ev_pool.push_back(r.a);
    c->ev_pool.push_back(r.b);
  }
  c->pending.clear();
}
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

// ---- algorithmic bytes per launch ------------
// Level-0 product: every stored off-diagonal block once with one index (76 B per edge), the diagonal
// block (48 B), the operand and the result (24 B each) per row; + the right-hand side (RESID, JACOBI)
// and the block-diagonal inverse (JACOBI).
double bytes_spmv0(const Sym0Dev& A, int mode) {
  return 76.0 * A.npairs + (96.0 + (mode != S0_AX ? 24.0 : 0.0) + (mode == S0_JACOBI ? 48.0 : 0.0)) * A.n;
}
// linearise + assemble: the row-parallel design reads each edge's operands once per endpoint row
// (2 x 128 B: indices, inverse measurement, information, two poses), writes the off-diagonal block once
// (72 B) and 72 B of (diagonal block, b) per row
double bytes_linearize(const sgo_ctx* c) { return 128.0 * c->S0.ncs + 72.0 * c->S0.nu + 72.0 * c->n; }
double bytes_chi2(const sgo_ctx* c) { return 96.0 * c->E + 24.0 * c->V; }

// Hilbert-curve index of the cell (x, y) of a 2^order x 2^order grid.
uint32_t hilbert_index(uint32_t x, uint32_t y, int order) {
  uint32_t d = 0;
  for (uint32_t s = 1u << (order - 1); s > 0; s >>= 1) {
    const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
    d += s * s * ((3u * rx) ^ ry);
    if (ry == 0) {   // rotate the quadrant
      if (rx == 1) {
        x = s - 1 - (x & (s - 1));
        y = s - 1 - (y & (s - 1));
      } else {
        x &= s - 1;
        y &= s - 1;
      }
      const uint32_t t = x;
      x = y;
      y = t;
    } else {
      x &= s - 1;
      y &= s - 1;
    }
  }
  return d;
}

// ---- structure build: SparseOptimizer::initializeOptimization + BlockSolver::buildStructure -
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

// Row plan, first half: hessian order, internal (Hilbert) row order, compact slots per row.
// `known_free`: the hessian order when the caller has already validated the edge list and listed the free active
// vertices (build_edges does both for the chi2 path): the pass over the edges is then not repeated.
int plan_rows_order(int V, const double* poses, const uint8_t* fixed, int E, const int32_t* ei, const int32_t* ej,
                    std::string* err, RowPlan& P, const std::vector<int>* known_free = nullptr) {
  const bool verbose = std::getenv("SGO_VERBOSE") != nullptr && (E > 200000 || std::atoi(std::getenv("SGO_VERBOSE")) > 1);
  double tl = wall_s();
  auto lap = [&](const char* what) {
    const double t = wall_s();
    if (verbose) std::fprintf(stderr, "[sgo]   plan %-18s %.1f ms\n", what, 1e3 * (t - tl));
    tl = t;
  };
  if (known_free) {
    P.free_id = *known_free;
  } else {
    std::vector<int> deg(V, 0);
    for (int e = 0; e < E; ++e) {
      int a = ei[e], b = ej[e];
      if (a < 0 || a >= V || b < 0 || b >= V) {
        *err = "edge " + std::to_string(e) + " references a vertex outside [0, V)";
        return SGO_EINVAL;
      }
      if (a == b) {
        *err = "edge " + std::to_string(e) + " is a self edge";
        return SGO_EINVAL;
      }
      deg[a]++;
      deg[b]++;
    }
    // active free vertices in ascending id = g2o's hessian order (initializeOptimization)
    P.free_id.clear();
    for (int v = 0; v < V; ++v)
      if (!fixed[v] && deg[v] > 0) P.free_id.push_back(v);
  }
  const int n = P.n = (int)P.free_id.size();
  lap("degrees");
  // internal row order: Hilbert index of the initial position (ties and non-finite poses: by id)
  P.hpos.assign(V, -1);
  P.row_vertex.assign(n, 0);
  P.row_of_asc.assign(n, 0);
  {
    double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
    for (int h = 0; h < n; ++h) {
      const double* q = poses + 3 * (size_t)P.free_id[h];
      for (int d = 0; d < 2; ++d)
        if (std::isfinite(q[d])) {
          lo[d] = std::min(lo[d], q[d]);
          hi[d] = std::max(hi[d], q[d]);
        }
    }
    const double ext = std::max(hi[0] - lo[0], hi[1] - lo[1]);
    const double scale = (ext > 0.0 && std::isfinite(ext)) ? 65535.0 / ext : 0.0;
    std::vector<uint64_t> key(n);
    parallel_for(n, [&](int h0, int h1) {
      for (int h = h0; h < h1; ++h) {
        const double* q = poses + 3 * (size_t)P.free_id[h];
        uint32_t d = 0;
        if (std::isfinite(q[0]) && std::isfinite(q[1]) && scale > 0.0)
          d = hilbert_index((uint32_t)((q[0] - lo[0]) * scale), (uint32_t)((q[1] - lo[1]) * scale), 16);
        key[h] = ((uint64_t)d << 32) | (uint32_t)h;
      }
    });
    // sorted in parallel: eight chunks by std::sort, then three rounds of pairwise merges (keys are distinct: the
    // low word is the vertex's hessian index, so the order does not depend on how the work is split)
    if (n >= 65536 && HostPool::get().size() >= 4) {
      constexpr int kParts = 8;
      int cut[kParts + 1];
      for (int q = 0; q <= kParts; ++q) cut[q] = (int)((long long)n * q / kParts);
      host_parallel_for(kParts, 1, [&](int q0, int q1, int) {
        for (int q = q0; q < q1; ++q) std::sort(key.begin() + cut[q], key.begin() + cut[q + 1]);
      });
      std::vector<uint64_t> tmp(n);
      std::vector<uint64_t>*src = &key, *dst = &tmp;
      for (int width = 1; width < kParts; width *= 2) {
        const int npairs = kParts / (2 * width);
        host_parallel_for(npairs, 1, [&](int q0, int q1, int) {
          for (int q = q0; q < q1; ++q) {
            const int a = cut[2 * width * q], m = cut[2 * width * q + width], b = cut[2 * width * (q + 1)];
            std::merge(src->begin() + a, src->begin() + m, src->begin() + m, src->begin() + b, dst->begin() + a);
          }
        });
        std::swap(src, dst);
      }
      if (src != &key) key.swap(tmp);
    } else {
      std::sort(key.begin(), key.end());
    }
    for (int r = 0; r < n; ++r) {
      const int h = (int)(key[r] & 0xffffffffu);
      P.row_of_asc[h] = r;
      P.row_vertex[r] = P.free_id[h];
      P.hpos[P.free_id[h]] = r;
    }
  }
  const std::vector<int>& hpos = P.hpos;
  lap("hilbert order");
  // compact slots: per row one slot per incident edge (edge order within the row)
  // A stable counting sort of the edge endpoints by row (a row's slots in edge order), in parallel over contiguous
  // chunks of the edge list: per-chunk counts per row, offsets by a prefix over (row, chunk), then every chunk places
  // its own slots -- the same layout as one sequential pass (which took 6 ms of the critical path on C4, 94 ms on C5).
  std::vector<int>& rowptr = P.rowptr;
  rowptr.assign((size_t)n + 1, 0);
  const int nchunk = (E >= 200000 && n > 0) ? std::max(1, std::min(HostPool::get().size(), 16)) : 1;
  auto chunk_lo = [&](int t) { return (int)((long long)E * t / nchunk); };
  std::vector<int>& ccnt = P.chunk_cnt;   // [chunk][row], turned into the chunk's first position per row
  ccnt.assign((size_t)nchunk * std::max(n, 1), 0);
  host_parallel_for(nchunk, 1, [&](int t0, int t1, int) {
    for (int t = t0; t < t1; ++t) {
      int* c = ccnt.data() + (size_t)t * std::max(n, 1);
      for (int e = chunk_lo(t); e < chunk_lo(t + 1); ++e) {
        const int hi = hpos[ei[e]], hj = hpos[ej[e]];
        if (hi >= 0) c[hi]++;
        if (hj >= 0) c[hj]++;
      }
    }
  });
  parallel_for(n, [&](int r0, int r1) {
    for (int r = r0; r < r1; ++r) {
      int tot = 0;
      for (int t = 0; t < nchunk; ++t) tot += ccnt[(size_t)t * n + r];
      rowptr[r + 1] = tot;
    }
  });
  for (int r = 0; r < n; ++r) rowptr[r + 1] += rowptr[r];
  const int ns = P.ns = rowptr[n];
  parallel_for(n, [&](int r0, int r1) {
    for (int r = r0; r < r1; ++r) {
      int at = rowptr[r];
      for (int t = 0; t < nchunk; ++t) {
        const int c = ccnt[(size_t)t * n + r];
        ccnt[(size_t)t * n + r] = at;
        at += c;
      }
    }
  });
  P.pos_i.assign(E, -1);
  P.pos_j.assign(E, -1);
  P.col.resize((size_t)std::max(ns, 1));
  host_parallel_for(nchunk, 1, [&](int t0, int t1, int) {
    for (int t = t0; t < t1; ++t) {
      int* fill = ccnt.data() + (size_t)t * std::max(n, 1);
      for (int e = chunk_lo(t); e < chunk_lo(t + 1); ++e) {
        const int hi = hpos[ei[e]], hj = hpos[ej[e]];
        if (hi >= 0) {
          P.col[fill[hi]] = hj;
          P.pos_i[e] = fill[hi]++;
        }
        if (hj >= 0) {
          P.col[fill[hj]] = hi;
          P.pos_j[e] = fill[hj]++;
        }
      }
    }
  });
  lap("slot positions");
  return SGO_OK;
}

// Row plan, second half: the tiles of the level-0 product kernel.
void plan_rows_tiles(int tile_div, RowPlan& P) {
  const bool verbose = std::getenv("SGO_VERBOSE") != nullptr && (P.ns > 400000 || std::atoi(std::getenv("SGO_VERBOSE")) > 1);
  double tl = wall_s();
  auto lap = [&](const char* what) {
    const double t = wall_s();
    if (verbose) std::fprintf(stderr, "[sgo]   plan %-18s %.1f ms\n", what, 1e3 * (t - tl));
    tl = t;
  };
  const int n = P.n, ns = P.ns;
  const std::vector<int>&rowptr = P.rowptr, &col = P.col;
  // ---- tiles (Tile0Dev): consecutive rows, cut so that the blocks are spread evenly over ~2 tiles per CU
  // and a tile's LDS -- operand slice + halo, owned sums, one staging slot per intra-tile transposed slot --
  // fits kTileLdsMax.  A pair inside a tile stores its block with the lower row only (the other row's slot
  // is TRANSPOSED); every other slot with a free column is OWNED.
  std::vector<TileDesc>& tiles = P.tiles;
  std::vector<int>& tile_of_row = P.tile_of_row;
  tile_of_row.assign(std::max(n, 1), 0);
  P.tiles_ok = true;
  {
    int lds_budget = kTileLdsMax - 1024;
    if (const char* e = std::getenv("SGO_TILE_LDS")) lds_budget = std::atoi(e);
    if (const char* e = std::getenv("SGO_TILE_DIV")) tile_div = std::max(1, std::atoi(e));
    long long nblk = 0;
    for (int k = 0; k < ns; ++k) nblk += col[k] >= 0;
    // A tile costs what it STORES (measured, C4: 4.8 cycles per stored block + 90 per wave group, against 20-47 k
    // cycles per tile when tiles were cut by slot count): its slots with a free column minus its intra-tile pairs,
    // which are stored once.  Tiles are cut greedily to a block target; the target is re-derived from the total
    // the cut produced (pairs that straddle two tiles are stored twice, so the total depends on the cut) until
    // the tiles number one per CU.
    long long target = std::max<long long>(512, (nblk / 2 * 5 / 4 + tile_div - 1) / tile_div);   // stored blocks per tile
    long long starget = std::max<long long>(512, (nblk / 2 + tile_div - 1) / tile_div);          // pairs per tile (fallback)
    std::vector<int> mark(std::max(n, 1), -1);
    for (int attempt = 0; attempt < 7; ++attempt) {
      long long lds = 0;
      bool too_many = false;
      if (attempt == 0) {
        // equal stored blocks per tile, a tile closed early when its LDS need (tracked exactly while rows are added:
        // rows, distinct outside columns, intra-tile pairs) would pass the budget
        // The greedy cut runs over kSeg row segments of equal slot counts in parallel (a segment starts a tile): one
        // sequential pass over the 2 M slots of C4 took 5-6 ms of the set-up's critical path.
        constexpr int kSeg = 8;
        const int nseg = (n >= 32768 && HostPool::get().size() >= 4) ? kSeg : 1;
        int seg_row[kSeg + 1];
        for (int q = 0; q <= nseg; ++q) {
          const long long want = (long long)ns * q / nseg;
          seg_row[q] = q == nseg ? n : (int)(std::lower_bound(rowptr.begin(), rowptr.begin() + n, (int)want) - rowptr.begin());
        }
        std::vector<std::vector<int>> seg_mark((size_t)nseg);
        std::vector<std::vector<TileDesc>> seg_tiles((size_t)nseg);
        long long seg_total[kSeg];
        int seg_by_target[kSeg], seg_stamp[kSeg];
        for (int q = 0; q < nseg; ++q) seg_stamp[q] = 1 << 20;
        for (int pass = 0; pass < 4; ++pass) {
          host_parallel_for(nseg, 1, [&](int q0, int q1, int) {
            for (int q = q0; q < q1; ++q) {
              std::vector<int>& mk = seg_mark[q];
              if (mk.empty()) mk.assign(std::max(n, 1), -1);
              std::vector<TileDesc>& out = seg_tiles[q];
              out.clear();
              int& stamp = seg_stamp[q];
              long long total = 0;
              int r = seg_row[q], by_target = 0;
              const int rend = seg_row[q + 1];
              while (r < rend) {
                TileDesc T{};
                T.row0 = r;
                ++stamp;
                long long blocks = 0, halo = 0, staged = 0;
                while (r < rend && (r == T.row0 || (blocks < target && r - T.row0 < 4096))) {
                  long long db = 0, dh = 0, ds = 0;
                  for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
                    const int cc = col[k];
                    if (cc < 0) continue;
                    if (cc >= T.row0 && cc < r) {
                      ++ds;            // the pair is inside the tile: counted as a block with its earlier row, staged here
                    } else {
                      ++db;
                      if (mk[cc] != stamp) {
                        mk[cc] = stamp;
                        ++dh;
                      }
                    }
                  }
                  const long long back = mk[r] == stamp ? 1 : 0;   // r was an outside column of the tile's earlier rows
                  const long long need = 24 * (2 * (long long)(r + 1 - T.row0) + (halo + dh - back) + (staged + ds));
                  if (r > T.row0 && need > lds_budget) break;   // (the marks this row left carry a stamp no later tile uses)
                  blocks += db;
                  halo += dh - back;
                  staged += ds;
                  ++r;
                }
                T.row1 = r;
                total += blocks;
                by_target += blocks >= target;
                out.push_back(T);
              }
              seg_total[q] = total;
              seg_by_target[q] = by_target;
            }
          });
          tiles.clear();
          long long total = 0;
          int by_target = 0;
          for (int q = 0; q < nseg; ++q) {
            tiles.insert(tiles.end(), seg_tiles[q].begin(), seg_tiles[q].end());
            total += seg_total[q];
            by_target += seg_by_target[q];
          }
          const int K = (int)tiles.size();
          // Graphs whose halo fills the LDS long before a CU's share of the blocks is reached (long-range closures;
          // C5: ~4000 tiles of ~245 rows): every tile is as large as the LDS allows -- the fewest pairs stored twice --
          // and with many tiles per CU the uneven block counts average out over a workgroup's tiles.  No larger block
          // target changes this cut, and the slot-balanced fallback would only find smaller tiles by repeated halving.
          if (K >= 4 * tile_div && 8 * by_target < K) {
            too_many = false;   // (an earlier pass with a smaller target may have set it): the cut is verified and taken
            break;
          }
          too_many = K > tile_div && target > 512;
          if (target <= 512 || (K <= tile_div && K >= tile_div - tile_div / 32)) break;
          target = std::max<long long>(512, total / tile_div + (K > tile_div ? total / tile_div / 64 + 1 : 1));
        }
        std::fill(mark.begin(), mark.end(), -1);
      } else {
        // the block-balanced cut did not fit the LDS (its largest tiles hold the most rows + staged entries): cut by
        // slot count instead -- rows, halo and staging then vary less -- and halve the tiles until they fit
        tiles.clear();
        int r = 0;
        while (r < n) {
          TileDesc T{};
          T.row0 = r;
          long long slots = 0;
          while (r < n && (r == T.row0 || (slots < 2 * starget && r - T.row0 < 4096))) {
            slots += rowptr[r + 1] - rowptr[r];
            ++r;
          }
          T.row1 = r;
          tiles.push_back(T);
        }
      }
      for (size_t t = 0; t < tiles.size(); ++t)
        for (int q = tiles[t].row0; q < tiles[t].row1; ++q) tile_of_row[q] = (int)t;
      // exact LDS need per tile: rows + halo columns + rows + staged entries, 24 B each (tiles in parallel on the host
      // pool, every worker with its own column marks)
      bool fits = !too_many;   // more tiles than CUs because the LDS closed tiles early: the slot-balanced cut is better
      if (fits) {
        const int ntl = (int)tiles.size();
        std::vector<long long> need_t((size_t)ntl, 0);
        std::vector<unsigned char> bad_t((size_t)ntl, 0);
        host_parallel_for(ntl, 8, [&](int t0, int t1, int) {
          std::vector<int> mk(std::max(n, 1), -1);
          for (int t = t0; t < t1; ++t) {
            const TileDesc& T = tiles[t];
            long long halo = 0, staged = 0;
            for (int k = rowptr[T.row0]; k < rowptr[T.row1]; ++k) {
              const int cc = col[k];
              if (cc < 0) continue;
              if (cc >= T.row0 && cc < T.row1) {
                staged += 1;   // each intra-tile pair has two slots, one of them staged: count halves below
              } else if (mk[cc] != t) {
                mk[cc] = t;
                ++halo;
              }
            }
            staged /= 2;
            const long long rows = T.row1 - T.row0;
            need_t[t] = 24 * (2 * rows + halo + staged);
            bad_t[t] = need_t[t] > lds_budget || rows + halo > 65000 || staged > 65000;
          }
        });
        for (int t = 0; t < ntl; ++t) {
          if (bad_t[t]) fits = false;
          lds = std::max(lds, need_t[t]);
        }
      }
      P.tile_lds = (int)lds;
      if (fits) break;
      if (attempt >= 1) {
        if (starget <= 64) {
          P.tiles_ok = false;   // e.g. a hub vertex whose row alone overflows the LDS: no tile view
          break;
        }
        starget = std::max<long long>(64, starget / 2);
      }
      if (attempt == 6) P.tiles_ok = false;
    }
  }
  lap("tiles");
  if (const char* e = std::getenv("SGO_SPMV0"))
    if (!std::strcmp(e, "group")) P.tiles_ok = false;   // experiments: force the wave-group kernel
  if (!P.tiles_ok) {   // one "tile" per row range of nothing: every pair stored once, with the lower row
    tiles.clear();
    for (int r = 0; r < n; ++r) tile_of_row[r] = 0;
  }
}

int plan_rows(int V, const double* poses, const uint8_t* fixed, int E, const int32_t* ei, const int32_t* ej, int tile_div,
              std::string* err, RowPlan& P) {
  const int rc = plan_rows_order(V, poses, fixed, E, ei, ej, err, P);
  if (rc != SGO_OK) return rc;
  plan_rows_tiles(tile_div, P);
  return SGO_OK;
}

// The edge arrays, poses and chi2 buffers of a graph: all that chi2 / per-edge chi2 / the single-launch direct path
// need.  Validates the edge list and fixes the hessian order (free active vertices in ascending id).
int build_edges(sgo_ctx* c, int V, const double* poses, const uint8_t* fixed, int E, const int32_t* ei, const int32_t* ej,
                const double* meas, const double* info, const double* phi) {
  std::vector<int> deg(V, 0);
  for (int e = 0; e < E; ++e) {
    const int a = ei[e], b = ej[e];
    if (a < 0 || a >= V || b < 0 || b >= V) {
      c->err = "edge " + std::to_string(e) + " references a vertex outside [0, V)";
      return SGO_EINVAL;
    }
    if (a == b) {
      c->err = "edge " + std::to_string(e) + " is a self edge";
      return SGO_EINVAL;
    }
    deg[a]++;
    deg[b]++;
  }
  c->free_id.clear();
  for (int v = 0; v < V; ++v)
    if (!fixed[v] && deg[v] > 0) c->free_id.push_back(v);
  c->V = V;
  c->E = E;
  c->n = (int)c->free_id.size();
  int rc;
  c->el.E = E;
  double *d_meas = nullptr, *d_info = nullptr;
  if ((rc = dalloc(c, &c->el.vi, (size_t)E)) || (rc = dalloc(c, &c->el.vj, (size_t)E)) || (rc = dalloc(c, &c->el.phi, (size_t)E)) ||
      (rc = dalloc(c, &c->el.zinv, 3 * (size_t)E)) || (rc = dalloc(c, &c->el.info, 6 * (size_t)E)) ||
      (rc = dalloc(c, &d_meas, 3 * (size_t)E)) || (rc = dalloc(c, &d_info, 6 * (size_t)E)))
    return rc;
  if (E > 0) {
    HIP_TRY(c, hipMemcpyAsync(c->el.vi, ei, sizeof(int32_t) * (size_t)E, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->el.vj, ej, sizeof(int32_t) * (size_t)E, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->el.phi, phi, sizeof(double) * (size_t)E, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_meas, meas, sizeof(double) * 3 * (size_t)E, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_info, info, sizeof(double) * 6 * (size_t)E, hipMemcpyHostToDevice, c->stream));
    launch_edge_prepare(c->stream, E, d_meas, d_info, c->el.zinv, c->el.info);
  }
  if ((rc = dalloc(c, &c->d_poses, 3 * (size_t)V))) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->d_poses, poses, sizeof(double) * 3 * (size_t)V, hipMemcpyHostToDevice, c->stream));
  if ((rc = dalloc(c, &c->d_e2, (size_t)E))) return rc;
  if ((rc = dalloc(c, &c->d_partials, 3 * (size_t)kMaxPartials))) return rc;
  if ((rc = dalloc(c, &c->d_hist, 2 * (size_t)(SGO_MAX_ITERS + 2)))) return rc;
  HIP_TRY(c, hipMemsetAsync(c->d_partials, 0, sizeof(double) * 3 * kMaxPartials, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));   // the caller's arrays may go away after sgo_set_graph_se2
  return SGO_OK;
}

// Row plan, level-0 structures and vectors of the PCG path, after build_edges.
int build_structure(sgo_ctx* c, int V, const double* poses, const uint8_t* fixed, int E, const int32_t* ei,
                    const int32_t* ej) {
  const double tb0 = wall_s();
  l0_discard(c);   // a helper thread of an earlier set-up that was never consumed
  // The plan's vectors (slot positions, columns, per-chunk counters: ~30 MB on C4) keep their storage between calls of
  // this thread: the reference re-initialises its graph before every optimize(20), and fresh pages cost 2-3 ms of page
  // faults per set-up on the critical path.
  static thread_local RowPlan plan_storage;
  RowPlan& P = plan_storage;
  P.tiles.clear();
  P.tile_lds = 0;
  P.tiles_ok = true;
  const int tile_div = c->cu_count > 0 ? c->cu_count : 256;   // one tile per CU: the larger the tiles, the fewer pairs straddle two of them
  {
    // (build_edges has validated the edges and listed the free active vertices of this very graph; the lazy path of
    // graphs that took the direct solver keeps the lists too)
    const int prc = plan_rows_order(V, poses, fixed, E, ei, ej, &c->err, P, (int)c->free_id.size() == c->n && c->V == V && c->E == E ? &c->free_id : nullptr);
    if (prc != SGO_OK) return prc;
  }
  const bool verbose = c->opts.verbose && (E > 200000 || c->opts.verbose > 1);
  double tl = wall_s();
  auto lap = [&](const char* what) {
    const double t = wall_s();
    if (verbose) std::fprintf(stderr, "[sgo]   build %-17s %.1f ms\n", what, 1e3 * (t - tl));
    tl = t;
  };
  c->row_of_asc = P.row_of_asc;
  const int n = P.n, ns = P.ns;
  const std::vector<int>&row_vertex = P.row_vertex, &rowptr = P.rowptr, &pos_i = P.pos_i, &pos_j = P.pos_j;
  std::vector<int>& col = P.col;
  std::vector<TileDesc>& tiles = P.tiles;
  std::vector<int>& tile_of_row = P.tile_of_row;
  std::vector<int> hcol;
  HostArena& ar = c->stage;
  try {
    ar.reserve((size_t)ns * (3 * sizeof(int) + 4 + sizeof(unsigned int)) + 64 * 64);
  } catch (const std::bad_alloc&) {
    c->err = "sgo_set_graph_se2: out of host memory for the staging buffers";
    return SGO_ENOMEM;
  }
  HostBuf<int> eidx(ar, ns), own(ar, (size_t)ns + 1);
  HostBuf<unsigned char> type(ar, ns), meta(ar, ns), flags(ar, ns), off1(ar, (size_t)std::max(ns, 1));
  HostBuf<unsigned int> cv(ar, (size_t)std::max(ns, 1));
  if (!cv.p) {
    c->err = "sgo_set_graph_se2: internal error (staging arena too small)";
    return SGO_EINVAL;
  }
  // every slot is written exactly once (each edge fills its one or two slots): the edge it came from and the
  // side; the operand arrays themselves are expanded on the device (k_slot_expand)
  parallel_for(E, [&](int e0, int e1) {
    for (int e = e0; e < e1; ++e) {
      const int ki = pos_i[e], kj = pos_j[e];
      if (ki >= 0) {
        eidx[ki] = e;
        flags[ki] = 0;
      }
      if (kj >= 0) {
        eidx[kj] = e;
        flags[kj] = (unsigned char)kSlotDir;
      }
    }
  });
  lap("edge operands");
  // logical structure for the multigrid set-up (diagonal slot first, then the row's block slots): the pattern depends
  // on the row plan only, not on the tiles
  HostLevel& H = c->H0;
  H.n = n;
  H.visit = c->row_of_asc;   // the multigrid aggregation walks level 0 along the trajectory (ascending vertex id)
  H.rowptr.assign((size_t)n + 1, 0);
  parallel_for(n, [&](int r0, int r1) {
    for (int r = r0; r < r1; ++r) {
      int nb = 1;
      for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) nb += col[k] >= 0;
      H.rowptr[r + 1] = nb;
    }
  });
  for (int r = 0; r < n; ++r) H.rowptr[r + 1] += H.rowptr[r];
  H.nslot = H.rowptr[n];
  H.row.resize(H.nslot);
  H.col.resize(H.nslot);
  parallel_for(n, [&](int r0, int r1) {
    for (int r = r0; r < r1; ++r) {
      int q = H.rowptr[r];
      H.row[q] = r;
      H.col[q] = r;
      ++q;
      for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        if (col[k] < 0) {
          flags[k] |= (unsigned char)kSlotFixedCol;
          continue;
        }
        H.row[q] = r;
        H.col[q] = col[k];
        ++q;
      }
    }
  });
  lap("logical pattern");
  int rc;
  // per-slot edge index / side flags and the operand arrays of k_linearize (expanded on the device further down)
  int* d_eidx = nullptr;
  if ((rc = upload(c, &d_eidx, eidx))) return rc;
  if ((rc = upload(c, &c->es.flags, flags))) return rc;
  if ((rc = dalloc(c, &c->es.vi, (size_t)ns)) || (rc = dalloc(c, &c->es.vj, (size_t)ns)) || (rc = dalloc(c, &c->es.zinv, 3 * (size_t)ns)) ||
      (rc = dalloc(c, &c->es.info, 6 * (size_t)ns)) || (rc = dalloc(c, &c->es.phi, (size_t)ns)))
    return rc;
  // Large graphs: the multigrid's host analysis of level 0 (greedy aggregation + patterns / product lists of the
  // smoothed transfer: C4 11 + 17 ms, the longest sequential piece of the set-up) needs the strength weights and the
  // logical pattern only.  The weights are made right here from the edge list (k_row_strength; a not yet expanded
  // operand array serves as scratch), and a helper thread does the analysis while this one cuts the
  // tiles, types the slots and uploads the level-0 storage; build_amg joins it.
  {
    bool pipeline = c->opts.solver == SGO_SOLVER_PCG_AMG && n >= 20000 && E > 0;
    if (const char* e = std::getenv("SGO_SETUP_PIPELINE")) pipeline = pipeline && std::atoi(e) != 0;
    if (pipeline) {
      int *d_rowptr = nullptr, *d_hrowptr = nullptr;
      if ((rc = upload(c, &d_rowptr, rowptr)) || (rc = upload(c, &d_hrowptr, H.rowptr))) return rc;
      double* d_w = c->es.info;   // scratch: nslot <= n + ns <= 2 ns doubles of the 6 ns the not yet expanded operand array holds
      launch_early_strength(c->stream, c->el, c->d_poses, n, d_rowptr, d_eidx, c->es.flags, d_hrowptr, d_w);
      c->l0_w.resize((size_t)H.nslot);
      HIP_TRY(c, hipMemcpyAsync(c->l0_w.data(), d_w, sizeof(double) * (size_t)H.nslot, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      c->l0_pre = amg_host_l0_new();
      AmgHostL0* pre = c->l0_pre;
      const HostLevel* Hp = &c->H0;
      const std::vector<double>* wp = &c->l0_w;
      ChunkArena* scr = &c->amg_scratch;
      c->l0_thread = std::thread([pre, Hp, wp, scr] {
        HostPool::lane() = 1;   // its own worker pool: runs beside this thread's regions instead of queueing with them
        amg_host_l0_run(pre, *Hp, *wp, AmgConfig(), scr);
      });
      lap("early strengths");
    }
  }
  plan_rows_tiles(tile_div, P);
  const int tile_lds = P.tile_lds;
  bool tiles_ok = P.tiles_ok;
  tl = wall_s();
  // slot types
  parallel_for(n, [&](int r0, int r1) {
    for (int r = r0; r < r1; ++r)
      for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const int cc = col[k];
        if (cc < 0) {
          type[k] = kSlotNoBlock;
          col[k] = r;   // never dereferenced as a column; keeps the index in range
        } else {
          type[k] = (unsigned char)((tile_of_row[cc] == tile_of_row[r] && cc < r) ? kSlotTransposed : kSlotOwned);
        }
      }
  });
  // storage index of every owned slot = its rank among the owned slots (own[k]: exclusive count);
  // tslot[k]: rank among the transposed slots
  std::vector<int> tslot((size_t)ns + 1);
  int nu = 0, ntr = 0;
  {
    // exclusive prefix counts in two passes over fixed chunks (counts per chunk, then the ranks inside each chunk)
    const int nchunk = std::max(1, std::min(256, ns / 8192));
    std::vector<int> cu((size_t)nchunk + 1, 0), ct((size_t)nchunk + 1, 0);
    auto chunk_lo = [&](int q) { return (int)((long long)ns * q / nchunk); };
    host_parallel_for(nchunk, 1, [&](int q0, int q1, int) {
      for (int q = q0; q < q1; ++q) {
        int a = 0, b = 0;
        for (int k = chunk_lo(q); k < chunk_lo(q + 1); ++k) {
          a += type[k] == kSlotOwned;
          b += type[k] == kSlotTransposed;
        }
        cu[q + 1] = a;
        ct[q + 1] = b;
      }
    });
    for (int q = 0; q < nchunk; ++q) {
      cu[q + 1] += cu[q];
      ct[q + 1] += ct[q];
    }
    host_parallel_for(nchunk, 1, [&](int q0, int q1, int) {
      for (int q = q0; q < q1; ++q) {
        int a = cu[q], b = ct[q];
        for (int k = chunk_lo(q); k < chunk_lo(q + 1); ++k) {
          own[k] = a;
          tslot[k] = b;
          a += type[k] == kSlotOwned;
          b += type[k] == kSlotTransposed;
        }
      }
    });
    nu = cu[nchunk];
    ntr = ct[nchunk];
  }
  own[ns] = nu;
  tslot[ns] = ntr;
  lap("types + ranks");
  // wave groups over the compact slots: whole rows packed up to 64 slots; a longer row is its own group
  std::vector<int> grp, grow;
  grp.push_back(0);
  {
    int cur = 0, first = 0;
    for (int r = 0; r < n; ++r) {
      const int len = rowptr[r + 1] - rowptr[r];
      if (cur > 0 && cur + len > 64) {
        grp.push_back(rowptr[r]);
        grow.push_back(first);
        first = r;
        cur = 0;
      }
      cur += len;
      if (cur >= 64) {  // full (or a long row): close the group here
        grp.push_back(rowptr[r + 1]);
        grow.push_back(first);
        first = r + 1;
        cur = 0;
      }
    }
    if (grp.back() != ns) {
      grp.push_back(ns);
      grow.push_back(first);
    }
  }
  const int ngrp = (int)grp.size() - 1;
  std::vector<int> gown(ngrp), gtr(ngrp), tref((size_t)std::max(ntr, 1));
  for (int g = 0; g < ngrp; ++g) {
    gown[g] = own[grp[g]];
    gtr[g] = tslot[grp[g]];
  }
  // transposed slots' references (the owner's slot of the same edge); meta bytes
  parallel_for(E, [&](int e0, int e1) {
    for (int e = e0; e < e1; ++e) {
      const int ki = pos_i[e], kj = pos_j[e];
      if (ki < 0 || kj < 0) continue;
      if (type[ki] == kSlotTransposed) tref[tslot[ki]] = own[kj];
      else if (type[kj] == kSlotTransposed) tref[tslot[kj]] = own[ki];
    }
  });
  parallel_for(ngrp, [&](int g0, int g1) {
    for (int g = g0; g < g1; ++g) {
      int r = grow[g];
      for (int k = grp[g]; k < grp[g + 1]; ++k) {
        while (k >= rowptr[r + 1]) ++r;
        const int off = (grp[g + 1] - grp[g] > 64) ? 0 : r - grow[g];
        meta[k] = (unsigned char)(off | (type[k] << 6));
      }
    }
  });
  lap("groups tref meta");
  // where the logical slots' blocks live in the symmetric storage (diagonal / stored block / stored block transposed)
  std::vector<int> lref(H.nslot);
  parallel_for(n, [&](int r0, int r1) {
    for (int r = r0; r < r1; ++r) {
      int q = H.rowptr[r];
      lref[q] = ~r;
      ++q;
      for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        if (type[k] == kSlotNoBlock) continue;
        lref[q] = type[k] == kSlotOwned ? (own[k] << 1) : ((tref[tslot[k]] << 1) | 1);
        ++q;
      }
    }
  });
  lap("logical view");
  // tile arrays: phase-1 groups over the owned slots (numbered like the storage), operand index and twin's
  // staging slot per owned slot, halo columns, staged-entry ranges per row
  std::vector<int> trowptr((size_t)n + 1), grp1, grow1;
  if (tiles_ok) {
    for (int r = 0; r <= n; ++r) trowptr[r] = tslot[rowptr[std::min(r, n)]];
    // per tile, on the host threads: halo numbering in first-seen order, operand index of every owned slot,
    // phase-1 groups; then the per-tile lists are strung together
    const int nt = (int)tiles.size();
    std::vector<std::vector<int>> t_hcol(nt), t_grp(nt), t_grow(nt);
    {
      const int T = std::max(1, std::min(HostPool::get().size(), nt));
      HostPool::get().run(T, [&](int w) {
          std::vector<int> hidx(std::max(n, 1), -1), hmark(std::max(n, 1), -1);
          for (int t = (int)((long long)nt * w / T); t < (int)((long long)nt * (w + 1) / T); ++t) {
            const TileDesc& TT = tiles[t];
            const int nr = TT.row1 - TT.row0;
            std::vector<int>&hc = t_hcol[t], &tg = t_grp[t], &tw = t_grow[t];
            int cur = 0, first = TT.row0;
            for (int r = TT.row0; r < TT.row1; ++r) {
              const int len = own[rowptr[r + 1]] - own[rowptr[r]];
              for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
                if (type[k] != kSlotOwned) continue;
                const int cc = col[k];
                unsigned ci;
                if (cc >= TT.row0 && cc < TT.row1) {
                  ci = (unsigned)(cc - TT.row0);
                } else {
                  if (hmark[cc] != t) {
                    hmark[cc] = t;
                    hidx[cc] = (int)hc.size();
                    hc.push_back(cc);
                  }
                  ci = (unsigned)(nr + hidx[cc]);
                }
                cv[own[k]] = ci | 0xFFFF0000u;   // twin's staging slot filled in below
              }
              if (len == 0) continue;
              if (cur > 0 && cur + len > 64) {   // group boundaries are recorded as END positions (owned-slot numbers)
                tg.push_back(own[rowptr[r]]);
                tw.push_back(first);
                cur = 0;
              }
              if (cur == 0) first = r;
              cur += len;
              if (cur >= 64) {
                tg.push_back(own[rowptr[r + 1]]);
                tw.push_back(first);
                cur = 0;
              }
            }
            if (cur > 0) {
              tg.push_back(own[rowptr[TT.row1]]);
              tw.push_back(first);
            }
          }
      });
    }
    grp1.push_back(0);
    for (int t = 0; t < nt; ++t) {
      TileDesc& T = tiles[t];
      T.e0 = trowptr[T.row0];
      T.nstaged = trowptr[T.row1] - T.e0;
      T.h0 = (int)hcol.size();
      T.g0 = (int)grow1.size();
      hcol.insert(hcol.end(), t_hcol[t].begin(), t_hcol[t].end());
      grp1.insert(grp1.end(), t_grp[t].begin(), t_grp[t].end());
      grow1.insert(grow1.end(), t_grow[t].begin(), t_grow[t].end());
      T.g1 = (int)grow1.size();
      T.h1 = (int)hcol.size();
    }
    const int ng1 = (int)grow1.size();
    std::atomic<bool> span_ok{true};
    parallel_for(ng1, [&](int ga, int gb) {
      for (int g = ga; g < gb; ++g) {
        int r = grow1[g];
        const bool longrow = grp1[g + 1] - grp1[g] > 64;
        for (int u = grp1[g]; u < grp1[g + 1]; ++u) {
          while (own[rowptr[r + 1]] <= u) ++r;
          if (!longrow && r - grow1[g] > 255) span_ok = false;   // hundreds of rows in a row that own nothing
          off1[u] = (unsigned char)(longrow ? 0 : r - grow1[g]);
        }
      }
    });
    if (!span_ok) tiles_ok = false;
    // twins: the owned slot of an intra-tile pair hands B^T x to the transposed slot's staging entry
    parallel_for(E, [&](int e0, int e1) {
      for (int e = e0; e < e1; ++e) {
        const int ki = pos_i[e], kj = pos_j[e];
        if (ki < 0 || kj < 0) continue;
        int ko, kt;
        if (type[ki] == kSlotTransposed) { kt = ki; ko = kj; }
        else if (type[kj] == kSlotTransposed) { kt = kj; ko = ki; }
        else continue;
        const int trow = col[ko];
        const unsigned vp = (unsigned)(tslot[kt] - tiles[tile_of_row[trow]].e0);
        cv[own[ko]] = (cv[own[ko]] & 0xFFFFu) | (vp << 16);
      }
    });
    if (!tiles_ok) {
      // cannot happen for tiles that fit the LDS unless rows own nothing en masse; the types were already
      // chosen for these tiles, and the wave-group kernel handles any mix of owned / transposed slots
      tiles.clear();
    }
  }

  lap("tile arrays");
  const double tb1 = wall_s();
  Sym0Dev& S = c->S0;
  S.n = n;
  S.nu = nu;
  S.npairs = (nu + ntr) / 2;   // owned = intra pairs + 2 x inter pairs, transposed = intra pairs
  S.ncs = ns;
  S.ngrp = ngrp;
  if ((rc = upload(c, &S.col, col))) return rc;
  if ((rc = upload(c, &S.meta, meta))) return rc;
  if ((rc = upload(c, &S.tref, tref))) return rc;
  if ((rc = upload(c, &S.grp, grp))) return rc;
  if ((rc = upload(c, &S.grow, grow))) return rc;
  if ((rc = upload(c, &S.gown, gown))) return rc;
  if ((rc = upload(c, &S.gtr, gtr))) return rc;
  if ((rc = dalloc(c, &S.ublk, 9 * (size_t)nu))) return rc;
  if ((rc = dalloc(c, &S.dblk, 6 * (size_t)n))) return rc;
  if ((rc = dalloc(c, &S.dinv, 6 * (size_t)n))) return rc;
  c->unit_row0.clear();   // first row of every level-0 work unit (tiles, or wave groups without a tile view)
  if (tiles_ok && !tiles.empty()) {
    for (const TileDesc& T : tiles) c->unit_row0.push_back(T.row0);
  } else {
    for (int g = 0; g < ngrp; ++g) c->unit_row0.push_back(grow[g]);
  }
  c->unit_row0.push_back(n);
  Tile0Dev& TL = c->T0;
  TL = Tile0Dev();
  if (tiles_ok && !tiles.empty()) {
    TL.ntile = (int)tiles.size();
    TL.lds_bytes = tile_lds;
    if (const char* e = std::getenv("SGO_TILE_THREADS")) TL.threads = std::atoi(e) == 512 ? 512 : 1024;
    if (hcol.empty()) hcol.push_back(0);
    if ((rc = upload(c, &TL.tile, tiles))) return rc;
    if ((rc = upload(c, &TL.cv, cv))) return rc;
    if ((rc = upload(c, &TL.off1, off1))) return rc;
    if ((rc = upload(c, &TL.grp1, grp1))) return rc;
    if ((rc = upload(c, &TL.grow1, grow1))) return rc;
    if ((rc = upload(c, &TL.trowptr, trowptr))) return rc;
    if ((rc = upload(c, &TL.hcol, hcol))) return rc;
    {
      int hs = 0;
      for (const TileDesc& T : tiles) hs = std::max(hs, std::min(T.h1 - T.h0, TL.threads));
      hs = std::max(64, (hs + 63) / 64 * 64);
      std::vector<int> hfirst((size_t)hs * tiles.size(), -1);
      for (size_t t = 0; t < tiles.size(); ++t)
        std::copy(hcol.begin() + tiles[t].h0, hcol.begin() + tiles[t].h0 + std::min(tiles[t].h1 - tiles[t].h0, hs),
                  hfirst.begin() + (size_t)hs * t);
      TL.hstride = hs;
      if ((rc = upload(c, &TL.hfirst, hfirst))) return rc;
    }
    if (c->opts.verbose)
      std::fprintf(stderr, "[sgo] level-0 tiles: %d tiles, %d B LDS, %d stored blocks for %d pairs (%.1f %% stored with both rows), %zu halo columns\n",
                   TL.ntile, TL.lds_bytes, nu, (nu + ntr) / 2, (nu + ntr) > 0 ? 100.0 * (nu - ntr) / (nu + ntr) : 0.0, hcol.size());
  }
  // logical view for the multigrid set-up kernels
  BsrDev& A = c->A;
  A.n = n;
  A.nslot = H.nslot;
  A.ngrp = 0;
  if ((rc = upload(c, &A.row, H.row))) return rc;
  if ((rc = upload(c, &A.col, H.col))) return rc;
  if ((rc = upload(c, &A.rowptr, H.rowptr))) return rc;
  {
    int* d_ref = nullptr;
    if ((rc = upload(c, &d_ref, lref))) return rc;
    A.ref = d_ref;
  }
  A.ublk = S.ublk;
  A.nu = (size_t)nu;
  A.dblk = S.dblk;
  A.dinv = S.dinv;
  // edge arrays in caller order: indices / kernel parameter straight from the caller's buffers; the inverse
  // measurements and the SoA information are made on the device from the raw rows, and the per-slot operand
  // arrays of k_linearize are expanded there too
  if (E > 0 && ns > 0) launch_slot_expand(c->stream, ns, d_eidx, c->el, c->es);
  if ((rc = upload(c, &c->d_free_id, row_vertex))) return rc;
  const size_t n3 = 3 * (size_t)n;
  if ((rc = dalloc(c, &c->d_dgb, 9 * (size_t)n))) return rc;
  if ((rc = dalloc(c, &c->d_b, n3))) return rc;
  if ((rc = dalloc(c, &c->d_x, n3))) return rc;
  if ((rc = dalloc(c, &c->d_r, n3))) return rc;
  if ((rc = dalloc(c, &c->d_z, n3))) return rc;
  if ((rc = dalloc(c, &c->d_p, n3))) return rc;
  if ((rc = dalloc(c, &c->d_q, n3))) return rc;
  if ((rc = dalloc(c, &c->d_s1, n3))) return rc;
  if ((rc = dalloc(c, &c->d_s2, n3))) return rc;
  if ((rc = dalloc(c, &c->d_xprev, n3))) return rc;
  if ((rc = dalloc(c, &c->d_zparts, 2 * (size_t)kMaxPartials))) return rc;
  if ((rc = dalloc(c, &c->d_S, 1))) return rc;
  HIP_TRY(c, hipMemsetAsync(c->d_S, 0, sizeof(PcgScalars), c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // host staging vectors die at return
  if (c->opts.verbose)
    std::fprintf(stderr, "[sgo] set_graph: host structure %.1f ms, alloc+upload %.1f ms (%d rows, %d stored blocks, %d slots)\n",
                 1e3 * (tb1 - tb0), 1e3 * (wall_s() - tb1), n, nu, ns);
  return SGO_OK;
}

// ---- one GN building block each ------------------------------------------------------------
int do_chi2(sgo_ctx* c, double* d_out2, double* d_e2) {
  int grid = 0;
  {
    Scope sc(c, K_CHI2, bytes_chi2(c));
    int e0 = 0, e1 = c->E;
    if (c->comm.nranks > 1 && !d_e2) sgo_shard_range(c->E, c->comm.nranks, c->comm.rank, &e0, &e1);
    launch_chi2(c->stream, c->el, e0, e1, c->d_poses, d_e2, c->d_partials, &grid);
  }
  {
    Scope sc(c, K_REDUCE2, 16.0 * grid);
    launch_reduce2(c->stream, c->d_partials, grid, d_out2);
  }
  if (c->comm.nranks > 1 && !d_e2 && !c->comm.allreduce_f64(d_out2, 2, c->stream, &c->err)) return SGO_ECOMM;
  return SGO_OK;
}

// PCG start state after k_finalize (x = 0, r = b, z = Dinv b, p = z; partials rz / bb with `grid`
// entries).  With the AMG preconditioner: refresh the coarse operators, z = M^-1 b, p = z.
int do_spmv(sgo_ctx* c, const double* x, double* y, bool dot, const PcgScalars* S, int* grid_out);

int start_pcg(sgo_ctx* c, int grid) {
  if (c->amg && c->warm_valid && c->d_xprev) {
    // Start from the previous Gauss-Newton step scaled by the energy-optimal factor: consecutive steps of a linearly
    // converging iteration are nearly parallel, ||b - gamma H x_prev|| is 0.2-0.45 ||b|| on C4 / C2 (scripts/
    // warm_probe.py), i.e. two PCG iterations for the price of one Hessian product.  Same stopping test, same
    // solution; only the path to it is shorter.
    const int maxit = c->pcg_softcap > 0 ? std::min(c->pcg_softcap, c->opts.pcg_maxit) : c->opts.pcg_maxit;
    int rc;
    if (!c->amg_skip_update && (rc = amg_update(c->amg, c->stream, &c->err))) return rc;
    {
      Scope sc(c, K_INIT_SCALARS, 16.0 * grid);   // ||b||^2, tolerance, iteration count (r.z is replaced below)
      launch_init_scalars(c->stream, c->d_S, c->d_partials, grid, c->d_partials + kMaxPartials, grid, c->opts.pcg_tol * c->tol_scale,
                          maxit, c->bb_ref, c->tol_cap);
    }
    int gq = 0, gd = 0;
    if ((rc = do_spmv(c, c->d_xprev, c->d_q, true, nullptr, &gq))) return rc;
    {
      Scope sc(c, K_DOT, 48.0 * c->n);
      launch_dot(c->stream, 3 * c->n, c->d_b, c->d_xprev, c->d_partials + 2 * kMaxPartials, nullptr, &gd);
    }
    {
      Scope sc(c, K_UPDATE_XR, 120.0 * c->n);
      launch_warm_start(c->stream, 3 * c->n, c->d_xprev, c->d_q, c->d_b, c->d_x, c->d_r, c->d_partials, gq,
                        c->d_partials + 2 * kMaxPartials, gd);
    }
    const int gz = amg_apply(c->amg, c->stream, c->d_r, c->d_z, c->d_r, c->d_zparts, nullptr, nullptr, false);
    if (amg_comm_failed(c->amg)) return SGO_ECOMM;
    HIP_TRY(c, hipMemcpyAsync(c->d_p, c->d_z, sizeof(double) * 3 * (size_t)c->n, hipMemcpyDeviceToDevice, c->stream));
    Scope sc(c, K_INIT_SCALARS, 8.0 * gz);
    launch_restart_scalars(c->stream, c->d_S, c->d_zparts, gz, maxit, 1);
  } else if (c->amg) {
    if (!c->amg_skip_update) {
      int rc = amg_update(c->amg, c->stream, &c->err);
      if (rc) return rc;
    }
    const int gz = amg_apply(c->amg, c->stream, c->d_r, c->d_z, c->d_r, c->d_zparts, nullptr, nullptr, true);
    HIP_TRY(c, hipMemcpyAsync(c->d_p, c->d_z, sizeof(double) * 3 * (size_t)c->n, hipMemcpyDeviceToDevice, c->stream));
    Scope sc(c, K_INIT_SCALARS, 8.0 * (gz + grid));
    launch_init_scalars(c->stream, c->d_S, c->d_zparts, gz, c->d_partials + kMaxPartials, grid, c->opts.pcg_tol * c->tol_scale,
                        c->pcg_softcap > 0 ? std::min(c->pcg_softcap, c->opts.pcg_maxit) : c->opts.pcg_maxit, c->bb_ref, c->tol_cap);
  } else {
    Scope sc(c, K_INIT_SCALARS, 16.0 * grid);
    launch_init_scalars(c->stream, c->d_S, c->d_partials, grid, c->d_partials + kMaxPartials, grid, c->opts.pcg_tol * c->tol_scale,
                        c->pcg_softcap > 0 ? std::min(c->pcg_softcap, c->opts.pcg_maxit) : c->opts.pcg_maxit, c->bb_ref, c->tol_cap);
  }
  return SGO_OK;
}

// buildSystem + preconditioner + PCG start state.  Multi-GPU: every rank linearises the whole graph (1.5 % of
// a GN iteration; sharding it would mean all-reducing the blocks, 72 B per edge, to save it).
int do_linearize(sgo_ctx* c) {
  {
    Scope sc(c, K_LINEARIZE, bytes_linearize(c));
    launch_linearize(c->stream, c->S0, 0, c->S0.ngrp, c->es, c->d_poses, c->d_dgb);
  }
  int grid = 0;
  {
    Scope sc(c, K_FINALIZE, (72.0 + 48.0 + 48.0 + 6 * 24.0) * c->n);
    launch_finalize(c->stream, c->S0, c->d_dgb, c->d_b, c->d_x, c->d_r, c->d_z, c->d_p,
                    c->amg ? amg_xs0(c->amg) : nullptr, c->amg ? amg_omega(c->amg) : 0.0, c->d_partials, &grid);
  }
  int rc = start_pcg(c, grid);
  if (rc) return rc;
  c->linearized = true;
  return SGO_OK;
}

// y = H x  (+ optional x.y partials).  The solve is replicated on every rank (identical H after
// the all-reduce in do_linearize), so no collective is needed here.
int do_spmv(sgo_ctx* c, const double* x, double* y, bool dot, const PcgScalars* S, int* grid_out) {
  Spmv0Args a{};
  a.x = x;
  a.y = y;
  a.S = S;
  if (c->comm.nranks > 1 || c->comm.active()) {
    // multi-GPU: this rank's range of tiles only, zeros elsewhere, all-reduce of the product vector (every row
    // has exactly one non-zero contributor: the sum is exact), then the dot product on the full vectors --
    // the same arithmetic on every rank, so the replicated PCG recurrences stay bit-identical across ranks
    a.u0 = c->shard_u0;
    a.u1 = c->shard_u1;
    HIP_TRY(c, hipMemsetAsync(y, 0, sizeof(double) * 3 * (size_t)c->n, c->stream));
    if (a.u1 > a.u0) {
      Scope sc(c, c->T0.ntile > 0 ? K_SPMV0T_AX : K_SPMV0_AX, bytes_spmv0(c->S0, S0_AX) * (a.u1 - a.u0) / std::max(1, c->shard_units));
      launch_spmv0_any(c->stream, c->S0, c->T0, S0_AX, a);
    }
    if (!c->comm.allreduce_f64(y, 3 * (size_t)c->n, c->stream, &c->err)) return SGO_ECOMM;
    if (dot) {
      int grid = 0;
      Scope sc(c, K_DOT, 48.0 * c->n);
      launch_dot(c->stream, 3 * c->n, x, y, c->d_partials, S, &grid);
      if (grid_out) *grid_out = grid;
    }
    return SGO_OK;
  }
  Scope sc(c, c->T0.ntile > 0 ? K_SPMV0T_AX : K_SPMV0_AX, bytes_spmv0(c->S0, S0_AX));
  if (dot) {
    a.dotA = x;
    a.partials = c->d_partials;
  }
  const int grid = launch_spmv0_any(c->stream, c->S0, c->T0, S0_AX, a);
  if (grid_out) *grid_out = grid;
  return SGO_OK;
}

int pcg_iteration(sgo_ctx* c) {
  int g1 = 0, g2 = 0, rc;
  if ((rc = do_spmv(c, c->d_p, c->d_q, true, c->d_S, &g1))) return rc;
  double* parts2 = c->d_partials + kMaxPartials;  // [0] = r.z (block-Jacobi only), [1] = r.r
  {
    // block-Jacobi: z = Dinv r; multigrid: xs = omega Dinv r, the cycle's first level-0 sweep from zero
    Scope sc(c, K_UPDATE_XR, (7 * 24.0 + 48.0) * c->n);
    launch_update_xr(c->stream, c->n, c->d_S, c->d_partials, g1, c->S0.dinv, c->d_p, c->d_q, c->d_x, c->d_r, c->d_z,
                     c->amg ? amg_xs0(c->amg) : nullptr, c->amg ? amg_omega(c->amg) : 0.0, parts2, &g2);
  }
  if (c->amg) {
    // the K-cycle is a (mildly) variable preconditioner: flexible beta from z.q
    const int gz = amg_apply(c->amg, c->stream, c->d_r, c->d_z, c->d_r, c->d_zparts, c->d_S, c->d_q, true);
    if (amg_comm_failed(c->amg)) {
      c->err = "collective failed inside the multigrid cycle";
      return SGO_ECOMM;
    }
    Scope sc(c, K_UPDATE_P, 3 * 24.0 * c->n);
    launch_update_p(c->stream, c->n, c->d_S, c->d_zparts, gz, parts2 + kMaxPartials, g2, c->d_zparts + kMaxPartials,
                    c->d_z, c->d_p);
  } else {
    Scope sc(c, K_UPDATE_P, 3 * 24.0 * c->n);
    launch_update_p(c->stream, c->n, c->d_S, parts2, g2, parts2 + kMaxPartials, g2, nullptr, c->d_z, c->d_p);
  }
  return SGO_OK;
}

int ensure_pcg_graph(sgo_ctx* c, int chunk) {
  if (c->pcg_exec && c->pcg_exec_chunk == chunk) return SGO_OK;
  if (c->pcg_exec) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    hipGraphExecDestroy(c->pcg_exec);
    c->pcg_exec = nullptr;
  }
  hipGraph_t graph = nullptr;
  HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  int rc = SGO_OK;
  for (int k = 0; k < chunk && rc == SGO_OK; ++k) rc = pcg_iteration(c);
  hipError_t e = hipStreamEndCapture(c->stream, &graph);
  if (rc != SGO_OK) {
    if (graph) hipGraphDestroy(graph);
    return rc;
  }
  if (e != hipSuccess) {
    c->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e);
    return SGO_EHIP;
  }
  e = hipGraphInstantiate(&c->pcg_exec, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (e != hipSuccess) {
    c->pcg_exec = nullptr;
    c->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e);
    return SGO_EHIP;
  }
  c->pcg_exec_chunk = chunk;
  return SGO_OK;
}

// Runs PCG from the state k_finalize left (x = 0, r = b, ...) until S.stop != 0.
// Runs PCG from the state k_finalize / start_pcg left (x = 0, r = b, ...) until S.stop != 0.
// Graph mode: a 2-iteration hipGraph is replayed; the first 0.8 * predicted - 4 iterations -- predicted
// = the count of the previous solve -- go out without any host check, after that one replay is always
// in flight while the host waits for the stop flag copied out after the previous one (kernels of
// iterations past convergence exit on the flag), so the GPU never idles on a host round trip and at
// most two replays of early-exit launches are wasted.
int run_pcg(sgo_ctx* c) {
  // collectives inside the loop: plain stream launches (RCCL calls are not captured into the hipGraph)
  // Opt-in (env SGO_COMM_GRAPH=1): the RCCL collectives are captured into the hipGraph with the kernels around them
  // (every rank replays the same graph the same number of times: the replay count follows the device-resident stop
  // flag, which is bit-identical on all ranks).  Measured with a 1-rank communicator: 154 -> 167 M edge-Jacobians/s on
  // C4; off by default because it has never run on more than one GPU.  Not possible with the host transport.
  const bool comm_graph = c->comm.handle != nullptr && !c->comm.host_fn && std::getenv("SGO_COMM_GRAPH") != nullptr;
  const bool graph = c->opts.use_graph && !c->opts.profile && (!(c->comm.nranks > 1 || c->comm.active()) || comm_graph);
  if (!graph) {
    const int chunk = std::max(1, c->opts.pcg_chunk);
    for (;;) {
      HIP_TRY(c, hipMemcpyAsync(c->h_S, c->d_S, sizeof(PcgScalars), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      if (c->h_S->stop) break;
      for (int k = 0; k < chunk; ++k) {
        int rc = pcg_iteration(c);
        if (rc) return rc;
      }
    }
    c->pcg_pred = c->h_S->iter;
    return SGO_OK;
  }
  constexpr int kUnit = 2;   // iterations per graph replay
  int rc = ensure_pcg_graph(c, kUnit);
  if (rc) return rc;
  // One replay (2 iterations, >= 100 us even on 1k-pose graphs) in flight hides the host's read of the
  // stop flag; more only adds early-exit launches past convergence (measured: 8 iterations in flight
  // cost 3.5 % on C4 and 10 % on C1).  pcg_chunk = 16 -> 1 replay; larger values scale it up.
  const int chunk_launches = std::max(1, c->opts.pcg_chunk / 16);
  // unchecked prefix: 80 % of the previous count minus a margin (a solve that converges earlier
  // than that only wastes ~1 us per early-exit launch; tighter margins measured no different)
  const int unchecked = std::max(0, (int)(0.8 * c->pcg_pred) - 4) / kUnit;
  for (int k = 0; k < unchecked; ++k) HIP_TRY(c, hipGraphLaunch(c->pcg_exec, c->stream));
  int slot = 0;
  HIP_TRY(c, hipMemcpyAsync(&c->h_S2[slot], c->d_S, sizeof(PcgScalars), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipEventRecord(c->ev_S[slot], c->stream));
  for (;;) {
    for (int k = 0; k < chunk_launches; ++k) HIP_TRY(c, hipGraphLaunch(c->pcg_exec, c->stream));  // speculative
    HIP_TRY(c, hipMemcpyAsync(&c->h_S2[slot ^ 1], c->d_S, sizeof(PcgScalars), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipEventRecord(c->ev_S[slot ^ 1], c->stream));
    HIP_TRY(c, hipEventSynchronize(c->ev_S[slot]));
    if (c->h_S2[slot].stop) break;
    slot ^= 1;
  }
  *c->h_S = c->h_S2[slot];
  c->pcg_pred = c->h_S->iter;
  return SGO_OK;
}

// (Re)build the multigrid hierarchy from the CURRENT level-0 values (requires do_linearize).
// An interrupted solve (iteration cap) continues with refreshed hierarchy values: z = M^-1 r for the current residual,
// p = z, recurrence scalars restarted (restarted PCG: x and r carry over).
int continue_pcg_with_fresh_values(sgo_ctx* c, int maxit) {
  int rc = amg_update(c->amg, c->stream, &c->err);
  if (rc) return rc;
  const int gz = amg_apply(c->amg, c->stream, c->d_r, c->d_z, c->d_r, c->d_zparts, nullptr, nullptr, false);
  if (amg_comm_failed(c->amg)) return SGO_ECOMM;
  HIP_TRY(c, hipMemcpyAsync(c->d_p, c->d_z, sizeof(double) * 3 * (size_t)c->n, hipMemcpyDeviceToDevice, c->stream));
  {
    Scope sc(c, K_INIT_SCALARS, 8.0 * gz);
    launch_restart_scalars(c->stream, c->d_S, c->d_zparts, gz, maxit, 0);
  }
  return run_pcg(c);
}

int build_amg(sgo_ctx* c) {
  // speculative replays of the captured PCG iteration (and the launches queued behind them) may still be
  // in flight: drain the stream before the exec and the old hierarchy's buffers go away
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->pcg_exec) {  // the captured PCG iteration references the old hierarchy's buffers
    hipGraphExecDestroy(c->pcg_exec);
    c->pcg_exec = nullptr;
  }
  c->pcg_pred = 0;  // the iteration count of the old hierarchy predicts nothing about the new one
  c->amg_best = 0;
  if (c->amg) {
    amg_destroy(c->amg);
    c->amg = nullptr;
  }
  c->amg_arena.rewind();
  AmgConfig cfg;
  AmgProf prof;
  prof.user = c;
  prof.begin = [](void* u, int kid, double bytes) {
    sgo_ctx* cc = (sgo_ctx*)u;
    cc->amg_scope = new Scope(cc, kid, bytes, kid == K_DENSE_INVERT);
  };
  prof.end = [](void* u) {
    sgo_ctx* cc = (sgo_ctx*)u;
    delete (Scope*)cc->amg_scope;
    cc->amg_scope = nullptr;
  };
  std::string aerr;
  l0_join(c, true);   // the helper thread's analysis of level 0, when set_graph started one (first build only)
  c->amg = amg_create(c->stream, c->A, c->S0, c->T0, c->H0, c->d_poses, c->d_free_id, cfg, prof, &aerr, &c->amg_scratch,
                      &c->amg_arena, c->l0_pre);
  l0_discard(c);
  if (c->amg) {
    if (c->comm.nranks > 1 || c->comm.active()) amg_set_shard(c->amg, &c->comm, c->shard_u0, c->shard_u1, c->shard_row0, c->shard_row1);
    amg_describe(c->amg, &c->solver_desc);
    c->solver_desc = "pcg_amg: " + c->solver_desc;
  } else {
    c->solver_desc = "pcg_block_jacobi (AMG unavailable: " + aerr + ")";
    if (c->opts.verbose) std::fprintf(stderr, "[sgo] %s\n", c->solver_desc.c_str());
  }
  return SGO_OK;
}

// Vectors over the free vertices cross the API in g2o's hessian order and live on the device in the
// internal (Hilbert) row order: permute on the way (test / single-step entry points only).
// build_structure + what follows from it (multi-GPU tile range, tolerance rule)
int build_rows(sgo_ctx* c, const double* poses, const uint8_t* fixed, const int32_t* ei, const int32_t* ej) {
  int rc = build_structure(c, c->V, poses, fixed, c->E, ei, ej);
  if (rc != SGO_OK) return rc;
  c->shard_units = c->T0.ntile > 0 ? c->T0.ntile : c->S0.ngrp;
  sgo_shard_range(c->shard_units, c->comm.nranks, c->comm.rank, &c->shard_u0, &c->shard_u1);
  c->shard_row0 = c->unit_row0.empty() ? 0 : c->unit_row0[c->shard_u0];
  c->shard_row1 = c->unit_row0.empty() ? c->n : c->unit_row0[c->shard_u1];
  // Chain-like graphs (fewer than ~1.5 edges per free pose: under 4 Hessian blocks per row) are the
  // ill-conditioned ones -- kappa(H) grows with the square of the chain length -- and a relative
  // residual of 1e-8 then leaves errors that show in chi2 (3000 poses / 3150 edges: iterates 3e-6 and
  // poses 7e-5 m from the direct-solver oracle at 1e-8, 1.4e-8 at 1e-9).  Their PCG iterations are the
  // cheap ones, so they get a 10x tighter tolerance than opts.pcg_tol.
  c->tol_scale = (c->n > 0 && (long long)c->A.nslot < 4LL * c->n) ? 0.1 : 1.0;   // logical slots: 2 per edge + 1 per row
  return SGO_OK;
}

// Graphs that optimize() through the single-launch direct path have their PCG-path structures built by the first
// entry point that needs them (sgo_linearize, sgo_hessian_apply, ...): the row order then follows the CURRENT poses.
int ensure_rows(sgo_ctx* c) {
  if (!c->rows_pending) return SGO_OK;
  c->rows_pending = false;
  std::vector<double> poses(3 * (size_t)c->V);
  HIP_TRY(c, hipMemcpyAsync(poses.data(), c->d_poses, sizeof(double) * poses.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const int rc = build_rows(c, poses.data(), c->lz_fixed.data(), c->lz_ei.data(), c->lz_ej.data());
  c->lz_fixed = std::vector<uint8_t>();
  c->lz_ei = std::vector<int32_t>();
  c->lz_ej = std::vector<int32_t>();
  return rc;
}

// Graphs that optimize() through the single-launch direct path build their multigrid hierarchy only when a
// single-step entry point (sgo_solve, sgo_precondition) or the PCG fallback asks for it.
int ensure_amg(sgo_ctx* c) {
  int rc = ensure_rows(c);
  if (rc != SGO_OK) return rc;
  if (!c->amg_pending) return SGO_OK;
  c->amg_pending = false;
  if ((rc = do_linearize(c)) != SGO_OK || (rc = build_amg(c)) != SGO_OK) return rc;
  c->linearized = false;
  return SGO_OK;
}

int vec_to_device(sgo_ctx* c, const double* host_asc, double* dev) {
  std::vector<double> tmp(3 * (size_t)c->n);
  for (int i = 0; i < c->n; ++i)
    for (int q = 0; q < 3; ++q) tmp[3 * (size_t)c->row_of_asc[i] + q] = host_asc[3 * (size_t)i + q];
  HIP_TRY(c, hipMemcpyAsync(dev, tmp.data(), sizeof(double) * tmp.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SGO_OK;
}
int vec_from_device(sgo_ctx* c, const double* dev, double* host_asc) {
  std::vector<double> tmp(3 * (size_t)c->n);
  HIP_TRY(c, hipMemcpyAsync(tmp.data(), dev, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < c->n; ++i)
    for (int q = 0; q < 3; ++q) host_asc[3 * (size_t)i + q] = tmp[3 * (size_t)c->row_of_asc[i] + q];
  return SGO_OK;
}

int check_graph(sgo_ctx* c) {
  if (!c) return SGO_EINVAL;
  if (!c->has_graph) {
    c->err = "no graph: call sgo_set_graph_se2 first";
    return SGO_ENOGRAPH;
  }
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) {
    c->err = std::string("hipSetDevice: ") + hipGetErrorString(e);
    return SGO_EHIP;
  }
  return SGO_OK;
}

}  // namespace

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

// =============================================================================== C-ABI
extern "C" {

int sgo_version(void) { return SGO_VERSION; }

void sgo_default_opts(sgo_opts* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->struct_size = (int32_t)sizeof(sgo_opts);
  o->solver = SGO_SOLVER_PCG_AMG;
  o->pcg_tol = 1e-8;
  o->pcg_maxit = 20000;
  o->pcg_chunk = 16;
  o->use_graph = 1;
  o->profile = 0;
  o->verbose = 0;
  o->direct_rows = 8192;
  o->pcg_tol_cap = 1e-6;
  o->pcg_warm_start = 1;
  if (const char* s = std::getenv("SGO_PCG_WARM")) o->pcg_warm_start = std::atoi(s);
  if (const char* s = std::getenv("SGO_PCG_TOL_CAP")) o->pcg_tol_cap = std::atof(s);
  if (const char* s = std::getenv("SGO_DIRECT_ROWS")) o->direct_rows = std::atoi(s);
  if (const char* s = std::getenv("SGO_SOLVER")) {
    if (!std::strcmp(s, "pcg") || !std::strcmp(s, "bj")) o->solver = SGO_SOLVER_PCG_BJ;
    else if (!std::strcmp(s, "amg")) o->solver = SGO_SOLVER_PCG_AMG;
  }
  if (const char* s = std::getenv("SGO_PCG_TOL")) o->pcg_tol = std::atof(s);
  if (const char* s = std::getenv("SGO_PCG_MAXIT")) o->pcg_maxit = std::atoi(s);
  if (const char* s = std::getenv("SGO_PCG_CHUNK")) o->pcg_chunk = std::atoi(s);
  if (const char* s = std::getenv("SGO_USE_GRAPH")) o->use_graph = std::atoi(s);
  if (const char* s = std::getenv("SGO_PROFILE")) o->profile = std::atoi(s);
  if (const char* s = std::getenv("SGO_VERBOSE")) o->verbose = std::atoi(s);
}

sgo_ctx* sgo_create(int device, const sgo_opts* opts) {
  if (device < 0) {
    const char* s = std::getenv("SGO_DEVICE");
    device = s ? std::atoi(s) : 0;
  }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_err = std::string("no HIP device available: ") + hipGetErrorString(e);
    return nullptr;
  }
  if (device >= count) {
    g_err = "device ordinal " + std::to_string(device) + " out of range (" + std::to_string(count) + " devices)";
    return nullptr;
  }
  if ((e = hipSetDevice(device)) != hipSuccess) {
    g_err = std::string("hipSetDevice: ") + hipGetErrorString(e);
    return nullptr;
  }
  sgo_ctx* c = new (std::nothrow) sgo_ctx();
  if (!c) {
    g_err = "out of host memory";
    return nullptr;
  }
  c->device = device;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->cu_count = prop.multiProcessorCount;
  }
  sgo_default_opts(&c->opts);
  if (opts) {
    size_t sz = std::min<size_t>(sizeof(sgo_opts), opts->struct_size > 0 ? (size_t)opts->struct_size : sizeof(sgo_opts));
    std::memcpy(&c->opts, opts, sz);
    c->opts.struct_size = (int32_t)sizeof(sgo_opts);
  }
  if (c->opts.pcg_tol <= 0) c->opts.pcg_tol = 1e-8;
  if (c->opts.pcg_maxit <= 0) c->opts.pcg_maxit = 20000;
  if (c->opts.pcg_chunk <= 0) c->opts.pcg_chunk = 16;
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipHostMalloc((void**)&c->h_S, sizeof(PcgScalars))) != hipSuccess ||
      (e = hipHostMalloc((void**)&c->h_S2, 2 * sizeof(PcgScalars))) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_S[0], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_S[1], hipEventDisableTiming)) != hipSuccess ||
      (e = hipHostMalloc((void**)&c->h_hist, sizeof(double) * 2 * (SGO_MAX_ITERS + 2))) != hipSuccess ||
      (e = hipHostMalloc((void**)&c->h_dres, sizeof(DirectResult))) != hipSuccess ||
      (e = hipMalloc((void**)&c->d_dres, sizeof(DirectResult))) != hipSuccess) {
    g_err = std::string("context setup: ") + hipGetErrorString(e);
    sgo_destroy(c);
    return nullptr;
  }
  return c;
}

void sgo_destroy(sgo_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  prof_flush(c);
  free_graph(c);
  c->graph_arena.release();
  c->amg_arena.release();
  c->comm.destroy();
  for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
  for (hipEvent_t e : c->iter_events) hipEventDestroy(e);
  if (c->h_S) hipHostFree(c->h_S);
  if (c->h_S2) hipHostFree(c->h_S2);
  for (hipEvent_t ev : c->ev_S)
    if (ev) hipEventDestroy(ev);
  if (c->h_hist) hipHostFree(c->h_hist);
  if (c->h_dres) hipHostFree(c->h_dres);
  if (c->d_dres) hipFree(c->d_dres);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

const char* sgo_solver_description(sgo_ctx* c) {
  if (!c || !c->has_graph) return "";
  try {
    c->solver_text = c->solver_desc;
    if (!c->direct && !c->direct_why.empty()) c->solver_text += "; direct path not used: " + c->direct_why;
    return c->solver_text.c_str();
  } catch (...) {   // no C++ exception crosses the C boundary
    return "";
  }
}

const char* sgo_last_error(sgo_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }

int sgo_set_graph_se2(sgo_ctx* c, int32_t V, const double* poses, const uint8_t* fixed, int32_t E, const int32_t* ei,
                      const int32_t* ej, const double* meas, const double* info, const double* phi) {
  try {
    if (!c) return SGO_EINVAL;
    if (V <= 0 || E < 0 || !poses || !fixed || (E > 0 && (!ei || !ej || !meas || !info || !phi))) {
      c->err = "sgo_set_graph_se2: bad argument";
      return SGO_EINVAL;
    }
    if (2 * (int64_t)E + (int64_t)V > (int64_t)INT32_MAX / 2) {  // slot indices are 32-bit (2E + n slots)
      c->err = "sgo_set_graph_se2: graph too large for 32-bit slot indices";
      return SGO_EINVAL;
    }
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) {
      c->err = std::string("hipSetDevice: ") + hipGetErrorString(e);
      return SGO_EHIP;
    }
    const double t0 = wall_s();
    hipStreamSynchronize(c->stream);
    free_graph(c);
    if (c->opts.verbose) std::fprintf(stderr, "[sgo] set_graph: release of the previous graph %.1f ms\n", 1e3 * (wall_s() - t0));
    int rc = build_edges(c, V, poses, fixed, E, ei, ej, meas, info, phi);
    if (rc != SGO_OK) {
      free_graph(c);
      return rc;
    }
    c->has_graph = true;
    c->solver_desc = "pcg_block_jacobi";
    c->direct_why.clear();
    if (c->opts.solver == SGO_SOLVER_PCG_AMG && c->n > 0 && c->opts.direct_rows > 0 && c->comm.nranks <= 1 && !c->comm.active()) {
      // Small graphs (the reference's own sizes): optimize() as ONE launch of a sparse direct solver when the
      // elimination analysis fits (sgo_direct.h); the multigrid hierarchy is then built only if a single-step
      // entry point asks for it.
      std::string derr;
      const double td0 = wall_s();
      c->direct = direct_create(c->stream, &c->graph_arena, V, c->n, c->free_id.data(), E, ei, ej, c->opts.direct_rows,
                                &c->direct_why, &derr);
      if (!c->direct && !derr.empty()) {
        c->err = derr;
        free_graph(c);
        return SGO_EHIP;
      }
      if (c->opts.verbose) std::fprintf(stderr, "[sgo] set_graph: direct-path analysis %.2f ms\n", 1e3 * (wall_s() - td0));
      if (c->direct) {
        const DirectInfo& di = direct_info(c->direct);
        c->solver_desc = "direct_ldlt: " + std::to_string(di.n_chain) + " chain poses in " + std::to_string(di.levels) +
                         " levels + " + std::to_string(di.n_sep) + " separators (dense), " + std::to_string(di.slots) +
                         " stored blocks, " + std::to_string(di.contributions) + " block products per factorisation; pcg_amg on demand";
        c->amg_pending = true;
        // the row plan and the level-0 structures of the PCG path wait for the first entry point that needs them
        c->rows_pending = true;
        c->lz_fixed.assign(fixed, fixed + V);
        c->lz_ei.assign(ei, ei + E);
        c->lz_ej.assign(ej, ej + E);
      } else if (c->opts.verbose) {
        std::fprintf(stderr, "[sgo] direct path not used: %s\n", c->direct_why.c_str());
      }
    }
    if (!c->rows_pending && (rc = build_rows(c, poses, fixed, ei, ej)) != SGO_OK) {
      free_graph(c);
      return rc;
    }
    if (c->opts.solver == SGO_SOLVER_PCG_AMG && c->n > 0 && !c->direct) {
      // the hierarchy is built from the Hessian at the initial poses (strength of connection)
      const double ta0 = wall_s();
      if ((rc = do_linearize(c)) != SGO_OK || (rc = build_amg(c)) != SGO_OK) {
        free_graph(c);
        return rc;
      }
      if (c->opts.verbose) std::fprintf(stderr, "[sgo] set_graph: multigrid set-up %.1f ms\n", 1e3 * (wall_s() - ta0));
      c->linearized = false;
    }
    c->setup_seconds = wall_s() - t0;
    if (c->opts.verbose)
      std::fprintf(stderr, "[sgo] solver: %s\n[sgo] set_graph: total %.1f ms\n", c->solver_desc.c_str(), 1e3 * c->setup_seconds);
    return SGO_OK;
  } SGO_CATCH(c)
}

int sgo_set_poses(sgo_ctx* c, const double* poses) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if (!poses) return SGO_EINVAL;
    HIP_TRY(c, hipMemcpyAsync(c->d_poses, poses, sizeof(double) * 3 * (size_t)c->V, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->linearized = false;
    return SGO_OK;
  } SGO_CATCH(c)
}

int sgo_get_poses(sgo_ctx* c, double* poses) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if (!poses) return SGO_EINVAL;
    HIP_TRY(c, hipMemcpyAsync(poses, c->d_poses, sizeof(double) * 3 * (size_t)c->V, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SGO_OK;
  } SGO_CATCH(c)
}

int sgo_num_free(sgo_ctx* c) {
  int rc = check_graph(c);
  return rc ? rc : c->n;
}

int sgo_free_ids(sgo_ctx* c, int32_t* out) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if (!out) return SGO_EINVAL;
    std::copy(c->free_id.begin(), c->free_id.end(), out);
    return c->n;
  } SGO_CATCH(c)
}

int sgo_chi2(sgo_ctx* c, double* plain, double* robust) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if ((rc = do_chi2(c, c->d_hist, nullptr))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->h_hist, c->d_hist, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (plain) *plain = c->h_hist[0];
    if (robust) *robust = c->h_hist[1];
    return SGO_OK;
  } SGO_CATCH(c)
}

int sgo_edge_chi2(sgo_ctx* c, double* e2) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if (!e2) return SGO_EINVAL;
    if ((rc = do_chi2(c, c->d_hist, c->d_e2))) return rc;
    HIP_TRY(c, hipMemcpyAsync(e2, c->d_e2, sizeof(double) * (size_t)c->E, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SGO_OK;
  } SGO_CATCH(c)
}

int sgo_closure_information(sgo_ctx* c, int32_t n, const sgo_match_window* win, const float* scores, int64_t n_scores,
                            double* cov, double* info) {
  try {
    if (!c) return SGO_EINVAL;
    if (n < 0 || n_scores < 0 || (n > 0 && (!win || !scores || !cov || !info))) {
      c->err = "sgo_closure_information: null buffer or negative count";
      return SGO_EINVAL;
    }
    if (n == 0) return SGO_OK;
    // every window must lie inside scores[]: the kernel trusts these bounds
    for (int q = 0; q < n; ++q) {
      const sgo_match_window& W = win[q];
      if (W.w_size < 0 || W.w_size > 1024 || W.scan_window < 0 || W.scan_window > 1024 || W.score_offset < 0) {
        c->err = "sgo_closure_information: window " + std::to_string(q) + " has a negative or oversized extent";
        return SGO_EINVAL;
      }
      const int64_t nw = 2 * (int64_t)W.w_size + 1, total = nw * nw * (2 * (int64_t)W.scan_window + 1);
      if (total > INT32_MAX || W.score_offset + total > n_scores) {
        c->err = "sgo_closure_information: window " + std::to_string(q) + " reaches past scores[n_scores]";
        return SGO_EINVAL;
      }
    }
    HIP_TRY(c, hipSetDevice(c->device));
    sgo_match_window* d_win = nullptr;
    float* d_sc = nullptr;
    double* d_out = nullptr;
    auto release = [&]() {
      if (d_win) hipFree(d_win);
      if (d_sc) hipFree(d_sc);
      if (d_out) hipFree(d_out);
    };
    if (hipMalloc(&d_win, sizeof(sgo_match_window) * (size_t)n) != hipSuccess ||
        hipMalloc(&d_sc, sizeof(float) * (size_t)std::max<int64_t>(n_scores, 1)) != hipSuccess ||
        hipMalloc(&d_out, sizeof(double) * 18 * (size_t)n) != hipSuccess) {
      release();
      c->err = "sgo_closure_information: out of device memory";
      return SGO_ENOMEM;
    }
    hipError_t e = hipMemcpyAsync(d_win, win, sizeof(sgo_match_window) * (size_t)n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(d_sc, scores, sizeof(float) * (size_t)n_scores, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      launch_closure_cov(c->stream, n, d_win, d_sc, d_out, d_out + 9 * (size_t)n);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(cov, d_out, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(info, d_out + 9 * (size_t)n, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    release();
    if (e != hipSuccess) {
      c->err = std::string("sgo_closure_information: ") + hipGetErrorString(e);
      return SGO_EHIP;
    }
    return SGO_OK;
  } SGO_CATCH(c)
}

int sgo_linearize(sgo_ctx* c, double* b, double* diag, double* plain, double* robust) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if (c->n == 0) return SGO_ENOTHING;
    if ((rc = ensure_amg(c))) return rc;
    if ((rc = do_chi2(c, c->d_hist, nullptr))) return rc;
    if ((rc = do_linearize(c))) return rc;
    std::vector<double> dgb(9 * (size_t)c->n);
    HIP_TRY(c, hipMemcpyAsync(dgb.data(), c->d_dgb, sizeof(double) * dgb.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->h_hist, c->d_hist, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < c->n; ++i) {   // i: hessian index (g2o order); its values sit in internal row row_of_asc[i]
      const double* d = &dgb[9 * (size_t)c->row_of_asc[i]];
      if (diag) {
        double* D = diag + 9 * (size_t)i;
        D[0] = d[0]; D[1] = d[1]; D[2] = d[2];
        D[3] = d[1]; D[4] = d[3]; D[5] = d[4];
        D[6] = d[2]; D[7] = d[4]; D[8] = d[5];
      }
      if (b) {
        b[3 * (size_t)i] = d[6];
        b[3 * (size_t)i + 1] = d[7];
        b[3 * (size_t)i + 2] = d[8];
      }
    }
    if (plain) *plain = c->h_hist[0];
    if (robust) *robust = c->h_hist[1];
    return SGO_OK;
  } SGO_CATCH(c)
}

int sgo_hessian_apply(sgo_ctx* c, const double* x, double* y) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if (!x || !y) return SGO_EINVAL;
    if (!c->linearized) {
      c->err = "sgo_hessian_apply: call sgo_linearize first";
      return SGO_EINVAL;
    }
    if ((rc = vec_to_device(c, x, c->d_s1))) return rc;
    if ((rc = do_spmv(c, c->d_s1, c->d_s2, false, nullptr, nullptr))) return rc;
    return vec_from_device(c, c->d_s2, y);
  } SGO_CATCH(c)
}

int sgo_precondition(sgo_ctx* c, const double* r, double* z) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if (!r || !z) return SGO_EINVAL;
    if (!c->linearized) {
      c->err = "sgo_precondition: call sgo_linearize first";
      return SGO_EINVAL;
    }
    if ((rc = vec_to_device(c, r, c->d_s1))) return rc;
    if (c->amg) amg_apply(c->amg, c->stream, c->d_s1, c->d_s2, nullptr, nullptr, nullptr);
    else launch_precond_bj(c->stream, c->n, c->S0.dinv, c->d_s1, c->d_s2, 1.0);
    return vec_from_device(c, c->d_s2, z);
  } SGO_CATCH(c)
}

int sgo_solve(sgo_ctx* c, double* x, double* relres) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if (!c->linearized) {
      c->err = "sgo_solve: call sgo_linearize first";
      return SGO_EINVAL;
    }
    // restart from the state of the last linearisation (idempotent re-finalize)
    int grid = 0;
    launch_finalize(c->stream, c->S0, c->d_dgb, c->d_b, c->d_x, c->d_r, c->d_z, c->d_p,
                    c->amg ? amg_xs0(c->amg) : nullptr, c->amg ? amg_omega(c->amg) : 0.0, c->d_partials, &grid);
    if ((rc = start_pcg(c, grid))) return rc;
    if ((rc = run_pcg(c))) return rc;
    if (x && (rc = vec_from_device(c, c->d_x, x))) return rc;
    if (relres) *relres = c->h_S->bb > 0 ? std::sqrt(c->h_S->rr / c->h_S->bb) : 0.0;
    if (c->h_S->stop == 3) {
      c->err = "PCG breakdown (p.Hp <= 0 or non-finite): Hessian not positive definite";
      return SGO_EINVAL;
    }
    return c->h_S->iter;
  } SGO_CATCH(c)
}

int sgo_optimize_gn(sgo_ctx* c, int32_t iters, sgo_stats* out) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if (iters < 0 || iters > SGO_MAX_ITERS) {
      c->err = "sgo_optimize_gn: iters must be in [0, SGO_MAX_ITERS]";
      return SGO_EINVAL;
    }
    if (out) {
      std::memset(out, 0, sizeof(*out));
      out->iters_requested = iters;
      out->seconds_setup = c->setup_seconds;
    }
    if (c->n == 0) return SGO_ENOTHING;
    const double t0 = wall_s();
    if (c->direct) {
      // ---- small-graph path: the whole call is one launch (sgo_direct.h)
      {
        Scope sc(c, K_DIRECT, direct_bytes(c->direct, c->E, iters));
        hipError_t he = direct_optimize(c->direct, c->stream, c->el, c->d_poses, iters, c->d_hist, c->d_dres);
        if (he != hipSuccess) {
          c->err = std::string("k_direct launch: ") + hipGetErrorString(he);
          return SGO_EHIP;
        }
      }
      HIP_TRY(c, hipMemcpyAsync(c->h_hist, c->d_hist, sizeof(double) * 2 * (size_t)(iters + 1), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipMemcpyAsync(c->h_dres, c->d_dres, sizeof(DirectResult), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      prof_flush(c);
      c->linearized = false;
      const DirectResult& R = *c->h_dres;
      const int done = R.done;
      if (R.fail) {
        c->err = std::string("direct factorisation failed in GN iteration ") + std::to_string(done) +
                 (R.fail == 1 ? " (a pivot block is not positive definite: Hessian not positive definite)"
                              : " (non-finite update)") + "; the step was not applied";
      }
      if (out) {
        out->iters_done = done;
        for (int k = 0; k <= done; ++k) {
          out->chi2[k] = c->h_hist[2 * k];
          out->robust_chi2[k] = c->h_hist[2 * k + 1];
        }
        const int timed = std::min(iters, done + (R.fail ? 1 : 0));
        for (int k = 0; k < timed; ++k) {
          out->pcg_iters[k] = 0;
          out->pcg_converged[k] = (k < done) ? 1 : 0;
          out->seconds_linearize[k] = 1e-8 * (double)(R.stamp[2 * k + 1] - R.stamp[2 * k]);
          out->seconds[k] = 1e-8 * (double)(R.stamp[2 * k + 2] - R.stamp[2 * k]);
          out->seconds_solve[k] = out->seconds[k] - out->seconds_linearize[k];
        }
        out->seconds_total = wall_s() - t0;
      }
      if (c->opts.verbose && done > 0)
        std::fprintf(stderr, "[sgo] direct: %.0f MHz shader clock during the call\n",
                     (double)R.cycles / (1e-2 * (double)(R.stamp[R.fail ? 2 * done + 2 : 2 * iters + 1] - R.stamp[0])));
      if (c->opts.verbose && done > 0)
        std::fprintf(stderr, "[sgo] direct, last iteration [us]: edges %.1f, assembly %.1f, sparse forward %.1f, separators %.1f + %.1f, "
                     "sparse backward %.1f, update %.1f\n", 1e-2 * (double)(R.phase[1] - R.phase[0]), 1e-2 * (double)(R.phase[2] - R.phase[1]),
                     1e-2 * (double)(R.phase[3] - R.phase[2]), 1e-2 * (double)(R.phase[4] - R.phase[3]), 1e-2 * (double)(R.phase[5] - R.phase[4]),
                     1e-2 * (double)(R.phase[6] - R.phase[5]), 1e-2 * (double)(R.phase[7] - R.phase[6]));
      if (c->opts.verbose > 1)
        for (int k = 0; k <= done; ++k)
          std::fprintf(stderr, "[sgo] iteration= %d\t chi2= %.9e\t robust= %.9e\t (direct)\n", k, c->h_hist[2 * k], c->h_hist[2 * k + 1]);
      return R.fail ? 0 : done;
    }
    if ((rc = ensure_amg(c))) return rc;
    // per-iteration time stamps: events are kept in the context and reused by later calls
    while (c->iter_events.size() < 3 * (size_t)iters + 1) {
      hipEvent_t e = nullptr;
      HIP_TRY(c, hipEventCreate(&e));
      c->iter_events.push_back(e);
    }
    std::vector<hipEvent_t>& ev = c->iter_events;
    struct SoftcapGuard {   // the bail-out cap and the absolute accuracy target apply to solves inside this call only
      sgo_ctx* c;
      ~SoftcapGuard() {
        c->pcg_softcap = 0;
        c->bb_ref = 0.0;
        c->warm_valid = false;
      }
    } softcap_guard{c};
    c->warm_valid = false;
    const bool warm_env = c->opts.pcg_warm_start != 0;
    c->bb_ref = 0.0;
    c->tol_cap = c->opts.pcg_tol_cap > 0.0 ? std::max(c->opts.pcg_tol_cap, c->opts.pcg_tol * c->tol_scale) : 0.0;
    int done = 0;
    bool failed = false;
    int rebuilds = 0;
    // opt-in (SGO_AMG_LAZY=1): +7 % on C4, -3 % on C2, -4 % on the full-information 100k / 1M graph
    static const bool lazy_env = std::getenv("SGO_AMG_LAZY") && std::atoi(std::getenv("SGO_AMG_LAZY")) != 0;
    bool lazy_ok = lazy_env && c->opts.pcg_tol_cap > 0.0, prev_skipped = false;
    double prev_bb = 0.0;
    int fresh_iter = 0;   // PCG iterations (at equal tolerance) of the last solve with refreshed values
    int fresh_actual = 0; // ... as counted
    int& best_pcg = c->amg_best;
    bool rebuild_next = false;
    for (int it = 0; it < iters; ++it) {
      hipEventRecord(ev[3 * it], c->stream);
      c->pcg_softcap = (c->amg && c->amg_best > 0 && rebuilds < 3 && !rebuild_next) ? 4 * c->amg_best + 40 : 0;
      // Opt-in: late Gauss-Newton iterations barely move the Hessian: once ||b|| has fallen below 5 % of the call's
      // first, every other iteration reuses the hierarchy's values (P, Galerkin operators, dense inverse: 1.3 ms on
      // C4) of the one before.  Any fixed SPD preconditioner gives the same solution; a stale one only costs PCG
      // iterations: the solve is capped at the refresh's worth of extra iterations, then continues from its current
      // iterate with refreshed values, and the reuse ends for this call.
      c->amg_skip_update = lazy_ok && c->amg && it >= 2 && !prev_skipped && !rebuild_next && c->bb_ref > 0.0 &&
                           prev_bb <= 0.0025 * c->bb_ref && fresh_actual > 0;
      const int normal_cap = c->pcg_softcap;
      if (c->amg_skip_update) {   // a solve on reused values may cost the refresh's worth of extra iterations, not more
        const int cap = fresh_actual + std::max(8, fresh_actual / 4);
        c->pcg_softcap = normal_cap > 0 ? std::min(normal_cap, cap) : cap;
      }
      const bool this_skipped = c->amg_skip_update;
      if ((rc = do_chi2(c, c->d_hist + 2 * it, nullptr)) || (rc = do_linearize(c))) {
        c->amg_skip_update = false;
        return rc;
      }
      c->amg_skip_update = false;
      if (rebuild_next && c->amg) {
        // The aggregation was made from the Hessian of an earlier linearisation and robust-kernel
        // re-weighting has changed the strength of connection since (see the rule below): redo the
        // set-up from the current values (same cost as in sgo_set_graph_se2).
        if ((rc = build_amg(c)) || (rc = do_linearize(c))) return rc;
        rebuild_next = false;
        ++rebuilds;
        if (c->opts.verbose) std::fprintf(stderr, "[sgo] multigrid hierarchy rebuilt before iteration %d\n", it);
      }
      hipEventRecord(ev[3 * it + 1], c->stream);
      int wasted = 0;
      if ((rc = run_pcg(c))) {
        return rc;
      }
      if (this_skipped) {
        c->pcg_softcap = normal_cap;
        if (c->h_S->stop == 2 && c->h_S->iter < c->opts.pcg_maxit) {
          // the reused values cost more than they save: refresh them, carry on from the current iterate, and stop
          // reusing for the rest of this call
          lazy_ok = false;
          if ((rc = continue_pcg_with_fresh_values(c, normal_cap > 0 ? std::min(normal_cap, c->opts.pcg_maxit) : c->opts.pcg_maxit)))
            return rc;
          if (c->opts.verbose) std::fprintf(stderr, "[sgo] iteration %d: reused hierarchy values refreshed after %d PCG iterations\n", it, fresh_actual);
        }
      }
      if (c->pcg_softcap > 0 && c->h_S->stop == 2 && c->h_S->iter < c->opts.pcg_maxit && c->amg) {
        // The solve ran into the bail-out cap (4x the best count of this hierarchy): the aggregation no
        // longer fits the re-weighted Hessian.  Redo the set-up from the current values and solve again
        // from x = 0 instead of grinding on (seen: 735 iterations where the rebuilt hierarchy needs 16).
        wasted = c->h_S->iter;
        c->pcg_softcap = 0;
        if ((rc = build_amg(c)) || (rc = do_linearize(c)) || (rc = run_pcg(c))) return rc;
        ++rebuilds;
        rebuild_next = false;
        if (c->opts.verbose)
          std::fprintf(stderr, "[sgo] iteration %d: solve abandoned after %d PCG iterations, hierarchy rebuilt\n", it, wasted);
      }
      const PcgScalars S = *c->h_S;
      if (it == 0 && c->tol_cap > 0.0 && S.stop == 1) c->bb_ref = S.bb;
      const bool was_skipped = prev_skipped = (c->amg != nullptr) && this_skipped;
      prev_bb = S.bb;
      if (c->amg && S.stop != 3) {
        // Iteration counts are compared at EQUAL tolerance: a solve that stopped at the absolute criterion (a looser
        // relative tolerance, see pcg_tol_cap) is scaled to what pcg_tol would have cost -- PCG converges linearly,
        // iterations ~ log(1 / tolerance) -- or the staleness rules below would take every tight solve that follows
        // a loose one for a stale hierarchy.
        const double tol0 = c->opts.pcg_tol * c->tol_scale, tolk = std::sqrt(S.tol2);
        const int eq_iter = (tolk > tol0 && tolk < 1.0 && tol0 > 0.0) ? (int)std::lround(S.iter * std::log(tol0) / std::log(tolk)) : S.iter;
        if (!was_skipped) {
          fresh_iter = eq_iter;
          fresh_actual = S.iter;
        } else if (eq_iter > fresh_iter + std::max(7, fresh_iter / 4)) {
          lazy_ok = false;   // ~7 iterations = the refresh's cost
        }
        if (!was_skipped && (best_pcg == 0 || eq_iter < best_pcg)) best_pcg = eq_iter;
        // (a solve on reused values says nothing about the aggregation: it feeds neither the best count nor the rules below)
        // Redo the aggregation from the current values when that pays: always when the count has more
        // than doubled, and when it is > 25 % above the best while the PCG iterations it would save
        // over the remaining GN iterations exceed the set-up's cost (~150 PCG iterations' worth: host
        // aggregation + one more linearisation).  Counts only -- no clocks -- so that every rank of a
        // multi-GPU run takes the same decision.
        const int left = iters - it - 1;
        const bool doubled = eq_iter > 2 * best_pcg + 10;
        const bool pays = 4 * eq_iter > 5 * best_pcg && (long long)(eq_iter - best_pcg) * left > 150;
        if (!was_skipped && rebuilds < 3 && (doubled || pays)) rebuild_next = true;
      }
      if (out) {
        out->pcg_iters[it] = S.iter + wasted;   // an abandoned solve's iterations count too
        out->pcg_converged[it] = S.stop == 1;
        out->pcg_relres[it] = S.bb > 0 ? std::sqrt(S.rr / S.bb) : 0.0;
      }
      if (S.stop != 1) {
        // Solver failure, as LinearSolverEigen::solve returning false (OptimizationAlgorithm::Fail): the
        // step is NOT applied, estimates stay at the last successful update and the call returns 0 like
        // g2o::SparseOptimizer::optimize.  stop == 3: p.Hp <= 0 or non-finite (H not positive definite);
        // stop == 2: pcg_maxit iterations without reaching pcg_tol (an inexact step is never applied).
        if (S.stop == 3) {
          c->err = "PCG breakdown in GN iteration " + std::to_string(it) + " (Hessian not positive definite";
          if (c->amg && amg_coarsest_not_spd(c->amg, c->stream)) c->err += "; its coarsest Galerkin operator has a non-positive pivot";
          c->err += ")";
        } else {
          c->err = "PCG did not reach pcg_tol within pcg_maxit = " + std::to_string(c->opts.pcg_maxit) +
                   " iterations in GN iteration " + std::to_string(it) + " (relative residual " +
                   std::to_string(S.bb > 0 ? std::sqrt(S.rr / S.bb) : 0.0) + "); the step was not applied";
        }
        failed = true;
        hipEventRecord(ev[3 * it + 2], c->stream);
        break;
      }
      {
        Scope sc(c, K_POSE_UPDATE, 72.0 * c->n);
        launch_pose_update(c->stream, c->n, c->d_free_id, c->d_x, c->d_poses);
      }
      if (warm_env && c->amg && c->d_xprev) {   // (multi-GPU: the same replicated arithmetic on every rank)
        HIP_TRY(c, hipMemcpyAsync(c->d_xprev, c->d_x, sizeof(double) * 3 * (size_t)c->n, hipMemcpyDeviceToDevice, c->stream));
        c->warm_valid = !rebuild_next;   // a rebuilt hierarchy starts cold (its first solve sets the reference counts)
      }
      hipEventRecord(ev[3 * it + 2], c->stream);
      c->linearized = false;
      ++done;
      if (c->opts.verbose)
        std::fprintf(stderr, "[sgo] iteration= %d\t pcg= %d\t relres= %.3e\t |b|= %.3e\n", it, S.iter,
                     S.bb > 0 ? std::sqrt(S.rr / S.bb) : 0.0, std::sqrt(S.bb));
    }
    c->pcg_softcap = 0;
    if ((rc = do_chi2(c, c->d_hist + 2 * done, nullptr))) {
      return rc;
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_hist, c->d_hist, sizeof(double) * 2 * (size_t)(done + 1), hipMemcpyDeviceToHost,
                              c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    prof_flush(c);
    if (out) {
      out->iters_done = done;
      for (int k = 0; k <= done; ++k) {
        out->chi2[k] = c->h_hist[2 * k];
        out->robust_chi2[k] = c->h_hist[2 * k + 1];
      }
      const int timed = std::min(iters, done + 1);
      for (int k = 0; k < timed; ++k) {
        float a = 0.f, b = 0.f;
        hipEventElapsedTime(&a, ev[3 * k], ev[3 * k + 1]);
        hipEventElapsedTime(&b, ev[3 * k + 1], ev[3 * k + 2]);
        out->seconds_linearize[k] = a * 1e-3;
        out->seconds_solve[k] = b * 1e-3;
        out->seconds[k] = (a + b) * 1e-3;
      }
      out->seconds_total = wall_s() - t0;
    }
    return failed ? 0 : done;   // g2o: optimize() returns 0 when the algorithm reported Fail
  } SGO_CATCH(c)
}

// Micro-benchmark of the level-0 product on the resident graph: `reps` back-to-back launches of
// k_spmv0<mode> (operand = the PCG direction buffer, whatever it holds), HIP events around them on the
// context's stream; returns the mean microseconds per launch (< 0 on error).  variant 16: the wave-group kernel even when the graph has a tile view; variant 32: per-phase s_memtime stamps of the tile kernel on stderr (diagnostic).
double sgo_debug_spmv0_us(sgo_ctx* c, int mode, int variant, int reps) {
  if (check_graph(c) != SGO_OK || reps < 1 || c->n == 0 || ensure_rows(c) != SGO_OK) return -1.0;
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1.0;
  Spmv0Args args{};
  args.x = c->d_p;
  args.y = c->d_s2;
  args.b = c->d_b;
  args.omega = 0.8;
  const bool tiled = c->T0.ntile > 0 && !(variant & 16);   // variant 16: the wave-group kernel
  long long* d_st = nullptr;
  if ((variant & 32) && tiled) {
    hipMalloc((void**)&d_st, sizeof(long long) * 8 * (size_t)c->T0.ntile);
    hipMemset(d_st, 0, sizeof(long long) * 8 * (size_t)c->T0.ntile);
    args.dbg_stamps = d_st;
  }
  if (tiled) launch_spmv0t(c->stream, c->S0, c->T0, mode, args);
  else launch_spmv0(c->stream, c->S0, mode, args);
  hipEventRecord(a, c->stream);
  for (int k = 0; k < reps; ++k) {
    if (tiled) launch_spmv0t(c->stream, c->S0, c->T0, mode, args);
    else launch_spmv0(c->stream, c->S0, mode, args);
  }
  hipEventRecord(b, c->stream);
  hipStreamSynchronize(c->stream);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a);
  hipEventDestroy(b);
  if (d_st) {
    std::vector<long long> st(8 * (size_t)c->T0.ntile);
    hipMemcpy(st.data(), d_st, sizeof(long long) * st.size(), hipMemcpyDeviceToHost);
    hipFree(d_st);
    double ph[6] = {0, 0, 0, 0, 0, 0};
    for (int t = 0; t < c->T0.ntile; ++t)
      for (int q = 0; q < 6; ++q) ph[q] += (double)(st[8 * t + q + 1] - st[8 * t + q]);
    std::fprintf(stderr, "[sgo] tile kernel phases, shader cycles of wave 0 (s_memtime), mean over %d tiles: phase 0 %.0f, barrier %.0f, "
                 "phase 1 %.0f, barrier %.0f, phase 2 %.0f, barrier %.0f\n", c->T0.ntile,
                 ph[0] / c->T0.ntile, ph[1] / c->T0.ntile, ph[2] / c->T0.ntile, ph[3] / c->T0.ntile, ph[4] / c->T0.ntile,
                 ph[5] / c->T0.ntile);
    if (std::getenv("SGO_TILE_DUMP")) {   // per tile: rows, groups, stored blocks, halo columns, staged entries, phase cycles
      std::vector<TileDesc> td((size_t)c->T0.ntile);
      hipMemcpy(td.data(), c->T0.tile, sizeof(TileDesc) * td.size(), hipMemcpyDeviceToHost);
      int ng = 0;
      for (const TileDesc& T : td) ng = std::max(ng, T.g1);
      std::vector<int> g1((size_t)ng + 1);
      hipMemcpy(g1.data(), c->T0.grp1, sizeof(int) * g1.size(), hipMemcpyDeviceToHost);
      for (int t = 0; t < c->T0.ntile; ++t)
        std::fprintf(stderr, "TILE %d %d %d %d %d %d %lld %lld %lld %lld %lld %lld\n", t, td[t].row1 - td[t].row0, td[t].g1 - td[t].g0,
                     g1[td[t].g1] - g1[td[t].g0], td[t].h1 - td[t].h0, td[t].nstaged, st[8 * t + 1] - st[8 * t], st[8 * t + 2] - st[8 * t + 1],
                     st[8 * t + 3] - st[8 * t + 2], st[8 * t + 4] - st[8 * t + 3], st[8 * t + 5] - st[8 * t + 4], st[8 * t + 6] - st[8 * t + 5]);
    }
    long long s0 = st[0], s1 = st[0], e0 = st[6], e1 = st[6], dmin = st[6] - st[0], dmax = dmin;
    double dsum = 0.0;
    for (int t = 0; t < c->T0.ntile; ++t) {
      const long long a0 = st[8 * t], a6 = st[8 * t + 6], dd = a6 - a0;
      s0 = std::min(s0, a0); s1 = std::max(s1, a0); e0 = std::min(e0, a6); e1 = std::max(e1, a6);
      dmin = std::min(dmin, dd); dmax = std::max(dmax, dd); dsum += (double)dd;
    }
    std::fprintf(stderr, "[sgo] tile kernel, per tile (wave 0): cycles min %lld mean %.0f max %lld; first stamps spread over %lld cycles, "
                 "last stamps over %lld; first start to last end %lld cycles\n", dmin, dsum / c->T0.ntile, dmax, s1 - s0, e1 - e0, e1 - s0);
  }
  return 1e3 * ms / reps;
}

// Test hook for the multi-GPU scheme: the coarse right-hand side the first half of a multigrid cycle makes from r
// (hessian order, [n][3]): first sweep from zero, level-0 residual pass, restriction.  Under sgo_debug_set_shard it
// is this rank's partial; the partials of all ranks sum to the single-rank vector.  Returns its length (3 x coarse
// nodes), 0 without a multi-level hierarchy, < 0 on error.  Requires sgo_linearize.
int sgo_debug_coarse_rhs(sgo_ctx* c, const double* r, double* out, int cap) {
  int rc = check_graph(c);
  if (rc) return rc;
  if (!r || !out || !c->linearized) return SGO_EINVAL;
  if (ensure_amg(c)) return 0;
  if (!c->amg) return 0;
  if ((rc = vec_to_device(c, r, c->d_s1))) return rc;
  const int n3 = amg_debug_coarse_rhs(c->amg, c->stream, c->d_s1, c->d_s2, 3 * c->n);
  if (n3 <= 0) return n3;
  if (cap < n3) return SGO_EINVAL;
  HIP_TRY(c, hipMemcpyAsync(out, c->d_s2, sizeof(double) * (size_t)n3, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return n3;
}

int sgo_kernel_profile(sgo_ctx* c, sgo_kernel_stat* out, int cap) {
  try {
    if (!c || (cap > 0 && !out)) return SGO_EINVAL;
    prof_flush(c);
    for (int k = 0; k < K_COUNT && k < cap; ++k) {
      out[k].name = kKernelNames[k];
      out[k].launches = c->prof_launches[k];
      out[k].ms = c->prof_ms[k];
      out[k].bytes = c->prof_bytes[k];
    }
    return K_COUNT;
  } SGO_CATCH(c)
}

double sgo_profile_overhead_ms(sgo_ctx* c) {
  if (!c) return -1.0;
  if (c->prof_null_ms < 0.0) prof_calibrate(c);
  return c->prof_null_ms;
}

int sgo_profile_reset(sgo_ctx* c) {
  try {
    if (!c) return SGO_EINVAL;
    prof_flush(c);
    for (int k = 0; k < K_COUNT; ++k) {
      c->prof_ms[k] = 0;
      c->prof_launches[k] = 0;
      c->prof_bytes[k] = 0;
    }
    return SGO_OK;
  } SGO_CATCH(c)
}

int sgo_comm_unique_id(void* id_out) {
  if (!id_out) return SGO_EINVAL;
  return comm_unique_id(id_out, &g_err) ? SGO_OK : SGO_ECOMM;
}

int sgo_comm_init(sgo_ctx* c, int nranks, int rank, const void* unique_id) {
  try {
    if (!c || nranks < 1 || rank < 0 || rank >= nranks || !unique_id) return SGO_EINVAL;
    hipSetDevice(c->device);
    if (c->has_graph) {
      c->err = "sgo_comm_init must precede sgo_set_graph_se2";
      return SGO_EINVAL;
    }
    return c->comm.init(nranks, rank, unique_id, &c->err) ? SGO_OK : SGO_ECOMM;
  } SGO_CATCH(c)
}

int sgo_comm_init_host(sgo_ctx* c, int nranks, int rank, sgo_host_allreduce_fn fn, void* user) {
  try {
    if (!c || nranks < 1 || rank < 0 || rank >= nranks || !fn) return SGO_EINVAL;
    hipSetDevice(c->device);
    if (c->has_graph) {
      c->err = "sgo_comm_init_host must precede sgo_set_graph_se2";
      return SGO_EINVAL;
    }
    return c->comm.init_host(nranks, rank, fn, user) ? SGO_OK : SGO_ECOMM;
  } SGO_CATCH(c)
}

int sgo_comm_size(sgo_ctx* c) { return c ? c->comm.nranks : SGO_EINVAL; }

int sgo_debug_set_shard(sgo_ctx* c, int nranks, int rank) {
  if (!c || nranks < 1 || rank < 0 || rank >= nranks) return SGO_EINVAL;
  if (c->has_graph) {
    c->err = "sgo_debug_set_shard must precede sgo_set_graph_se2";
    return SGO_EINVAL;
  }
  c->comm.destroy();
  c->comm.nranks = nranks;  // no handle: Comm::allreduce_* are no-ops
  c->comm.rank = rank;
  return SGO_OK;
}

int sgo_plan_rows(int32_t V, const double* poses, const uint8_t* fixed, int32_t E, const int32_t* ei, const int32_t* ej,
                  int32_t nranks, int32_t* n_free, int32_t* row_vertex, int32_t* ntiles, int32_t* tile_row_begin,
                  int32_t tile_cap, int32_t* rank_row_begin) {
  if (V <= 0 || E < 0 || !poses || !fixed || (E > 0 && (!ei || !ej)) || nranks < 1 || !n_free) return SGO_EINVAL;
  try {
    RowPlan P;
    std::string err;
    const int rc = plan_rows(V, poses, fixed, E, ei, ej, 256, &err, P);
    if (rc != SGO_OK) {
      g_err = err;
      return rc;
    }
    *n_free = P.n;
    if (row_vertex) std::copy(P.row_vertex.begin(), P.row_vertex.end(), row_vertex);
    const int nt = (int)P.tiles.size();
    if (ntiles) *ntiles = nt;
    if (tile_row_begin) {
      if (tile_cap < nt + 1) {
        g_err = "sgo_plan_rows: tile_cap too small";
        return SGO_EINVAL;
      }
      for (int t = 0; t < nt; ++t) tile_row_begin[t] = P.tiles[t].row0;
      tile_row_begin[nt] = P.n;
    }
    if (rank_row_begin) {
      for (int r = 0; r <= nranks; ++r) {
        if (nt == 0) {   // no tile view: the wave-group kernel is sharded by row groups; report an even row split
          rank_row_begin[r] = (int32_t)((long long)P.n * r / nranks);
          continue;
        }
        int32_t b = 0, e = 0;
        sgo_shard_range(nt, nranks, std::min(r, nranks - 1), &b, &e);
        rank_row_begin[r] = r == nranks ? P.n : P.tiles[b].row0;
        (void)e;
      }
    }
    return SGO_OK;
  } catch (const std::bad_alloc&) {
    g_err = "sgo_plan_rows: out of host memory";
    return SGO_ENOMEM;
  }
}

void sgo_shard_range(int32_t count, int32_t nranks, int32_t rank, int32_t* begin, int32_t* end) {
  if (nranks < 1) nranks = 1;
  if (rank < 0) rank = 0;
  if (rank >= nranks) rank = nranks - 1;
  const long long lo = (long long)count * rank / nranks, hi = (long long)count * (rank + 1) / nranks;
  if (begin) *begin = (int32_t)lo;
  if (end) *end = (int32_t)hi;
}

}  // extern "C"
