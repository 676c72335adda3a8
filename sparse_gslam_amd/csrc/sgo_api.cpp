// sgo_api.cpp -- the C-ABI of include/sgo.h (argument checks, error convention, no exception across the boundary) over
// the host runtime of sgo_plan.cpp / sgo_structure.cpp / sgo_solve.cpp and the HIP kernels.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "sgo_ctx.h"

using namespace sgo;

namespace sgo {
thread_local std::string g_err;  // for ctx == NULL
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
      (e = hipHostMalloc((void**)&c->h_Sz, sizeof(PcgScalars))) != hipSuccess ||
      (e = hipHostGetDevicePointer((void**)&c->d_Sz, c->h_Sz, 0)) != hipSuccess ||
      (e = hipHostMalloc((void**)&c->h_dchg, 6 * sizeof(double) * kMaxPartials)) != hipSuccess ||
      (e = hipHostGetDevicePointer((void**)&c->h_dchg_dev, c->h_dchg, 0)) != hipSuccess ||
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
  c->amg_arena_prev.release();
  c->amg_tmp_arena.release();
  c->comm.destroy();
  overlay_release(c->ov);
  for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
  for (hipEvent_t e : c->iter_events) hipEventDestroy(e);
  if (c->h_S) hipHostFree(c->h_S);
  if (c->h_S2) hipHostFree(c->h_S2);
  if (c->h_Sz) hipHostFree(c->h_Sz);
  if (c->h_dchg) hipHostFree(c->h_dchg);
  if (c->h_pose_stage) hipHostFree(c->h_pose_stage);
  for (hipEvent_t ev : c->ev_S)
    if (ev) hipEventDestroy(ev);
  if (c->h_hist) hipHostFree(c->h_hist);
  if (c->halo_send) hipFree(c->halo_send);
  if (c->halo_recv) hipFree(c->halo_recv);
  if (c->h_dres) hipHostFree(c->h_dres);
  if (c->d_dres) hipFree(c->d_dres);
  if (c->d_comm_flag) hipFree(c->d_comm_flag);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

const char* sgo_solver_description(sgo_ctx* c) {
  if (!c || !c->has_graph) return "";
  try {
    c->solver_text = c->solver_desc;
    if (!c->direct && !c->direct_why.empty()) c->solver_text += "; direct path not used: " + c->direct_why;
    if (!c->direct && !c->mf && !c->mf_why.empty()) c->solver_text += "; multifrontal path not used: " + c->mf_why;
    if (c->ov.active)
      c->solver_text += "; incremental overlay: " + std::to_string(c->ov.new_vertex.size()) + " appended rows (" + std::to_string(c->ov.dev.nx) +
                        " hubs), " + std::to_string(c->ov.dev.nt) + " touched rows, " + std::to_string(c->ov.dev.el.cnt) + " appended edges (" +
                        std::to_string(c->ov.updates) + " updates)";
    if (!c->update_note.empty()) c->solver_text += "; last update: " + c->update_note;
    if (!c->lag_note.empty()) c->solver_text += "; " + c->lag_note;
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
    read_call_knobs(c);   // (which set-up the multigrid hierarchy takes is decided here: the helper thread of the pipeline depends on it)
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
      // Mid-size graphs (the reference's largest: a few thousand poses, a closure every few poses): a multifrontal sparse
      // Cholesky factorisation, one launch per level of the elimination tree (sgo_mfront.h), when the tree's critical path
      // is short enough.  SGO_MFRONT=0 keeps the multigrid PCG; SGO_MFRONT_ROWS bounds the graph size.
      bool mf_on = true;
      int mf_rows = 12288;
      if (const char* s = std::getenv("SGO_MFRONT")) mf_on = std::atoi(s) != 0;
      if (const char* s = std::getenv("SGO_MFRONT_ROWS")) mf_rows = std::atoi(s);
      c->mf_why.clear();
      if (!c->direct && mf_on && mf_rows > 0) {
        std::string merr;
        const double tm0 = wall_s();
        c->mf = mfront_create(c->stream, &c->graph_arena, V, c->n, c->free_id.data(), poses, E, ei, ej, mf_rows, &c->mf_order_hint, &c->mf_why, &merr);
        if (!c->mf && !merr.empty()) {
          c->err = merr;
          free_graph(c);
          return SGO_EHIP;
        }
        if (c->opts.verbose) std::fprintf(stderr, "[sgo] set_graph: multifrontal analysis %.2f ms\n", 1e3 * (wall_s() - tm0));
        if (c->mf) {
          const MfrontInfo& mi = mfront_info(c->mf);
          char buf[320];
          std::snprintf(buf, sizeof buf, "multifrontal_cholesky: %d fronts in %d levels (nested dissection in %s order), largest front %d rows "
                        "(%d own + %d boundary poses), %.0f Mflop per factorisation, %.1f on the critical path in %d panels, %.1f MB of fronts; "
                        "pcg_amg on demand", mi.fronts, mi.height + 1, mi.order_kind == 0 ? "Hilbert" : "id", mi.max_dim, mi.max_own, mi.max_bnd,
                        1e-6 * mi.flops, 1e-6 * mi.crit_flops, mi.crit_panels, 1e-6 * (double)mi.arena_bytes);
          c->solver_desc = buf;
          c->amg_pending = true;
          c->rows_pending = true;
          c->lz_fixed.assign(fixed, fixed + V);
          c->lz_ei.assign(ei, ei + E);
          c->lz_ej.assign(ej, ej + E);
        } else if (c->opts.verbose) {
          std::fprintf(stderr, "[sgo] multifrontal path not used: %s\n", c->mf_why.c_str());
        }
      }
    }
    if (!c->rows_pending) {
      // a graph whose initial poses contradict its closures gets its ROW ORDER from spanning-tree positions (sgo_plan.cpp)
      const double tp0 = wall_s();
      if (!plan_order_positions(V, poses, fixed, E, ei, ej, meas, c->order_xy)) c->order_xy.clear();
      else if (c->opts.verbose)
        std::fprintf(stderr, "[sgo] set_graph: the initial poses contradict the closures: row order from spanning-tree positions (%.1f ms)\n",
                     1e3 * (wall_s() - tp0));
    }
    if (!c->rows_pending && (rc = build_rows(c, poses, fixed, ei, ej)) != SGO_OK) {
      free_graph(c);
      return rc;
    }
    if (c->opts.solver != SGO_SOLVER_PCG_AMG) c->solver_desc += multi_gpu_description(c);
    if (c->opts.solver == SGO_SOLVER_PCG_AMG && c->n > 0 && !c->direct && !c->mf) {
      // the hierarchy is built from the Hessian at the initial poses (strength of connection)
      const double ta0 = wall_s();
      if ((rc = do_linearize(c)) != SGO_OK || (rc = build_amg(c)) != SGO_OK) {
        free_graph(c);
        return rc;
      }
      if (c->opts.verbose) std::fprintf(stderr, "[sgo] set_graph: multigrid set-up %.1f ms\n", 1e3 * (wall_s() - ta0));
      c->linearized = false;
    }
    // what an incremental update (sgo_update_graph_se2) appends to
    c->ov.base_V = V;
    c->ov.base_E = E;
    c->ov.base_n = c->n;
    c->ov.fixed.assign(fixed, fixed + V);
    c->setup_seconds = wall_s() - t0;
    if (c->opts.verbose)
      std::fprintf(stderr, "[sgo] solver: %s\n[sgo] set_graph: total %.1f ms\n", c->solver_desc.c_str(), 1e3 * c->setup_seconds);
    return SGO_OK;
  } SGO_CATCH(c)
}

// Incremental re-initialisation: see include/sgo.h and sgo_overlay.h.
int sgo_update_graph_se2(sgo_ctx* c, int32_t V, const double* poses, const uint8_t* fixed, int32_t E, const int32_t* ei,
                         const int32_t* ej, const double* meas, const double* info, const double* phi, int32_t n_resident_edges) {
  try {
    if (!c) return SGO_EINVAL;
    if (V <= 0 || E < 0 || !poses || !fixed || (E > 0 && (!ei || !ej || !meas || !info || !phi)) || n_resident_edges < 0 || n_resident_edges > E) {
      c->err = "sgo_update_graph_se2: bad argument";
      return SGO_EINVAL;
    }
    const double t0 = wall_s();
    Overlay& ov = c->ov;
    std::string why;
    const int res_E = c->has_graph ? ov.base_E + (int)ov.ei.size() : 0;
    bool env_on = true;
    if (const char* e = std::getenv("SGO_INCREMENTAL")) env_on = std::atoi(e) != 0;
    if (!c->has_graph) why = "no resident graph";
    else if (!env_on) why = "disabled (SGO_INCREMENTAL=0)";
    else if (n_resident_edges == 0) why = "the caller reports no common prefix";
    // (c->direct / c->mf themselves, not only the pending flags: a single-step entry point -- sgo_linearize, sgo_solve -- builds the
    // PCG structures of such a graph on demand and clears the flags, while sgo_optimize_gn keeps taking the factorisation path,
    // which knows nothing of an overlay)
    else if (c->mf) why = "the resident graph takes the multifrontal path (its set-up is cheap)";
    else if (c->direct) why = "the resident graph takes the single-launch direct path (its set-up is cheap)";
    else if (c->rows_pending || c->amg_pending) why = "the resident graph's PCG structures are not built";
    else if (!c->amg || c->opts.solver != SGO_SOLVER_PCG_AMG) why = "no multigrid hierarchy resident";
    else if (c->comm.nranks > 1 || c->comm.active()) why = "multi-GPU contexts re-partition";
    else if (n_resident_edges != res_E || V < c->V) why = "the resident graph is not a prefix of the new one";
    else if (V - ov.base_V > kOvMaxVerts || E - ov.base_E > kOvMaxEdges) why = "the appended part has outgrown the overlay";
    else if (std::memcmp(fixed, ov.fixed.data(), (size_t)c->V) != 0) why = "fixed flags of resident vertices changed";
    else if (ov.hpos.size() != (size_t)ov.base_V) why = "no row map of the resident structure";
    // an overlay that costs too many PCG iterations (counts only: the first solves of the optimize() calls, which run at the same
    // relative tolerance) is dropped for a fresh hierarchy that knows the closures
    else if (ov.active && c->its_base > 0 && 4 * c->its_last > 5 * c->its_base + 12) why = "the overlay costs too many PCG iterations";
    if (why.empty()) {
      for (int e = n_resident_edges; e < E; ++e) {
        const int a = ei[e], b = ej[e];
        if (a < 0 || a >= V || b < 0 || b >= V) {
          c->err = "edge " + std::to_string(e) + " references a vertex outside [0, V)";
          return SGO_EINVAL;
        }
        if (a == b) {
          c->err = "edge " + std::to_string(e) + " is a self edge";
          return SGO_EINVAL;
        }
      }
      hipError_t he = hipSetDevice(c->device);
      if (he != hipSuccess) {
        c->err = std::string("hipSetDevice: ") + hipGetErrorString(he);
        return SGO_EHIP;
      }
      // the update is tried on copies of the host state: a shape the overlay cannot take leaves the resident graph as it was
      const size_t old_ne = ov.ei.size();
      const int dE = E - n_resident_edges;
      ov.ei.insert(ov.ei.end(), ei + n_resident_edges, ei + E);
      ov.ej.insert(ov.ej.end(), ej + n_resident_edges, ej + E);
      const std::vector<unsigned char> old_fixed = ov.fixed;
      ov.fixed.assign(fixed, fixed + V);
      double tl = wall_s();
      auto lap = [&](const char* what) {
        if (c->opts.verbose > 2) hipStreamSynchronize(c->stream);   // (diagnostic: charge the device time to the phase that queued it)
        const double t = wall_s();
        if (c->opts.verbose > 1) std::fprintf(stderr, "[sgo]   update %-16s %.3f ms\n", what, 1e3 * (t - tl));
        tl = t;
      };
      hipStreamSynchronize(c->stream);   // (nothing of an earlier optimize() may still read the overlay's structure arrays)
      lap("stream drain");
      bool ok = overlay_upload_edges(ov, c->stream, (int)old_ne, 0, nullptr, nullptr, nullptr, nullptr, nullptr, &c->err);   // (allocates on first use)
      const OverlayDev old_dev = ov.dev;   // (with the device pointers in place: restored when the new shape is refused)
      const std::vector<int> old_new = ov.new_vertex;
      if (ok) ok = overlay_upload_edges(ov, c->stream, (int)old_ne, dE, ei + n_resident_edges, ej + n_resident_edges,
                                     meas + 3 * (size_t)n_resident_edges, info + 6 * (size_t)n_resident_edges, phi + n_resident_edges, &c->err);
      lap("edge upload");
      if (ok) ok = overlay_build(ov, V, c->stream, &why, &c->err);
      lap("overlay build");
      if (ok && upload_poses(c, poses, V) != SGO_OK) {
        ok = false;
        why = "pose upload failed";
      }
      if (ok && hipStreamSynchronize(c->stream) != hipSuccess) {   // the caller's arrays may go away after the call
        ok = false;
        why = "device error";
      }
      lap("poses + sync");
      if (ok) {
        ov.active = ov.dev.k + ov.dev.nt + ov.dev.nx > 0;
        ov.updates++;
        c->V = V;
        c->E = E;
        c->linearized = false;
        c->warm_valid = false;
        // the hierarchy's own staleness rule compares a solve with the best count seen so far (sgo_solve.cpp): the solves after
        // an update have another right-hand side (the appended poses' residual) -- their first one sets a new reference
        c->amg_best = 0;
        c->setup_seconds = wall_s() - t0;
        c->update_note = "incremental (" + std::to_string(dE) + " edges appended in " + std::to_string(1e3 * c->setup_seconds).substr(0, 5) + " ms)";
        if (c->opts.verbose)
          std::fprintf(stderr, "[sgo] update_graph: %d edges / %d vertices appended as overlay (%d chain rows, %d hubs, %d touched) in %.2f ms\n", dE,
                       V - (int)old_fixed.size(), ov.dev.k, ov.dev.nx, ov.dev.nt, 1e3 * c->setup_seconds);
        return SGO_OK;
      }
      ov.ei.resize(old_ne);
      ov.ej.resize(old_ne);
      ov.fixed = old_fixed;
      ov.dev = old_dev;
      ov.new_vertex = old_new;
      if (why.empty()) why = "overlay set-up failed";
    }
    if (c->opts.verbose) std::fprintf(stderr, "[sgo] update_graph: full set-up (%s)\n", why.c_str());
    const int rc = sgo_set_graph_se2(c, V, poses, fixed, E, ei, ej, meas, info, phi);
    if (rc == SGO_OK) c->update_note = "full set-up (" + why + ")";
    return rc;
  } SGO_CATCH(c)
}

int sgo_set_poses(sgo_ctx* c, const double* poses) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if (!poses) return SGO_EINVAL;
    if ((rc = upload_poses(c, poses, c->V))) return rc;
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
  return rc ? rc : c->n + (c->ov.active ? (int)c->ov.new_vertex.size() : 0);
}

int sgo_free_ids(sgo_ctx* c, int32_t* out) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if (!out) return SGO_EINVAL;
    if (c->ov.active && !c->ov.new_vertex.empty()) {   // g2o's hessian order over the resident and the appended free poses
      std::merge(c->free_id.begin(), c->free_id.end(), c->ov.new_vertex.begin(), c->ov.new_vertex.end(), out);
      return c->n + (int)c->ov.new_vertex.size();
    }
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
    if (c->ov.active) {
      c->err = "single-step entry points need a full set-up: the resident graph carries an incremental overlay (call sgo_set_graph_se2)";
      return SGO_EINVAL;
    }
    if (c->n == 0) return SGO_ENOTHING;
    if ((rc = ensure_amg(c))) return rc;
    if ((rc = do_chi2(c, c->d_hist, nullptr))) return rc;
    if ((rc = do_linearize(c))) return rc;
    if (c->owner && !halo_gather_slices(c->halo, c->stream, c->d_dgb, 9, &c->err)) return SGO_ECOMM;   // every rank reports all rows
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
    if (c->ov.active) {
      c->err = "single-step entry points need a full set-up: the resident graph carries an incremental overlay (call sgo_set_graph_se2)";
      return SGO_EINVAL;
    }
    if (!x || !y) return SGO_EINVAL;
    if (!c->linearized) {
      c->err = "sgo_hessian_apply: call sgo_linearize first";
      return SGO_EINVAL;
    }
    if ((rc = vec_to_device(c, x, c->d_s1))) return rc;
    if ((rc = do_spmv(c, c->d_s1, c->d_s2, false, nullptr, nullptr))) return rc;
    if (c->owner && !halo_gather_slices(c->halo, c->stream, c->d_s2, 3, &c->err)) return SGO_ECOMM;
    return vec_from_device(c, c->d_s2, y);
  } SGO_CATCH(c)
}

int sgo_precondition(sgo_ctx* c, const double* r, double* z) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if (c->ov.active) {
      c->err = "single-step entry points need a full set-up: the resident graph carries an incremental overlay (call sgo_set_graph_se2)";
      return SGO_EINVAL;
    }
    if (!r || !z) return SGO_EINVAL;
    if (!c->linearized) {
      c->err = "sgo_precondition: call sgo_linearize first";
      return SGO_EINVAL;
    }
    if ((rc = vec_to_device(c, r, c->d_s1))) return rc;
    if (c->amg) amg_apply(c->amg, c->stream, c->d_s1, c->d_s2, nullptr, nullptr, nullptr);
    else if (c->owner) launch_precond_bj(c->stream, c->halo.row1 - c->halo.row0, c->S0.dinv + 6 * (size_t)c->halo.row0, c->d_s1 + 3 * (size_t)c->halo.row0, c->d_s2 + 3 * (size_t)c->halo.row0, 1.0);
    else launch_precond_bj(c->stream, c->n, c->S0.dinv, c->d_s1, c->d_s2, 1.0);
    if (c->amg && amg_comm_failed(c->amg)) return SGO_ECOMM;
    if (c->owner && !halo_gather_slices(c->halo, c->stream, c->d_s2, 3, &c->err)) return SGO_ECOMM;
    return vec_from_device(c, c->d_s2, z);
  } SGO_CATCH(c)
}

int sgo_solve(sgo_ctx* c, double* x, double* relres) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if (c->ov.active) {
      c->err = "single-step entry points need a full set-up: the resident graph carries an incremental overlay (call sgo_set_graph_se2)";
      return SGO_EINVAL;
    }
    if (!c->linearized) {
      c->err = "sgo_solve: call sgo_linearize first";
      return SGO_EINVAL;
    }
    read_call_knobs(c);
    // restart from the state of the last linearisation (idempotent re-finalize)
    int grid = 0;
    launch_finalize(c->stream, c->S0, c->owner ? c->halo.row0 : 0, c->owner ? c->halo.row1 : c->n, c->d_dgb, c->d_b, c->d_x, c->d_r,
                    c->d_z, c->d_p, c->amg ? amg_xs0(c->amg) : nullptr, c->amg ? amg_omega(c->amg) : 0.0, c->d_partials, &grid);
    if ((rc = start_pcg(c, grid))) return rc;
    if ((rc = run_pcg(c))) return rc;
    if (c->owner && !halo_gather_slices(c->halo, c->stream, c->d_x, 3, &c->err)) return SGO_ECOMM;
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
    return optimize_gn(c, iters, out);
  } SGO_CATCH(c)
}

double sgo_debug_spmv0_us(sgo_ctx* c, int mode, int variant, int reps) { return debug_spmv0_us(c, mode, variant, reps); }

// Test hook for the multi-GPU scheme: the coarse right-hand side the first half of a multigrid cycle makes from r
// (hessian order, [n][3]): first sweep from zero, level-0 residual pass, restriction.  Under sgo_debug_set_shard it
// is this rank's partial; the partials of all ranks sum to the single-rank vector.  Returns its length (3 x coarse
// nodes), 0 without a multi-level hierarchy, < 0 on error.  Requires sgo_linearize.
int sgo_debug_coarse_rhs(sgo_ctx* c, const double* r, double* out, int cap) {
  int rc = check_graph(c);
  if (rc) return rc;
  if (!r || !out || !c->linearized || c->ov.active) return SGO_EINVAL;
  if (c->owner) {   // this rank holds its own rows' blocks and transfer entries only: the hook walks whole-graph arrays
    c->err = "sgo_debug_coarse_rhs: not available in multi-GPU row-owner mode";
    return SGO_EINVAL;
  }
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

// Diagnostic (env SGO_LANCZOS=1 at sgo_set_graph_se2): alpha / beta of every PCG iteration of the last solve, pairs in
// iteration order; returns the number of iterations written (the Lanczos matrix of the preconditioned operator follows
// from them: scripts/ritz_probe.py), < 0 on error.
int sgo_debug_lanczos(sgo_ctx* c, double* out, int cap) {
  try {
    int rc = check_graph(c);
    if (rc) return rc;
    if (!c->d_lanczos || !out || !c->h_S) return SGO_EINVAL;
    const int n = std::min(std::min(c->h_S->iter, (int)kLanczosMax), cap);
    if (n > 0) {
      std::vector<double> t(3 * (size_t)n);
      HIP_TRY(c, hipMemcpyAsync(t.data(), c->d_lanczos, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      for (int j = 0; j < n; ++j) {
        out[2 * j] = t[3 * (size_t)j];
        out[2 * j + 1] = t[3 * (size_t)j + 1];
      }
    }
    return n;
  } SGO_CATCH(c)
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

int sgo_kernel_profile_samples(sgo_ctx* c, int slot, float* out_ms, int cap) {
  try {
    if (!c || slot < 0 || slot >= K_COUNT || cap < 0 || (cap > 0 && !out_ms)) return SGO_EINVAL;
    prof_flush(c);
    const std::vector<float>& v = c->prof_samples[slot];
    const int n = (int)std::min<size_t>(v.size(), (size_t)cap);
    std::copy(v.begin(), v.begin() + n, out_ms);
    return n;
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
      c->prof_samples[k].clear();
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

int sgo_comm_host_allgather(sgo_ctx* c, sgo_host_allgather_fn fn) {
  if (!c || !c->comm.host_fn) return SGO_EINVAL;
  if (c->has_graph) {
    c->err = "sgo_comm_host_allgather must precede sgo_set_graph_se2";
    return SGO_EINVAL;
  }
  c->comm.host_gather_fn = fn;
  return SGO_OK;
}

int64_t sgo_debug_level0_bytes(sgo_ctx* c) {
  if (check_graph(c) != SGO_OK) return -1;
  return (int64_t)c->level0_bytes + (c->amg ? (int64_t)amg_level0_bytes(c->amg) : 0);
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


}  // extern "C"
