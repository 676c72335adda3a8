// sgo_solve.cpp -- host side of the Gauss-Newton iteration.
//
// Mirrors the control flow of g2o's SparseOptimizer::optimize() with OptimizationAlgorithmGaussNewton as sparse-gslam
// configures it (src/sparse_gslam/src/graphs.cpp:17-23; called at submap_loop_closer.cpp:286-288 and
// log_runner.cpp:203-204):
//     for k in 0..iters:  computeActiveErrors; buildSystem; solve; update
// with the linear solve done by preconditioned CG on the device instead of LinearSolverEigen.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "sgo_ctx.h"
#include "sgo_rules.h"

namespace sgo {

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
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      c->prof_ms[r.kid] += ms;
      if (c->prof_samples[r.kid].size() < sgo_ctx::kProfSamples) c->prof_samples[r.kid].push_back(ms);
    }
    c->ev_pool.push_back(r.a);
    c->ev_pool.push_back(r.b);
  }
  c->pending.clear();
}

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

// ---- one GN building block each ------------------------------------------------------------
int do_chi2(sgo_ctx* c, double* d_out2, double* d_e2) {
  int grid = 0;
  {
    Scope sc(c, K_CHI2, bytes_chi2(c));
    int e0 = 0, e1 = c->el.E;   // (the resident edge list; c->E also counts the edges an incremental update appended: el2 below)
    if (c->comm.nranks > 1 && !c->replicated && !d_e2) sgo_shard_range(c->el.E, c->comm.nranks, c->comm.rank, &e0, &e1);
    launch_chi2(c->stream, c->el, e0, e1, c->d_poses, d_e2, c->d_partials, &grid, c->ov.active ? &c->ov.dev.el : nullptr);
  }
  {
    Scope sc(c, K_REDUCE2, 16.0 * grid);
    launch_reduce2(c->stream, c->d_partials, grid, d_out2);
  }
  if (c->comm.nranks > 1 && !c->replicated && !d_e2 && !c->comm.allreduce_f64(d_out2, 2, c->stream, &c->err)) return SGO_ECOMM;
  return SGO_OK;
}

// PCG start state after k_finalize (x = 0, r = b, z = Dinv b, p = z; partials rz / bb with `grid`
// entries).  With the AMG preconditioner: refresh the coarse operators, z = M^-1 b, p = z.
int do_spmv(sgo_ctx* c, const double* x, double* y, bool dot, const PcgScalars* S, int* grid_out);

// Row-owner mode (multi-GPU): every vector lives on this rank's rows [row0, row1); a product needs the neighbours'
// boundary rows of its operand (one exchange), a dot product the other ranks' partial sums (they ride on an exchange and
// are reduced in rank order by the consumer: bit-identical scalars, hence identical decisions, on every rank).
static bool xch(sgo_ctx* c, double* vec, const HaloScalars& sc) {
  return halo_exchange(c->halo, c->stream, vec, 3, c->halo.bnd, c->halo.bmax, sc, &c->err);
}
static HaloScalars scal(const double* p0, int n0, const double* p1 = nullptr, int n1 = 0, const double* p2 = nullptr, int n2 = 0) {
  HaloScalars sc;
  sc.parts[0] = p0; sc.n[0] = n0;
  sc.parts[1] = p1; sc.n[1] = n1;
  sc.parts[2] = p2; sc.n[2] = n2;
  return sc;
}

static void set_probe(sgo_ctx* c, int k, double max);
static int start_pcg_owner(sgo_ctx* c, int grid) {
  set_probe(c, 0, 0.0);   // (the progress probe belongs to the single-GPU lagged refresh: never armed here)
  const HaloDev& H = c->halo;
  const int G = H.G, nr = H.row1 - H.row0;
  const size_t o3 = 3 * (size_t)H.row0;
  const int maxit = c->pcg_softcap > 0 ? std::min(c->pcg_softcap, c->opts.pcg_maxit) : c->opts.pcg_maxit;
  const double tol = c->opts.pcg_tol * c->tol_scale;
  double* bb_parts = c->d_partials + kMaxPartials;   // k_finalize: [0] r.z (block-Jacobi), [1] b.b over the owned rows
  int rc;
  if (c->amg && !c->amg_skip_update && (rc = amg_update(c->amg, c->stream, &c->err))) return rc;
  if (c->amg && amg_comm_failed(c->amg)) return SGO_ECOMM;
  // the block-diagonal inverse of the neighbours' boundary rows (the halo-row recurrences of the iterations apply it)
  if (c->amg && !halo_exchange(H, c->stream, c->S0.dinv, 6, H.bnd, H.bmax, HaloScalars(), &c->err)) return SGO_ECOMM;
  if (!c->amg) {   // block-Jacobi: k_finalize left z = Dinv b, p = z and the partials of r.z
    if (!xch(c, c->d_p, scal(c->d_partials, grid, bb_parts, grid))) return SGO_ECOMM;
    Scope sc(c, K_INIT_SCALARS, 16.0 * G);
    launch_init_scalars(c->stream, c->d_S, H.gparts, G, H.gparts + G, G, tol, maxit, c->bb_ref, c->tol_cap);
    return SGO_OK;
  }
  if (c->warm_valid && c->d_xprev) {
    // warm start (see start_pcg): x_prev is the previous step, gathered to every rank before the pose update
    int gq = 0, gd = 0;
    // (the product's kernel stores both of its dot-product rows: k_finalize's b.b partials move out of its way first)
    double* bb_keep = c->d_zparts + kMaxPartials;
    HIP_TRY(c, hipMemcpyAsync(bb_keep, bb_parts, sizeof(double) * (size_t)grid, hipMemcpyDeviceToDevice, c->stream));
    if ((rc = do_spmv(c, c->d_xprev, c->d_q, true, nullptr, &gq))) return rc;
    {
      Scope sc(c, K_DOT, 48.0 * nr);
      launch_dot(c->stream, 3 * nr, c->d_b + o3, c->d_xprev + o3, c->d_partials + 2 * kMaxPartials, nullptr, &gd);
    }
    if (!xch(c, nullptr, scal(bb_keep, grid, c->d_partials, gq, c->d_partials + 2 * kMaxPartials, gd))) return SGO_ECOMM;
    {
      Scope sc(c, K_INIT_SCALARS, 16.0 * G);   // ||b||^2, tolerance, iteration count (r.z is replaced below)
      launch_init_scalars(c->stream, c->d_S, H.gparts, G, H.gparts, G, tol, maxit, c->bb_ref, c->tol_cap);
    }
    {
      Scope sc(c, K_UPDATE_XR, 120.0 * nr);
      launch_warm_start(c->stream, 3 * nr, c->d_xprev + o3, c->d_q + o3, c->d_b + o3, c->d_x + o3, c->d_r + o3, H.gparts + G, G,
                        H.gparts + 2 * G, G);
    }
    const int gz = amg_apply(c->amg, c->stream, c->d_r, c->d_z, c->d_r, c->d_zparts, nullptr, nullptr, 0);
    if (amg_comm_failed(c->amg)) return SGO_ECOMM;
    HIP_TRY(c, hipMemcpyAsync(c->d_p + o3, c->d_z + o3, sizeof(double) * 3 * (size_t)nr, hipMemcpyDeviceToDevice, c->stream));
    if (!xch(c, c->d_p, scal(c->d_zparts, gz))) return SGO_ECOMM;
    {
      Scope sc(c, K_INIT_SCALARS, 8.0 * G);
      launch_restart_scalars(c->stream, c->d_S, H.gparts, G, maxit, 1);
    }
    if (!xch(c, c->d_r, HaloScalars())) return SGO_ECOMM;   // the halo rows' copies of r start here
    return SGO_OK;
  }
  const int gz = amg_apply(c->amg, c->stream, c->d_r, c->d_z, c->d_r, c->d_zparts, nullptr, nullptr, 1);
  if (amg_comm_failed(c->amg)) return SGO_ECOMM;
  HIP_TRY(c, hipMemcpyAsync(c->d_p + o3, c->d_z + o3, sizeof(double) * 3 * (size_t)nr, hipMemcpyDeviceToDevice, c->stream));
  if (!xch(c, c->d_p, scal(c->d_zparts, gz, bb_parts, grid))) return SGO_ECOMM;
  {
    Scope sc(c, K_INIT_SCALARS, 16.0 * G);
    launch_init_scalars(c->stream, c->d_S, H.gparts, G, H.gparts + G, G, tol, maxit, c->bb_ref, c->tol_cap);
  }
  if (!xch(c, c->d_r, HaloScalars())) return SGO_ECOMM;   // the halo rows' copies of r start here
  return SGO_OK;
}

// what k_update_p records besides the recurrence: the pinned mirror of the scalars and (diagnostic, env SGO_LANCZOS=1) the
// recurrence's coefficients per iteration -- the Lanczos matrix of M^-1 H follows from them (scripts/ritz_probe.py)
static RecDev rec_dev(const sgo_ctx* c) {
  RecDev r;
  r.mirror = c->d_Sz;
  r.lanczos = c->d_lanczos;
  return r;
}

static void read_diag_change(sgo_ctx* c) {   // (after a synchronisation behind the launch)
  const volatile double* h = c->h_dchg;
  for (int k = 0; k < 3; ++k) {
    double v = 0.0;
    for (int i = 0; i < c->dchg_grid; ++i) v += h[(size_t)k * kMaxPartials + i];
    c->last_dchg[k] = v;
  }
  c->amg_dchg_pending = false;
}
static void set_probe(sgo_ctx* c, int k, double max) {
  if (k == c->probe_dev_k && max == c->probe_dev_max) return;
  launch_set_probe(c->stream, c->d_S, k, max);
  c->probe_dev_k = k;
  c->probe_dev_max = max;
}

int start_pcg(sgo_ctx* c, int grid) {
  if (c->owner) return start_pcg_owner(c, grid);
  const int maxit = c->pcg_softcap > 0 ? std::min(c->pcg_softcap, c->opts.pcg_maxit) : c->opts.pcg_maxit;
  const double tol = c->opts.pcg_tol * c->tol_scale;
  if (c->amg) {
    int rc = c->amg_skip_update ? SGO_OK : amg_update(c->amg, c->stream, &c->err);
    if (rc) return rc;
    const bool warm = c->warm_valid && c->d_xprev;
    if (warm) {
      // Start from the previous Gauss-Newton step scaled by the energy-optimal factor: consecutive steps of a linearly
      // converging iteration are nearly parallel, ||b - gamma H x_prev|| is 0.2-0.45 ||b|| on C4 / C2 (scripts/
      // warm_probe.py), i.e. two PCG iterations for the price of one Hessian product.  Same stopping test, same
      // solution; only the path to it is shorter.
      {
        Scope sc(c, K_INIT_SCALARS, 16.0 * grid);   // ||b||^2, tolerance, iteration count (r.z is replaced below)
        launch_init_scalars(c->stream, c->d_S, c->d_partials, grid, c->d_partials + kMaxPartials, grid, tol, maxit, c->bb_ref, c->tol_cap);
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
    }
    // (cold: k_finalize left x = 0, r = b and xs = omega Dinv b, the cycle's first sweep from zero)
    const int xs_ready = warm ? 0 : 1;
    const int gz = amg_apply(c->amg, c->stream, c->d_r, c->d_z, c->d_r, c->d_zparts, nullptr, nullptr, xs_ready);
    if (amg_comm_failed(c->amg)) return SGO_ECOMM;
    HIP_TRY(c, hipMemcpyAsync(c->d_p, c->d_z, sizeof(double) * 3 * (size_t)c->n, hipMemcpyDeviceToDevice, c->stream));
    if (warm) {
      Scope sc(c, K_INIT_SCALARS, 8.0 * gz);
      launch_restart_scalars(c->stream, c->d_S, c->d_zparts, gz, maxit, 1);
    } else {
      Scope sc(c, K_INIT_SCALARS, 8.0 * (gz + grid));
      launch_init_scalars(c->stream, c->d_S, c->d_zparts, gz, c->d_partials + kMaxPartials, grid, tol, maxit, c->bb_ref, c->tol_cap);
    }
    // progress probe of the lagged refresh (PcgScalars): a fresh solve records, a solve behind kept operators is held to the record
    set_probe(c, c->amg_lag_on ? c->amg_probe_k : 0, c->amg_skip_update ? c->amg_probe_max : 0.0);
  } else {
    Scope sc(c, K_INIT_SCALARS, 16.0 * grid);
    launch_init_scalars(c->stream, c->d_S, c->d_partials, grid, c->d_partials + kMaxPartials, grid, tol, maxit, c->bb_ref, c->tol_cap);
    set_probe(c, 0, 0.0);   // (a probe left armed by an earlier solve behind a hierarchy this graph has since lost would raise stop = 4 here)
  }
  return SGO_OK;
}

// The movement of the level-0 diagonal blocks up to which a solve keeps its coarse operators: what costs this graph's solves four
// PCG iterations by the slope learned from its kept solves (sgo_ctx.h), at most amg_lag_tau.
static double lag_allowed(const sgo_ctx* c) { return rules::lag_allowed(c->amg_lag_tau, c->amg_lag_slope); }

// A solve behind KEPT coarse operators was stopped (its progress probe, or its iteration cap): the operators are refreshed and the
// solve carries on from its current x and r -- z = M^-1 r with the new cycle, p = z, the recurrence restarts (what the kept
// iterations gained in the residual stays; the Krylov memory goes, about two iterations' worth).  Single GPU only.
static int refresh_and_continue(sgo_ctx* c, int maxit) {
  int rc = amg_update(c->amg, c->stream, &c->err);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->d_dref, c->S0.dblk, sizeof(double) * 6 * (size_t)c->n, hipMemcpyDeviceToDevice, c->stream));
  c->amg_ref_valid = true;
  c->amg_skip_update = false;
  const int gz = amg_apply(c->amg, c->stream, c->d_r, c->d_z, c->d_r, c->d_zparts, nullptr, nullptr, 0);
  HIP_TRY(c, hipMemcpyAsync(c->d_p, c->d_z, sizeof(double) * 3 * (size_t)c->n, hipMemcpyDeviceToDevice, c->stream));
  {
    Scope sc(c, K_INIT_SCALARS, 8.0 * gz);
    launch_restart_scalars(c->stream, c->d_S, c->d_zparts, gz, maxit, 0);
  }
  set_probe(c, 0, 0.0);
  return SGO_OK;
}

// buildSystem + preconditioner + PCG start state.  Multi-GPU: every rank linearises the whole graph (1.5 % of
// a GN iteration; sharding it would mean all-reducing the blocks, 72 B per edge, to save it).
int do_linearize(sgo_ctx* c) {
  const int g0 = c->owner ? c->halo.g0 : 0, g1 = c->owner ? c->halo.g1 : c->S0.ngrp;   // row-owner mode: this rank's rows
  const int row0 = c->owner ? c->halo.row0 : 0, row1 = c->owner ? c->halo.row1 : c->n;
  {
    Scope sc(c, K_LINEARIZE, bytes_linearize(c) * (g1 - g0) / std::max(1, c->S0.ngrp));
    launch_linearize(c->stream, c->S0, g0, g1, c->es, c->d_poses, c->d_dgb);
  }
  if (c->ov.active) {
    // incremental set-up (sgo_overlay.h): the appended edges' linearisation, the elimination of the appended chain; the
    // touched rows' right-hand sides in dgb receive g = b_T - H_TN H_NN^-1 b_N before k_finalize reads them
    launch_ov_lin(c->stream, c->ov.dev, c->d_poses);
    launch_ov_solve(c->stream, c->ov.dev, c->d_dgb);
  }
  int grid = 0;
  {
    Scope sc(c, K_FINALIZE, (72.0 + 48.0 + 48.0 + 6 * 24.0) * (row1 - row0));
    launch_finalize(c->stream, c->S0, row0, row1, c->d_dgb, c->d_b, c->d_x, c->d_r, c->d_z, c->d_p,
                    c->amg ? amg_xs0(c->amg) : nullptr, c->amg ? amg_omega(c->amg) : 0.0, c->d_partials, &grid);
  }
  if (c->amg && c->amg_lag_on && !c->owner && c->d_dref) {   // (an incremental overlay changes nothing here: the hierarchy is the resident rows')
    // Lagged refresh: a solve keeps the coarse operators of the previous one while the level-0 diagonal blocks have moved little
    // (lag_allowed: relative, summed over the rows) since those operators were made.
    c->amg_skip_update = false;
    c->amg_dchg_pending = false;
    // The host waits for the sums -- the one round trip of the scheme, with the launches of the refresh not yet queued behind it:
    // 40-50 us -- only when the movement of the PREVIOUS iteration says this one may keep (a converging call's steps shrink);
    // otherwise the solve refreshes, the sums are read behind the solve's own synchronisation, and the kernel that measures
    // stores the new reference on its way.
    const bool wait = c->amg_ref_valid && ((c->amg_lag_expect && c->amg_probe_max > 0.0) || c->amg_force_keep);
    c->dchg_grid = launch_diag_change(c->stream, 0, c->n, c->S0.dblk, c->d_dref, !wait, c->h_dchg_dev);
    c->amg_dchg_pending = c->amg_ref_valid;
    if (wait) {
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      read_diag_change(c);
      const double rel = c->last_dchg[1] > 0.0 ? c->last_dchg[0] / c->last_dchg[1] : 1.0;
      c->amg_skip_update = rel <= lag_allowed(c) && c->last_dchg[2] <= c->amg_lag_rows && c->amg_probe_max > 0.0;
      if (c->amg_force_keep) {   // calibration (SGO_AMG_LAG_FORCE=N): every solve after iteration N keeps, whatever moved
        c->amg_skip_update = true;
        c->amg_lag_cap = 0;
        c->amg_probe_max = 0.0;
      }
      if (!c->amg_skip_update)
        HIP_TRY(c, hipMemcpyAsync(c->d_dref, c->S0.dblk, sizeof(double) * 6 * (size_t)c->n, hipMemcpyDeviceToDevice, c->stream));
    }
    if (c->amg_skip_update && c->amg_lag_cap > 0) c->pcg_softcap = c->pcg_softcap > 0 ? std::min(c->pcg_softcap, c->amg_lag_cap) : c->amg_lag_cap;
    if (!c->amg_skip_update) c->amg_ref_valid = true;
  } else {
    c->amg_skip_update = false;
    c->amg_ref_valid = false;
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
  if (c->owner) {
    // row-owner mode: this rank's tiles; x must hold the neighbours' boundary rows (the callers exchange them); the
    // dot product rides on the kernel as partials over the owned rows
    a.u0 = c->halo.u0;
    a.u1 = c->halo.u1;
    Scope sc(c, K_SPMV0T_AX, bytes_spmv0(c->S0, S0_AX) / c->halo.G);
    if (dot) {
      a.dotA = x;
      a.partials = c->d_partials;
    }
    const int grid = launch_spmv0_any(c->stream, c->S0, c->T0, S0_AX, a);
    if (grid_out) *grid_out = grid;
    return SGO_OK;
  }
  if (multi_rank(c)) {
    // multi-GPU: this rank's range of tiles only, zeros elsewhere, all-reduce of the product vector (every row
    // has exactly one non-zero contributor: the sum is exact), then the dot product on the full vectors --
    // the same arithmetic on every rank, so the replicated PCG recurrences stay bit-identical across ranks
    a.u0 = c->shard_u0;
    a.u1 = c->shard_u1;
    // (with a communicator the ranks' slices are all-gathered -- one contributor per row: nothing to sum, no zero fill;
    // the rank-emulation hook has no communicator: zeros elsewhere, so that a test can add the ranks' partial results)
    if (!c->gather_slices) HIP_TRY(c, hipMemsetAsync(y, 0, sizeof(double) * 3 * (size_t)c->n, c->stream));
    if (a.u1 > a.u0) {
      Scope sc(c, c->T0.ntile > 0 ? K_SPMV0T_AX : K_SPMV0_AX, bytes_spmv0(c->S0, S0_AX) * (a.u1 - a.u0) / std::max(1, c->shard_units));
      launch_spmv0_any(c->stream, c->S0, c->T0, S0_AX, a);
    }
    if (c->gather_slices) {
      if (!halo_gather_slices(c->halo, c->stream, y, 3, &c->err)) return SGO_ECOMM;
    } else if (!c->comm.allreduce_f64(y, 3 * (size_t)c->n, c->stream, &c->err)) {
      return SGO_ECOMM;
    }
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
  if (c->ov.active) launch_ov_ax(c->stream, c->ov.dev, x, y, dot ? c->d_partials : nullptr, S);   // + U M U^T x (sgo_overlay.h)
  if (grid_out) *grid_out = grid;
  return SGO_OK;
}

// One PCG iteration in row-owner mode: two exchanges -- q's boundary with the partial sums of p.q | z's boundary with the
// partial sums of r.z, r.r, z.q -- and the all-reduce of the coarse right-hand side; all vector work on the owned rows.  The
// copies of the neighbours' boundary rows ("halo rows") of r, xs and p are kept current by repeating the recurrences on them
// (k_update_*_rows: same inputs, same arithmetic as the owner's: bit-identical), and the cycle prolongates the replicated
// coarse solution on them itself (k_prolong_rows, sgo_amg.hip): no exchange of xs, p or the corrected xs.
static int pcg_iteration_owner(sgo_ctx* c) {
  const HaloDev& H = c->halo;
  const int G = H.G, nr = H.row1 - H.row0;
  const size_t o3 = 3 * (size_t)H.row0;
  int g1 = 0, g2 = 0, rc;
  if ((rc = do_spmv(c, c->d_p, c->d_q, true, c->d_S, &g1))) return rc;
  double* parts2 = c->d_partials + kMaxPartials;  // [0] = r.z (block-Jacobi only), [1] = r.r
  if (c->amg) {
    if (!xch(c, c->d_q, scal(c->d_partials, g1))) return SGO_ECOMM;
    {
      Scope sc(c, K_UPDATE_XR, (7 * 24.0 + 48.0) * nr);
      launch_update_xr(c->stream, nr, c->d_S, H.gparts, G, c->S0.dinv + 6 * (size_t)H.row0, c->d_p + o3, c->d_q + o3, c->d_x + o3,
                       c->d_r + o3, c->d_z + o3, amg_xs0(c->amg) + o3, amg_omega(c->amg), parts2, &g2);
    }
    launch_update_xr_rows(c->stream, H.nhalo, H.halo_rows, c->d_S, c->S0.dinv, c->d_p, c->d_q, c->d_r, amg_xs0(c->amg), amg_omega(c->amg));
    const int gz = amg_apply(c->amg, c->stream, c->d_r, c->d_z, c->d_r, c->d_zparts, c->d_S, c->d_q, 2);
    if (amg_comm_failed(c->amg)) {
      c->err = "collective failed inside the multigrid cycle";
      return SGO_ECOMM;
    }
    if (!xch(c, c->d_z, scal(c->d_zparts, gz, parts2 + kMaxPartials, g2, c->d_zparts + kMaxPartials, gz))) return SGO_ECOMM;
    {
      Scope sc(c, K_UPDATE_P, 3 * 24.0 * nr);
      launch_update_p(c->stream, nr, c->d_S, H.gparts, G, H.gparts + G, G, H.gparts + 2 * G, c->d_z + o3, c->d_p + o3);
    }
    launch_update_p_rows(c->stream, H.nhalo, H.halo_rows, c->d_S, c->d_z, c->d_p);
    return SGO_OK;
  }
  // block-Jacobi: p.q | r.z, r.r | p
  if (!xch(c, nullptr, scal(c->d_partials, g1))) return SGO_ECOMM;
  {
    Scope sc(c, K_UPDATE_XR, (7 * 24.0 + 48.0) * nr);
    launch_update_xr(c->stream, nr, c->d_S, H.gparts, G, c->S0.dinv + 6 * (size_t)H.row0, c->d_p + o3, c->d_q + o3, c->d_x + o3,
                     c->d_r + o3, c->d_z + o3, nullptr, 0.0, parts2, &g2);
  }
  if (!xch(c, nullptr, scal(parts2, g2, parts2 + kMaxPartials, g2))) return SGO_ECOMM;
  {
    Scope sc(c, K_UPDATE_P, 3 * 24.0 * nr);
    launch_update_p(c->stream, nr, c->d_S, H.gparts, G, H.gparts + G, G, nullptr, c->d_z + o3, c->d_p + o3);
  }
  if (!xch(c, c->d_p, HaloScalars())) return SGO_ECOMM;
  return SGO_OK;
}

int pcg_iteration(sgo_ctx* c) {
  if (c->owner) return pcg_iteration_owner(c);
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
                    c->d_z, c->d_p, rec_dev(c));
  } else {
    Scope sc(c, K_UPDATE_P, 3 * 24.0 * c->n);
    RecDev rec;
    rec.mirror = c->d_Sz;
    launch_update_p(c->stream, c->n, c->d_S, parts2, g2, parts2 + kMaxPartials, g2, nullptr, c->d_z, c->d_p, rec);
  }
  return SGO_OK;
}

int ensure_pcg_graph(sgo_ctx* c, int chunk, bool* captured) {
  const int key = c->ov.active ? 1 : 0;
  if (captured) *captured = false;
  if (c->pcg_exec && c->pcg_exec_chunk == chunk && c->pcg_exec_key == key) return SGO_OK;
  if (captured) *captured = true;   // (attempted: every rank of a multi-GPU run gets here at the same solve)
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
  c->pcg_exec_key = key;
  return SGO_OK;
}

// Runs PCG from the state k_finalize left (x = 0, r = b, ...) until S.stop != 0.
// Runs PCG from the state k_finalize / start_pcg left (x = 0, r = b, ...) until S.stop != 0.
// Graph mode: a 2-iteration hipGraph is replayed; the first 0.8 * predicted - 4 iterations -- predicted
// = the count of the previous solve -- go out without any host check, after that one replay is always
// in flight while the host waits for the stop flag copied out after the previous one (kernels of
// iterations past convergence exit on the flag), so the GPU never idles on a host round trip and at
// most two replays of early-exit launches are wasted.
void read_call_knobs(sgo_ctx* c) {
  sgo_ctx::CallKnobs k;
  if (const char* e = std::getenv("SGO_COMM_GRAPH")) k.comm_graph = std::atoi(e) != 0 ? 1 : 0;
  if (const char* e = std::getenv("SGO_PCG_STALL_WINDOW")) k.stall_window = std::max(0, std::atoi(e));
  if (const char* e = std::getenv("SGO_FIRST_SOLVE_CAP")) k.first_solve_cap = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("SGO_TEST_FAIL_TRIAL_BUILD")) k.fail_trial_build = std::atoi(e) != 0;
  if (const char* e = std::getenv("SGO_TEST_FAIL_DEVICE_SETUP")) k.fail_device_setup = std::atoi(e) != 0;
  if (const char* e = std::getenv("SGO_AMG_KEEP_AGG")) k.keep_agg = std::atoi(e) != 0;
  if (const char* e = std::getenv("SGO_AMG_SETUP")) k.setup_mode = std::string(e) == "host" ? 0 : (std::string(e) == "rebuilds" ? 1 : 2);
  if (const char* e = std::getenv("SGO_AMG_LISTS"))   // (test hook: product lists made on the host -- they belong to the host set-up)
    if (std::string(e) == "host") k.setup_mode = 0;
  if (const char* e = std::getenv("SGO_AMG_FORCE_REBUILD")) k.force_rebuild = std::atoi(e) != 0;
  if (const char* e = std::getenv("SGO_AMG_AGG")) k.dev_aggregation = std::string(e) == "device";
  if (const char* e = std::getenv("SGO_AMG_REBUILD_COST")) k.rebuild_cost = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("SGO_AMG_LAG")) k.lag_on = std::atoi(e) != 0;
  if (const char* e = std::getenv("SGO_AMG_LAG_TAU")) k.lag_tau = std::atof(e);
  if (const char* e = std::getenv("SGO_AMG_LAG_SLOPE")) k.lag_slope = std::atof(e);
  if (const char* e = std::getenv("SGO_AMG_LAG_FORCE")) k.lag_force_from = std::atoi(e);
  c->knobs = k;
  c->test_fail_trial_build = k.fail_trial_build;
}

int run_pcg(sgo_ctx* c) {
  // Multi-GPU with an RCCL communicator: the collectives are captured into the hipGraph with the kernels around them (every
  // rank replays the same graph the same number of times: the replay count follows snapshots of the device-resident stop flag
  // taken at fixed points of the stream, bit-identical on all ranks; the exchanges of the set-up and of the solve's start have
  // run eagerly before, so the communicator's channels exist when the capture begins).  Measured with a 1-rank communicator
  // on C4: 191 -> 203 M edge-Jacobians/s.  Not possible with the host transport (a host callback inside the loop).
  // DEFAULT: on for a 1-rank communicator (the measured case: no peer to wait for), OFF for nranks > 1 -- RCCL with more than
  // one rank has never executed on this code (no multi-GPU node: DESIGN.md section 6), and a hang there would cost the whole
  // run; SGO_COMM_GRAPH=1 opts in, =0 opts out.  Whether the capture worked is decided COLLECTIVELY (an eager all-reduce of
  // the ranks' failure flags right after the attempt): ranks that replay graphs and ranks that launch plainly would issue
  // different numbers of collectives per solve and hang each other, so either all ranks replay or all fall back.
  bool comm_graph = c->comm.handle != nullptr && !c->comm.host_fn && !c->comm_graph_failed && c->comm.nranks <= 1;
  if (c->knobs.comm_graph >= 0)
    comm_graph = c->comm.handle != nullptr && !c->comm.host_fn && !c->comm_graph_failed && c->knobs.comm_graph != 0;
  // Stagnation guard (single GPU: the decision reads a clock-free but rank-local snapshot): a solve whose r.r has not reached a new
  // minimum for max(3000, 30 x the previous solve's count) iterations is ended as one that ran out of iterations -- the systems
  // PCG cannot finish in double precision otherwise grind on to pcg_maxit, three times per call.
  c->pcg_stalled = false;
  c->pcg_stall_window = multi_rank(c) ? 0 : std::max(3000, 30 * std::max(1, c->pcg_pred));
  // (once a solve of this call has been accepted at the floating-point floor of its system -- optimize_gn, solve_backward_error --
  // the call is at the edge of double precision: its later solves wait a tenth as long before they are looked at)
  if (c->floor_seen && c->pcg_stall_window > 0) c->pcg_stall_window = std::max(300, 3 * std::max(1, c->pcg_pred));
  if (c->knobs.stall_window >= 0) c->pcg_stall_window = multi_rank(c) ? 0 : c->knobs.stall_window;   // (test hook; 0: no guard)
  double stall_rr = -1.0;
  int stall_it = 0;
  auto stalled = [&](const volatile PcgScalars* S) -> bool {
    if (c->pcg_stall_window <= 0 || c->pcg_stalled) return false;
    const double rr = S->rr;
    const int it = S->iter;
    if (stall_rr < 0.0 || rr < stall_rr) {
      stall_rr = rr;
      stall_it = it;
      return false;
    }
    return it - stall_it > c->pcg_stall_window;
  };
  bool graph = c->opts.use_graph && !c->opts.profile && (!multi_rank(c) || comm_graph);
  constexpr int kUnit = 2;   // iterations per graph replay (1: 44.6, 2: 41.2, 4: 42, 8: 47 us per PCG iteration on C2 -- a replay costs
                             // ~7 us, an iteration past convergence eight early-exit nodes)
  if (graph) {
    bool attempted = false;
    const int grc = ensure_pcg_graph(c, kUnit, &attempted);
    if (multi_rank(c)) {
      int failed = grc != SGO_OK ? 1 : 0;
      if (attempted && c->comm.nranks > 1) {   // the ranks agree on the outcome before anyone replays
        if (!c->d_comm_flag) HIP_TRY(c, hipMalloc((void**)&c->d_comm_flag, sizeof(int)));
        HIP_TRY(c, hipMemcpyAsync(c->d_comm_flag, &failed, sizeof(int), hipMemcpyHostToDevice, c->stream));
        std::string cerr;
        if (!c->comm.allreduce_i32(c->d_comm_flag, 1, c->stream, &cerr)) {
          c->err = "all-reduce of the capture flags: " + cerr;
          return SGO_ECOMM;
        }
        HIP_TRY(c, hipMemcpyAsync(&failed, c->d_comm_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
      }
      if (failed) {
        if (c->pcg_exec) {   // (this rank's capture worked, another rank's did not)
          hipGraphExecDestroy(c->pcg_exec);
          c->pcg_exec = nullptr;
        }
        c->comm_graph_failed = true;
        graph = false;
        if (c->opts.verbose)
          std::fprintf(stderr, "[sgo] capturing the collectives failed on %d rank(s)%s%s: plain stream launches on all ranks from here on\n", failed,
                       grc != SGO_OK ? "; here: " : "", grc != SGO_OK ? c->err.c_str() : "");
      }
    } else if (grc != SGO_OK) {
      return grc;
    }
  }
  if (!graph) {
    // (profile mode: the stop flag is read after EVERY iteration, so that no early-exit launch past convergence is among
    // the timed launches -- their 1.5-us dispatches would pull the per-kernel figures down)
    const int chunk = c->opts.profile ? 1 : std::max(1, c->opts.pcg_chunk);
    for (;;) {
      HIP_TRY(c, hipMemcpyAsync(c->h_S, c->d_S, sizeof(PcgScalars), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      if (c->h_S->stop) break;
      if (stalled(c->h_S)) {
        c->pcg_stalled = true;
        launch_force_stop(c->stream, c->d_S, nullptr);
        continue;
      }
      for (int k = 0; k < chunk; ++k) {
        int rc = pcg_iteration(c);
        if (rc) return rc;
      }
    }
    c->pcg_pred = c->h_S->iter;
    return SGO_OK;
  }
  // One replay (2 iterations, >= 100 us even on 1k-pose graphs) in flight hides the host's read of the
  // stop flag; more only adds early-exit launches past convergence (measured: 8 iterations in flight
  // cost 3.5 % on C4 and 10 % on C1).  pcg_chunk = 16 -> 1 replay; larger values scale it up.
  const int chunk_launches = std::max(1, c->opts.pcg_chunk / 16);
  // unchecked prefix: 80 % of the previous count minus a margin (a solve that converges earlier
  // than that only wastes ~1 us per early-exit launch; tighter margins measured no different)
  const int pred = c->pcg_pred;
  const int unchecked = std::max(0, (int)(0.8 * pred) - 4) / kUnit;
  for (int k = 0; k < unchecked; ++k) HIP_TRY(c, hipGraphLaunch(c->pcg_exec, c->stream));
  // The stop flag is read from the pinned mirror every k_update_p rewrites (RecDev::mirror) behind an event: no copy kernel
  // between replays (4.3 us each on the stream, measured).  The mirror may already show the state of the replay in flight --
  // a later iteration of the same solve, equally valid; once a stop flag is set nothing moves any more.
  static const bool mirror_env = !(std::getenv("SGO_MIRROR") && std::atoi(std::getenv("SGO_MIRROR")) == 0);
  const bool mirror = mirror_env && !multi_rank(c) && !c->owner;   // (the multi-rank iterations do not pass the mirror: they keep the copy)
  int slot = 0;
  auto snapshot = [&](int sl) -> hipError_t {
    if (!mirror) {
      hipError_t e = hipMemcpyAsync(&c->h_S2[sl], c->d_S, sizeof(PcgScalars), hipMemcpyDeviceToHost, c->stream);
      if (e != hipSuccess) return e;
    }
    return hipEventRecord(c->ev_S[sl], c->stream);
  };
  if (mirror) {   // the start kernels' state (a solve may be over before its first iteration): one copy per solve
    HIP_TRY(c, hipMemcpyAsync(c->h_Sz, c->d_S, sizeof(PcgScalars), hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(c, snapshot(slot));
  for (;;) {
    for (int k = 0; k < chunk_launches; ++k) HIP_TRY(c, hipGraphLaunch(c->pcg_exec, c->stream));  // speculative
    HIP_TRY(c, snapshot(slot ^ 1));
    HIP_TRY(c, hipEventSynchronize(c->ev_S[slot]));
    if (mirror ? ((volatile PcgScalars*)c->h_Sz)->stop : c->h_S2[slot].stop) break;
    if (stalled(mirror ? (volatile PcgScalars*)c->h_Sz : (volatile PcgScalars*)&c->h_S2[slot])) {
      c->pcg_stalled = true;
      launch_force_stop(c->stream, c->d_S, mirror ? c->d_Sz : nullptr);
    }
    slot ^= 1;
  }
  if (mirror) {
    HIP_TRY(c, hipEventSynchronize(c->ev_S[slot ^ 1]));   // (the replay in flight rewrites the mirror: let it finish before the final read)
    *c->h_S = *c->h_Sz;
    c->pcg_pred = c->h_S->iter;
    return SGO_OK;
  }
  *c->h_S = c->h_S2[slot];
  c->pcg_pred = c->h_S->iter;
  return SGO_OK;
}

// (Re)build the multigrid hierarchy from the CURRENT level-0 values (requires do_linearize).
std::string multi_gpu_description(const sgo_ctx* c) {
  if (c->owner)
    return "; multi-GPU row-owner mode: rank " + std::to_string(c->halo.me) + " of " + std::to_string(c->halo.G) + " owns rows [" +
           std::to_string(c->halo.row0) + ", " + std::to_string(c->halo.row1) + "), largest boundary " + std::to_string(c->halo.bmax) + " rows";
  if (multi_rank(c)) return "; multi-GPU all-reduce mode (" + std::to_string(c->comm.nranks) + " ranks)";
  if (c->replicated) return "; multi-GPU: replicated on all " + std::to_string(c->comm.nranks) + " ranks (fewer level-0 work units than ranks)";
  return "";
}

int build_amg(sgo_ctx* c, bool keep_old, bool keep_agg) {
  // speculative replays of the captured PCG iteration (and the launches queued behind them) may still be
  // in flight: drain the stream before the exec and the old hierarchy's buffers go away
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->amg_ref_valid = false;   // (the lagged refresh's reference blocks belong to the old hierarchy's coarse operators)
  if (c->pcg_exec) {  // the captured PCG iteration references the old hierarchy's buffers
    hipGraphExecDestroy(c->pcg_exec);
    c->pcg_exec = nullptr;
  }
  c->pcg_pred = 0;  // the iteration count of the old hierarchy predicts nothing about the new one
  c->amg_best = 0;
  c->agg_best = 0;
  if (c->amg_prev) {   // (whatever an earlier trial kept is gone now)
    amg_destroy(c->amg_prev);
    c->amg_prev = nullptr;
  }
  if (keep_old && c->amg) {   // trial rebuild: the old hierarchy stays intact in its arena, the new one goes into the other
    c->amg_prev = c->amg;
    c->amg = nullptr;
    c->amg_prev_desc = c->solver_desc;
    std::swap(c->amg_arena.chunks, c->amg_arena_prev.chunks);
    std::swap(c->amg_arena.next_chunk, c->amg_arena_prev.next_chunk);
  }
  const bool reuse_agg = keep_agg && !keep_old && c->amg && !c->owner;
  if (reuse_agg) amg_kept_aggregates(c->amg, &c->kept_agg);
  if (c->amg) {
    amg_destroy(c->amg);
    c->amg = nullptr;
  }
  c->amg_arena.rewind();
  if (keep_old && c->amg_prev && c->test_fail_trial_build) {   // test hook (SGO_TEST_FAIL_TRIAL_BUILD, read once per sgo_optimize_gn): the trial's set-up "fails"
    c->solver_desc = "pcg_block_jacobi (AMG unavailable: test hook)";
    return SGO_OK;
  }
  AmgConfig cfg;
  cfg.theta_scale = c->amg_theta_scale;
  cfg.filtered_smoothing = !c->amg_no_filter;
  cfg.keep_agg = reuse_agg ? &c->kept_agg : nullptr;
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
  AmgHalo ah;
  std::vector<double> w0;
  if (c->owner) {
    // row-owner mode: a rank holds current blocks for its own rows only; the strength weights of ALL level-0 slots come
    // from the (replicated) edge list at the current poses, as in the set-up pipeline
    if (!(c->l0_pre && amg_host_l0_ready(c->l0_pre))) {
      double* d_w = (double*)c->amg_arena.take(sizeof(double) * (size_t)std::max(c->H0.nslot, 1));
      if (!d_w) {
        c->err = "out of device memory";
        return SGO_ENOMEM;
      }
      launch_early_strength(c->stream, c->el, c->d_poses, c->n, c->d_rowptr, c->d_eidx, c->es.flags, c->d_hrowptr, d_w);
      w0.resize((size_t)c->H0.nslot);
      HIP_TRY(c, hipMemcpyAsync(w0.data(), d_w, sizeof(double) * w0.size(), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    ah.dev = &c->halo;
    ah.G = c->halo_host.G;
    ah.bmax = c->halo_host.bmax;
    ah.bnd_host = c->halo_host.bnd.data();
    ah.user = c;
    ah.reserve = [](void* u, size_t doubles) { return halo_reserve((sgo_ctx*)u, doubles) == SGO_OK; };
    ah.w0 = &w0;
  }
  // The set-up ON THE DEVICE (sgo_amg_dev.inc): the rebuilds inside sgo_optimize_gn on one GPU (SGO_AMG_SETUP=host: the host
  // set-up for them too; =device: every set-up, sgo_set_graph_se2's included, whose level 0 the helper thread has not made ahead).
  const double t_create0 = wall_s();
  const bool pre0_ready = c->l0_pre && amg_host_l0_ready(c->l0_pre);
  const bool pre0_agg = pre0_ready && amg_host_l0_agg_only(c->l0_pre);
  const bool dev_setup = !c->owner && !multi_rank(c) && (pre0_agg || (!pre0_ready && (c->knobs.setup_mode == 2 || (c->knobs.setup_mode == 1 && c->in_optimize))));
  if (dev_setup)
    c->amg = amg_create_dev(c->stream, c->A, c->S0, c->T0, c->H0, c->d_poses, c->d_free_id, cfg, prof, &aerr, &c->amg_scratch, &c->amg_arena,
                            &c->amg_tmp_arena, c->knobs.dev_aggregation, pre0_agg ? c->l0_pre : nullptr);
  if (dev_setup && c->amg && c->knobs.fail_device_setup) {   // test hook (SGO_TEST_FAIL_DEVICE_SETUP): the device set-up "fails"
    amg_destroy(c->amg);
    c->amg = nullptr;
    aerr = "test hook";
  }
  if (dev_setup && !c->amg) {
    // (the device set-up could not be made -- out of device memory for its sort buffers, say --: the host set-up, which needs
    // none, before the graph is left to block-Jacobi; a helper thread's aggregation-only result is of no use to it)
    if (c->opts.verbose) std::fprintf(stderr, "[sgo] device set-up failed (%s): host set-up\n", aerr.c_str());
    c->amg_arena.rewind();
    aerr.clear();
  }
  if (!c->amg)
    c->amg = amg_create(c->stream, c->A, c->S0, c->T0, c->H0, c->d_poses, c->d_free_id, cfg, prof, &aerr, &c->amg_scratch,
                        &c->amg_arena, pre0_agg ? nullptr : c->l0_pre, c->owner ? &ah : nullptr);
  l0_discard(c);
  if (c->opts.verbose)
    std::fprintf(stderr, "[sgo] multigrid set-up (%s): %.2f ms\n", dev_setup ? (c->knobs.dev_aggregation ? "aggregation and patterns on the device" : "host aggregation, patterns on the device") : "host", 1e3 * (wall_s() - t_create0));
  if (c->amg && amg_comm_failed(c->amg)) {
    c->err = "collective failed during the multigrid set-up";
    return SGO_ECOMM;
  }
  if (c->amg) {
    if (!c->owner && multi_rank(c))
      amg_set_shard(c->amg, &c->comm, c->shard_u0, c->shard_u1, c->shard_row0, c->shard_row1, c->gather_slices ? &c->halo : nullptr);
    // (a graph on the single-launch direct or the multifrontal path gets here through a single-step entry point -- sgo_linearize,
    // sgo_solve --: sgo_optimize_gn keeps running the factorisation, and the description keeps saying so)
    std::string amg_desc;
    amg_describe(c->amg, &amg_desc);
    if (c->direct || c->mf) {
      const size_t cut = c->solver_desc.find("; single-step entry points: ");
      if (cut != std::string::npos) c->solver_desc.resize(cut);
      c->solver_desc += "; single-step entry points: pcg_amg: " + amg_desc;
    } else {
      c->solver_desc = "pcg_amg: " + amg_desc;
      c->solver_desc += multi_gpu_description(c);
    }
  } else {
    if (c->direct || c->mf) c->solver_desc += "; single-step entry points: pcg_block_jacobi (AMG unavailable: " + aerr + ")";
    else c->solver_desc = "pcg_block_jacobi (AMG unavailable: " + aerr + ")";
    if (c->opts.verbose) std::fprintf(stderr, "[sgo] %s\n", c->solver_desc.c_str());
  }
  if (c->d_dref_agg && c->S0.dblk && !c->owner) {   // the blocks this aggregation was made from (optimize_gn's movement rule)
    HIP_TRY(c, hipMemcpyAsync(c->d_dref_agg, c->S0.dblk, sizeof(double) * 6 * (size_t)c->n, hipMemcpyDeviceToDevice, c->stream));
    c->agg_ref_valid = true;
  }
  return SGO_OK;
}

// The hierarchy a trial rebuild made is dropped for the one it replaced (build_amg(c, true) kept it): the old one's values are
// refreshed by the next solve's amg_update like any other iteration's.
int revert_amg(sgo_ctx* c) {
  if (!c->amg_prev) return SGO_OK;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->pcg_exec) {   // (the captured PCG iteration references the dropped hierarchy's buffers)
    hipGraphExecDestroy(c->pcg_exec);
    c->pcg_exec = nullptr;
  }
  if (c->amg) amg_destroy(c->amg);
  c->amg = c->amg_prev;
  c->amg_prev = nullptr;
  std::swap(c->amg_arena.chunks, c->amg_arena_prev.chunks);
  std::swap(c->amg_arena.next_chunk, c->amg_arena_prev.next_chunk);
  c->amg_arena_prev.rewind();
  c->solver_desc = c->amg_prev_desc;
  c->pcg_pred = 0;
  c->amg_best = 0;
  c->agg_best = 0;
  c->amg_ref_valid = false;   // its coarse operators are two iterations old: the next solve refreshes
  c->amg_probe_max = 0.0;
  c->agg_ref_valid = false;   // (and the blocks it was aggregated from are not known any more: the rule is off for this graph anyway)
  return SGO_OK;
}

// Vectors over the free vertices cross the API in g2o's hessian order and live on the device in the
// internal (Hilbert) row order: permute on the way (test / single-step entry points only).
// build_structure + what follows from it (multi-GPU tile range, tolerance rule)
int build_rows(sgo_ctx* c, const double* poses, const uint8_t* fixed, const int32_t* ei, const int32_t* ej) {
  int rc = build_structure(c, c->V, poses, fixed, c->E, ei, ej);
  if (rc != SGO_OK) return rc;
  c->shard_units = c->T0.ntile > 0 ? c->T0.ntile : c->S0.ngrp;
  // A graph with fewer work units than ranks (a few hundred poses on 8 GPUs) would leave some rank an EMPTY range, which the
  // level-0 kernels and the transfers read as "all" (u1 == 0 / row1 == 0): such a graph is not sharded at all -- every rank
  // runs the single-GPU computation on its own copy (same arithmetic: bit-identical ranks, no collective).
  c->replicated = (c->comm.nranks > 1 || c->comm.active()) && !c->owner && !c->gather_slices && c->shard_units < c->comm.nranks;
  if (c->replicated) {
    c->shard_u0 = c->shard_u1 = 0;
    c->shard_row0 = 0;
    c->shard_row1 = c->n;
  } else {
    sgo_shard_range(c->shard_units, c->comm.nranks, c->comm.rank, &c->shard_u0, &c->shard_u1);
    c->shard_row0 = c->unit_row0.empty() ? 0 : c->unit_row0[c->shard_u0];
    c->shard_row1 = c->unit_row0.empty() ? c->n : c->unit_row0[c->shard_u1];
  }
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
  std::vector<double> poses(3 * (size_t)c->V);
  HIP_TRY(c, hipMemcpyAsync(poses.data(), c->d_poses, sizeof(double) * poses.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const int rc = build_rows(c, poses.data(), c->lz_fixed.data(), c->lz_ei.data(), c->lz_ej.data());
  if (rc != SGO_OK) return rc;   // (still pending: a later call tries again instead of running on half-built structures)
  c->rows_pending = false;
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
  if ((rc = do_linearize(c)) != SGO_OK || (rc = build_amg(c)) != SGO_OK) return rc;
  c->amg_pending = false;
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

// The normwise backward error of the solve's current x, eta = |r| / (|H| |x| + |b|) with |H| taken as 2 max_i |D_i|_F (an
// UNDER-estimate for rows of high degree: eta errs on the large side).  A solve that stops making progress with eta at a few
// thousand units of roundoff has the solution double precision can give for this system -- a backward-stable direct solver
// (LinearSolverEigen's LDL^T, graphs.cpp:19) returns one of the same quality and g2o applies it --, whatever |r| / |b| says: seen
// from BASELINE.md's dead-reckoned start on large graphs with full information matrices, where undamped Gauss-Newton + DCS blows up
// (steps of 10^9 m, |x| / |b| ~ 20, |H| ~ 4 10^9: relative residuals of 10^-5 .. 10^-6 at eta = 10^-16 .. 3 10^-14; restarting the
// recurrence from b - H x changes nothing there: NOTES.md section 29).  Rare path: x and the diagonal blocks are read on the host.
static int solve_backward_error(sgo_ctx* c, double* eta) {
  std::vector<double> hx(3 * (size_t)c->n), hd(6 * (size_t)c->n);
  HIP_TRY(c, hipMemcpyAsync(hx.data(), c->d_x, sizeof(double) * hx.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(hd.data(), c->S0.dblk, sizeof(double) * hd.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  double xx = 0.0, dmax = 0.0;
  for (double v : hx) xx += v * v;
  for (int i = 0; i < c->n; ++i) {
    const double* d = hd.data() + 6 * (size_t)i;
    dmax = std::max(dmax, std::sqrt(d[0] * d[0] + 2 * d[1] * d[1] + 2 * d[2] * d[2] + d[3] * d[3] + 2 * d[4] * d[4] + d[5] * d[5]));
  }
  const double den = 2.0 * dmax * std::sqrt(xx) + std::sqrt(c->h_S->bb);
  *eta = (den > 0.0 && std::isfinite(den) && std::isfinite(c->h_S->rr)) ? std::sqrt(c->h_S->rr) / den : 1.0;
  if (c->opts.verbose)
    std::fprintf(stderr, "[sgo] solve stopped without reaching pcg_tol: |r| %.3e |b| %.3e |x| %.3e max|D| %.3e: normwise backward error %.2e\n", std::sqrt(c->h_S->rr),
                 std::sqrt(c->h_S->bb), std::sqrt(xx), dmax, *eta);
  return SGO_OK;
}
constexpr double kFloorEta = 1e-12;   // ~ 4 500 units of roundoff

int optimize_gn(sgo_ctx* c, int32_t iters, sgo_stats* out) {
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
    read_call_knobs(c);
    if (c->direct || c->mf) {
      // ---- small-graph path: the whole call is one launch (sgo_direct.h); mid-size path: one launch per level of the
      // elimination tree and Gauss-Newton iteration, no host round trip inside the call (sgo_mfront.h)
      if (c->direct) {
        Scope sc(c, K_DIRECT, direct_bytes(c->direct, c->E, iters));
        hipError_t he = direct_optimize(c->direct, c->stream, c->el, c->d_poses, iters, c->d_hist, c->d_dres);
        if (he != hipSuccess) {
          c->err = std::string("k_direct launch: ") + hipGetErrorString(he);
          return SGO_EHIP;
        }
      } else {
        Scope sc(c, K_MFRONT, mfront_bytes(c->mf, c->E, iters), true);
        hipError_t he = mfront_optimize(c->mf, c->stream, c->el, c->d_poses, iters, c->d_hist, c->d_dres);
        if (he != hipSuccess) {
          c->err = std::string("multifrontal launch: ") + hipGetErrorString(he);
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
        c->err = std::string(c->direct ? "direct" : "multifrontal") + " factorisation failed in GN iteration " + std::to_string(done) +
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
      if (c->opts.verbose && done > 0 && c->direct)
        std::fprintf(stderr, "[sgo] direct: %.0f MHz shader clock during the call\n",
                     (double)R.cycles / (1e-2 * (double)(R.stamp[R.fail ? 2 * done + 2 : 2 * iters + 1] - R.stamp[0])));
      if (c->opts.verbose && done > 0 && c->direct)
        std::fprintf(stderr, "[sgo] direct, last iteration [us]: edges %.1f, assembly %.1f, sparse forward %.1f, separators %.1f + %.1f, "
                     "sparse backward %.1f, update %.1f\n", 1e-2 * (double)(R.phase[1] - R.phase[0]), 1e-2 * (double)(R.phase[2] - R.phase[1]),
                     1e-2 * (double)(R.phase[3] - R.phase[2]), 1e-2 * (double)(R.phase[4] - R.phase[3]), 1e-2 * (double)(R.phase[5] - R.phase[4]),
                     1e-2 * (double)(R.phase[6] - R.phase[5]), 1e-2 * (double)(R.phase[7] - R.phase[6]));
      if (c->opts.verbose > 1)
        for (int k = 0; k <= done; ++k)
          std::fprintf(stderr, "[sgo] iteration= %d\t chi2= %.9e\t robust= %.9e\t (%s)\n", k, c->h_hist[2 * k], c->h_hist[2 * k + 1], c->direct ? "direct" : "multifrontal");
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
        c->amg_lag_on = false;
        c->amg_skip_update = false;
        c->in_optimize = false;
        c->floor_seen = false;
      }
    } softcap_guard{c};
    // Lagged refresh of the coarse operators (do_linearize): on by default on one GPU (the sharded modes refresh before every solve:
    // a decision the ranks must take alike, not exercised on hardware); SGO_AMG_LAG=0 refreshes before every solve.
    // The call's first solve always refreshes (the incremental set-up's staleness rule compares first solves, sgo_ctx.h).
    c->amg_lag_tau = c->knobs.lag_tau;
    c->amg_lag_rows = 32.0;
    if (c->amg_lag_n == 0 || std::abs(c->n - c->amg_lag_n) > c->amg_lag_n / 10) {   // another graph: its sensitivity is not known yet
      c->amg_lag_slope = rules::kLagSlopeStart;
      c->amg_lag_slope_seen = false;
    }
    c->amg_lag_n = c->n;
    if (c->knobs.lag_slope > 0.0) {   // test hook (SGO_AMG_LAG_SLOPE): the sensitivity every call starts from (1: keep whatever SGO_AMG_LAG_TAU allows)
      c->amg_lag_slope = c->knobs.lag_slope;
      c->amg_lag_slope_seen = false;
    }
    c->amg_lag_on = c->knobs.lag_on && !multi_rank(c);   // (replicated ranks: the whole single-GPU computation each)
    c->amg_ref_valid = false;
    c->amg_probe_max = 0.0;
    c->amg_lag_expect = false;
    if (c->amg_probe_k < 4) c->amg_probe_k = 6;
    const int force_from = c->knobs.lag_force_from;   // (calibration hook, scripts/lag_calib.py)
    int fresh_pcg = 0;   // the count of the last solve behind freshly made coarse operators
    int kept_solves = 0;
    int floor_solves = 0;   // solves accepted at the floating-point floor of their system (solve_backward_error)
    int call_best = 0;   // the fewest (equal-tolerance) iterations a fresh solve of this call has taken
    std::string agg_note;
    int trial = 0, trial_old = 0, trial_best = 0, trial_seen = 0;   // the re-aggregation rule's trial: 1 = rebuild pending, 2 = judging the new hierarchy's first solves
    c->lag_note.clear();
    c->warm_valid = false;
    const bool warm_env = c->opts.pcg_warm_start != 0;
    c->bb_ref = 0.0;
    c->tol_cap = c->opts.pcg_tol_cap > 0.0 ? std::max(c->opts.pcg_tol_cap, c->opts.pcg_tol * c->tol_scale) : 0.0;
    int done = 0;
    bool failed = false;
    int rebuilds = 0;
    bool rebuild_next = c->knobs.force_rebuild && c->amg != nullptr;   // (test hook: the set-up is redone before the call's first solve)
    c->in_optimize = true;
    // (a cap on the set-ups redone inside one call, against thrashing: three for a short call, one per three Gauss-Newton
    // iterations for a long one.  Round 5: with three flat, a call whose weights keep changing -- DCS from a dead-reckoned
    // start -- used them up by iteration 6 and then had no safety net left: its 10th solve ground on to pcg_maxit.)
    const int max_rebuilds = rules::max_rebuilds(iters);
    double its_sum = 0.0;
    for (int it = 0; it < iters; ++it) {
      hipEventRecord(ev[3 * it], c->stream);
      c->pcg_softcap = (c->amg && c->amg_best > 0 && rebuilds < max_rebuilds && !rebuild_next) ? rules::bail_out_cap(c->amg_best) : 0;
      // A hierarchy that has never solved anything gets 600 iterations (the hardest first solves seen take 200-350: C4 from a
      // dead-reckoned start): one that needs more has a coarse space that does not carry the slow modes, and the set-up is
      // redone with HALF the strength thresholds (larger aggregates, sparser coarse operators) instead of grinding on to
      // pcg_maxit.  (SGO_FIRST_SOLVE_CAP: test hook.)
      const int first_solve_cap = c->knobs.first_solve_cap;
      if (c->amg && c->amg_best == 0 && rebuilds < max_rebuilds && c->amg_theta_scale > 0.2) c->pcg_softcap = first_solve_cap;
      c->amg_lag_cap = rules::lag_cap(fresh_pcg);
      c->amg_force_keep = force_from >= 0 && it > force_from;
      if (trial == 2) c->amg_ref_valid = false;   // (the trial's solves are fresh ones)
      if (rebuild_next) c->amg_ref_valid = false;
      if ((rc = do_chi2(c, c->d_hist + 2 * it, nullptr)) || (rc = do_linearize(c))) return rc;
      c->agg_grid = 0;
      if (it == 0 && c->amg && c->amg_lag_on && c->agg_ref_valid && !c->owner && iters > 1)   // (read behind the first solve's synchronisation)
        c->agg_grid = launch_diag_change(c->stream, 0, c->n, c->S0.dblk, c->d_dref_agg, false, c->h_dchg_dev + 3 * (size_t)kMaxPartials);
      if (rebuild_next && c->amg) {
        // The aggregation was made from the Hessian of an earlier linearisation and robust-kernel
        // re-weighting has changed the strength of connection since (see the rule below): redo the
        // set-up from the current values (same cost as in sgo_set_graph_se2).
        if (rebuilds + 1 >= max_rebuilds) c->amg_no_filter = true;   // (the call's last rebuild: see the abandoned solve below)
        const bool was_trial = trial == 1;
        rc = build_amg(c, was_trial, c->knobs.keep_agg);
        if (was_trial && c->amg_prev && (rc != SGO_OK || !c->amg) && rc != SGO_ECOMM) {
          // A TRIAL whose set-up did not come about (out of device memory -- a trial holds two hierarchies --, or the graph at its
          // current values cannot be coarsened): the hierarchy that was parked for the comparison is intact and comes back; the
          // re-aggregation rule is off for this graph, and the call carries on as if it had never tried.  (Without this the solve
          // ran behind block-Jacobi under the old hierarchy's bail-out cap and failed, with the working hierarchy unused in
          // amg_prev for every later call: ADVICE round 5.)
          const std::string why = c->err;
          if ((rc = revert_amg(c))) return rc;
          trial = 0;
          c->agg_rule_off = true;
          rebuild_next = false;
          c->err.clear();
          agg_note = "a re-aggregation was attempted in the last sgo_optimize_gn and its set-up failed" + (why.empty() ? std::string() : " (" + why + ")") +
                     ": the previous hierarchy stays";
          if (c->opts.verbose) std::fprintf(stderr, "[sgo] iteration %d: the trial set-up failed; the previous hierarchy is back\n", it);
          if ((rc = do_linearize(c))) return rc;
        } else {
        if (rc || (rc = do_linearize(c))) return rc;
        if (trial == 1) trial = 2;
        rebuild_next = false;
        ++rebuilds;
        call_best = 0;   // (another hierarchy: its first solve sets the reference)
        if (c->opts.verbose) std::fprintf(stderr, "[sgo] multigrid hierarchy rebuilt before iteration %d\n", it);
        }
      }
      hipEventRecord(ev[3 * it + 1], c->stream);
      int wasted = 0;
      if ((rc = run_pcg(c))) {
        return rc;
      }
      bool kept_interrupted = false;
      if (c->amg_skip_update && c->amg && (c->h_S->stop == 4 || (c->h_S->stop == 2 && c->h_S->iter < c->opts.pcg_maxit && c->pcg_softcap > 0))) {
        // A solve behind KEPT coarse operators (lagged refresh, do_linearize) that its progress probe stopped -- after a third of the
        // last fresh count its residual was more than half a decade behind that solve's -- or that reached its cap, the
        // fresh count + 3: the operators are refreshed and the solve carries on from where it is.
        const int at = c->h_S->iter;
        c->pcg_softcap = (c->amg_best > 0 && rebuilds < max_rebuilds) ? rules::bail_out_cap(c->amg_best) + at : 0;
        const int maxit = c->pcg_softcap > 0 ? std::min(c->pcg_softcap, c->opts.pcg_maxit) : c->opts.pcg_maxit;
        c->pcg_pred = std::max(2, fresh_pcg);
        if ((rc = refresh_and_continue(c, maxit)) || (rc = run_pcg(c))) return rc;
        kept_interrupted = true;
        const double moved = c->last_dchg[1] > 0.0 ? c->last_dchg[0] / c->last_dchg[1] : 0.0;
        // (counted as sixteen iterations over -- but never past what forbids keeping over 0.01 % of movement: an interruption at next
        // to no movement was not the movement's doing, and a slope that only kept solves can lower must not lock them out)
        c->amg_lag_slope = rules::lag_slope_after_interrupt(c->amg_lag_slope, moved);
        c->amg_lag_slope_seen = true;
        if (c->opts.verbose)
          std::fprintf(stderr, "[sgo] iteration %d: solve behind kept coarse operators interrupted after %d PCG iterations, operators refreshed\n", it, at);
      }
      // A solve that STAGNATED (run_pcg's guard) or ran out of iterations with x at the floating-point floor of its system: accepted,
      // like a backward-stable direct solver's solution (solve_backward_error above).  Single GPU (the guard's domain).
      bool floor_accept = false;
      auto check_floor = [&]() -> int {
        if (floor_accept || !(c->h_S->stop == 2 && (c->pcg_stalled || c->h_S->iter >= c->opts.pcg_maxit) && !c->owner && !multi_rank(c))) return SGO_OK;
        double eta = 1.0;
        const int r2 = solve_backward_error(c, &eta);
        if (r2) return r2;
        floor_accept = eta <= kFloorEta;
        if (floor_accept) {
          ++floor_solves;
          c->floor_seen = true;
          if (c->opts.verbose)
            std::fprintf(stderr, "[sgo] iteration %d: the solve's x is at the floating-point floor of its system (backward error %.1e <= %.0e): step applied\n", it, eta, kFloorEta);
        }
        return SGO_OK;
      };
      if ((rc = check_floor())) return rc;
      if (!floor_accept && c->pcg_softcap > 0 && c->h_S->stop == 2 && c->h_S->iter < c->opts.pcg_maxit && c->amg) {
        // The solve ran into the bail-out cap (4x the best count of this hierarchy): the aggregation no
        // longer fits the re-weighted Hessian.  Redo the set-up from the current values and solve again
        // from x = 0 instead of grinding on (seen: 735 iterations where the rebuilt hierarchy needs 16).
        wasted = c->h_S->iter;
        c->pcg_softcap = 0;
        // (a hierarchy with FILTERED transfers -- sgo_amg_host.h -- that never solved anything: what the filter dropped is not
        // negligible together on this graph, it keeps the tentative transfers on those levels from here on.  One that HAS solved
        // and is abandoned now is stale -- connections it lumped into the diagonal have become strong, and such a hierarchy does
        // not degrade gracefully --: the rebuild filters by the current values)
        if (c->amg_best == 0 && amg_has_filtered(c->amg)) c->amg_no_filter = true;
        else if (c->amg_best == 0) c->amg_theta_scale *= 0.5;   // this hierarchy never worked: coarsen more aggressively
        // The hierarchy the call's LAST rebuild leaves has no safety net behind it (no bail-out cap without a rebuild to follow):
        // it keeps the tentative transfers, which go stale gracefully -- 190 -> 240 iterations where a stale filtered one can
        // grind on to pcg_maxit (seen with another aggregation threshold from the dead-reckoned start: 11 059, then 20 000)
        if (rebuilds + 1 >= max_rebuilds) c->amg_no_filter = true;
        if ((rc = build_amg(c, false, c->knobs.keep_agg && c->amg_best != 0)) || (rc = do_linearize(c)) || (rc = run_pcg(c))) return rc;
        ++rebuilds;
        call_best = 0;   // (another hierarchy: its first solve sets the reference)
        rebuild_next = false;
        if (c->opts.verbose)
          std::fprintf(stderr, "[sgo] iteration %d: solve abandoned after %d PCG iterations, hierarchy rebuilt\n", it, wasted);
        if ((rc = check_floor())) return rc;
      }
      if (!floor_accept && c->h_S->stop == 3 && c->amg && amg_has_filtered(c->amg) && rebuilds < max_rebuilds) {
        // A breakdown (p.Hp <= 0 or a non-finite scalar) behind a hierarchy with FILTERED transfers: the Gauss-Newton Hessian is
        // positive semi-definite by construction, so the first suspect is the preconditioner -- the set-up is redone with the
        // tentative transfers and the solve repeated once; a Hessian that really is indefinite fails again and is reported.
        wasted += c->h_S->iter;
        c->amg_no_filter = true;
        c->pcg_softcap = 0;
        if ((rc = build_amg(c)) || (rc = do_linearize(c)) || (rc = run_pcg(c))) return rc;
        ++rebuilds;
        call_best = 0;   // (another hierarchy: its first solve sets the reference)
        rebuild_next = false;
        if (c->opts.verbose)
          std::fprintf(stderr, "[sgo] iteration %d: PCG breakdown behind a filtered hierarchy, rebuilt with tentative transfers and solved again\n", it);
        if ((rc = check_floor())) return rc;
      }
      if (c->amg_dchg_pending) read_diag_change(c);   // (written before the solve began; the solve's end was waited for)
      if (c->amg_lag_on) {
        // what the blocks moved by in this iteration, against the operators' reference: when they were refreshed now, the NEXT
        // iteration's movement against them will be about this iteration's one-step movement or less
        const double moved = c->last_dchg[1] > 0.0 ? c->last_dchg[0] / c->last_dchg[1] : 1.0;
        c->amg_lag_expect = c->amg_ref_valid && moved <= 2.0 * lag_allowed(c) && c->last_dchg[2] <= 4.0 * c->amg_lag_rows;
      }
      const PcgScalars S = *c->h_S;
      int& best_pcg = c->amg_best;
      if (it == 0 && c->tol_cap > 0.0 && S.stop == 1) c->bb_ref = S.bb;
      if (c->amg && S.stop != 3) {
        // Iteration counts are compared at EQUAL tolerance: a solve that stopped at the absolute criterion (a looser
        // relative tolerance, see pcg_tol_cap) is scaled to what pcg_tol would have cost -- PCG converges linearly,
        // iterations ~ log(1 / tolerance) -- or the staleness rules below would take every tight solve that follows
        // a loose one for a stale hierarchy.
        const double tol0 = c->opts.pcg_tol * c->tol_scale, tolk = std::sqrt(S.tol2);
        const int eq_iter = rules::equal_tolerance_count(S.iter, tol0, tolk);
        if (kept_interrupted || floor_accept) {
          // (counts of a solve that changed its preconditioner half-way, or that ended at the floating-point floor, say nothing
          // about either hierarchy state)
        } else if (c->amg_skip_update) {
          ++kept_solves;
          // behind kept coarse operators: judged against the last fresh solve only -- when the lag has cost more than a refresh is
          // worth (~ 8-10 PCG iterations on C2 / C4), the next solve refreshes whatever the blocks' movement says
          if (rules::kept_solve_too_slow(eq_iter, fresh_pcg)) c->amg_ref_valid = false;
          // ... and every kept solve teaches what movement costs on this graph (the first observation replaces the cautious start,
          // later ones raise the slope at once and lower it slowly)
          const double moved = c->last_dchg[1] > 0.0 ? c->last_dchg[0] / c->last_dchg[1] : 0.0;
          c->amg_lag_slope = rules::lag_slope_after_kept(c->amg_lag_slope, c->amg_lag_slope_seen, eq_iter - fresh_pcg, moved);
          c->amg_lag_slope_seen = true;
        } else {
        fresh_pcg = eq_iter;
        c->amg_lag_slope = rules::lag_slope_after_fresh(c->amg_lag_slope);   // (a high slope is re-examined in time)
        if (trial == 2) {
          // The re-made hierarchy's first two (fresh, warm-started) solves against the old one's first solve of this call (cold): a
          // hierarchy that is better shows it at once -- 22 / 24 against 33 on the C4-sized session --; one that is not -- the
          // re-aggregation can lose a smoothed transfer it had, 42 / 38 against 27 on a 40 k / 60 k graph -- is dropped for the old.
          trial_best = trial_seen == 0 ? eq_iter : std::min(trial_best, eq_iter);
          if (++trial_seen == 2) {
            trial = 0;
            if (rules::trial_reverts(trial_best, trial_old)) {
              if ((rc = revert_amg(c))) return rc;
              call_best = 0;
              c->agg_rule_off = true;
              rebuild_next = false;
              fresh_pcg = 0;
              agg_note = "a re-aggregated hierarchy was tried in the last sgo_optimize_gn and dropped (" + std::to_string(trial_best) + " PCG iterations against the old one's " + std::to_string(trial_old) + ")";
              if (c->opts.verbose) std::fprintf(stderr, "[sgo] iteration %d: the re-aggregated hierarchy is no better (%d against %d): the old one is back\n", it, trial_best, trial_old);
            }
          }
        }
        if (c->agg_best == 0 || eq_iter < c->agg_best) c->agg_best = eq_iter;   // (an incremental update resets amg_best, not this)
        // the progress probe of the solves that keep these operators: iteration fresh / 3 (at least 4), half a decade of slack
        if (S.probe_k > 0 && S.iter >= S.probe_k && S.probe_rel > 0.0) c->amg_probe_max = 10.0 * S.probe_rel;
        else if (S.probe_k > 0) c->amg_probe_max = 0.0;
        const int k_new = rules::probe_iteration(S.iter);
        if (S.probe_k < 4 || std::abs(k_new - S.probe_k) > 1) {   // (the record was taken at another iteration: the next fresh solve records anew)
          c->amg_probe_k = k_new;
          c->amg_probe_max = 0.0;
        }
        if (best_pcg == 0 || eq_iter < best_pcg) best_pcg = eq_iter;
        // Redo the aggregation from the current values when that pays: always when the count has more
        // than doubled, and when it is > 25 % above the best while the PCG iterations it would save
        // over the remaining GN iterations exceed the set-up's cost (~150 PCG iterations' worth: host
        // aggregation + one more linearisation).  Counts only -- no clocks -- so that every rank of a
        // multi-GPU run takes the same decision.
        // (crediting a rebuild only for as long as the previous one of this call stayed good -- weights that keep changing -- was
        // measured in round 5 on C4 from a dead-reckoned start and is worse: the hierarchy left in place went from 74 to 400
        // iterations two solves later; median 29.8 against 23.3 ms per Gauss-Newton iteration)
        // (against the best count of THIS call: the first solve of a call from a far start is harder than the last solves of the call
        // before whatever the hierarchy -- C5 re-optimised from its initial poses: 88 iterations against the 31 the previous call ended
        // with -- and with the best carried over, every such call redid the set-up after its first solve, at a state the NEXT call's
        // start then found useless: 2.2 instead of 1.4 s per call.  Staleness across calls is the movement rule's business, below.)
        if (call_best == 0 || eq_iter < call_best) call_best = eq_iter;
        const int left = iters - it - 1;
        if (rebuilds < max_rebuilds && rules::staleness(eq_iter, call_best, left, c->knobs.rebuild_cost).rebuild()) rebuild_next = true;
        // The aggregation's own staleness, across calls: the hierarchy was aggregated from blocks that have since moved a lot -- a
        // graph set up at poor poses and optimised since -- and this call's first solve needs visibly more iterations than the
        // hierarchy's best.  The count rules above weigh a rebuild against the iterations left in THIS call; the reference calls
        // optimize(20) after every closure (slc.cpp:286-287), and the excess is paid in every call (bench.py's incremental session:
        // 24.7 iterations per solve behind the hierarchy of the initial poses, 19.1 behind one made after the first optimize).
        if (c->agg_grid > 0) {
          double v[3] = {0.0, 0.0, 0.0};
          const volatile double* h = c->h_dchg + 3 * (size_t)kMaxPartials;
          for (int k = 0; k < 3; ++k)
            for (int i = 0; i < c->agg_grid; ++i) v[k] += h[(size_t)k * kMaxPartials + i];
          c->agg_grid = 0;
          const bool moved_far = rules::moved_far(v[0], v[1], v[2], c->n);
          if (c->opts.verbose) std::fprintf(stderr, "[sgo] since the aggregation: blocks moved by %.2f %%, %.0f rows by a quarter; first solve %d, best of this aggregation %d\n", v[1] > 0 ? 100.0 * v[0] / v[1] : 0.0, v[2], eq_iter, c->agg_best);
          if (rules::reaggregate(moved_far, c->agg_rule_off, iters - it, c->agg_best, eq_iter, rebuilds, max_rebuilds, rebuild_next)) {
            rebuild_next = true;
            trial = 1;
            trial_old = eq_iter;
            char nb[160];
            std::snprintf(nb, sizeof nb, "hierarchy re-aggregated in the last sgo_optimize_gn (the blocks had moved by %.0f %% since it was made)", 100.0 * v[0] / v[1]);
            agg_note = nb;
            if (c->opts.verbose)
              std::fprintf(stderr, "[sgo] the blocks have moved by %.1f %% (%.0f rows by a quarter) since the hierarchy was aggregated, first solve %d against its best %d: rebuild\n",
                           100.0 * v[0] / v[1], v[2], eq_iter, c->agg_best);
          }
        }
        }
      }
      its_sum += S.iter + wasted;
      if (it == 0 && S.stop == 1) {   // the call's first solve: what the incremental set-up's staleness rule compares (sgo_ctx.h)
        c->its_last = S.iter;
        if (!c->ov.active && c->its_base == 0) c->its_base = S.iter;
      }
      if (out) {
        out->pcg_iters[it] = S.iter + wasted;   // an abandoned solve's iterations count too
        out->pcg_converged[it] = S.stop == 1 ? 1 : (floor_accept ? 2 : 0);
        out->pcg_relres[it] = S.bb > 0 ? std::sqrt(S.rr / S.bb) : 0.0;
      }
      if (S.stop != 1 && !floor_accept) {
        // Solver failure, as LinearSolverEigen::solve returning false (OptimizationAlgorithm::Fail): the
        // step is NOT applied, estimates stay at the last successful update and the call returns 0 like
        // g2o::SparseOptimizer::optimize.  stop == 3: p.Hp <= 0 or non-finite (H not positive definite);
        // stop == 2: pcg_maxit iterations without reaching pcg_tol (an inexact step is never applied).
        if (S.stop == 3) {
          c->err = "PCG breakdown in GN iteration " + std::to_string(it) + " (Hessian not positive definite";
          if (c->amg && amg_coarsest_not_spd(c->amg, c->stream)) c->err += "; its coarsest Galerkin operator has a non-positive pivot";
          c->err += ")";
        } else {
          c->err = (c->pcg_stalled ? "PCG stagnated (no new minimum of the residual in the last " + std::to_string(c->pcg_stall_window) + " of " +
                                         std::to_string(S.iter) + " iterations; pcg_maxit = "
                                   : "PCG did not reach pcg_tol within pcg_maxit = ") + std::to_string(c->opts.pcg_maxit) +
                   (c->pcg_stalled ? ")" : " iterations") + " in GN iteration " + std::to_string(it) + " (relative residual " +
                   std::to_string(S.bb > 0 ? std::sqrt(S.rr / S.bb) : 0.0) + "); the step was not applied";
        }
        failed = true;
        hipEventRecord(ev[3 * it + 2], c->stream);
        break;
      }
      // (row-owner mode: every rank holds all poses; the ranks' slices of the step are gathered first)
      if (c->owner && !halo_gather_slices(c->halo, c->stream, c->d_x, 3, &c->err)) return SGO_ECOMM;
      if (c->ov.active) launch_ov_finish(c->stream, c->ov.dev, c->d_x, c->d_poses);   // x_N and the appended poses' update
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
        std::fprintf(stderr, "[sgo] iteration= %d\t pcg= %d\t relres= %.3e\t |b|= %.3e\t diag moved %.3e (%.0f rows > 1/4; kept up to %.1e)%s\n", it, S.iter,
                     S.bb > 0 ? std::sqrt(S.rr / S.bb) : 0.0, std::sqrt(S.bb), c->last_dchg[1] > 0 ? c->last_dchg[0] / c->last_dchg[1] : 0.0,
                     c->last_dchg[2], lag_allowed(c), c->amg_skip_update ? " coarse operators kept" : "");
    }
    c->pcg_softcap = 0;
    (void)its_sum;
    if (kept_solves > 0)
      c->lag_note = "last sgo_optimize_gn: " + std::to_string(kept_solves) + " of " + std::to_string(done) + " solves kept the coarse operators of the one before";
    if (!agg_note.empty()) c->lag_note += (c->lag_note.empty() ? "" : "; ") + agg_note;
    if (floor_solves > 0)
      c->lag_note += std::string(c->lag_note.empty() ? "last sgo_optimize_gn: " : "; ") + std::to_string(floor_solves) +
                     " solve(s) stopped at the floating-point floor of their system (normwise backward error <= 1e-12 without reaching pcg_tol): "
                     "steps applied, as a backward-stable direct solver's would be";
    if ((rc = do_chi2(c, c->d_hist + 2 * done, nullptr))) {
      return rc;
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_hist, c->d_hist, sizeof(double) * 2 * (size_t)(done + 1), hipMemcpyDeviceToHost,
                              c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->amg_prev && trial == 0) {   // a trial that was accepted: the replaced hierarchy goes; its arena keeps its chunks for the next
      amg_destroy(c->amg_prev);      // trial (released and re-grown every time, the chunk size doubled until each trial took a gigabyte)
      c->amg_prev = nullptr;
      c->amg_arena_prev.rewind();
    }
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
}

// Micro-benchmark of the level-0 product on the resident graph: `reps` back-to-back launches of
// k_spmv0<mode> (operand = the PCG direction buffer, whatever it holds), HIP events around them on the
// context's stream; returns the mean microseconds per launch (< 0 on error).  variant 16: the wave-group kernel even when the graph has a tile view; variant 32: per-phase s_memtime stamps of the tile kernel on stderr (diagnostic).
double debug_spmv0_us(sgo_ctx* c, int mode, int variant, int reps) {
  if (check_graph(c) != SGO_OK || reps < 1 || c->n == 0 || ensure_rows(c) != SGO_OK || c->owner || c->ov.active) return -1.0;
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1.0;
  Spmv0Args args{};
  args.x = c->d_p;
  args.y = c->d_s2;
  args.b = c->d_b;
  args.omega = 0.8;
  args.force_f64 = (variant & 64) != 0;   // variant 64: RESID / JACOBI on the fp64 blocks although an fp32 copy exists
  const bool tiled = c->T0.ntile > 0 && !(variant & 16);   // variant 16: the wave-group kernel
  long long* d_st = nullptr;
  if ((variant & 32) && tiled) {
    hipMalloc((void**)&d_st, sizeof(long long) * 8 * (size_t)c->T0.ntile);
    hipMemset(d_st, 0, sizeof(long long) * 8 * (size_t)c->T0.ntile);
    args.dbg_stamps = d_st;
  }
  if (tiled) launch_spmv0t(c->stream, c->S0, c->T0, mode, args);
  else launch_spmv0(c->stream, c->S0, mode, args);
  if (variant & 128) {   // variant 128: as nodes of a replayed hipGraph (16 launches per graph), the way the solve runs them
    hipGraph_t g = nullptr;
    hipGraphExec_t ex = nullptr;
    hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
    for (int k = 0; k < 16; ++k) {
      if (tiled) launch_spmv0t(c->stream, c->S0, c->T0, mode, args);
      else launch_spmv0(c->stream, c->S0, mode, args);
    }
    hipStreamEndCapture(c->stream, &g);
    if (!g || hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) return -1.0;
    hipGraphLaunch(ex, c->stream);
    hipEventRecord(a, c->stream);
    for (int k = 0; k < (reps + 15) / 16; ++k) hipGraphLaunch(ex, c->stream);
    hipEventRecord(b, c->stream);
    hipStreamSynchronize(c->stream);
    hipGraphExecDestroy(ex);
    hipGraphDestroy(g);
    reps = (reps + 15) / 16 * 16;
  } else {
  hipEventRecord(a, c->stream);
  for (int k = 0; k < reps; ++k) {
    if (tiled) launch_spmv0t(c->stream, c->S0, c->T0, mode, args);
    else launch_spmv0(c->stream, c->S0, mode, args);
  }
  hipEventRecord(b, c->stream);
  }
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

}  // namespace sgo
