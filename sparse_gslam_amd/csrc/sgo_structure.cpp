// sgo_structure.cpp -- the device-resident graph: edge arrays in caller order (chi2), the level-0 Hessian's symmetric
// storage with its compact slot list, tile view and logical view, the per-slot operand arrays of k_linearize and the
// PCG vectors (Sym0Dev / Tile0Dev / BsrDev / EdgeSlotsDev of sgo_internal.h), built from the row plan of sgo_plan.cpp.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "sgo_ctx.h"

namespace sgo {

// Joins the helper thread of the level-0 analysis; `keep`: leave its result for build_amg, otherwise drop it.
void l0_join(sgo_ctx* c, bool keep) {
  if (c->l0_thread.joinable()) c->l0_thread.join();
  if (!keep && c->l0_pre) {
    amg_host_l0_free(c->l0_pre);
    c->l0_pre = nullptr;
  }
  // (c->l0_w keeps its storage: a fresh 17-MB vector per call is 4000 page faults on the set-up's critical path)
}

int upload_poses(sgo_ctx* c, const double* poses, int V) {
  const size_t cnt = 3 * (size_t)V;
  if (cnt > c->pose_stage_cap) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // (an earlier copy may still read the old buffer)
    if (c->h_pose_stage) hipHostFree(c->h_pose_stage);
    c->h_pose_stage = nullptr;
    c->pose_stage_cap = 0;
    const size_t cap = cnt + cnt / 4 + 3 * (size_t)kOvMaxVerts;
    if (hipHostMalloc((void**)&c->h_pose_stage, sizeof(double) * cap) != hipSuccess) {
      c->h_pose_stage = nullptr;
      c->err = "out of pinned host memory (pose staging)";
      return SGO_ENOMEM;
    }
    c->pose_stage_cap = cap;
  } else {
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // the previous upload has left the staging buffer
  }
  const double t0 = wall_s();
  std::memcpy(c->h_pose_stage, poses, sizeof(double) * cnt);
  const double t1 = wall_s();
  HIP_TRY(c, hipMemcpyAsync(c->d_poses, c->h_pose_stage, sizeof(double) * cnt, hipMemcpyHostToDevice, c->stream));
  if (c->opts.verbose > 1)
    std::fprintf(stderr, "[sgo]   pose staging: host copy %.3f ms, queueing the upload %.3f ms\n", 1e3 * (t1 - t0), 1e3 * (wall_s() - t1));
  return SGO_OK;
}

int halo_reserve(sgo_ctx* c, size_t packet_doubles) {
  // the receive buffer holds one packet per rank: it is remade when the packets grow AND when the communicator has
  // more ranks than the buffers were made for (a context may be given another communicator between graphs)
  const int ranks = std::max(1, c->comm.nranks);
  if (packet_doubles > c->halo_cap || ranks > c->halo_ranks) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->halo_send) hipFree(c->halo_send);
    if (c->halo_recv) hipFree(c->halo_recv);
    c->halo_send = c->halo_recv = nullptr;
    const size_t cap = std::max(c->halo_cap, packet_doubles + packet_doubles / 4 + 64);
    c->halo_cap = 0;
    c->halo_ranks = 0;
    if (hipMalloc((void**)&c->halo_send, sizeof(double) * cap) != hipSuccess ||
        hipMalloc((void**)&c->halo_recv, sizeof(double) * cap * (size_t)ranks) != hipSuccess) {
      c->err = "out of device memory (multi-GPU exchange buffers)";
      return SGO_ENOMEM;
    }
    c->halo_cap = cap;
    c->halo_ranks = ranks;
  }
  c->halo.send = c->halo_send;
  c->halo.recv = c->halo_recv;
  c->halo.cap = c->halo_cap;
  return SGO_OK;
}

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
  if (c->amg_prev) {
    amg_destroy(c->amg_prev);
    c->amg_prev = nullptr;
  }
  c->agg_rule_off = false;
  if (c->direct) {
    direct_destroy(c->direct);
    c->direct = nullptr;
  }
  if (c->mf) {
    mfront_destroy(c->mf);
    c->mf = nullptr;
  }
  c->amg_pending = false;
  c->amg_theta_scale = 1.0;
  c->amg_no_filter = false;
  c->order_xy.clear();
  c->rows_pending = false;
  c->amg_arena.rewind();
  c->amg_arena_prev.rewind();
  c->graph_arena.rewind();   // the caller has synchronised the stream: nothing in flight reads these arrays
  c->pcg_pred = 0;
  c->A = BsrDev();
  c->S0 = Sym0Dev();
  c->T0 = Tile0Dev();
  c->es = EdgeSlotsDev();
  c->el = EdgeListDev();
  c->d_xprev = nullptr;
  c->d_dref = nullptr;
  c->d_dref_agg = nullptr;
  c->agg_ref_valid = false;
  c->amg_ref_valid = false;
  c->warm_valid = false;
  c->has_graph = false;
  c->linearized = false;
  c->owner = false;
  c->gather_slices = false;
  c->replicated = false;
  c->halo = HaloDev();
  c->halo_failed = false;
  c->ov.active = false;
  c->ov.ei.clear();
  c->ov.ej.clear();
  c->ov.new_vertex.clear();
  c->ov.hpos.clear();
  c->ov.updates = 0;
  c->ov.dev.k = c->ov.dev.nt = c->ov.dev.nnz = c->ov.dev.nx = 0;
  c->ov.dev.ncol = 1;
  c->ov.dev.el.cnt = 0;
  c->pcg_exec_key = 0;
  c->d_lanczos = nullptr;   // (lived in the graph arena)
  c->its_base = c->its_last = 0;
  c->update_note.clear();
  c->lag_note.clear();
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
    launch_edge_prepare(c->stream, E, d_meas, d_info, c->el.zinv, c->el.info, (size_t)E, 0);
  }
  // (capacity for the vertices / edges an incremental update may append, sgo_overlay.h)
  if ((rc = dalloc(c, &c->d_poses, 3 * ((size_t)V + kOvMaxVerts)))) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->d_poses, poses, sizeof(double) * 3 * (size_t)V, hipMemcpyHostToDevice, c->stream));
  if ((rc = dalloc(c, &c->d_e2, (size_t)E + kOvMaxEdges))) return rc;
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
  // one tile per CU -- the larger the tiles, the fewer pairs straddle two of them -- of every rank (multi-GPU: a rank owns
  // 1 / nranks of the tiles and wants its own 256 CUs busy)
  const int tile_div = kTileDiv * std::max(1, c->comm.nranks);
  {
    // (build_edges has validated the edges and listed the free active vertices of this very graph; the lazy path of
    // graphs that took the direct solver keeps the lists too)
    const int prc = plan_rows_order(V, poses, fixed, E, ei, ej, &c->err, P, (int)c->free_id.size() == c->n && c->V == V && c->E == E ? &c->free_id : nullptr,
                                    c->order_xy.size() == 2 * (size_t)V ? c->order_xy.data() : nullptr);
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
  c->ov.hpos = P.hpos;   // vertex -> row of the resident structure: what an incremental update classifies appended edges by
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
  c->d_eidx = d_eidx;
  c->d_rowptr = c->d_hrowptr = nullptr;
  if ((rc = upload(c, &c->es.flags, flags))) return rc;
  // (with a communicator the operand arrays are allocated further down, for this rank's slots only in row-owner mode)
  const bool es_early = !c->comm.active();
  c->es.stride = ns;
  if (es_early && ((rc = dalloc(c, &c->es.vi, (size_t)ns)) || (rc = dalloc(c, &c->es.vj, (size_t)ns)) || (rc = dalloc(c, &c->es.zinv, 3 * (size_t)ns)) ||
                   (rc = dalloc(c, &c->es.info, 6 * (size_t)ns)) || (rc = dalloc(c, &c->es.phi, (size_t)ns))))
    return rc;
  // Large graphs: the multigrid's host analysis of level 0 (greedy aggregation + patterns / product lists of the
  // smoothed transfer: C4 11 + 17 ms, the longest sequential piece of the set-up) needs the strength weights and the
  // logical pattern only.  The weights are made right here from the edge list (k_row_strength; a not yet expanded
  // operand array serves as scratch), and a helper thread does the analysis while this one cuts the
  // tiles, types the slots and uploads the level-0 storage; build_amg joins it.
  {
    bool pipeline = c->opts.solver == SGO_SOLVER_PCG_AMG && n >= 20000 && E > 0;
    if (const char* e = std::getenv("SGO_SETUP_PIPELINE")) pipeline = pipeline && std::atoi(e) != 0;
    if (pipeline || c->comm.active()) {   // (row-owner mode makes every later hierarchy's strength weights this way too)
      if ((rc = upload(c, &c->d_rowptr, rowptr)) || (rc = upload(c, &c->d_hrowptr, H.rowptr))) return rc;
    }
    if (pipeline) {
      int *d_rowptr = c->d_rowptr, *d_hrowptr = c->d_hrowptr;
      double* d_w = c->es.info;   // scratch: nslot <= n + ns <= 2 ns doubles of the 6 ns the not yet expanded operand array holds
      if (!es_early && (rc = dalloc(c, &d_w, (size_t)H.nslot))) return rc;
      launch_early_strength(c->stream, c->el, c->d_poses, n, d_rowptr, d_eidx, c->es.flags, d_hrowptr, d_w);
      c->l0_w.resize((size_t)H.nslot);
      HIP_TRY(c, hipMemcpyAsync(c->l0_w.data(), d_w, sizeof(double) * (size_t)H.nslot, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      c->l0_pre = amg_host_l0_new();
      AmgHostL0* pre = c->l0_pre;
      const HostLevel* Hp = &c->H0;
      const std::vector<double>* wp = &c->l0_w;
      ChunkArena* scr = &c->amg_scratch;
      // (one GPU, patterns on the device -- sgo_amg_dev.inc, the default --: the helper thread makes the aggregation only)
      const bool agg_only = c->knobs.setup_mode == 2 && !c->comm.active();
      c->l0_thread = std::thread([pre, Hp, wp, scr, agg_only] {
        HostPool::lane() = 1;   // its own worker pool: runs beside this thread's regions instead of queueing with them
        amg_host_l0_run(pre, *Hp, *wp, AmgConfig(), scr, agg_only);
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
  // wave groups over the compact slots: whole rows packed up to 64 slots; a longer row is its own group; a tile starts
  // a group, so that a range of tiles (one rank's rows, multi-GPU) is a range of groups
  std::vector<int> grp, grow;
  std::vector<int> tile_g0;   // first group of every tile (+ the group count at the end)
  grp.push_back(0);
  {
    int cur = 0, first = 0;
    size_t nt = 0;
    for (int r = 0; r < n; ++r) {
      const int len = rowptr[r + 1] - rowptr[r];
      const bool tile_start = tiles_ok && nt < tiles.size() && tiles[nt].row0 == r;
      if (cur > 0 && (cur + len > 64 || tile_start)) {
        grp.push_back(rowptr[r]);
        grow.push_back(first);
        first = r;
        cur = 0;
      }
      if (tile_start) {
        tile_g0.push_back((int)grow.size());
        ++nt;
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
    tile_g0.push_back((int)grow.size());
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
  // ---- multi-GPU, row-owner mode: the boundary rows of every rank (the rows with an edge into another rank's range)
  c->owner = false;
  c->gather_slices = false;
  c->replicated = false;
  c->halo = HaloDev();
  c->halo_host = HaloHost();
  // (the rank-emulation hook without a communicator keeps the all-reduce mode's sharded passes; fewer tiles than ranks:
  // some rank's range would be empty -- build_rows then runs the graph replicated.  SGO_OWNER_MIN_ROWS: test hook)
  const int owner_min_rows = std::getenv("SGO_OWNER_MIN_ROWS") ? std::atoi(std::getenv("SGO_OWNER_MIN_ROWS")) : 2048;
  if (c->comm.active() && tiles_ok && (int)tiles.size() >= c->comm.nranks && n >= owner_min_rows) {
    const int G = c->comm.nranks, nt = (int)tiles.size();
    HaloHost& HH = c->halo_host;
    HH.G = G;
    HH.me = c->comm.rank;
    HH.rank_row.assign((size_t)G + 1, n);
    std::vector<int> rank_u((size_t)G + 1, nt);
    for (int q = 0; q < G; ++q) {
      int b = 0, e = 0;
      sgo_shard_range(nt, G, q, &b, &e);
      rank_u[q] = b;
      HH.rank_row[q] = b < nt ? tiles[b].row0 : n;
    }
    std::vector<unsigned char> isb((size_t)std::max(n, 1), 0);
    parallel_for(n, [&](int r0, int r1) {
      int q = 0;
      for (int r = r0; r < r1; ++r) {
        while (r >= HH.rank_row[q + 1]) ++q;
        const int lo = HH.rank_row[q], hi = HH.rank_row[q + 1];
        for (int k = rowptr[r]; k < rowptr[r + 1]; ++k)
          if (type[k] != kSlotNoBlock && (col[k] < lo || col[k] >= hi)) {
            isb[r] = 1;
            break;
          }
      }
    });
    std::vector<std::vector<int>> lists((size_t)G);
    long long total = 0;
    int bmax = 0, maxrows = 0;
    for (int q = 0; q < G; ++q) {
      for (int r = HH.rank_row[q]; r < HH.rank_row[q + 1]; ++r)
        if (isb[r]) lists[q].push_back(r);
      total += (long long)lists[q].size();
      bmax = std::max(bmax, (int)lists[q].size());
      maxrows = std::max(maxrows, HH.rank_row[q + 1] - HH.rank_row[q]);
    }
    // Row ownership pays when the boundaries are thin (spatially local closures in Hilbert order: a few per cent of the
    // rows); graphs with random long-range closures have every row on a boundary and keep the all-reduce of the
    // product vectors ("keep both modes").  SGO_COMM_MODE=owner|allreduce forces one.
    bool owner = 4 * total <= (long long)n;
    if (const char* e = std::getenv("SGO_COMM_MODE")) owner = !std::strcmp(e, "owner") ? true : (!std::strcmp(e, "allreduce") ? false : owner);
    // (every rank owns at least one tile -- nt >= G above -- and so at least one row: the level-0 kernels read an empty
    // range, u1 == 0, as "all tiles")
    if (owner) {
      HH.bmax = std::max(bmax, 1);
      HH.bnd.assign((size_t)G * HH.bmax, -1);
      for (int q = 0; q < G; ++q) std::copy(lists[q].begin(), lists[q].end(), HH.bnd.begin() + (size_t)q * HH.bmax);
      HaloDev& H = c->halo;
      H.comm = &c->comm;
      H.G = G;
      H.me = HH.me;
      H.row0 = HH.rank_row[HH.me];
      H.row1 = HH.rank_row[HH.me + 1];
      H.u0 = rank_u[HH.me];
      H.u1 = rank_u[HH.me + 1];
      H.g0 = tile_g0[H.u0];
      H.g1 = tile_g0[H.u1];
      H.bmax = HH.bmax;
      H.maxrows = maxrows;
      H.failed = &c->halo_failed;
      int *d_bnd = nullptr, *d_rr = nullptr, *d_hr = nullptr;
      std::vector<int> others;
      for (int q = 0; q < G; ++q)
        if (q != HH.me) others.insert(others.end(), lists[q].begin(), lists[q].end());
      H.nhalo = (int)others.size();
      if (others.empty()) others.push_back(0);
      if ((rc = upload(c, &d_bnd, HH.bnd)) || (rc = upload(c, &d_rr, HH.rank_row)) || (rc = upload(c, &d_hr, others))) return rc;
      HIP_TRY(c, hipStreamSynchronize(c->stream));   // `others` is a local
      H.bnd = d_bnd;
      H.rank_row = d_rr;
      H.halo_rows = d_hr;
      if ((rc = dalloc(c, &H.gparts, (size_t)kHaloScalars * G))) return rc;
      HIP_TRY(c, hipMemsetAsync(H.gparts, 0, sizeof(double) * kHaloScalars * G, c->stream));
      if ((rc = halo_reserve(c, std::max<size_t>((size_t)kHaloScalars + 3 * (size_t)HH.bmax, 9 * (size_t)maxrows)))) return rc;
      c->owner = true;
      if (c->opts.verbose)
        std::fprintf(stderr, "[sgo] row-owner mode: rank %d of %d owns tiles [%d, %d) = rows [%d, %d); boundary rows %lld of %d (largest rank %d)\n",
                     H.me, G, H.u0, H.u1, H.row0, H.row1, total, n, bmax);
    } else {
      // all-reduce mode: every rank holds the whole graph and evaluates the level-0 products for the rows of its tiles;
      // the product vectors travel as an all-gather of the ranks' (contiguous) slices -- half the bytes of an all-reduce
      // with zero fill, one contributor per row either way
      HaloDev& H = c->halo;
      H.comm = &c->comm;
      H.G = G;
      H.me = HH.me;
      H.row0 = HH.rank_row[HH.me];
      H.row1 = HH.rank_row[HH.me + 1];
      H.u0 = rank_u[HH.me];
      H.u1 = rank_u[HH.me + 1];
      H.maxrows = maxrows;
      H.failed = &c->halo_failed;
      int* d_rr = nullptr;
      if ((rc = upload(c, &d_rr, HH.rank_row))) return rc;
      H.rank_row = d_rr;
      if ((rc = halo_reserve(c, 3 * (size_t)maxrows))) return rc;
      c->gather_slices = true;
      if (c->opts.verbose) std::fprintf(stderr, "[sgo] multi-GPU: %lld of %d rows are boundary rows: all-reduce mode\n", total, n);
    }
  }
  // Row-owner mode: the per-slot and per-block arrays are allocated and uploaded for this rank's rows only; the base
  // pointers are shifted so that the kernels keep addressing them by global slot / storage numbers.
  const bool own_only = c->owner;
  const int orow0 = own_only ? c->halo.row0 : 0, orow1 = own_only ? c->halo.row1 : n;
  const size_t K0 = (size_t)rowptr[orow0], K1 = (size_t)rowptr[orow1];            // compact slots
  const size_t U0 = (size_t)own[K0], U1 = (size_t)own[K1];                        // stored blocks
  const size_t R0 = (size_t)tslot[K0], R1 = (size_t)tslot[K1];                    // transposed slots
  c->level0_bytes = 0;
  auto up_range = [&](auto** p, const auto* host, size_t lo, size_t hi) -> int {
    using T = std::remove_const_t<std::remove_pointer_t<decltype(host)>>;
    T* q = nullptr;
    int r = dalloc(c, &q, hi - lo);
    if (r) return r;
    if (hi > lo) HIP_TRY(c, hipMemcpyAsync(q, host + lo, (hi - lo) * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *p = q - lo;   // (global numbers address the rank's range)
    c->level0_bytes += (long long)((hi - lo) * sizeof(T));
    return SGO_OK;
  };
  Sym0Dev& S = c->S0;
  S.n = n;
  S.nu = nu;
  S.npairs = (nu + ntr) / 2;   // owned = intra pairs + 2 x inter pairs, transposed = intra pairs
  S.ncs = ns;
  S.ngrp = ngrp;
  S.nus = (int)(U1 - U0);
  if ((rc = up_range(&S.col, col.data(), K0, K1))) return rc;
  if ((rc = up_range(&S.meta, meta.data(), K0, K1))) return rc;
  if ((rc = up_range(&S.tref, tref.data(), R0, R1))) return rc;
  if ((rc = upload(c, &S.grp, grp))) return rc;
  if ((rc = upload(c, &S.grow, grow))) return rc;
  if ((rc = upload(c, &S.gown, gown))) return rc;
  if ((rc = upload(c, &S.gtr, gtr))) return rc;
  {
    double* ub = nullptr;
    if ((rc = dalloc(c, &ub, 9 * (U1 - U0)))) return rc;
    S.ublk = ub - 2 * U0;                       // pair p of block u: ublk + 2 (p nus + u)
    S.ublk8 = ub + 8 * (U1 - U0) - U0;          // component 8 of block u: ublk8 + u
    c->level0_bytes += (long long)(72 * (U1 - U0));
    // fp32 copy for the preconditioner's level-0 passes (tile view only; SGO_PRECOND_F32=0: the passes read the fp64 blocks)
    S.fblk = nullptr;
    S.fblk8 = nullptr;
    bool f32 = tiles_ok && !tiles.empty() && c->opts.solver == SGO_SOLVER_PCG_AMG;
    if (const char* e = std::getenv("SGO_PRECOND_F32")) f32 = f32 && std::atoi(e) != 0;
    if (f32) {
      float* fb = nullptr;
      if ((rc = dalloc(c, &fb, 9 * (U1 - U0) + 4))) return rc;
      S.fblk = fb - 4 * U0;                     // quad q of block u: fblk + 4 (q nus + u)
      S.fblk8 = fb + 8 * (U1 - U0) - U0;
      c->level0_bytes += (long long)(36 * (U1 - U0));
    }
  }
  if ((rc = dalloc(c, &S.dblk, 6 * (size_t)n))) return rc;
  if ((rc = dalloc(c, &S.dinv, 6 * (size_t)n))) return rc;
  if (!es_early) {   // the operand arrays of k_linearize: this rank's slots
    const size_t nsl = K1 - K0;
    int *vi = nullptr, *vj = nullptr;
    double *zi = nullptr, *in = nullptr, *ph = nullptr;
    if ((rc = dalloc(c, &vi, nsl)) || (rc = dalloc(c, &vj, nsl)) || (rc = dalloc(c, &zi, 3 * nsl)) || (rc = dalloc(c, &in, 6 * nsl)) ||
        (rc = dalloc(c, &ph, nsl)))
      return rc;
    c->es.stride = (int)nsl;
    c->es.vi = vi - K0;
    c->es.vj = vj - K0;
    c->es.zinv = zi - K0;
    c->es.info = in - K0;
    c->es.phi = ph - K0;
    c->level0_bytes += (long long)(88 * nsl);
  } else {
    c->level0_bytes += (long long)(88 * (size_t)ns);
  }
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
    if (hcol.empty()) hcol.push_back(0);
    if ((rc = upload(c, &TL.tile, tiles))) return rc;
    if ((rc = up_range(&TL.cv, cv.data(), U0, U1))) return rc;
    if ((rc = up_range(&TL.off1, off1.data(), U0, U1))) return rc;
    if ((rc = upload(c, &TL.grp1, grp1))) return rc;
    if ((rc = upload(c, &TL.grow1, grow1))) return rc;
    if ((rc = upload(c, &TL.trowptr, trowptr))) return rc;
    if ((rc = upload(c, &TL.hcol, hcol))) return rc;
    {
      int hs = 0;
      for (const TileDesc& T : tiles) hs = std::max(hs, std::min(T.h1 - T.h0, 1024));
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
  // (the logical view stays whole on every rank: the product-list kernels of the multigrid set-up walk other ranks' rows' patterns)
  if ((rc = upload(c, &A.row, H.row))) return rc;
  if ((rc = upload(c, &A.col, H.col))) return rc;
  if ((rc = upload(c, &A.rowptr, H.rowptr))) return rc;
  {
    int* d_ref = nullptr;
    if ((rc = upload(c, &d_ref, lref))) return rc;
    A.ref = d_ref;
  }
  A.ublk = S.ublk;
  A.ublk8 = S.ublk8;
  A.nus = (size_t)S.nus;
  A.nu = (size_t)nu;
  A.dblk = S.dblk;
  A.dinv = S.dinv;
  // edge arrays in caller order: indices / kernel parameter straight from the caller's buffers; the inverse
  // measurements and the SoA information are made on the device from the raw rows, and the per-slot operand
  // arrays of k_linearize are expanded there too
  if (E > 0 && K1 > K0) launch_slot_expand(c->stream, (int)K0, (int)K1, d_eidx, c->el, c->es);
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
  if ((rc = dalloc(c, &c->d_dref, 6 * (size_t)std::max(n, 1))) || (rc = dalloc(c, &c->d_dref_agg, 6 * (size_t)std::max(n, 1)))) return rc;
  if ((rc = dalloc(c, &c->d_zparts, 2 * (size_t)kMaxPartials))) return rc;
  if ((rc = dalloc(c, &c->d_S, 1))) return rc;
  // diagnostic record of the PCG recurrence's coefficients (scripts/ritz_probe.py)
  c->d_lanczos = nullptr;
  if (std::getenv("SGO_LANCZOS") && (rc = dalloc(c, &c->d_lanczos, 3 * (size_t)kLanczosMax))) return rc;
  HIP_TRY(c, hipMemsetAsync(c->d_S, 0, sizeof(PcgScalars), c->stream));
  c->probe_dev_k = 0;   // (what set_probe, sgo_solve.cpp, believes the device holds)
  c->probe_dev_max = 0.0;
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // host staging vectors die at return
  if (c->opts.verbose)
    std::fprintf(stderr, "[sgo] set_graph: host structure %.1f ms, alloc+upload %.1f ms (%d rows, %d stored blocks, %d slots)\n",
                 1e3 * (tb1 - tb0), 1e3 * (wall_s() - tb1), n, nu, ns);
  return SGO_OK;
}

}  // namespace sgo
