// sgo_plan.cpp -- host-only plan of the level-0 rows: SparseOptimizer::initializeOptimization + BlockSolver::buildStructure
// (g2o's hessian order), the internal Hilbert row order, the compact slot positions of every edge and the tiles of the
// level-0 product kernel.  No GPU involved: also behind sgo_plan_rows for the multi-process tests.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "sgo_ctx.h"

using namespace sgo;

namespace sgo {

// ---- structure build: SparseOptimizer::initializeOptimization + BlockSolver::buildStructure -
// The hessian index map of g2o -- free active vertices in ascending id -- is what the API speaks
// (c->free_id, sgo_free_ids, sgo_linearize, ...).  Internally the rows are numbered along a Hilbert
// curve through the initial poses (c->row_of_asc maps one to the other), which makes the symmetric
// storage of Sym0Dev work: the endpoints of almost every edge end up a few hundred rows apart.
// Host-only plan of the level-0 rows (no GPU involved; also behind sgo_plan_rows for the multi-process tests):
// g2o's hessian order, the internal Hilbert row order, the compact slot positions of every edge and the tiles.
// Row plan, first half: hessian order, internal (Hilbert) row order, compact slots per row.
// `known_free`: the hessian order when the caller has already validated the edge list and listed the free active
// vertices (build_edges does both for the chi2 path): the pass over the edges is then not repeated.
// Positions for the ROW ORDER of a graph whose initial poses contradict its closures (a dead-reckoned start: BASELINE.md's
// literal workload).  The Hilbert order exists to put the two endpoints of an edge a few hundred rows apart; it does so when
// the poses are roughly where the edges say.  Poses chained through odometry alone drift by tens of metres over 10^5 steps, the
// closures then connect rows that are far apart along the curve, and the level-0 tiles of C4 hold 61 % of their pairs twice and
// 260 k halo columns instead of 23 % and 69 k -- every level-0 pass a third slower.  The order is a property of the GRAPH, not
// of the estimate: when more than a quarter of a sample of the non-odometry edges are off by more than max(1, 2 |z|) in
// translation, the positions that order the rows come from a breadth-first spanning tree over ALL edges from the fixed
// vertices (vertex 0 without any): closures shortcut the drift, a pose is a handful of composed measurements away from its
// root.  The estimates are not touched.  Returns false (xy untouched) for a consistent graph or without measurements.
bool plan_order_positions(int V, const double* poses, const uint8_t* fixed, int E, const int32_t* ei, const int32_t* ej,
                          const double* meas, std::vector<double>& xy) {
  if (!meas || E < 64 || V < 64) return false;
  // ---- the check: every (E / 4096)-th edge that is not a step along the trajectory
  {
    const int step = std::max(1, E / 4096);
    int seen = 0, off = 0;
    for (int e = 0; e < E; e += step) {
      const int a = ei[e], b = ej[e];
      if (a < 0 || a >= V || b < 0 || b >= V || a == b || a - b == 1 || b - a == 1) continue;
      const double* pa = poses + 3 * (size_t)a;
      const double* pb = poses + 3 * (size_t)b;
      const double* z = meas + 3 * (size_t)e;
      if (!std::isfinite(pa[0] + pa[1] + pa[2] + pb[0] + pb[1])) continue;
      const double c = std::cos(pa[2]), s = std::sin(pa[2]);
      const double dx = pa[0] + c * z[0] - s * z[1] - pb[0], dy = pa[1] + s * z[0] + c * z[1] - pb[1];
      const double lim = std::max(1.0, 2.0 * std::sqrt(z[0] * z[0] + z[1] * z[1]));
      ++seen;
      if (dx * dx + dy * dy > lim * lim) ++off;
    }
    if (seen < 32 || 4 * off <= seen) return false;
  }
  // ---- adjacency (edge id and direction per entry), breadth-first composition of the measurements
  std::vector<int> ptr((size_t)V + 1, 0);
  for (int e = 0; e < E; ++e) {
    const int a = ei[e], b = ej[e];
    if (a < 0 || a >= V || b < 0 || b >= V || a == b) continue;   // (reported by the plan proper)
    ptr[(size_t)a + 1]++;
    ptr[(size_t)b + 1]++;
  }
  for (int v = 0; v < V; ++v) ptr[(size_t)v + 1] += ptr[v];
  std::vector<int> adj((size_t)ptr[V]), fill(ptr.begin(), ptr.end() - 1);
  for (int e = 0; e < E; ++e) {
    const int a = ei[e], b = ej[e];
    if (a < 0 || a >= V || b < 0 || b >= V || a == b) continue;
    adj[(size_t)fill[a]++] = 2 * e;        // this vertex is the edge's first endpoint
    adj[(size_t)fill[b]++] = 2 * e + 1;    // ... its second
  }
  std::vector<double> q(poses, poses + 3 * (size_t)V);
  std::vector<unsigned char> done((size_t)V, 0);
  std::vector<int> queue;
  queue.reserve(V);
  for (int v = 0; v < V; ++v)
    if (fixed[v]) {
      done[v] = 1;
      queue.push_back(v);
    }
  size_t head = 0;
  for (int seed = 0; seed <= V; ++seed) {   // components without a fixed vertex start from their lowest id, at its given pose
    for (; head < queue.size(); ++head) {
      const int v = queue[head];
      const double* pv = &q[3 * (size_t)v];
      for (int t = ptr[v]; t < ptr[(size_t)v + 1]; ++t) {
        const int e = adj[t] >> 1;
        const bool first = !(adj[t] & 1);
        const int u = first ? ej[e] : ei[e];
        if (done[u]) continue;
        const double* z = meas + 3 * (size_t)e;
        double* pu = &q[3 * (size_t)u];
        if (first) {   // X_u = X_v * Z
          const double c = std::cos(pv[2]), s = std::sin(pv[2]);
          pu[0] = pv[0] + c * z[0] - s * z[1];
          pu[1] = pv[1] + s * z[0] + c * z[1];
          pu[2] = pv[2] + z[2];
        } else {       // X_u = X_v * Z^-1
          const double th = pv[2] - z[2], c = std::cos(th), s = std::sin(th);
          pu[0] = pv[0] - (c * z[0] - s * z[1]);
          pu[1] = pv[1] - (s * z[0] + c * z[1]);
          pu[2] = th;
        }
        done[u] = 1;
        queue.push_back(u);
      }
    }
    if (seed == V) break;
    if (!done[seed] && ptr[(size_t)seed + 1] > ptr[seed]) {
      done[seed] = 1;
      queue.push_back(seed);
    }
  }
  xy.resize(2 * (size_t)V);
  for (int v = 0; v < V; ++v) {
    xy[2 * (size_t)v] = q[3 * (size_t)v];
    xy[2 * (size_t)v + 1] = q[3 * (size_t)v + 1];
  }
  return true;
}

int plan_rows_order(int V, const double* poses, const uint8_t* fixed, int E, const int32_t* ei, const int32_t* ej,
                    std::string* err, RowPlan& P, const std::vector<int>* known_free, const double* order_xy) {
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
    // (order_xy: the positions the order follows when they are not the initial poses -- plan_order_positions)
    const double* pos = order_xy ? order_xy : poses;
    const size_t pstride = order_xy ? 2 : 3;
    for (int h = 0; h < n; ++h) {
      const double* q = pos + pstride * (size_t)P.free_id[h];
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
        const double* q = pos + pstride * (size_t)P.free_id[h];
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
    if (const char* e = std::getenv("SGO_TILE_LDS")) lds_budget = std::atoi(e);   // test hook: small tiles on small graphs
    long long nblk = 0;
    for (int k = 0; k < ns; ++k) nblk += col[k] >= 0;
    // Small graphs take the wave-group kernel (k_spmv0: one lane per slot, no LDS tiles): below ~150 k connected pairs a
    // level-0 pass is a chain of dependent round trips, not a stream, and the tile kernel's three phases + two barriers
    // are the longer chain (measured per pass, tile / wave-group: 40 k pairs 5.4 / 4.4 us, 100 k 8.6 / 5.1, 200 k 7.0 / 7.7,
    // 400 k 10 / 15, 1 M 18-21 / 33-37).  Multi-GPU runs keep the tiles (the unit of the partition); SGO_SPMV0=tile forces them.
    {
      const char* e = std::getenv("SGO_SPMV0");
      const bool force_tile = e && !std::strcmp(e, "tile");
      if (!force_tile && tile_div <= kTileDiv && nblk / 2 < kSmallGraphPairs) {
        P.tiles_ok = false;
        tiles.clear();
        for (int r = 0; r < n; ++r) tile_of_row[r] = 0;
        lap("tiles (small graph: wave-group kernel)");
        return;
      }
    }
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
              std::string* err, RowPlan& P, const double* meas) {
  std::vector<double> xy;
  const bool alt = plan_order_positions(V, poses, fixed, E, ei, ej, meas, xy);
  const int rc = plan_rows_order(V, poses, fixed, E, ei, ej, err, P, nullptr, alt ? xy.data() : nullptr);
  if (rc != SGO_OK) return rc;
  plan_rows_tiles(tile_div, P);
  return SGO_OK;
}

}  // namespace sgo

extern "C" {

int sgo_plan_rows(int32_t V, const double* poses, const uint8_t* fixed, int32_t E, const int32_t* ei, const int32_t* ej,
                  int32_t nranks, int32_t* n_free, int32_t* row_vertex, int32_t* ntiles, int32_t* tile_row_begin,
                  int32_t tile_cap, int32_t* rank_row_begin, const double* meas) {
  if (V <= 0 || E < 0 || !poses || !fixed || (E > 0 && (!ei || !ej)) || nranks < 1 || !n_free) return SGO_EINVAL;
  try {
    RowPlan P;
    std::string err;
    const int rc = plan_rows(V, poses, fixed, E, ei, ej, kTileDiv * std::max(1, (int)nranks), &err, P, meas);   // as sgo_set_graph_se2 cuts them for this world size
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

int sgo_mfront_plan(int32_t V, const double* poses, const uint8_t* fixed, int32_t E, const int32_t* ei, const int32_t* ej,
                    int32_t leaf, double max_crit_mflop, int64_t* stats, int32_t* elim_vertex, int32_t* front_of_elim) {
  if (V <= 0 || E < 0 || !poses || !fixed || (E > 0 && (!ei || !ej)) || !stats) return SGO_EINVAL;
  try {
    std::vector<int> deg((size_t)V, 0), free_id;
    for (int e = 0; e < E; ++e) {
      if (ei[e] < 0 || ei[e] >= V || ej[e] < 0 || ej[e] >= V || ei[e] == ej[e]) {
        g_err = "sgo_mfront_plan: bad edge " + std::to_string(e);
        return SGO_EINVAL;
      }
      deg[ei[e]]++;
      deg[ej[e]]++;
    }
    for (int v = 0; v < V; ++v)
      if (!fixed[v] && deg[v] > 0) free_id.push_back(v);
    MfLimits lim;
    lim.max_rows = 1 << 30;
    if (leaf > 0) lim.leaf = leaf;
    if (max_crit_mflop > 0.0) {
      lim.max_crit_flops = 1e6 * max_crit_mflop;
      lim.max_degree = 1e9;   // (an explicit budget: analyse whatever the density)
    }
    MfPlan P;
    std::string why;
    std::memset(stats, 0, sizeof(int64_t) * 12);
    stats[0] = (int64_t)free_id.size();
    if (!mfront_analyze(V, (int)free_id.size(), free_id.data(), poses, E, ei, ej, lim, &P, &why)) {
      g_err = why;
      return SGO_ENOTHING;   // the graph does not qualify (sgo_last_error says why)
    }
    stats[1] = (int64_t)P.fronts.size();
    stats[2] = P.height + 1;
    stats[3] = P.max_dim;
    stats[4] = P.max_own;
    stats[5] = P.max_bnd;
    stats[6] = (int64_t)P.flops;
    stats[7] = (int64_t)P.crit_flops;
    stats[8] = P.crit_panels;
    stats[9] = P.arena_doubles * 8;
    stats[10] = P.order_kind;
    stats[11] = (int64_t)P.targets.size();
    if (elim_vertex) std::copy(P.elim_vertex.begin(), P.elim_vertex.end(), elim_vertex);
    if (front_of_elim)
      for (size_t f = 0; f < P.fronts.size(); ++f)
        for (int q = 0; q < P.fronts[f].own; ++q) front_of_elim[P.fronts[f].e0 + q] = (int32_t)f;
    return SGO_OK;
  } catch (const std::bad_alloc&) {
    g_err = "sgo_mfront_plan: out of host memory";
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
