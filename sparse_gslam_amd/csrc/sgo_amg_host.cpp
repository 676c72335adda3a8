// sgo_amg_host.cpp -- see sgo_amg_host.h
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "sgo_amg_host.h"
#include "sgo_hostpool.h"

namespace sgo {

// wave groups over segments [ptr[i], ptr[i+1]): whole segments packed up to 64 items; a longer
// segment is its own group (same rule as the level-0 row groups in sgo_structure.cpp).  Round 6: the rule is applied
// independently to chunks of kGroupChunk segments -- a chunk starts a new group --, literally as the device's list builder does
// (k_group_chunks, sgo_amg.hip): a hierarchy set up on the device (sgo_amg_dev.inc) then has the SAME groups as one set up here,
// and with them the same association order inside every wavefront segmented scan (its DPP steps are tied to the lanes' positions):
// the two are bit-identical, which is what tests/test_gpu_device_setup.py asserts.  The grouping never changes WHICH terms a sum has.
std::vector<int> make_groups(const std::vector<int>& ptr) {
  std::vector<int> grp;
  const int nseg = (int)ptr.size() - 1;
  for (int s0 = 0; s0 < nseg; s0 += kGroupChunk) {
    const int s1 = std::min(nseg, s0 + kGroupChunk);
    int cur = 0, start = ptr[s0];
    bool open = false;
    for (int f = s0; f < s1; ++f) {
      const int b = ptr[f], len = ptr[f + 1] - b;
      if (open && cur + len > 64) {
        grp.push_back(start);
        open = false;
        cur = 0;
      }
      if (!open) {
        start = b;
        open = true;
      }
      cur += len;
      if (cur >= 64) {
        grp.push_back(start);
        open = false;
        cur = 0;
      }
    }
    if (open) grp.push_back(start);
  }
  grp.push_back(nseg >= 0 ? ptr[nseg] : 0);   // (the list is closed by the total)
  return grp;
}

namespace {

// Greedy root-node aggregation (Vanek et al.) on the strength graph
//   strong(i,j)  <=>  w_ij >= theta * sqrt(w_ii w_jj)
int aggregate(const HostLevel& L, const std::vector<double>& w, double theta, std::vector<int>& agg, ChunkArena* scratch = nullptr) {
  const int n = L.n;
  agg.assign(n, -1);
  const int* visit = L.visit.size() == (size_t)n ? L.visit.data() : nullptr;
  // The strong neighbours of every node (and their weights), compacted IN VISITING ORDER by the host pool: the three
  // sequential passes below then stream through two flat arrays instead of hopping through the rows -- the visiting
  // order (along the trajectory) is not the row order (along the Hilbert curve), and the hops were most of their time.
  std::vector<int> sptr((size_t)n + 1, 0);
  host_parallel_for(n, 2048, [&](int lo, int hi, int) {
    for (int t = lo; t < hi; ++t) {
      const int i = visit ? visit[t] : t;
      const double di = w[L.rowptr[i]];
      int cnt = 0;
      for (int k = L.rowptr[i] + 1; k < L.rowptr[i + 1]; ++k) {
        const int j = L.col[k];
        if (j == i) continue;
        const double th = theta * theta * di * w[L.rowptr[j]];   // w_ij >= theta sqrt(d_i d_j), squared
        cnt += (w[k] > 0.0 && w[k] * w[k] >= th) ? 1 : 0;
      }
      sptr[t + 1] = cnt;
    }
  });
  for (int t = 0; t < n; ++t) sptr[t + 1] += sptr[t];
  // (from the set-up's scratch arena when there is one: its pages are warm, a fresh 24 MB would be faulted in here)
  std::vector<int> scol_own;
  std::vector<double> sw_own;
  int* scol;
  double* sw;
  if (scratch) {
    scol = (int*)scratch->take(sizeof(int) * (size_t)std::max(sptr[n], 1));
    sw = (double*)scratch->take(sizeof(double) * (size_t)std::max(sptr[n], 1));
  } else {
    scol_own.resize((size_t)std::max(sptr[n], 1));
    sw_own.resize((size_t)std::max(sptr[n], 1));
    scol = scol_own.data();
    sw = sw_own.data();
  }
  host_parallel_for(n, 2048, [&](int lo, int hi, int) {
    for (int t = lo; t < hi; ++t) {
      const int i = visit ? visit[t] : t;
      const double di = w[L.rowptr[i]];
      int q = sptr[t];
      for (int k = L.rowptr[i] + 1; k < L.rowptr[i + 1]; ++k) {
        const int j = L.col[k];
        if (j == i) continue;
        const double th = theta * theta * di * w[L.rowptr[j]];
        if (w[k] > 0.0 && w[k] * w[k] >= th) {
          scol[q] = j;
          sw[q] = w[k];
          ++q;
        }
      }
    }
  });
  int nc = 0;
  // pass 1: a node all of whose strong neighbours are free roots a new aggregate
  for (int t = 0; t < n; ++t) {
    const int i = visit ? visit[t] : t;
    if (agg[i] >= 0 || sptr[t] == sptr[t + 1]) continue;
    bool ok = true;
    for (int q = sptr[t]; q < sptr[t + 1] && ok; ++q)
      if (agg[scol[q]] >= 0) ok = false;
    if (!ok) continue;
    agg[i] = nc;
    for (int q = sptr[t]; q < sptr[t + 1]; ++q) agg[scol[q]] = nc;
    ++nc;
  }
  // pass 2: leftovers join the aggregate of their strongest aggregated strong neighbour
  std::vector<int> agg1(agg);
  for (int t = 0; t < n; ++t) {
    const int i = visit ? visit[t] : t;
    if (agg1[i] >= 0) continue;
    double best = -1.0;
    int ba = -1;
    for (int q = sptr[t]; q < sptr[t + 1]; ++q)
      if (agg1[scol[q]] >= 0 && sw[q] > best) {
        best = sw[q];
        ba = agg1[scol[q]];
      }
    if (ba >= 0) agg[i] = ba;
  }
  // pass 3: whatever is left forms aggregates with its free strong neighbours
  for (int t = 0; t < n; ++t) {
    const int i = visit ? visit[t] : t;
    if (agg[i] >= 0) continue;
    agg[i] = nc;
    for (int q = sptr[t]; q < sptr[t + 1]; ++q)
      if (agg[scol[q]] < 0) agg[scol[q]] = nc;
    ++nc;
  }
  return nc;
}

// Renumber the aggregates in the order in which they first appear along the rows (the coarse level then
// inherits the fine level's locality); `visit_c` receives the new ids in creation order, i.e. the order in
// which the next level's aggregation should visit them to continue along the trajectory.
void renumber_aggregates(std::vector<int>& agg, int nc, std::vector<int>& visit_c) {
  std::vector<int> newid((size_t)nc, -1);
  int next = 0;
  for (size_t i = 0; i < agg.size(); ++i)
    if (newid[agg[i]] < 0) newid[agg[i]] = next++;
  for (size_t i = 0; i < agg.size(); ++i) agg[i] = newid[agg[i]];
  visit_c.assign(newid.begin(), newid.end());
}

// Per-row counting sort of products by the local index q of their target: count(q) for every
// product, then start(), then place() returns each product's destination (stable).
struct RowSorter {
  std::vector<int> off;
  void begin(int ntargets) { off.assign((size_t)ntargets + 1, 0); }
  void count(int q) { off[(size_t)q + 1]++; }
  void start(int base, int first_target, int* ptr) {
    const int nt = (int)off.size() - 1;
    for (int q = 0; q < nt; ++q) off[q + 1] += off[q];
    for (int q = 0; q < nt; ++q) {
      off[q] += base;
      ptr[first_target + q] = off[q];
    }
  }
  int place(int q) { return off[q]++; }
};

// Returns false (nothing usable in `o`) when the product lists would exceed `budget` products: graphs
// with many long-range edges make the smoothed coarse operators nearly dense, and the caller then
// keeps the tentative prolongator for this level.
bool sa_symbolic(const HostLevel& H, const std::vector<int>& agg, int nc, const std::vector<int>& mem_ptr,
                 const std::vector<int>& mem, long long budget, bool lists_on_device, SaHost& o, const unsigned char* strong = nullptr) {
  const int n = H.n;
  o.lists_on_device = lists_on_device;
  // (filtered smoothing: a slot the filter drops contributes neither to P's pattern nor to its values)
  auto kept = [&](int k) { return strong == nullptr || strong[k] != 0; };
  const bool verbose = std::getenv("SGO_VERBOSE") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    const auto t1 = std::chrono::steady_clock::now();
    if (verbose && n > 20000) std::fprintf(stderr, "[sgo]   sa_symbolic %-12s %.1f ms\n", what, 1e3 * std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  };
  // ---- P: row i holds the aggregates of the columns of row i (its own among them: diagonal slot)
  o.p_rowptr.assign((size_t)n + 1, 0);
  o.val_rowptr.assign((size_t)n + 1, 0);
  host_parallel_for(n, 512, [&](int lo, int hi, int) {
    std::vector<int> mark((size_t)nc, -1);
    for (int i = lo; i < hi; ++i) {
      int cnt = 0, nk = 0;
      for (int k = H.rowptr[i]; k < H.rowptr[i + 1]; ++k) {
        if (!kept(k)) continue;
        ++nk;
        const int a = agg[H.col[k]];
        if (mark[a] != i) {
          mark[a] = i;
          ++cnt;
        }
      }
      o.p_rowptr[i + 1] = cnt;
      o.val_rowptr[i + 1] = nk;
    }
  });
  for (int i = 0; i < n; ++i) {
    o.p_rowptr[i + 1] += o.p_rowptr[i];
    o.val_rowptr[i + 1] += o.val_rowptr[i];
  }
  const int np = o.p_rowptr[n], nval = o.val_rowptr[n];
  o.p_row.resize(np);
  o.p_col.resize(np);
  o.val_src.resize(nval);
  o.val_tgt.resize(nval);
  std::vector<int> val_ptr((size_t)np + 1);
  host_parallel_for(n, 512, [&](int lo, int hi, int) {
    std::vector<int> mark((size_t)nc, -1), pos((size_t)nc, 0), uniq;
    RowSorter rs;
    for (int i = lo; i < hi; ++i) {
      uniq.clear();
      for (int k = H.rowptr[i]; k < H.rowptr[i + 1]; ++k) {
        if (!kept(k)) continue;
        const int a = agg[H.col[k]];
        if (mark[a] != i) {
          mark[a] = i;
          uniq.push_back(a);
        }
      }
      std::sort(uniq.begin(), uniq.end());
      const int e0 = o.p_rowptr[i];
      for (size_t q = 0; q < uniq.size(); ++q) {
        pos[uniq[q]] = (int)q;
        o.p_row[e0 + q] = i;
        o.p_col[e0 + q] = uniq[q];
      }
      rs.begin((int)uniq.size());
      for (int k = H.rowptr[i]; k < H.rowptr[i + 1]; ++k)
        if (kept(k)) rs.count(pos[agg[H.col[k]]]);
      rs.start(o.val_rowptr[i], e0, val_ptr.data());
      for (int k = H.rowptr[i]; k < H.rowptr[i + 1]; ++k) {
        if (!kept(k)) continue;
        const int q = pos[agg[H.col[k]]], dst = rs.place(q);
        o.val_src[dst] = k;
        o.val_tgt[dst] = e0 + q;
      }
    }
  });
  val_ptr[np] = nval;
  o.val_grp = make_groups(val_ptr);
  lap("P");
  // ---- early verdict on the coarse operator's size from every 32nd coarse row (the exact count comes after the AP
  // pattern and product lists, which cost several times this whole function's share so far; graphs with long-range
  // closures fail it -- C5: 0.37 s of lists made for nothing).  Row a of P^T A P holds the columns of the AP rows of the
  // rows with a P entry in column a, i.e. of the members of a and their neighbours.
  if (nc >= 256 && (long long)H.nslot <= 32LL * n) {   // (denser levels: the product count below says no at once)
    const int step = 32, nsample = (nc + step - 1) / step;
    std::vector<long long> cnt((size_t)nsample, 0);
    host_parallel_for(nsample, 4, [&](int s0, int s1, int) {
      std::vector<int> mark_c((size_t)nc, -1), mark_r((size_t)n, -1);
      for (int sidx = s0; sidx < s1; ++sidx) {
        const int a = sidx * step;
        long long c_a = 0;
        auto visit_row = [&](int i) {
          if (mark_r[i] == a) return;
          mark_r[i] = a;
          for (int k = H.rowptr[i]; k < H.rowptr[i + 1]; ++k) {
            const int j = H.col[k];
            for (int e = o.p_rowptr[j]; e < o.p_rowptr[j + 1]; ++e) {
              const int c = o.p_col[e];
              if (mark_c[c] != a) {
                mark_c[c] = a;
                ++c_a;
              }
            }
          }
        };
        for (int t = mem_ptr[a]; t < mem_ptr[a + 1]; ++t) {
          const int m = mem[t];
          for (int k = H.rowptr[m]; k < H.rowptr[m + 1]; ++k) visit_row(H.col[k]);   // (the diagonal slot is m itself)
        }
        cnt[sidx] = c_a;
      }
    });
    long long sum = 0;
    for (long long v : cnt) sum += v;
    const double estimate = (double)sum * nc / nsample;
    lap("coarse size estimate");
    if (estimate > 1.5 * std::max(H.nslot, 4096)) return false;
  }
  // ---- AP: row i holds the union of the P rows of the columns of row i
  std::vector<int> ap_rowptr((size_t)n + 1, 0);
  std::vector<long long> app((size_t)n + 1, 0);
  // (device-lists mode: the sorted distinct columns are collected in the same traversal, per task, and copied to
  // their place once the row pointers are known -- one walk over the 13 M (slot, P entry) pairs of C4 instead of two)
  {   // the number of products is a sum of P row lengths over the slots: checked before any pattern is collected
      // (dense coarse levels of graphs with long-range closures fail it: C5 level 2, 59 ms of pattern work for nothing)
    std::vector<long long> part((size_t)std::max(1, std::min(HostPool::get().size(), n / 2048)) + 1, 0);
    host_parallel_for(n, 2048, [&](int lo, int hi, int task) {
      long long sum = 0;
      for (int k = H.rowptr[lo]; k < H.rowptr[hi]; ++k) sum += o.p_rowptr[H.col[k] + 1] - o.p_rowptr[H.col[k]];
      part[task] = sum;
    });
    long long total = 0;
    for (long long v : part) total += v;
    if (total > budget) return false;
  }
  const int ap_tasks = std::max(1, std::min(HostPool::get().size(), n / 512));
  std::vector<std::vector<int>> ap_local((size_t)(lists_on_device ? ap_tasks : 0));
  if (lists_on_device) {
    host_parallel_for(n, 512, [&](int lo, int hi, int task) {
      std::vector<int> mark((size_t)nc, -1), uniq;
      std::vector<int>& loc = ap_local[task];
      loc.clear();
      for (int i = lo; i < hi; ++i) {
        uniq.clear();
        long long prod = 0;
        for (int k = H.rowptr[i]; k < H.rowptr[i + 1]; ++k) {
          const int j = H.col[k];
          prod += o.p_rowptr[j + 1] - o.p_rowptr[j];
          for (int e = o.p_rowptr[j]; e < o.p_rowptr[j + 1]; ++e) {
            const int c = o.p_col[e];
            if (mark[c] != i) {
              mark[c] = i;
              uniq.push_back(c);
            }
          }
        }
        std::sort(uniq.begin(), uniq.end());
        loc.insert(loc.end(), uniq.begin(), uniq.end());
        ap_rowptr[i + 1] = (int)uniq.size();
        app[i + 1] = prod;
      }
    });
  } else
  host_parallel_for(n, 512, [&](int lo, int hi, int) {
    std::vector<int> mark((size_t)nc, -1);
    for (int i = lo; i < hi; ++i) {
      int cnt = 0;
      long long prod = 0;
      for (int k = H.rowptr[i]; k < H.rowptr[i + 1]; ++k) {
        const int j = H.col[k];
        prod += o.p_rowptr[j + 1] - o.p_rowptr[j];
        for (int e = o.p_rowptr[j]; e < o.p_rowptr[j + 1]; ++e) {
          const int c = o.p_col[e];
          if (mark[c] != i) {
            mark[c] = i;
            ++cnt;
          }
        }
      }
      ap_rowptr[i + 1] = cnt;
      app[i + 1] = prod;
    }
  });
  for (int i = 0; i < n; ++i) {
    ap_rowptr[i + 1] += ap_rowptr[i];
    app[i + 1] += app[i];
  }
  lap("AP count");
  if (app[n] > budget || ap_rowptr[n] < 0) return false;
  // ---- entries by coarse column (restriction walks P^T)
  std::vector<int> t_ptr((size_t)nc + 1, 0);
  for (int e = 0; e < np; ++e) t_ptr[(size_t)o.p_col[e] + 1]++;
  for (int a = 0; a < nc; ++a) t_ptr[a + 1] += t_ptr[a];
  std::vector<int> t_idx(np);
  o.t_pos.resize(np);
  o.t_row.resize(np);
  o.t_col.resize(np);
  {
    std::vector<int> fill(t_ptr.begin(), t_ptr.end() - 1);
    for (int e = 0; e < np; ++e) {
      const int t = fill[o.p_col[e]]++;
      t_idx[t] = e;
      o.t_pos[e] = t;
      o.t_row[t] = o.p_row[e];
      o.t_col[t] = o.p_col[e];
    }
  }
  o.t_grp = make_groups(t_ptr);
  o.r_grp = make_groups(o.p_rowptr);
  lap("P^T lists");
  o.nap = ap_rowptr[n];
  const int nprod_ap = (int)app[n];
  o.n_ap_prod = app[n];
  UVec& ap_col = o.ap_col;
  ap_col.resize((size_t)o.nap);
  std::vector<int> ap_ptr((size_t)(lists_on_device ? 0 : o.nap) + 1);
  if (lists_on_device) {
    // pattern only: the tasks' column lists to their place (and the row of every entry, for the device's walk)
    o.ap_row.resize((size_t)o.nap);
    host_parallel_for(n, 512, [&](int lo, int hi, int task) {
      const std::vector<int>& loc = ap_local[task];
      std::copy(loc.begin(), loc.end(), ap_col.begin() + ap_rowptr[lo]);
      for (int i = lo; i < hi; ++i)
        for (int f = ap_rowptr[i]; f < ap_rowptr[i + 1]; ++f) o.ap_row[f] = i;
    });
    ap_local.clear();
    lap("AP pattern");
  } else {
  o.ap_a.resize(nprod_ap);
  o.ap_b.resize(nprod_ap);
  o.ap_tgt.resize(nprod_ap);
  host_parallel_for(n, 256, [&](int lo, int hi, int) {
    std::vector<int> mark((size_t)nc, -1), pos((size_t)nc, 0), uniq;
    RowSorter rs;
    for (int i = lo; i < hi; ++i) {
      uniq.clear();
      for (int k = H.rowptr[i]; k < H.rowptr[i + 1]; ++k) {
        const int j = H.col[k];
        for (int e = o.p_rowptr[j]; e < o.p_rowptr[j + 1]; ++e) {
          const int c = o.p_col[e];
          if (mark[c] != i) {
            mark[c] = i;
            uniq.push_back(c);
          }
        }
      }
      std::sort(uniq.begin(), uniq.end());
      const int f0 = ap_rowptr[i];
      for (size_t q = 0; q < uniq.size(); ++q) {
        pos[uniq[q]] = (int)q;
        ap_col[f0 + q] = uniq[q];
      }
      rs.begin((int)uniq.size());
      for (int k = H.rowptr[i]; k < H.rowptr[i + 1]; ++k) {
        const int j = H.col[k];
        for (int e = o.p_rowptr[j]; e < o.p_rowptr[j + 1]; ++e) rs.count(pos[o.p_col[e]]);
      }
      rs.start((int)app[i], f0, ap_ptr.data());
      int *pa = o.ap_a.data(), *pb = o.ap_b.data(), *pt = o.ap_tgt.data();
      for (int k = H.rowptr[i]; k < H.rowptr[i + 1]; ++k) {
        const int j = H.col[k];
        for (int e = o.p_rowptr[j]; e < o.p_rowptr[j + 1]; ++e) {
          const int q = pos[o.p_col[e]], dst = rs.place(q);
          pa[dst] = k;
          pb[dst] = e;
          pt[dst] = f0 + q;
        }
      }
    }
  });
  ap_ptr[o.nap] = nprod_ap;
  o.ap_grp = make_groups(ap_ptr);
  lap("AP fill");
  }
  // ---- A_c = P^T AP: coarse row a collects, over the entries (i, a) of column a of P, row i of AP
  HostLevel& C = o.Hc;
  C.n = nc;
  C.rowptr.assign((size_t)nc + 1, 0);
  std::vector<long long> rpp((size_t)nc + 1, 0);
  const int rap_tasks = std::max(1, std::min(HostPool::get().size(), nc / 64));
  std::vector<std::vector<int>> rap_local((size_t)(lists_on_device ? rap_tasks : 0));
  if (lists_on_device) {   // counts and sorted distinct columns in one traversal (see A P above)
    host_parallel_for(nc, 64, [&](int lo, int hi, int task) {
      std::vector<int> mark((size_t)nc, -1), uniq;
      std::vector<int>& loc = rap_local[task];
      loc.clear();
      for (int a = lo; a < hi; ++a) {
        uniq.clear();
        long long prod = 0;
        bool diag = false;
        for (int t = t_ptr[a]; t < t_ptr[a + 1]; ++t) {
          const int i = o.t_row[t];
          for (int f = ap_rowptr[i]; f < ap_rowptr[i + 1]; ++f) {
            const int c = ap_col[f];
            prod += c >= a;
            if (mark[c] != a) {
              mark[c] = a;
              if (c != a) uniq.push_back(c);
              else diag = true;
            }
          }
        }
        std::sort(uniq.begin(), uniq.end());
        loc.push_back(a);   // diagonal slot first (BsrDev convention; every aggregate reaches itself: its own members' rows)
        loc.insert(loc.end(), uniq.begin(), uniq.end());
        C.rowptr[a + 1] = (int)uniq.size() + 1;
        rpp[a + 1] = prod;
        (void)diag;
      }
    });
  } else
  host_parallel_for(nc, 64, [&](int lo, int hi, int) {
    std::vector<int> mark((size_t)nc, -1);
    for (int a = lo; a < hi; ++a) {
      int cnt = 0;
      long long prod = 0;
      for (int t = t_ptr[a]; t < t_ptr[a + 1]; ++t) {
        const int i = o.t_row[t];
        for (int f = ap_rowptr[i]; f < ap_rowptr[i + 1]; ++f) {
          const int c = ap_col[f];
          prod += c >= a;   // A_c is symmetric: only its upper triangle is computed, the lower is mirrored (rap_mirror)
          if (mark[c] != a) {
            mark[c] = a;
            ++cnt;
          }
        }
      }
      C.rowptr[a + 1] = cnt;
      rpp[a + 1] = prod;
    }
  });
  for (int a = 0; a < nc; ++a) {
    C.rowptr[a + 1] += C.rowptr[a];
    rpp[a + 1] += rpp[a];
  }
  lap("RAP count");
  if (app[n] + rpp[nc] > budget || C.rowptr[nc] > std::max(H.nslot, 4096)) return false;
  C.nslot = C.rowptr[nc];
  const int nprod_rap = (int)rpp[nc];
  o.n_rap_prod = rpp[nc];
  C.row.resize(C.nslot);
  C.col.resize(C.nslot);
  std::vector<int> rap_ptr((size_t)(lists_on_device ? 0 : C.nslot) + 1);
  if (lists_on_device) {
    host_parallel_for(nc, 64, [&](int lo, int hi, int task) {
      const std::vector<int>& loc = rap_local[task];
      std::copy(loc.begin(), loc.end(), C.col.begin() + C.rowptr[lo]);
      for (int a = lo; a < hi; ++a)
        for (int k = C.rowptr[a]; k < C.rowptr[a + 1]; ++k) C.row[k] = a;
    });
    rap_local.clear();
  } else {
  o.rap_a.resize(nprod_rap);
  o.rap_b.resize(nprod_rap);
  o.rap_tgt.resize(nprod_rap);
  host_parallel_for(nc, 64, [&](int lo, int hi, int) {
    std::vector<int> mark((size_t)nc, -1), pos((size_t)nc, 0), uniq;
    RowSorter rs;
    for (int a = lo; a < hi; ++a) {
      uniq.clear();
      for (int t = t_ptr[a]; t < t_ptr[a + 1]; ++t) {
        const int i = o.t_row[t];
        for (int f = ap_rowptr[i]; f < ap_rowptr[i + 1]; ++f) {
          const int c = ap_col[f];
          if (mark[c] != a) {
            mark[c] = a;
            if (c != a) uniq.push_back(c);
          }
        }
      }
      std::sort(uniq.begin(), uniq.end());
      const int s0 = C.rowptr[a];   // diagonal slot first (BsrDev convention), then ascending columns
      C.row[s0] = a;
      C.col[s0] = a;
      pos[a] = 0;
      for (size_t q = 0; q < uniq.size(); ++q) {
        pos[uniq[q]] = (int)q + 1;
        C.row[s0 + 1 + q] = a;
        C.col[s0 + 1 + q] = uniq[q];
      }
      rs.begin((int)uniq.size() + 1);
      for (int t = t_ptr[a]; t < t_ptr[a + 1]; ++t) {
        const int i = o.t_row[t];
        for (int f = ap_rowptr[i]; f < ap_rowptr[i + 1]; ++f)
          if (ap_col[f] >= a) rs.count(pos[ap_col[f]]);
      }
      rs.start((int)rpp[a], s0, rap_ptr.data());
      int *pa = o.rap_a.data(), *pb = o.rap_b.data(), *pt = o.rap_tgt.data();
      for (int t = t_ptr[a]; t < t_ptr[a + 1]; ++t) {
        const int e = t_idx[t], i = o.t_row[t];
        for (int f = ap_rowptr[i]; f < ap_rowptr[i + 1]; ++f) {
          if (ap_col[f] < a) continue;
          const int q = pos[ap_col[f]], dst = rs.place(q);
          pa[dst] = e;
          pb[dst] = f;
          pt[dst] = s0 + q;
        }
      }
    }
  });
  rap_ptr[C.nslot] = nprod_rap;
  o.rap_grp = make_groups(rap_ptr);
  }
  // slot (a, c), c > a  ->  slot (c, a): where the numeric kernel stores the transposed block
  o.rap_mirror.assign((size_t)C.nslot, -1);
  host_parallel_for(nc, 256, [&](int lo, int hi, int) {
    for (int a = lo; a < hi; ++a)
      for (int k = C.rowptr[a] + 1; k < C.rowptr[a + 1]; ++k) {
        const int c = C.col[k];
        if (c < a) continue;
        const int* b = C.col.data() + C.rowptr[c] + 1;
        const int* e = C.col.data() + C.rowptr[c + 1];
        const int* it = std::lower_bound(b, e, a);
        o.rap_mirror[k] = (it != e && *it == a) ? (int)(it - C.col.data()) : -1;
      }
  });
  {   // every lower slot must be some upper slot's mirror (the pattern of P^T A P is symmetric when A's is)
    long long lower = 0, mirrored = 0;
    for (int a = 0; a < nc; ++a)
      for (int k = C.rowptr[a] + 1; k < C.rowptr[a + 1]; ++k) {
        lower += C.col[k] < a;
        mirrored += o.rap_mirror[k] >= 0;
      }
    if (lower != mirrored) return false;
  }
  lap("RAP fill");
  if (lists_on_device) {
    o.ap_rowptr = std::move(ap_rowptr);
    o.t_ptr = std::move(t_ptr);
    o.t_idx = std::move(t_idx);
  }
  return true;
}


}  // namespace

// The aggregation alone (what host_coarsen does first): for a set-up that makes the patterns on the device (sgo_amg_dev.inc) from
// the host's aggregates -- the greedy walk along the trajectory is sequential by nature and its aggregates are visibly better than
// a parallel independent-set aggregation's (NOTES.md section 28), so it stays here; it is a fifth of the host set-up's time.
int host_aggregate(const HostLevel& H, const std::vector<double>& w, const AmgConfig& cfg, int l, ChunkArena* scratch, std::vector<int>& agg,
                   std::vector<int>& visit_c, double* theta_used) {
  const double theta_l = (l == 0 ? cfg.theta : cfg.theta_coarse) * cfg.theta_scale;
  if (scratch) scratch->rewind();
  *theta_used = theta_l;
  int nc = aggregate(H, w, theta_l, agg, scratch);
  if (nc > 0.9 * H.n) {   // stalled: treat every connection as strong
    nc = aggregate(H, w, 0.0, agg, scratch);
    *theta_used = 0.0;
  }
  if (nc > 0.9 * H.n || nc < 1) return 0;   // cannot coarsen further
  visit_c.clear();
  if (!H.visit.empty()) renumber_aggregates(agg, nc, visit_c);
  return nc;
}

void host_coarsen(const HostLevel& H, const std::vector<double>& w, const AmgConfig& cfg, int l, ChunkArena* scratch, HostCoarse& o) {
  const int n = H.n;
  std::vector<int>& agg = o.agg;
  const double theta_l = (l == 0 ? cfg.theta : cfg.theta_coarse) * cfg.theta_scale;
  const auto tA = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point t) {
    return 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
  };
  if (scratch) scratch->rewind();   // (the previous level's lists have been uploaded: amg_create synchronises per level)
  double theta_used = theta_l;
  int nc;
  if (o.reuse_agg && (int)agg.size() == n && o.nc > 0) {
    nc = o.nc;   // (kept from the hierarchy this one replaces: already renumbered, visit_c given)
  } else {
  o.reuse_agg = false;
  nc = aggregate(H, w, theta_l, agg, scratch);
  if (nc > 0.9 * n) {   // stalled: treat every connection as strong
    nc = aggregate(H, w, 0.0, agg, scratch);
    theta_used = 0.0;
  }
  if (nc > 0.9 * n || nc < 1) {                        // cannot coarsen further
    o.stop = true;
    return;
  }
  if (!H.visit.empty()) renumber_aggregates(agg, nc, o.visit_c);
  }
  o.nc = nc;
  o.t_agg = ms_since(tA);

  // members by aggregate
  std::vector<int>&mem_ptr = o.mem_ptr, &mem = o.mem;
  mem_ptr.assign((size_t)nc + 1, 0);
  mem.resize(n);
  for (int i = 0; i < n; ++i) mem_ptr[agg[i] + 1]++;
  for (int a = 0; a < nc; ++a) mem_ptr[a + 1] += mem_ptr[a];
  {
    std::vector<int> fill(mem_ptr.begin(), mem_ptr.end() - 1);
    for (int i = 0; i < n; ++i) mem[fill[agg[i]]++] = i;
  }
  HostLevel& Hc = o.Hc;
  std::vector<int>&order = o.order, &tgt = o.tgt, &cptr = o.cptr;
  SaHost& sa = o.sa;
  if (scratch) {   // everything of the previous level has been uploaded (stream synchronised by the caller)
    scratch->rewind();
    sa.ap_a.arena = sa.ap_b.arena = sa.ap_tgt.arena = scratch;
    sa.val_src.arena = sa.val_tgt.arena = sa.ap_col.arena = sa.ap_row.arena = scratch;
    sa.rap_a.arena = sa.rap_b.arena = sa.rap_tgt.arena = scratch;
  }
  bool smooth = cfg.smooth;
  if (smooth) {
    // product lists of at most 16 per fine slot (C4 needs 9-11, chain-like graphs 4-6) and a coarse
    // operator with no more blocks than the fine one (C4 0.06x, chains 0.5x); beyond that the smoothed coarse operator is
    // nearly dense (5 % random long-range closures on C4: 33 products per slot, 4.4x the blocks,
    // 2x slower than the tentative hierarchy): this level keeps the tentative prolongator
    const long long budget = std::min<long long>(1500000000LL, std::max<long long>(16LL * H.nslot, 2000000LL));
    try {
      smooth = sa_symbolic(H, agg, nc, mem_ptr, mem, budget, cfg.lists_on_device, sa);
    } catch (const std::bad_alloc&) {
      smooth = false;
    }
    // (not on a nearly dense level -- more than 48 slots per row: a refused attempt there costs more than the whole level, C4r's
    // 897-row level 11 ms -- and a sample of the rows says first whether the filter would drop most connections at all.  The
    // SMALL levels of a chain-like hierarchy do matter: with the attempt limited to levels of >= 4 096 rows C4's first solve from
    // the dead-reckoned start took 91 instead of 56 iterations.)
    bool try_filtered = !smooth && theta_used > 0.0 && cfg.filtered_smoothing && n >= 512 && (long long)H.nslot <= 48LL * n;
    if (try_filtered) {
      long long seen = 0, weak = 0;
      for (int i = 0; i < n; i += 16) {
        const double di = w[H.rowptr[i]];
        for (int k = H.rowptr[i] + 1; k < H.rowptr[i + 1]; ++k) {
          const int j = H.col[k];
          if (j == i) continue;
          ++seen;
          weak += !(w[k] > 0.0 && w[k] * w[k] >= cfg.theta_filter * cfg.theta_filter * di * w[H.rowptr[j]]);
        }
      }
      try_filtered = 5 * weak >= 2 * seen;   // (the full count below decides at one half)
    }
    if (try_filtered) {
      // Second attempt: FILTERED smoothing.  The whole operator made P too dense -- many connections per row, e.g. the 10^6
      // closures DCS has switched off at a dead-reckoned start (BASELINE.md's literal workload), each of them negligible next to
      // the odometry chain --: smooth T with the operator of the strong connections only (the aggregation's own criterion), keep
      // every connection in the Galerkin products.  Refused like the first when the coarse operator would still be too dense
      // (random long-range closures at full weight are strong: C4r, C5 keep the tentative transfer).
      sa = SaHost();
      if (scratch) {
        scratch->rewind();   // (the refused attempt's lists)
        sa.ap_a.arena = sa.ap_b.arena = sa.ap_tgt.arena = scratch;
        sa.val_src.arena = sa.val_tgt.arena = sa.ap_col.arena = sa.ap_row.arena = scratch;
        sa.rap_a.arena = sa.rap_b.arena = sa.rap_tgt.arena = scratch;
      }
      std::vector<unsigned char> strong((size_t)H.nslot, 1);
      long long nweak = 0;
      {
        std::vector<long long> part((size_t)std::max(1, std::min(HostPool::get().size(), n / 2048)) + 1, 0);
        host_parallel_for(n, 2048, [&](int lo, int hi, int task) {
          long long cnt = 0;
          for (int i = lo; i < hi; ++i) {
            const double di = w[H.rowptr[i]];
            for (int k = H.rowptr[i] + 1; k < H.rowptr[i + 1]; ++k) {
              const int j = H.col[k];
              if (j == i) continue;   // (level 0: the block-less slot of an edge to a fixed vertex aliases the diagonal)
              const double th = cfg.theta_filter * cfg.theta_filter * di * w[H.rowptr[j]];   // (aggregate()'s criterion, at the filter's threshold)
              if (!(w[k] > 0.0 && w[k] * w[k] >= th)) {
                strong[k] = 0;
                ++cnt;
              }
            }
          }
          part[task] = cnt;
        });
        for (long long v : part) nweak += v;
      }
      // (worth the second pass over the patterns only when the filter drops most of the connections -- at the dead-reckoned
      // start 89 % --: with few of them negligible P's pattern is nearly the refused one.  C5 / C4r, whose random closures are at
      // full weight, paid 20 % of their set-up for an attempt that could not succeed.)
      if (2 * nweak >= (long long)H.nslot - n) {
        try {
          smooth = sa_symbolic(H, agg, nc, mem_ptr, mem, budget, cfg.lists_on_device, sa, strong.data());
        } catch (const std::bad_alloc&) {
          smooth = false;
        }
      }
      if (smooth) {
        sa.filtered = true;
        sa.strong = std::move(strong);
      }
    }
    if (smooth) Hc = std::move(sa.Hc);
    else sa = SaHost();
    o.t_sort = ms_since(tA) - o.t_agg;
  }
  if (!smooth) {
    // coarse slots: unique (agg[row], agg[col]); diagonal first in each row, then ascending columns; the fine slots
    // behind each coarse slot in ascending order.  Coarse row a collects the slots of its member rows (members and
    // slots ascending), stably sorted by the coarse column code: rows in parallel on the host pool (the two global
    // counting sorts this replaces were 0.25 s of sequential work on C5).
    const int ns = H.nslot;
    std::vector<int> cbase((size_t)nc + 1, 0), cs_cnt((size_t)nc + 1, 0);
    for (int a = 0; a < nc; ++a) {
      int len = 0;
      for (int t = mem_ptr[a]; t < mem_ptr[a + 1]; ++t) len += H.rowptr[mem[t] + 1] - H.rowptr[mem[t]];
      cbase[a + 1] = cbase[a] + len;
    }
    order.resize(ns);
    tgt.resize(ns);   // contribution -> coarse slot; cptr: coarse slot -> contribution range
    std::vector<uint32_t> code_sorted((size_t)ns);
    host_parallel_for(nc, 64, [&](int a0, int a1, int) {
      std::vector<std::pair<uint32_t, int>> items;
      for (int a = a0; a < a1; ++a) {
        items.clear();
        for (int t = mem_ptr[a]; t < mem_ptr[a + 1]; ++t) {
          const int i = mem[t];
          for (int k = H.rowptr[i]; k < H.rowptr[i + 1]; ++k) {
            const int cc = agg[H.col[k]];
            items.emplace_back(cc == a ? 0u : (uint32_t)cc + 1u, k);   // diagonal sorts first
          }
        }
        std::stable_sort(items.begin(), items.end(), [](const std::pair<uint32_t, int>& x, const std::pair<uint32_t, int>& y) { return x.first < y.first; });
        int distinct = 0;
        uint32_t prev = 0xFFFFFFFFu;
        for (size_t q = 0; q < items.size(); ++q) {
          order[(size_t)cbase[a] + q] = items[q].second;
          code_sorted[(size_t)cbase[a] + q] = items[q].first;
          tgt[(size_t)cbase[a] + q] = distinct - (items[q].first == prev ? 1 : 0);   // local slot number, made global below
          if (items[q].first != prev) {
            prev = items[q].first;
            ++distinct;
          }
        }
        cs_cnt[a + 1] = distinct;
      }
    });
    o.t_sort = ms_since(tA) - o.t_agg;
    for (int a = 0; a < nc; ++a) cs_cnt[a + 1] += cs_cnt[a];
    Hc.n = nc;
    Hc.nslot = cs_cnt[nc];
    Hc.row.resize(Hc.nslot);
    Hc.col.resize(Hc.nslot);
    Hc.rowptr.assign(cs_cnt.begin(), cs_cnt.end());
    cptr.assign((size_t)Hc.nslot + 1, 0);
    std::atomic<bool> diag_ok{true};
    host_parallel_for(nc, 64, [&](int a0, int a1, int) {
      for (int a = a0; a < a1; ++a) {
        const int s0 = cs_cnt[a];
        for (int t = cbase[a]; t < cbase[a + 1]; ++t) {
          const bool first = t == cbase[a] || code_sorted[t] != code_sorted[t - 1];
          tgt[t] += s0;
          if (first) {
            const int cs = tgt[t];
            cptr[cs] = t;
            Hc.row[cs] = a;
            Hc.col[cs] = code_sorted[t] == 0 ? a : (int)(code_sorted[t] - 1);
          }
        }
        if (cbase[a + 1] == cbase[a] || code_sorted[cbase[a]] != 0) diag_ok = false;
      }
    });
    cptr[Hc.nslot] = ns;
    if (!diag_ok) {
      o.err = "amg_create: internal error (coarse diagonal slot missing)";
      return;
    }
    o.grp_g = make_groups(cptr);
  }
  o.smooth = smooth;
  o.grp_c = make_groups(Hc.rowptr);
  o.t_all = ms_since(tA);
}


}  // namespace sgo
