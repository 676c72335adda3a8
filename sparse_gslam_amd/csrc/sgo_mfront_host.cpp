// sgo_mfront_host.cpp -- symbolic analysis of the multifrontal path (sgo_mfront.h): nested dissection, front structures,
// assembly lists, level schedule.  Host only (no HIP call): also behind sgo_mfront_plan for the CPU tests.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <numeric>

#include "sgo_mfront.h"

namespace sgo {
namespace {

// Minimum vertex cover of a bipartite graph (left vertices 0 .. nl, right 0 .. nr, adjacency of the left side) by Koenig's
// theorem: maximum matching by augmenting paths, Z = everything reachable from the unmatched left vertices by alternating
// paths, cover = (left \ Z) u (right n Z).
struct Bipartite {
  int nl = 0, nr = 0;
  std::vector<std::vector<int>> adj;
  std::vector<int> match_l, match_r;
  std::vector<char> seen;
  bool augment(int u) {
    for (int v : adj[u]) {
      if (seen[v]) continue;
      seen[v] = 1;
      if (match_r[v] < 0 || augment(match_r[v])) {
        match_l[u] = v;
        match_r[v] = u;
        return true;
      }
    }
    return false;
  }
  void cover(std::vector<char>& left_in, std::vector<char>& right_in) {
    match_l.assign(nl, -1);
    match_r.assign(nr, -1);
    // greedy start, then augmenting paths
    for (int u = 0; u < nl; ++u)
      for (int v : adj[u])
        if (match_r[v] < 0) {
          match_l[u] = v;
          match_r[v] = u;
          break;
        }
    for (int u = 0; u < nl; ++u)
      if (match_l[u] < 0) {
        seen.assign(nr, 0);
        augment(u);
      }
    std::vector<char> zl(nl, 0), zr(nr, 0);
    std::vector<int> stack;
    for (int u = 0; u < nl; ++u)
      if (match_l[u] < 0) {
        zl[u] = 1;
        stack.push_back(u);
      }
    while (!stack.empty()) {
      const int u = stack.back();
      stack.pop_back();
      for (int v : adj[u]) {
        if (zr[v] || match_l[u] == v) continue;
        zr[v] = 1;
        const int w = match_r[v];
        if (w >= 0 && !zl[w]) {
          zl[w] = 1;
          stack.push_back(w);
        }
      }
    }
    left_in.assign(nl, 0);
    right_in.assign(nr, 0);
    for (int u = 0; u < nl; ++u) left_in[u] = !zl[u];
    for (int v = 0; v < nr; ++v) right_in[v] = zr[v];
  }
};

struct Node {
  std::vector<int> own;   // rows (numbers in the chosen order)
  int kid[2] = {-1, -1};
};

struct Dissector {
  const std::vector<int>& adjp;
  const std::vector<int>& adj;
  int leaf;
  std::vector<Node> nodes;
  std::vector<int> mark;      // 0 outside, 1 part A, 2 part B of the set being split
  std::vector<int> local;     // index of a cut vertex in its side of the bipartite graph

  Dissector(const std::vector<int>& ap, const std::vector<int>& a, int n, int leaf_)
      : adjp(ap), adj(a), leaf(leaf_), mark((size_t)std::max(n, 1), 0), local((size_t)std::max(n, 1), -1) {}

  // separator of the split of the sorted set `v` at position `half`; returns its size
  size_t split(const std::vector<int>& v, size_t half, std::vector<int>* sep) {
    for (size_t i = 0; i < v.size(); ++i) mark[v[i]] = i < half ? 1 : 2;
    Bipartite G;
    std::vector<int> lv, rv;
    for (size_t i = 0; i < half; ++i) {
      const int a = v[i];
      int la = -1;
      for (int k = adjp[a]; k < adjp[a + 1]; ++k) {
        const int b = adj[k];
        if (mark[b] != 2) continue;
        if (la < 0) {
          la = (int)lv.size();
          lv.push_back(a);
          G.adj.emplace_back();
        }
        if (local[b] < 0) {
          local[b] = (int)rv.size();
          rv.push_back(b);
        }
        G.adj[la].push_back(local[b]);
      }
    }
    G.nl = (int)lv.size();
    G.nr = (int)rv.size();
    std::vector<char> lin, rin;
    G.cover(lin, rin);
    sep->clear();
    for (int u = 0; u < G.nl; ++u)
      if (lin[u]) sep->push_back(lv[u]);
    for (int w = 0; w < G.nr; ++w)
      if (rin[w]) sep->push_back(rv[w]);
    for (int b : rv) local[b] = -1;
    for (int a : v) mark[a] = 0;
    return sep->size();
  }

  int dissect(std::vector<int>& v) {   // v sorted ascending
    if ((int)v.size() <= leaf) {
      Node nd;
      nd.own = v;
      nodes.push_back(std::move(nd));
      return (int)nodes.size() - 1;
    }
    // three candidate cuts; the smallest separator wins (ties: the most balanced)
    std::vector<int> best, cand;
    size_t best_half = v.size() / 2;
    const double fr[3] = {0.5, 0.42, 0.58};
    for (int q = 0; q < 3; ++q) {
      const size_t half = std::min(v.size() - 1, std::max<size_t>(1, (size_t)(fr[q] * (double)v.size())));
      split(v, half, &cand);
      if (q == 0 || cand.size() < best.size()) {
        best.swap(cand);
        best_half = half;
      }
      if (best.size() <= 3) break;   // small enough: the other cuts cannot save much
    }
    std::sort(best.begin(), best.end());
    std::vector<int> A, B;
    for (size_t i = 0; i < v.size(); ++i) {
      if (std::binary_search(best.begin(), best.end(), v[i])) continue;
      (i < best_half ? A : B).push_back(v[i]);
    }
    std::vector<int>().swap(v);
    int kids[2] = {-1, -1}, nk = 0;
    if (!A.empty()) kids[nk++] = dissect(A);
    if (!B.empty()) kids[nk++] = dissect(B);
    // (an empty separator -- the halves are not connected -- leaves a front without pivots that only merges its children's
    // update matrices; with one child it is not needed at all)
    if (best.empty() && nk == 1) return kids[0];
    Node nd;
    nd.own = best;
    nd.kid[0] = kids[0];
    nd.kid[1] = kids[1];
    nodes.push_back(std::move(nd));
    return (int)nodes.size() - 1;
  }
};

}  // namespace

bool mfront_analyze(int V, int n, const int* free_id, const double* poses, int E, const int* ei, const int* ej, const MfLimits& lim,
                    MfPlan* plan, std::string* why) {
  auto no = [&](const std::string& w) {
    if (why) *why = w;
    return false;
  };
  if (n <= 0) return no("no free pose");
  if (n > lim.max_rows) return no(std::to_string(n) + " free poses > " + std::to_string(lim.max_rows));
  // Cheap refusal before any analysis: graphs with many closures per pose have large separators whatever the order (a
  // Manhattan world with four edges per pose: 580-row fronts, 0.24 Gflop on the critical path), and their analysis costs
  // tens of milliseconds of the set-up
  {
    std::vector<char> is_free((size_t)V, 0);
    for (int h = 0; h < n; ++h) is_free[free_id[h]] = 1;
    long long inner = 0;
    for (int e = 0; e < E; ++e) inner += is_free[ei[e]] && is_free[ej[e]];
    if ((double)inner > lim.max_degree * (double)n) {
      char buf[128];
      std::snprintf(buf, sizeof buf, "%.2f edges per free pose > %.2f", (double)inner / (double)n, lim.max_degree);
      return no(buf);
    }
  }
  MfPlan best;
  bool have = false;
  std::string last_why;
  for (int kind = 0; kind < ((n <= lim.both_orders_rows) ? 2 : 1); ++kind) {
    if (lim.only_kind >= 0 && kind != lim.only_kind) continue;
    // ---- row order
    std::vector<int> row_vertex(n), row_of((size_t)V, -1);
    if (kind == 0) {
      double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
      for (int h = 0; h < n; ++h) {
        const double* q = poses + 3 * (size_t)free_id[h];
        for (int d = 0; d < 2; ++d)
          if (std::isfinite(q[d])) {
            lo[d] = std::min(lo[d], q[d]);
            hi[d] = std::max(hi[d], q[d]);
          }
      }
      const double ext = std::max(hi[0] - lo[0], hi[1] - lo[1]);
      if (!(ext > 0.0) || !std::isfinite(ext)) continue;   // no geometry to order by
      const double scale = 65535.0 / ext;
      std::vector<uint64_t> key(n);
      for (int h = 0; h < n; ++h) {
        const double* q = poses + 3 * (size_t)free_id[h];
        uint32_t d = 0;
        if (std::isfinite(q[0]) && std::isfinite(q[1])) d = hilbert_index((uint32_t)((q[0] - lo[0]) * scale), (uint32_t)((q[1] - lo[1]) * scale), 16);
        key[h] = ((uint64_t)d << 32) | (uint32_t)h;
      }
      std::sort(key.begin(), key.end());
      for (int r = 0; r < n; ++r) row_vertex[r] = free_id[(int)(key[r] & 0xffffffffu)];
    } else {
      for (int r = 0; r < n; ++r) row_vertex[r] = free_id[r];
    }
    for (int r = 0; r < n; ++r) row_of[row_vertex[r]] = r;
    // ---- adjacency of the free rows (unique neighbours)
    std::vector<int> adjp((size_t)n + 1, 0), adj;
    {
      std::vector<uint64_t> pairs;
      pairs.reserve(2 * (size_t)E);
      for (int e = 0; e < E; ++e) {
        const int a = row_of[ei[e]], b = row_of[ej[e]];
        if (a < 0 || b < 0 || a == b) continue;
        pairs.push_back(((uint64_t)(uint32_t)a << 32) | (uint32_t)b);
        pairs.push_back(((uint64_t)(uint32_t)b << 32) | (uint32_t)a);
      }
      std::sort(pairs.begin(), pairs.end());
      pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
      adj.resize(pairs.size());
      for (size_t k = 0; k < pairs.size(); ++k) {
        ++adjp[(size_t)(pairs[k] >> 32) + 1];
        adj[k] = (int)(pairs[k] & 0xffffffffu);
      }
      for (int r = 0; r < n; ++r) adjp[r + 1] += adjp[r];
    }
    // ---- nested dissection
    Dissector D(adjp, adj, n, lim.leaf);
    std::vector<int> all(n);
    std::iota(all.begin(), all.end(), 0);
    if (n > 4096) {
      // Larger graphs: the three top separators first (three linear passes).  The fronts of the root's children hold their own
      // separator plus, as boundary, most of the root's: when that estimate -- two of some ten levels -- already takes half the budget the full analysis
      // -- tens of milliseconds of every set-up of a graph that then takes the multigrid path anyway -- is not run.
      std::vector<int> s0, sa, sb, A, B;
      D.split(all, (size_t)n / 2, &s0);
      std::sort(s0.begin(), s0.end());
      for (int r = 0; r < n; ++r) {
        if (std::binary_search(s0.begin(), s0.end(), r)) continue;
        (r < n / 2 ? A : B).push_back(r);
      }
      if (A.size() > 1) D.split(A, A.size() / 2, &sa);
      if (B.size() > 1) D.split(B, B.size() / 2, &sb);
      const double o = 3.0 * (double)std::max(sa.size(), sb.size()), mm = o + 3.0 * (double)s0.size(), r0 = 3.0 * (double)s0.size();
      const double est = r0 * r0 * r0 / 3.0 + o * mm * mm - mm * o * o + o * o * o / 3.0;
      if (est > 0.5 * lim.max_crit_flops) {
        char buf[200];
        std::snprintf(buf, sizeof buf, "about %.0f Mflop on the critical path of the elimination tree already in its two top levels "
                      "(separators of %zu and %zu poses): more than half of the budget of %.0f", 1e-6 * est, s0.size(), std::max(sa.size(), sb.size()), 1e-6 * lim.max_crit_flops);
        last_why = buf;
        continue;
      }
    }
    const int root = D.dissect(all);
    (void)root;
    std::vector<Node>& nodes = D.nodes;
    const int nf = (int)nodes.size();
    // ---- elimination positions: post-order = order of creation
    MfPlan P;
    P.n = n;
    P.order_kind = kind;
    P.elim_vertex.resize(n);
    std::vector<int> elim_of_row(n, -1);
    P.fronts.resize(nf);
    {
      int c = 0;
      for (int f = 0; f < nf; ++f) {
        MfFront& F = P.fronts[f];
        F.e0 = c;
        F.own = (int)nodes[f].own.size();
        for (int r : nodes[f].own) {
          elim_of_row[r] = c;
          P.elim_vertex[c] = row_vertex[r];
          ++c;
        }
        F.kid[0] = nodes[f].kid[0];
        F.kid[1] = nodes[f].kid[1];
        for (int k = 0; k < 2; ++k)
          if (F.kid[k] >= 0) P.fronts[F.kid[k]].parent = f;
      }
      if (c != n) return no("internal: nested dissection lost a pose");
    }
    // ---- boundaries, bottom-up
    std::vector<std::vector<int>> bnd(nf);
    bool fits = true;
    for (int f = 0; f < nf && fits; ++f) {
      MfFront& F = P.fronts[f];
      std::vector<int>& b = bnd[f];
      const int last = F.e0 + F.own;
      for (int r : nodes[f].own)
        for (int k = adjp[r]; k < adjp[r + 1]; ++k) {
          const int p = elim_of_row[adj[k]];
          if (p >= last) b.push_back(p);
        }
      for (int k = 0; k < 2; ++k)
        if (F.kid[k] >= 0)
          for (int p : bnd[F.kid[k]])
            if (p >= last) b.push_back(p);
      std::sort(b.begin(), b.end());
      b.erase(std::unique(b.begin(), b.end()), b.end());
      F.nb = (int)b.size();
      F.height = 0;
      for (int k = 0; k < 2; ++k)
        if (F.kid[k] >= 0) F.height = std::max(F.height, P.fronts[F.kid[k]].height + 1);
      if (3 * (F.own + F.nb) + 1 > kMfMaxDim + 1) {
        fits = false;
        last_why = "a front has " + std::to_string(3 * (F.own + F.nb)) + " rows (" + std::to_string(F.own) + " own + " + std::to_string(F.nb) +
                   " boundary poses) > " + std::to_string(kMfMaxDim);
      }
    }
    if (!fits) continue;
    // ---- figures of merit, level lists
    P.height = 0;
    for (const MfFront& F : P.fronts) P.height = std::max(P.height, F.height);
    std::vector<double> fl(nf);
    std::vector<double> crit((size_t)P.height + 1, 0.0);
    std::vector<int> critp((size_t)P.height + 1, 0);
    long long off = 0;
    for (int f = 0; f < nf; ++f) {
      MfFront& F = P.fronts[f];
      const double s = 3.0 * F.own, m = 3.0 * (F.own + F.nb) + 1.0;
      // sum over the own pivots k of (m - k)^2 multiply-adds / 2 x 2 flops
      fl[f] = s * m * m - m * s * (s - 1.0) + (s - 1.0) * s * (2.0 * s - 1.0) / 6.0;
      P.flops += fl[f];
      crit[F.height] = std::max(crit[F.height], fl[f]);
      critp[F.height] = std::max(critp[F.height], (3 * F.own + kMfPanel - 1) / kMfPanel);
      P.max_dim = std::max(P.max_dim, 3 * (F.own + F.nb));
      P.max_own = std::max(P.max_own, F.own);
      P.max_bnd = std::max(P.max_bnd, F.nb);
      F.ld = (3 * (F.own + F.nb) + 1 + 1) & ~1;   // even: columns start 16-byte aligned
      F.off = off;
      off += (long long)F.ld * 3 * (F.own + F.nb);
      F.bnd_off = (int)P.bnd.size();
      P.bnd.insert(P.bnd.end(), bnd[f].begin(), bnd[f].end());
    }
    P.arena_doubles = off + 64;
    for (int h = 0; h <= P.height; ++h) {
      P.crit_flops += crit[h];
      P.crit_panels += critp[h];
    }
    if (P.crit_flops > lim.max_crit_flops) {
      char buf[160];
      std::snprintf(buf, sizeof buf, "%.0f Mflop on the critical path of the elimination tree (largest front %d rows) > %.0f", 1e-6 * P.crit_flops,
                    P.max_dim, 1e-6 * lim.max_crit_flops);
      last_why = buf;
      continue;
    }
    if (P.arena_doubles * 8 > lim.max_arena_bytes) {
      last_why = "frontal matrices of " + std::to_string(P.arena_doubles * 8 >> 20) + " MiB";
      continue;
    }
    if (have && best.crit_flops <= P.crit_flops) continue;
    // ---- extend-add maps
    for (int f = 0; f < nf; ++f) {
      MfFront& F = P.fronts[f];
      for (int k = 0; k < 2; ++k) {
        if (F.kid[k] < 0) continue;
        F.map_off[k] = (int)P.cmap.size();
        const std::vector<int>& cb = bnd[F.kid[k]];
        const std::vector<int>& pb = bnd[f];
        for (int p : cb) {
          int l;
          if (p < F.e0 + F.own) {
            l = p - F.e0;
          } else {
            const auto it = std::lower_bound(pb.begin(), pb.end(), p);
            if (it == pb.end() || *it != p) return no("internal: a child's boundary pose is missing from its parent's front");
            l = F.own + (int)(it - pb.begin());
          }
          if (l < 0) return no("internal: a child's boundary pose precedes its parent's front");
          P.cmap.push_back(l);
        }
      }
    }
    // ---- assembly lists: every edge goes to the front of its first-eliminated endpoint
    {
      std::vector<int> front_of_elim(n);
      for (int f = 0; f < nf; ++f)
        for (int q = 0; q < P.fronts[f].own; ++q) front_of_elim[P.fronts[f].e0 + q] = f;
      struct Item {
        int front, li, lj, val;
      };
      std::vector<Item> items;
      items.reserve(3 * (size_t)E);
      auto local_of = [&](int f, int p) {
        const MfFront& F = P.fronts[f];
        if (p < F.e0 + F.own) return p - F.e0;
        const std::vector<int>& pb = bnd[f];
        return F.own + (int)(std::lower_bound(pb.begin(), pb.end(), p) - pb.begin());
      };
      for (int e = 0; e < E; ++e) {
        const int ra = row_of[ei[e]], rb = row_of[ej[e]];
        const int pa = ra >= 0 ? elim_of_row[ra] : -1, pb = rb >= 0 ? elim_of_row[rb] : -1;
        if (pa < 0 && pb < 0) continue;
        const int home = front_of_elim[(pa >= 0 && (pb < 0 || pa < pb)) ? pa : pb];
        const int la = pa >= 0 ? local_of(home, pa) : -1, lb = pb >= 0 ? local_of(home, pb) : -1;
        if (la >= 0) items.push_back({home, la, la, (e << 2) | 0});
        if (lb >= 0) items.push_back({home, lb, lb, (e << 2) | 1});
        if (la >= 0 && lb >= 0) {
          if (la > lb) items.push_back({home, la, lb, (e << 2) | 2});
          else items.push_back({home, lb, la, (e << 2) | 3});
        }
      }
      std::sort(items.begin(), items.end(), [](const Item& x, const Item& y) {
        if (x.front != y.front) return x.front < y.front;
        if (x.li != y.li) return x.li < y.li;
        if (x.lj != y.lj) return x.lj < y.lj;
        return x.val < y.val;
      });
      P.contrib.resize(items.size());
      size_t i = 0;
      for (int f = 0; f < nf; ++f) {
        P.fronts[f].tgt0 = (int)P.targets.size();
        while (i < items.size() && items[i].front == f) {
          MfTarget T;
          T.li = items[i].li;
          T.lj = items[i].lj;
          T.c0 = (int)i;
          while (i < items.size() && items[i].front == f && items[i].li == T.li && items[i].lj == T.lj) {
            P.contrib[i] = items[i].val;
            ++i;
          }
          T.c1 = (int)i;
          P.targets.push_back(T);
        }
        P.fronts[f].tgt1 = (int)P.targets.size();
      }
    }
    // ---- levels, the largest fronts first (they start first)
    P.level_ptr.assign((size_t)P.height + 2, 0);
    for (const MfFront& F : P.fronts) ++P.level_ptr[(size_t)F.height + 1];
    for (int h = 0; h <= P.height; ++h) P.level_ptr[h + 1] += P.level_ptr[h];
    P.level_front.resize(nf);
    {
      std::vector<int> fill(P.level_ptr.begin(), P.level_ptr.end() - 1);
      for (int f = 0; f < nf; ++f) P.level_front[fill[P.fronts[f].height]++] = f;
      for (int h = 0; h <= P.height; ++h)
        std::sort(P.level_front.begin() + P.level_ptr[h], P.level_front.begin() + P.level_ptr[h + 1], [&](int a, int b) {
          if (fl[a] != fl[b]) return fl[a] > fl[b];
          return a < b;
        });
    }
    best = std::move(P);
    have = true;
  }
  if (!have) return no(last_why.empty() ? "no usable row order" : last_why);
  *plan = std::move(best);
  return true;
}

}  // namespace sgo
