// oracle/sgo_oracle.cpp -- plain C++17 fp64 CPU restatement of the g2o Gauss-Newton
// SE(2) pose-graph path that sparse-gslam drives.
//
// TEST INFRASTRUCTURE ONLY.  The product (sparse_gslam_amd/csrc, libsgo.so) never links,
// loads or calls this file.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it, as the checker / as the reported CPU baseline.
//
// PARITY UNPINNED: the arithmetic lives in g2o (ros-gbp/libg2o-release, noetic branch =
// g2o 2020.5.29), an un-vendored third-party dependency that is absent from
// /root/reference and unbuildable in this image (no Eigen, no network); the reference has
// no tests or golden vectors for this path.  This file restates the
// published g2o algorithm, anchored on the reference call sites cited per function, and is
// cross-checked against the independent numpy/scipy restatement (oracle/np_oracle.py) and
// hand-derived known answers (tests/test_oracle_kat.py).
//
// The linear solver here is an up-looking sparse LDL^T with a minimum-degree ordering: the
// same class of method as the reference's g2o::LinearSolverEigen (Eigen SimplicialLDLT +
// AMD; src/sparse_gslam/src/graphs.cpp:19) -- exact sparse direct factorisation in fp64.
//
// Arrays: poses[V][3] (x,y,theta), fixed[V] (u8), ei/ej[E] (i32 vertex ids),
// meas[E][3], info[E][6] (upper triangle o11,o12,o13,o22,o23,o33), phi[E] (<0: no kernel).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <utility>
#include <vector>

namespace {

constexpr double kPi = 3.14159265358979323846;

// g2o::normalize_theta (g2o/stuff/misc.h); used via SE2 at slc.cpp:217-219,275.
inline double normalize_theta(double t) {
  if (t >= -kPi && t < kPi) return t;
  double m = std::floor(t / (2 * kPi));
  t = t - m * 2 * kPi;
  if (t >= kPi) t -= 2 * kPi;
  if (t < -kPi) t += 2 * kPi;
  return t;
}

struct SE2 {
  double x, y, th;
};
// g2o::SE2::operator* : t = ta + R(tha) tb ; th = normalize(tha + thb)
inline SE2 mul(const SE2& a, const SE2& b) {
  double c = std::cos(a.th), s = std::sin(a.th);
  return {a.x + c * b.x - s * b.y, a.y + s * b.x + c * b.y, normalize_theta(a.th + b.th)};
}
// g2o::SE2::inverse : th' = normalize(-th); t' = R(th') * (-t)
inline SE2 inv(const SE2& a) {
  double th = normalize_theta(-a.th);
  double c = std::cos(th), s = std::sin(th);
  return {c * (-a.x) - s * (-a.y), s * (-a.x) + c * (-a.y), th};
}

// EdgeSE2::computeError (bound at slc.cpp:214-217,273-276; log_runner.cpp:183-184):
//   e = toVector( Z^-1 * (Xi^-1 * Xj) )
inline void edge_error(const double* xi, const double* xj, const double* z, double* e) {
  SE2 Xi{xi[0], xi[1], xi[2]}, Xj{xj[0], xj[1], xj[2]}, Z{z[0], z[1], z[2]};
  SE2 d = mul(inv(Z), mul(inv(Xi), Xj));
  e[0] = d.x;
  e[1] = d.y;
  e[2] = d.th;
}

// EdgeSE2::linearizeOplus (analytic).  A = de/dXi, B = de/dXj, row-major 3x3.
inline void edge_jacobians(const double* xi, const double* xj, const double* z, double* A,
                           double* B) {
  SE2 Zi = inv(SE2{z[0], z[1], z[2]});
  double si = std::sin(xi[2]), ci = std::cos(xi[2]);
  double dx = xj[0] - xi[0], dy = xj[1] - xi[1];
  double a[9] = {-ci, -si, -si * dx + ci * dy, si, -ci, -ci * dx - si * dy, 0, 0, -1};
  double b[9] = {ci, si, 0, -si, ci, 0, 0, 0, 1};
  double cz = std::cos(Zi.th), sz = std::sin(Zi.th);
  double R[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double sa = 0, sb = 0;
      for (int k = 0; k < 3; ++k) {
        sa += R[3 * r + k] * a[3 * k + c];
        sb += R[3 * r + k] * b[3 * k + c];
      }
      A[3 * r + c] = sa;
      B[3 * r + c] = sb;
    }
}

inline void info_full(const double* u, double* O) {
  O[0] = u[0]; O[1] = u[1]; O[2] = u[2];
  O[3] = u[1]; O[4] = u[3]; O[5] = u[4];
  O[6] = u[2]; O[7] = u[4]; O[8] = u[5];
}

// RobustKernelDCS::robustify (kernel attached at slc.cpp:41,57,283).  phi < 0: no kernel.
inline void dcs_rho(double e2, double phi, double* rho0, double* rho1) {
  if (phi < 0) {
    *rho0 = e2;
    *rho1 = 1.0;
    return;
  }
  double scale = (2.0 * phi) / (phi + e2);
  if (scale >= 1.0) {
    *rho0 = e2;
    *rho1 = 1.0;
  } else {
    *rho0 = scale * e2 * scale;
    *rho1 = scale * scale;
  }
}

struct Graph {
  int V, E;
  const uint8_t* fixed;
  const int32_t *ei, *ej;
  const double *meas, *info, *phi;
};

// ---------------------------------------------------------------- block system (upper)
// BlockSolver<BlockSolverTraits<3,3>>::buildStructure (graphs.cpp:18): index map = non-fixed
// vertices in ascending id; Hpp upper block-triangular.
struct System {
  int n = 0;                       // number of free vertices
  std::vector<int> hidx;           // vertex -> hessian index or -1
  std::vector<int> free_id;        // hessian index -> vertex
  // unique upper off-diagonal block pairs (r < c), sorted by (c, r)
  std::vector<std::pair<int, int>> pairs;
  std::vector<int> edge_blk;       // edge -> index into pairs, or -1
  std::vector<uint8_t> edge_tr;    // edge stores (j,i): accumulate transposed
  std::vector<double> diag;        // n x 9
  std::vector<double> off;         // pairs x 9  (block (r,c), r<c)
  std::vector<double> b;           // 3n
};

void build_structure(const Graph& g, System& S) {
  S.hidx.assign(g.V, -1);
  S.free_id.clear();
  for (int v = 0; v < g.V; ++v)
    if (!g.fixed[v]) {
      S.hidx[v] = (int)S.free_id.size();
      S.free_id.push_back(v);
    }
  S.n = (int)S.free_id.size();
  std::vector<std::pair<int, int>> raw;
  raw.reserve(g.E);
  for (int e = 0; e < g.E; ++e) {
    int hi = S.hidx[g.ei[e]], hj = S.hidx[g.ej[e]];
    if (hi >= 0 && hj >= 0 && hi != hj) raw.emplace_back(std::max(hi, hj), std::min(hi, hj));
  }
  std::sort(raw.begin(), raw.end());
  raw.erase(std::unique(raw.begin(), raw.end()), raw.end());
  S.pairs.resize(raw.size());
  for (size_t k = 0; k < raw.size(); ++k) S.pairs[k] = {raw[k].second, raw[k].first};  // (r,c)
  S.edge_blk.assign(g.E, -1);
  S.edge_tr.assign(g.E, 0);
  for (int e = 0; e < g.E; ++e) {
    int hi = S.hidx[g.ei[e]], hj = S.hidx[g.ej[e]];
    if (hi >= 0 && hj >= 0 && hi != hj) {
      std::pair<int, int> key(std::max(hi, hj), std::min(hi, hj));
      size_t k = std::lower_bound(raw.begin(), raw.end(), key) - raw.begin();
      S.edge_blk[e] = (int)k;
      S.edge_tr[e] = hi > hj;
    }
  }
  S.diag.assign((size_t)S.n * 9, 0.0);
  S.off.assign(S.pairs.size() * 9, 0.0);
  S.b.assign((size_t)S.n * 3, 0.0);
}

inline void mat3_AtB(const double* A, const double* B, double* C) {  // C = A^T B
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      C[3 * r + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
}
inline void mat3_mul(const double* A, const double* B, double* C) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}

// BlockSolver::buildSystem = for each active edge: linearizeOplus + constructQuadraticForm.
// Returns plain / robust chi2 of the linearisation point (computeActiveErrors).
void build_system(const Graph& g, const double* poses, System& S, double* chi2, double* rchi2) {
  std::fill(S.diag.begin(), S.diag.end(), 0.0);
  std::fill(S.off.begin(), S.off.end(), 0.0);
  std::fill(S.b.begin(), S.b.end(), 0.0);
  double c2 = 0, rc2 = 0;
  for (int e = 0; e < g.E; ++e) {
    const double* xi = poses + 3 * (size_t)g.ei[e];
    const double* xj = poses + 3 * (size_t)g.ej[e];
    const double* z = g.meas + 3 * (size_t)e;
    double err[3], A[9], B[9], O[9];
    edge_error(xi, xj, z, err);
    info_full(g.info + 6 * (size_t)e, O);
    double Oe[3];
    for (int r = 0; r < 3; ++r) Oe[r] = O[3 * r] * err[0] + O[3 * r + 1] * err[1] + O[3 * r + 2] * err[2];
    double e2 = err[0] * Oe[0] + err[1] * Oe[1] + err[2] * Oe[2];
    double rho0, rho1;
    dcs_rho(e2, g.phi[e], &rho0, &rho1);
    c2 += e2;
    rc2 += rho0;
    int hi = S.hidx[g.ei[e]], hj = S.hidx[g.ej[e]];
    if (hi < 0 && hj < 0) continue;
    edge_jacobians(xi, xj, z, A, B);
    double Ow[9], OA[9], OB[9], T[9];
    for (int k = 0; k < 9; ++k) Ow[k] = rho1 * O[k];   // robustInformation: rho[1]*Omega
    double Owe[3] = {rho1 * Oe[0], rho1 * Oe[1], rho1 * Oe[2]};
    mat3_mul(Ow, A, OA);
    mat3_mul(Ow, B, OB);
    if (hi >= 0) {
      mat3_AtB(A, OA, T);
      for (int k = 0; k < 9; ++k) S.diag[(size_t)hi * 9 + k] += T[k];
      for (int r = 0; r < 3; ++r)
        S.b[(size_t)hi * 3 + r] -= A[r] * Owe[0] + A[3 + r] * Owe[1] + A[6 + r] * Owe[2];
    }
    if (hj >= 0) {
      mat3_AtB(B, OB, T);
      for (int k = 0; k < 9; ++k) S.diag[(size_t)hj * 9 + k] += T[k];
      for (int r = 0; r < 3; ++r)
        S.b[(size_t)hj * 3 + r] -= B[r] * Owe[0] + B[3 + r] * Owe[1] + B[6 + r] * Owe[2];
    }
    if (S.edge_blk[e] >= 0) {
      mat3_AtB(A, OB, T);  // H_ij = A^T Ow B  (block row hi, col hj)
      double* dst = &S.off[(size_t)S.edge_blk[e] * 9];
      if (!S.edge_tr[e])
        for (int k = 0; k < 9; ++k) dst[k] += T[k];
      else
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) dst[3 * r + c] += T[3 * c + r];
    } else if (hi >= 0 && hi == hj) {  // self edge (not produced by the reference; kept total)
      mat3_AtB(A, OB, T);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) S.diag[(size_t)hi * 9 + 3 * r + c] += T[3 * r + c] + T[3 * c + r];
    }
  }
  *chi2 = c2;
  *rchi2 = rc2;
}

// ---------------------------------------------------------------- sparse direct LDL^T
// Minimum-degree ordering on the block graph (explicit elimination graph).
std::vector<int> min_degree_order(int n, const std::vector<std::pair<int, int>>& pairs) {
  std::vector<std::vector<int>> adj(n);
  for (auto& p : pairs) {
    adj[p.first].push_back(p.second);
    adj[p.second].push_back(p.first);
  }
  for (auto& a : adj) std::sort(a.begin(), a.end());
  using QE = std::pair<int, int>;
  std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq;
  for (int v = 0; v < n; ++v) pq.emplace((int)adj[v].size(), v);
  std::vector<uint8_t> done(n, 0);
  std::vector<int> perm;
  perm.reserve(n);
  std::vector<int> tmp;
  while (!pq.empty()) {
    auto [d, v] = pq.top();
    pq.pop();
    if (done[v] || d != (int)adj[v].size()) continue;
    done[v] = 1;
    perm.push_back(v);
    std::vector<int> N;
    N.swap(adj[v]);
    for (int u : N) {
      tmp.clear();
      tmp.reserve(adj[u].size() + N.size());
      std::set_union(adj[u].begin(), adj[u].end(), N.begin(), N.end(), std::back_inserter(tmp));
      tmp.erase(std::remove_if(tmp.begin(), tmp.end(), [&](int w) { return w == u || w == v; }),
                tmp.end());
      adj[u].swap(tmp);
      pq.emplace((int)adj[u].size(), u);
    }
  }
  return perm;
}

struct LDL {
  int n = 0;                      // scalar dimension
  std::vector<int> perm_blk;      // new block position -> old block
  std::vector<int> inv_blk;       // old block -> new block position
  std::vector<int> Ap, Ai;        // permuted upper-triangular CSC pattern
  std::vector<int> Asrc;          // for each Ai entry: source (>=0: off index*9+k ; <0: diag)
  std::vector<int> Lp, Parent, Lnz, Li;
  std::vector<double> Ax, Lx, D, Y;
  std::vector<int> Pattern, Flag;
  bool analysed = false;
};

// LinearSolverEigen::solve pattern: symbolic analysis once per optimize() call, numeric
// factorisation every iteration.
void ldl_analyse(const System& S, LDL& F) {
  int nb = S.n;
  F.perm_blk = min_degree_order(nb, S.pairs);
  F.inv_blk.assign(nb, 0);
  for (int k = 0; k < nb; ++k) F.inv_blk[F.perm_blk[k]] = k;
  int n = 3 * nb;
  F.n = n;
  // column lists of the permuted upper triangle
  struct Ent { int row; int src; };
  std::vector<std::vector<Ent>> cols(n);
  for (int blk = 0; blk < nb; ++blk) {
    int pb = F.inv_blk[blk];
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r <= c; ++r) cols[3 * pb + c].push_back({3 * pb + r, -(blk * 9 + 3 * r + c) - 1});
  }
  for (size_t k = 0; k < S.pairs.size(); ++k) {
    int pr = F.inv_blk[S.pairs[k].first], pc = F.inv_blk[S.pairs[k].second];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        // block (first, second) element (r,c); place into upper triangle of permuted matrix
        int R = 3 * pr + r, C = 3 * pc + c;
        int src = (int)(k * 9 + 3 * r + c);
        if (R < C) cols[C].push_back({R, src});
        else cols[R].push_back({C, src});
      }
  }
  F.Ap.assign(n + 1, 0);
  F.Ai.clear();
  F.Asrc.clear();
  for (int c = 0; c < n; ++c) {
    std::sort(cols[c].begin(), cols[c].end(), [](const Ent& a, const Ent& b) { return a.row < b.row; });
    for (auto& e : cols[c]) {
      F.Ai.push_back(e.row);
      F.Asrc.push_back(e.src);
    }
    F.Ap[c + 1] = (int)F.Ai.size();
  }
  F.Ax.assign(F.Ai.size(), 0.0);
  // symbolic (elimination tree + column counts)
  F.Parent.assign(n, -1);
  F.Lnz.assign(n, 0);
  F.Flag.assign(n, 0);
  F.Lp.assign(n + 1, 0);
  for (int k = 0; k < n; ++k) {
    F.Parent[k] = -1;
    F.Flag[k] = k;
    F.Lnz[k] = 0;
    for (int p = F.Ap[k]; p < F.Ap[k + 1]; ++p) {
      int i = F.Ai[p];
      if (i < k)
        for (; F.Flag[i] != k; i = F.Parent[i]) {
          if (F.Parent[i] == -1) F.Parent[i] = k;
          F.Lnz[i]++;
          F.Flag[i] = k;
        }
    }
  }
  for (int k = 0; k < n; ++k) F.Lp[k + 1] = F.Lp[k] + F.Lnz[k];
  F.Li.assign(F.Lp[n], 0);
  F.Lx.assign(F.Lp[n], 0.0);
  F.D.assign(n, 0.0);
  F.Y.assign(n, 0.0);
  F.Pattern.assign(n, 0);
  F.analysed = true;
}

// returns true on success; false mirrors Eigen::NumericalIssue (zero / non-finite pivot).
bool ldl_factor(const System& S, LDL& F) {
  int n = F.n;
  for (size_t p = 0; p < F.Ai.size(); ++p) {
    int s = F.Asrc[p];
    F.Ax[p] = s >= 0 ? S.off[s] : S.diag[-(s + 1)];
  }
  for (int k = 0; k < n; ++k) {
    F.Y[k] = 0.0;
    int top = n;
    F.Flag[k] = k;
    F.Lnz[k] = 0;
    for (int p = F.Ap[k]; p < F.Ap[k + 1]; ++p) {
      int i = F.Ai[p];
      if (i <= k) {
        F.Y[i] += F.Ax[p];
        int len = 0;
        for (; F.Flag[i] != k; i = F.Parent[i]) {
          F.Pattern[len++] = i;
          F.Flag[i] = k;
        }
        while (len > 0) F.Pattern[--top] = F.Pattern[--len];
      }
    }
    F.D[k] = F.Y[k];
    F.Y[k] = 0.0;
    for (; top < n; ++top) {
      int i = F.Pattern[top];
      double yi = F.Y[i];
      F.Y[i] = 0.0;
      int p2 = F.Lp[i] + F.Lnz[i];
      for (int p = F.Lp[i]; p < p2; ++p) F.Y[F.Li[p]] -= F.Lx[p] * yi;
      double lki = yi / F.D[i];
      F.D[k] -= lki * yi;
      F.Li[p2] = k;
      F.Lx[p2] = lki;
      F.Lnz[i]++;
    }
    if (F.D[k] == 0.0 || !std::isfinite(F.D[k])) return false;
  }
  return true;
}

void ldl_solve(const System& S, const LDL& F, std::vector<double>& x) {
  int n = F.n, nb = S.n;
  std::vector<double> y(n);
  for (int blk = 0; blk < nb; ++blk)
    for (int c = 0; c < 3; ++c) y[3 * F.inv_blk[blk] + c] = S.b[3 * (size_t)blk + c];
  for (int j = 0; j < n; ++j) {
    int p2 = F.Lp[j] + F.Lnz[j];
    for (int p = F.Lp[j]; p < p2; ++p) y[F.Li[p]] -= F.Lx[p] * y[j];
  }
  for (int j = 0; j < n; ++j) y[j] /= F.D[j];
  for (int j = n - 1; j >= 0; --j) {
    int p2 = F.Lp[j] + F.Lnz[j];
    for (int p = F.Lp[j]; p < p2; ++p) y[j] -= F.Lx[p] * y[F.Li[p]];
  }
  x.resize(n);
  for (int blk = 0; blk < nb; ++blk)
    for (int c = 0; c < 3; ++c) x[3 * (size_t)blk + c] = y[3 * F.inv_blk[blk] + c];
}

// ---------------------------------------------------------------- block-Jacobi PCG (CPU)
inline bool inv3(const double* M, double* R) {
  double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
  double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
  if (det == 0.0 || !std::isfinite(det)) return false;
  double id = 1.0 / det;
  R[0] = c00 * id; R[1] = (M[2] * M[7] - M[1] * M[8]) * id; R[2] = (M[1] * M[5] - M[2] * M[4]) * id;
  R[3] = c01 * id; R[4] = (M[0] * M[8] - M[2] * M[6]) * id; R[5] = (M[2] * M[3] - M[0] * M[5]) * id;
  R[6] = c02 * id; R[7] = (M[1] * M[6] - M[0] * M[7]) * id; R[8] = (M[0] * M[4] - M[1] * M[3]) * id;
  return true;
}

void sym_spmv(const System& S, const double* p, double* q) {
  int n = S.n;
  for (int i = 0; i < n; ++i) {
    const double* D = &S.diag[(size_t)i * 9];
    for (int r = 0; r < 3; ++r) q[3 * i + r] = D[3 * r] * p[3 * i] + D[3 * r + 1] * p[3 * i + 1] + D[3 * r + 2] * p[3 * i + 2];
  }
  for (size_t k = 0; k < S.pairs.size(); ++k) {
    int r0 = S.pairs[k].first, c0 = S.pairs[k].second;
    const double* B = &S.off[k * 9];
    for (int r = 0; r < 3; ++r) {
      q[3 * r0 + r] += B[3 * r] * p[3 * c0] + B[3 * r + 1] * p[3 * c0 + 1] + B[3 * r + 2] * p[3 * c0 + 2];
      q[3 * c0 + r] += B[r] * p[3 * r0] + B[3 + r] * p[3 * r0 + 1] + B[6 + r] * p[3 * r0 + 2];
    }
  }
}

int pcg_solve(const System& S, std::vector<double>& x, double tol, int maxit) {
  int n = S.n, N = 3 * n;
  std::vector<double> Dinv((size_t)n * 9), r(S.b), z(N), p(N), q(N);
  for (int i = 0; i < n; ++i)
    if (!inv3(&S.diag[(size_t)i * 9], &Dinv[(size_t)i * 9])) return -1;
  x.assign(N, 0.0);
  auto prec = [&]() {
    for (int i = 0; i < n; ++i) {
      const double* M = &Dinv[(size_t)i * 9];
      for (int k = 0; k < 3; ++k) z[3 * i + k] = M[3 * k] * r[3 * i] + M[3 * k + 1] * r[3 * i + 1] + M[3 * k + 2] * r[3 * i + 2];
    }
  };
  auto dot = [&](const std::vector<double>& a, const std::vector<double>& b2) {
    double s = 0;
    for (int i = 0; i < N; ++i) s += a[i] * b2[i];
    return s;
  };
  prec();
  p = z;
  double rz = dot(r, z), bn = std::sqrt(dot(S.b, S.b));
  int it = 0;
  while (it < maxit && std::sqrt(dot(r, r)) > tol * bn) {
    sym_spmv(S, p.data(), q.data());
    double pq = dot(p, q);
    if (!(pq > 0)) return -1;
    double alpha = rz / pq;
    for (int i = 0; i < N; ++i) {
      x[i] += alpha * p[i];
      r[i] -= alpha * q[i];
    }
    prec();
    double rzn = dot(r, z), beta = rzn / rz;
    for (int i = 0; i < N; ++i) p[i] = z[i] + beta * p[i];
    rz = rzn;
    ++it;
  }
  return it;
}

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------- threaded PCG (CPU-baseline variants B / C)
// The same block-Jacobi PCG on a row-wise copy of H (both triangles, CSR of 3x3 blocks) so that the product
// is a gather per row and the rows can be split over OpenMP threads ("CPU PCG ...
// plus an OpenMP all-cores variant").  Runs at most maxit iterations; returns the iterations done.
struct RowCsr {
  std::vector<int> ptr, col;
  std::vector<double> blk;   // 9 per entry, diagonal first
};
void to_rows(const System& S, RowCsr& R) {
  const int n = S.n;
  R.ptr.assign((size_t)n + 1, 0);
  for (int i = 0; i < n; ++i) R.ptr[i + 1] = 1;
  for (auto& pr : S.pairs) {
    R.ptr[pr.first + 1]++;
    R.ptr[pr.second + 1]++;
  }
  for (int i = 0; i < n; ++i) R.ptr[i + 1] += R.ptr[i];
  R.col.resize(R.ptr[n]);
  R.blk.resize((size_t)R.ptr[n] * 9);
  std::vector<int> fill(R.ptr.begin(), R.ptr.end() - 1);
  for (int i = 0; i < n; ++i) {
    R.col[fill[i]] = i;
    std::copy(&S.diag[(size_t)i * 9], &S.diag[(size_t)i * 9] + 9, &R.blk[(size_t)fill[i] * 9]);
    fill[i]++;
  }
  for (size_t k = 0; k < S.pairs.size(); ++k) {
    const int r0 = S.pairs[k].first, c0 = S.pairs[k].second;
    const double* B = &S.off[k * 9];
    R.col[fill[r0]] = c0;
    std::copy(B, B + 9, &R.blk[(size_t)fill[r0] * 9]);
    fill[r0]++;
    R.col[fill[c0]] = r0;
    double* T = &R.blk[(size_t)fill[c0] * 9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) T[3 * r + c] = B[3 * c + r];
    fill[c0]++;
  }
}
int pcg_rows(const System& S, const RowCsr& R, std::vector<double>& x, double tol, int maxit, int threads, bool* converged) {
  const int n = S.n, N = 3 * n;
  std::vector<double> Dinv((size_t)n * 9), r(S.b), z(N), p(N), q(N);
  for (int i = 0; i < n; ++i)
    if (!inv3(&S.diag[(size_t)i * 9], &Dinv[(size_t)i * 9])) return -1;
  x.assign(N, 0.0);
  double rz = 0, bb = 0;
#pragma omp parallel for num_threads(threads) reduction(+ : rz, bb) schedule(static)
  for (int i = 0; i < n; ++i) {
    const double* M = &Dinv[(size_t)i * 9];
    for (int k = 0; k < 3; ++k) {
      z[3 * i + k] = M[3 * k] * r[3 * i] + M[3 * k + 1] * r[3 * i + 1] + M[3 * k + 2] * r[3 * i + 2];
      p[3 * i + k] = z[3 * i + k];
      rz += r[3 * i + k] * z[3 * i + k];
      bb += r[3 * i + k] * r[3 * i + k];
    }
  }
  double rr = bb;
  int it = 0;
  *converged = false;
  while (it < maxit) {
    if (rr <= tol * tol * bb) {
      *converged = true;
      break;
    }
    double pq = 0;
#pragma omp parallel for num_threads(threads) reduction(+ : pq) schedule(static)
    for (int i = 0; i < n; ++i) {
      double a0 = 0, a1 = 0, a2 = 0;
      for (int k = R.ptr[i]; k < R.ptr[i + 1]; ++k) {
        const double* B = &R.blk[(size_t)k * 9];
        const double* v = &p[3 * (size_t)R.col[k]];
        a0 += B[0] * v[0] + B[1] * v[1] + B[2] * v[2];
        a1 += B[3] * v[0] + B[4] * v[1] + B[5] * v[2];
        a2 += B[6] * v[0] + B[7] * v[1] + B[8] * v[2];
      }
      q[3 * i] = a0; q[3 * i + 1] = a1; q[3 * i + 2] = a2;
      pq += p[3 * i] * a0 + p[3 * i + 1] * a1 + p[3 * i + 2] * a2;
    }
    if (!(pq > 0)) return -1;
    const double alpha = rz / pq;
    double rzn = 0;
    rr = 0;
#pragma omp parallel for num_threads(threads) reduction(+ : rzn, rr) schedule(static)
    for (int i = 0; i < n; ++i) {
      const double* M = &Dinv[(size_t)i * 9];
      for (int k = 0; k < 3; ++k) {
        x[3 * i + k] += alpha * p[3 * i + k];
        r[3 * i + k] -= alpha * q[3 * i + k];
      }
      for (int k = 0; k < 3; ++k) {
        z[3 * i + k] = M[3 * k] * r[3 * i] + M[3 * k + 1] * r[3 * i + 1] + M[3 * k + 2] * r[3 * i + 2];
        rzn += r[3 * i + k] * z[3 * i + k];
        rr += r[3 * i + k] * r[3 * i + k];
      }
    }
    const double beta = rzn / rz;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int i = 0; i < N; ++i) p[i] = z[i] + beta * p[i];
    rz = rzn;
    ++it;
  }
  return it;
}

}  // namespace

extern "C" {

double sgo_oracle_normalize_theta(double t) { return normalize_theta(t); }

void sgo_oracle_se2_mul(const double* a, const double* b, double* out) {
  SE2 r = mul(SE2{a[0], a[1], a[2]}, SE2{b[0], b[1], b[2]});
  out[0] = r.x; out[1] = r.y; out[2] = r.th;
}
void sgo_oracle_se2_inv(const double* a, double* out) {
  SE2 r = inv(SE2{a[0], a[1], a[2]});
  out[0] = r.x; out[1] = r.y; out[2] = r.th;
}

// per-edge quantities for n independent (xi, xj, z, info, phi) tuples
void sgo_oracle_edges(int n, const double* xi, const double* xj, const double* z, const double* info,
                      const double* phi, double* e, double* A, double* B, double* e2, double* rho0,
                      double* rho1) {
  for (int k = 0; k < n; ++k) {
    double err[3], O[9];
    edge_error(xi + 3 * k, xj + 3 * k, z + 3 * k, err);
    if (e) std::memcpy(e + 3 * k, err, sizeof err);
    if (A && B) edge_jacobians(xi + 3 * k, xj + 3 * k, z + 3 * k, A + 9 * k, B + 9 * k);
    if (info) {
      info_full(info + 6 * k, O);
      double c = 0;
      for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) c += err[r] * O[3 * r + q] * err[q];
      if (e2) e2[k] = c;
      if (phi && rho0 && rho1) dcs_rho(c, phi[k], rho0 + k, rho1 + k);
    }
  }
}

// computeActiveErrors + activeChi2 / activeRobustChi2
void sgo_oracle_chi2(int V, const double* poses, int E, const int32_t* ei, const int32_t* ej,
                     const double* meas, const double* info, const double* phi, double* chi2,
                     double* rchi2) {
  (void)V;
  double c2 = 0, rc2 = 0;
  for (int e = 0; e < E; ++e) {
    double err[3], O[9];
    edge_error(poses + 3 * (size_t)ei[e], poses + 3 * (size_t)ej[e], meas + 3 * (size_t)e, err);
    info_full(info + 6 * (size_t)e, O);
    double c = 0;
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) c += err[r] * O[3 * r + q] * err[q];
    double r0, r1;
    dcs_rho(c, phi[e], &r0, &r1);
    c2 += c;
    rc2 += r0;
  }
  *chi2 = c2;
  *rchi2 = rc2;
}

// buildSystem at `poses`: b (3n), block diagonal (n x 9, row-major 3x3), n = #free vertices.
// Returns n, or -1 if the out buffers are too small (cap_n).
int sgo_oracle_linearize(int V, const double* poses, const uint8_t* fixed, int E, const int32_t* ei,
                         const int32_t* ej, const double* meas, const double* info,
                         const double* phi, int cap_n, double* b, double* diag, double* chi2,
                         double* rchi2) {
  Graph g{V, E, fixed, ei, ej, meas, info, phi};
  System S;
  build_structure(g, S);
  if (S.n > cap_n) return -1;
  build_system(g, poses, S, chi2, rchi2);
  std::memcpy(b, S.b.data(), sizeof(double) * S.b.size());
  std::memcpy(diag, S.diag.data(), sizeof(double) * S.diag.size());
  return S.n;
}

// y = H x at the linearisation point `poses` (for SpMV parity).  x, y are 3n.
int sgo_oracle_hessian_apply(int V, const double* poses, const uint8_t* fixed, int E,
                             const int32_t* ei, const int32_t* ej, const double* meas,
                             const double* info, const double* phi, const double* x, double* y) {
  Graph g{V, E, fixed, ei, ej, meas, info, phi};
  System S;
  build_structure(g, S);
  double c, rc;
  build_system(g, poses, S, &c, &rc);
  sym_spmv(S, x, y);
  return S.n;
}

// CPU-baseline timing of ONE Gauss-Newton iteration with the block-Jacobi PCG on `threads` OpenMP threads
// (1 = variant B of SURVEY.md section 8(d), all cores = variant C): linearise + assemble at `poses`, then at
// most pcg_maxit PCG iterations.  out[0] = seconds of linearise + assemble (single thread, as the GN loop
// above), out[1] = seconds of the PCG iterations, out[2] = iterations run, out[3] = 1 if pcg_tol was reached.
int sgo_oracle_pcg_timing(int V, const double* poses, const uint8_t* fixed, int E, const int32_t* ei, const int32_t* ej,
                          const double* meas, const double* info, const double* phi, double pcg_tol, int pcg_maxit,
                          int threads, double* out) {
  Graph g{V, E, fixed, ei, ej, meas, info, phi};
  System S;
  build_structure(g, S);
  if (S.n == 0) return -1;
  double c2, rc2;
  double t0 = now_s();
  build_system(g, poses, S, &c2, &rc2);
  out[0] = now_s() - t0;
  RowCsr R;
  to_rows(S, R);
  std::vector<double> x;
  bool conv = false;
  t0 = now_s();
  const int k = pcg_rows(S, R, x, pcg_tol, pcg_maxit, threads < 1 ? 1 : threads, &conv);
  out[1] = now_s() - t0;
  out[2] = (double)k;
  out[3] = conv ? 1.0 : 0.0;
  return k;
}

// SparseOptimizer::optimize(iters) with OptimizationAlgorithmGaussNewton (slc.cpp:286-288,
// log_runner.cpp:203-204).  solver: 0 = sparse direct LDL^T, 1 = block-Jacobi PCG.
// chi2/rchi2 have iters+1 slots: [k] = value at the start of iteration k, [done] = final.
// Returns iterations done (0 = solver failed in the first iteration, -1 = nothing to optimise).
int sgo_oracle_gn(int V, double* poses, const uint8_t* fixed, int E, const int32_t* ei,
                  const int32_t* ej, const double* meas, const double* info, const double* phi,
                  int iters, int solver, double pcg_tol, int pcg_maxit, double* chi2, double* rchi2,
                  int* pcg_iters, double* seconds) {
  Graph g{V, E, fixed, ei, ej, meas, info, phi};
  System S;
  build_structure(g, S);
  if (S.n == 0) return -1;
  LDL F;
  std::vector<double> x;
  int done = 0;
  for (int it = 0; it < iters; ++it) {
    double t0 = now_s();
    double c2, rc2;
    build_system(g, poses, S, &c2, &rc2);
    if (chi2) chi2[it] = c2;
    if (rchi2) rchi2[it] = rc2;
    bool ok = true;
    int k = 0;
    if (solver == 0) {
      if (!F.analysed) ldl_analyse(S, F);
      ok = ldl_factor(S, F);
      if (ok) ldl_solve(S, F, x);
    } else {
      k = pcg_solve(S, x, pcg_tol, pcg_maxit);
      ok = k >= 0;
    }
    if (ok)
      for (double v : x)
        if (!std::isfinite(v)) { ok = false; break; }
    if (!ok) break;
    if (pcg_iters) pcg_iters[it] = k;
    // SparseOptimizer::update -> VertexSE2::oplusImpl: additive t, wrapped theta
    for (int h = 0; h < S.n; ++h) {
      double* P = poses + 3 * (size_t)S.free_id[h];
      P[0] += x[3 * (size_t)h];
      P[1] += x[3 * (size_t)h + 1];
      P[2] = normalize_theta(P[2] + x[3 * (size_t)h + 2]);
    }
    if (seconds) seconds[it] = now_s() - t0;
    ++done;
  }
  double c2, rc2;
  sgo_oracle_chi2(V, poses, E, ei, ej, meas, info, phi, &c2, &rc2);
  if (chi2) chi2[done] = c2;
  if (rchi2) rchi2[done] = rc2;
  return done;
}

}  // extern "C"
