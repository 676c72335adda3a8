// sgo_mfront.hip -- numeric phase of the multifrontal path (sgo_mfront.h): per Gauss-Newton iteration
//   k_mf_edges    EdgeSE2::computeError + linearizeOplus + robust weighting, the 6x6 element of every edge (D_ii, D_jj, H_ij,
//                 b_i, b_j) and chi2 / robust chi2 (g2o: OptimizationAlgorithmGaussNewton::solve -> computeActiveErrors,
//                 linearizeOplus, constructQuadraticForm; src/sparse_gslam/src/graphs.cpp:9-37 chooses the algorithm)
//   k_mf_factor   one launch per level of the elimination tree, one workgroup per front: assembly, extend-add, partial
//                 Cholesky (LinearSolverEigen / CHOLMOD's numeric factorisation, graphs.cpp:19)
//   k_mf_solve    one launch per level, top-down: backward substitution
//   k_mf_update   SparseOptimizer::update -> VertexSE2::oplusImpl
// A front's matrix is column-major with leading dimension ld, lower triangle, rows 0 .. m-1 = its poses' scalar rows (own
// first, then boundary, both in elimination order) and row m = the right-hand side: the Cholesky factor of [[H, b], [b^T, .]]
// carries L^-1 b in its last row, and the Schur complement's last row is the children's contribution to the parent's
// right-hand side.  Bounds: assembly and extend-add are L2 traffic (a front is read and written once), the panel loop is a
// chain of barrier-separated steps (16-column panels: update on the matrix cores -> 16x16 Cholesky + inverse in one wave's
// registers -> panel solve on the matrix cores), the Schur complement is fp64 MFMA work on operands that sit in L2.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "sgo_device.h"
#include "sgo_internal.h"
#include "sgo_mfront.h"

namespace sgo {
namespace {

constexpr int kMfNW = kMfThreads / 64;
constexpr int kElemStride = 28;   // kMfElem padded to 16-byte multiples

struct MfFrontDev {
  int e0, own3, m, ld;
  long long off;
  int nb, bnd_off;
  int kid[2];
  int map_off[2];
  int tgt0, tgt1;
};

struct MfDev {
  int n = 0, E = 0, nfront = 0;
  const MfFrontDev* fronts = nullptr;
  const int* level_front = nullptr;
  const int* bnd = nullptr;
  const int* cmap = nullptr;
  const MfTarget* targets = nullptr;
  const int* contrib = nullptr;
  const int* elim_vertex = nullptr;
  double* arena = nullptr;
  double* elem = nullptr;      // [E][kElemStride]
  double* x = nullptr;         // [3 n] by elimination position
  double* partials = nullptr;  // [2][kMaxPartials]
  long long* dbg = nullptr;    // diagnostic runs (SGO_MFRONT_DEBUG): [nfront][8] s_memtime cycles of the factor kernel's phases
  int* flags = nullptr;        // [0] fail (1 not positive definite, 2 non-finite update)  [1] iteration of the failure
                               // [2] a back-substitution produced a non-finite value  [3] updates applied  [4] ticket of k_mf_edges
};

typedef double mf_d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double mf_readlane(double v, int l) {   // l wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// acc += sum_{k < K} L[ra][k] L[rb][k] for the 16 x 16 tile whose operand rows this lane addresses through pa / pb (= the
// front's matrix + (lane >> 4) * ld + row: rows beyond the front are CLAMPED by the caller, not masked -- a tile element
// depends on its own row and column only, the caller does not store the others).  The loads of eight (then four) MFMA steps
// are issued together: the operands sit in L2, and a step that waits for its own two loads costs a round trip (measured: 900
// cycles per step before, the whole K loop of a panel update was latency).
__device__ __forceinline__ void mf_tile_dot(const double* __restrict__ pa, const double* __restrict__ pb, size_t ld, int K, int lk, mf_d4& acc) {
  int kk = 0;
  for (; kk + 32 <= K; kk += 32) {
    double av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      av[u] = pa[(size_t)(kk + 4 * u) * ld];
      bv[u] = pb[(size_t)(kk + 4 * u) * ld];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
  }
  for (; kk + 16 <= K; kk += 16) {
    double av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      av[u] = pa[(size_t)(kk + 4 * u) * ld];
      bv[u] = pb[(size_t)(kk + 4 * u) * ld];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
  }
  if (kk < K) {   // K = 3 x own poses: up to four steps more, the last one partly beyond K
    double av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool kv = kk + 4 * u + lk < K;
      av[u] = kv ? pa[(size_t)(kk + 4 * u) * ld] : 0.0;
      bv[u] = kv ? pb[(size_t)(kk + 4 * u) * ld] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
  }
}

// 1 / sqrt(x) by the hardware estimate and two Newton steps (sqrt and the IEEE division are ~60 dependent instructions on the
// critical path of every pivot of the 16 x 16 factorisation)
__device__ __forceinline__ double mf_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x * r, r, 1.5);
  r = r * fma(-0.5 * x * r, r, 1.5);
  return r;
}

// ---------------------------------------------------------------------------- k_mf_edges
__global__ __launch_bounds__(kBlock) void k_mf_edges(MfDev M, EdgeListDev el, const double* __restrict__ poses, int it, int chi2_only,
                                                     double* __restrict__ hist, DirectResult* __restrict__ res) {
  if (blockIdx.x == 0 && threadIdx.x == 0) res->stamp[2 * it] = (unsigned long long)wall_clock64();
  if (M.flags[0]) return;
  const size_t ns = (size_t)el.E;
  double acc[2] = {0.0, 0.0};
  for (int e = blockIdx.x * kBlock + threadIdx.x; e < M.E; e += gridDim.x * kBlock) {
    const int vi = el.vi[e], vj = el.vj[e];
    const double xi = poses[3 * (size_t)vi], yi = poses[3 * (size_t)vi + 1], ti = poses[3 * (size_t)vi + 2];
    const double xj = poses[3 * (size_t)vj], yj = poses[3 * (size_t)vj + 1], tj = poses[3 * (size_t)vj + 2];
    const double zx = el.zinv[e], zy = el.zinv[ns + e], zt = el.zinv[2 * ns + e];
    double sz, cz;
    sincos(zt, &sz, &cz);
    double er[3];
    edge_error(xi, yi, ti, xj, yj, tj, zx, zy, zt, sz, cz, er);
    const double o00 = el.info[e], o01 = el.info[ns + e], o02 = el.info[2 * ns + e];
    const double o11 = el.info[3 * ns + e], o12 = el.info[4 * ns + e], o22 = el.info[5 * ns + e];
    double oe[3] = {o00 * er[0] + o01 * er[1] + o02 * er[2], o01 * er[0] + o11 * er[1] + o12 * er[2], o02 * er[0] + o12 * er[1] + o22 * er[2]};
    const double e2 = er[0] * oe[0] + er[1] * oe[1] + er[2] * oe[2];
    double r0, w;
    dcs(e2, el.phi[e], &r0, &w);
    acc[0] += e2;
    acc[1] += r0;
    if (chi2_only) continue;
    // EdgeSE2::linearizeOplus: A = d e / d x_i, B = d e / d x_j (rows: error components), with Rz of the inverse measurement
    double si, ci;
    sincos(ti, &si, &ci);
    const double ddx = xj - xi, ddy = yj - yi;
    const double a02 = -si * ddx + ci * ddy, a12 = -ci * ddx - si * ddy;
    const double A[3][3] = {{cz * (-ci) - sz * si, cz * (-si) - sz * (-ci), cz * a02 - sz * a12},
                            {sz * (-ci) + cz * si, sz * (-si) + cz * (-ci), sz * a02 + cz * a12},
                            {0.0, 0.0, -1.0}};
    const double B[3][3] = {{cz * ci - sz * (-si), cz * si - sz * ci, 0.0}, {sz * ci + cz * (-si), sz * si + cz * ci, 0.0}, {0.0, 0.0, 1.0}};
    const double W[3][3] = {{w * o00, w * o01, w * o02}, {w * o01, w * o11, w * o12}, {w * o02, w * o12, w * o22}};
    double WA[3][3], WB[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        WA[a][b] = W[a][0] * A[0][b] + W[a][1] * A[1][b] + W[a][2] * A[2][b];
        WB[a][b] = W[a][0] * B[0][b] + W[a][1] * B[1][b] + W[a][2] * B[2][b];
      }
    double* out = M.elem + (size_t)kElemStride * e;
    int q = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = a; b < 3; ++b) out[q++] = A[0][a] * WA[0][b] + A[1][a] * WA[1][b] + A[2][a] * WA[2][b];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = a; b < 3; ++b) out[q++] = B[0][a] * WB[0][b] + B[1][a] * WB[1][b] + B[2][a] * WB[2][b];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) out[q++] = A[0][a] * WB[0][b] + A[1][a] * WB[1][b] + A[2][a] * WB[2][b];
#pragma unroll
    for (int a = 0; a < 3; ++a) out[q++] = -w * (A[0][a] * oe[0] + A[1][a] * oe[1] + A[2][a] * oe[2]);
#pragma unroll
    for (int a = 0; a < 3; ++a) out[q++] = -w * (B[0][a] * oe[0] + B[1][a] * oe[1] + B[2][a] * oe[2]);
  }
  block_sum_store<2>(acc, M.partials, kMaxPartials);
  // the last workgroup to arrive sums the partials in a fixed order (an integer ticket: no floating-point atomics)
  __shared__ int s_last;
  __threadfence();
  if (threadIdx.x == 0) s_last = atomicAdd(M.flags + 4, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  const double c0 = block_reduce_parts(M.partials, (int)gridDim.x);
  const double c1 = block_reduce_parts(M.partials + kMaxPartials, (int)gridDim.x);
  if (threadIdx.x == 0) {
    hist[2 * it] = c0;
    hist[2 * it + 1] = c1;
    M.flags[4] = 0;
  }
}

// ---------------------------------------------------------------------------- k_mf_factor
__global__ __launch_bounds__(kMfThreads) void k_mf_factor(MfDev M, int lvl0, int it, int stamp_slot, DirectResult* __restrict__ res) {
  extern __shared__ double Pn[];            // panel: column c at Pn + c * ldp, rows relative to k0
  __shared__ double Yt[kMfPanel * kMfPanel];   // Yt[t * 16 + c] = (L11^-1)[c][t]
  __shared__ int s_fail;
  if (stamp_slot >= 0 && blockIdx.x == 0 && threadIdx.x == 0) res->stamp[stamp_slot] = (unsigned long long)wall_clock64();
  if (M.flags[0]) return;
  const MfFrontDev F = M.fronts[M.level_front[lvl0 + blockIdx.x]];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int m = F.m, s3 = F.own3, ld = F.ld;
  if (m == 0) return;
  double* __restrict__ A = M.arena + F.off;
  if (tid == 0) s_fail = 0;
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = M.dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
  auto lapse = [&](int k) {
    if (M.dbg && tid == 0) {
      const long long now = (long long)__builtin_amdgcn_s_memtime();
      ph[k] += now - tprev;
      tprev = now;
    }
  };
  // ---- A. clear
  {
    const long long tot = (long long)ld * m / 2;   // ld is even
    double2* A2 = reinterpret_cast<double2*>(A);
    for (long long i = tid; i < tot; i += kMfThreads) A2[i] = make_double2(0.0, 0.0);
  }
  __syncthreads();
  lapse(0);
  // ---- B. the edges whose first-eliminated endpoint is a pose of this front: one thread per 3x3 target, contributions in edge order
  for (int t = F.tgt0 + tid; t < F.tgt1; t += kMfThreads) {
    const MfTarget T = M.targets[t];
    if (T.li == T.lj) {
      double D[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
      int cv4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) cv4[q] = M.contrib[min(T.c0 + q, T.c1 - 1)];
      for (int c = T.c0; c < T.c1; ++c) {
        const int v = (c - T.c0 < 4) ? cv4[c - T.c0] : M.contrib[c];
        const double* el = M.elem + (size_t)kElemStride * (v >> 2);
        const int side = v & 1;
#pragma unroll
        for (int q = 0; q < 6; ++q) D[q] += el[6 * side + q];
#pragma unroll
        for (int q = 0; q < 3; ++q) b[q] += el[21 + 3 * side + q];
      }
      const size_t c0 = 3 * (size_t)T.li;
      A[c0 * ld + c0] = D[0];
      A[c0 * ld + c0 + 1] = D[1];
      A[c0 * ld + c0 + 2] = D[2];
      A[(c0 + 1) * ld + c0 + 1] = D[3];
      A[(c0 + 1) * ld + c0 + 2] = D[4];
      A[(c0 + 2) * ld + c0 + 2] = D[5];
#pragma unroll
      for (int q = 0; q < 3; ++q) A[(c0 + q) * ld + m] = b[q];
    } else {
      double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int c = T.c0; c < T.c1; ++c) {
        const int v = M.contrib[c];
        const double* el = M.elem + (size_t)kElemStride * (v >> 2) + 12;
        if ((v & 3) == 2) {
#pragma unroll
          for (int q = 0; q < 9; ++q) H[q] += el[q];
        } else {
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) H[3 * a + b] += el[3 * b + a];
        }
      }
      const size_t r0 = 3 * (size_t)T.li, c0 = 3 * (size_t)T.lj;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) A[(c0 + b) * ld + r0 + a] = H[3 * a + b];
    }
  }
  __syncthreads();
  lapse(1);
  // ---- C. extend-add: the children's update matrices (a child's boundary maps monotonically into this front: lower stays lower)
  for (int k = 0; k < 2; ++k) {
    if (F.kid[k] < 0) continue;
    const MfFrontDev C = M.fronts[F.kid[k]];
    const int nb3 = C.m - C.own3;
    if (nb3 > 0) {
      const double* __restrict__ U = M.arena + C.off + (size_t)C.own3 * C.ld + C.own3;
      const int* __restrict__ map = M.cmap + F.map_off[k];
      // rows in strips of 64 (one wave per strip and column), four columns of a strip in flight per wave: every element is a
      // dependent read-modify-write through L2, so the loads of several are issued before the first store
      const int nstrip = (nb3 + 1 + 63) >> 6;
      for (int w = wave; w < nstrip * ((nb3 + 3) >> 2); w += kMfNW) {
        const int strip = w % nstrip, j0 = 4 * (w / nstrip);
        const int i = 64 * strip + lane;
        if (i > nb3) continue;
        const int row = (i == nb3) ? m : 3 * map[i / 3] + i % 3;
        double u[4], a[4];
        size_t at[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = min(j0 + q, nb3 - 1);
          at[q] = (3 * (size_t)map[j / 3] + j % 3) * ld + row;
          u[q] = U[(size_t)j * C.ld + i];
          a[q] = A[at[q]];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (j0 + q < nb3 && i >= j0 + q) A[at[q]] = a[q] + u[q];
      }
    }
    __syncthreads();
  }
  lapse(2);
  // ---- D. own columns in panels of 16: left-looking update, 16x16 Cholesky + inverse, panel solve
  const int ldp = (m + 2) | 1;
  for (int k0 = 0; k0 < s3; k0 += kMfPanel) {
    const int wp = min(kMfPanel, s3 - k0), R = m + 1 - k0;
    const bool cv = lr < wp;
    // D1: P = A[k0.., k0 .. k0+wp) - L[k0.., 0 .. k0) L[k0 .. k0+wp, 0 .. k0)^T  ->  LDS
    for (int rt = wave; rt < ((R + 15) >> 4); rt += kMfNW) {
      const int r0 = k0 + 16 * rt;
      mf_d4 acc = {0.0, 0.0, 0.0, 0.0};
      double cur[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) cur[q] = A[(size_t)(k0 + min(lr, wp - 1)) * ld + min(r0 + lk + 4 * q, m)];
      mf_tile_dot(A + (size_t)lk * ld + min(r0 + lr, m), A + (size_t)lk * ld + k0 + min(lr, wp - 1), (size_t)ld, k0, lk, acc);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = r0 + lk + 4 * q;
        if (row <= m) Pn[lr * ldp + row - k0] = cv ? cur[q] - acc[q] : 0.0;
      }
    }
    __syncthreads();
    lapse(3);
    // D2: wave 0: Cholesky of the wp x wp diagonal block and the inverse of its factor, rows on the lanes (lane & 15), columns in
    // registers, pivot column / finished inverse row broadcast by v_readlane
    if (wave == 0) {
      const int i = lr;
      double a[kMfPanel], y[kMfPanel];
#pragma unroll
      for (int c = 0; c < kMfPanel; ++c) {
        a[c] = (i < wp && c < wp) ? Pn[c * ldp + i] : (c == i ? 1.0 : 0.0);
        y[c] = (c == i) ? 1.0 : 0.0;
      }
      bool ok = true;
#pragma unroll
      for (int j = 0; j < kMfPanel; ++j) {
        const double d = mf_readlane(a[j], j);
        ok = ok && d > 0.0 && isfinite(d);
        const double inv = mf_rsqrt(d);
        const double lij = a[j] * inv;   // lane j: sqrt(d)
        a[j] = lij;
        const bool below = i > j;
#pragma unroll
        for (int c = j + 1; c < kMfPanel; ++c) a[c] -= lij * mf_readlane(lij, c);
#pragma unroll
        for (int c = 0; c <= j; ++c) {
          if (i == j) y[c] *= inv;
          const double yjc = mf_readlane(y[c], j);
          if (below) y[c] -= lij * yjc;
        }
      }
      if (lane < kMfPanel) {
#pragma unroll
        for (int c = 0; c < kMfPanel; ++c) {
          if (i < wp && c <= i) Pn[c * ldp + i] = a[c];
          Yt[c * kMfPanel + i] = (c <= i) ? y[c] : 0.0;
        }
      }
      if (!ok && lane == 0) s_fail = 1;
    }
    __syncthreads();
    lapse(4);
    if (s_fail) {
      if (tid == 0) {
        M.flags[1] = it;
        M.flags[0] = 1;
      }
      return;
    }
    // D3: rows below the diagonal block: L21 = P21 L11^-T on the matrix cores (a row tile is private to its wave)
    {
      const int R2 = R - wp;
      for (int rt = wave; rt < ((R2 + 15) >> 4); rt += kMfNW) {
        const int rr0 = wp + 16 * rt;
        const bool rv = rr0 + lr < R;
        double av[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) av[s] = rv ? Pn[(4 * s + lk) * ldp + rr0 + lr] : 0.0;
        mf_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], Yt[(4 * s + lk) * kMfPanel + lr], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = rr0 + lk + 4 * q;
          if (row < R && cv) Pn[lr * ldp + row] = acc[q];
        }
      }
    }
    __syncthreads();
    lapse(5);
    // D4: the finished columns back to the front's matrix
    for (int c = wave; c < wp; c += kMfNW)
      for (int r = c + lane; r < R; r += 64) A[(size_t)(k0 + c) * ld + k0 + r] = Pn[c * ldp + r];
    __syncthreads();
    lapse(6);
  }
  // ---- E. update matrix: U = A22 - L21 L21^T (lower triangle of the boundary rows + the right-hand side row), in place
  const int nb3 = m - s3;
  if (s3 > 0 && nb3 > 0) {
    const int nct = (nb3 + 15) >> 4, nrt = (nb3 + 1 + 15) >> 4;
    int cnt = 0;
    for (int ct = 0; ct < nct; ++ct)
      for (int rt = ct; rt < nrt; ++rt, ++cnt) {
        if ((cnt & (kMfNW - 1)) != wave) continue;
        const int r0 = s3 + 16 * rt, c0 = s3 + 16 * ct;
        mf_d4 acc = {0.0, 0.0, 0.0, 0.0};
        double cur[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = A[(size_t)min(c0 + lr, m - 1) * ld + min(r0 + lk + 4 * q, m)];
        mf_tile_dot(A + (size_t)lk * ld + min(r0 + lr, m), A + (size_t)lk * ld + min(c0 + lr, m), (size_t)ld, s3, lk, acc);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = r0 + lk + 4 * q, col = c0 + lr;
          if (row <= m && col < m && row >= col) A[(size_t)col * ld + row] = cur[q] - acc[q];
        }
      }
  }
  if (M.dbg) {
    __syncthreads();
    lapse(7);
    if (tid == 0)
      for (int k = 0; k < 8; ++k) M.dbg[8 * (size_t)M.level_front[lvl0 + blockIdx.x] + k] = ph[k];
  }
}

// ---------------------------------------------------------------------------- k_mf_solve
__global__ __launch_bounds__(kMfThreads) void k_mf_solve(MfDev M, int lvl0) {
  __shared__ double xs[kMfMaxDim + 1];
  __shared__ double tt[kMfPanel];
  __shared__ double Ld[kMfPanel * (kMfPanel + 1)];
  if (M.flags[0]) return;
  const MfFrontDev F = M.fronts[M.level_front[lvl0 + blockIdx.x]];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = F.m, s3 = F.own3, ld = F.ld;
  if (s3 == 0) return;
  const double* __restrict__ A = M.arena + F.off;
  for (int i = tid; i < m - s3; i += kMfThreads) xs[s3 + i] = M.x[3 * (size_t)M.bnd[F.bnd_off + i / 3] + i % 3];
  __syncthreads();
  for (int c0 = ((s3 - 1) / kMfPanel) * kMfPanel; c0 >= 0; c0 -= kMfPanel) {
    const int wp = min(kMfPanel, s3 - c0);
    // t_c = y_c - sum over the rows below the block of L[r][c] x_r: one wave per column (columns are contiguous)
    for (int cw = wave; cw < wp; cw += kMfNW) {
      const double* col = A + (size_t)(c0 + cw) * ld;
      double sum = 0.0;
      for (int r = c0 + wp + lane; r < m; r += 64) sum += col[r] * xs[r];
      sum = wave_sum(sum);
      if (lane == 0) tt[cw] = col[m] - sum;
    }
    if (tid < kMfPanel * kMfPanel) {
      const int r = tid & 15, c = tid >> 4;
      if (r < wp && c <= r) Ld[r * (kMfPanel + 1) + c] = A[(size_t)(c0 + c) * ld + c0 + r];
    }
    __syncthreads();
    if (wave == 0) {
      const int i = lane & 15;
      double t = (i < wp) ? tt[i] : 0.0;
#pragma unroll
      for (int r = kMfPanel - 1; r >= 0; --r) {
        if (r < wp) {   // uniform
          const double xr = mf_readlane(t, r) / Ld[r * (kMfPanel + 1) + r];
          if (i < r) t -= Ld[r * (kMfPanel + 1) + i] * xr;
          else if (i == r) t = xr;
        }
      }
      if (lane < wp) {
        xs[c0 + lane] = t;
        M.x[3 * (size_t)F.e0 + c0 + lane] = t;
        if (!isfinite(t)) M.flags[2] = 1;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------- k_mf_update
__global__ __launch_bounds__(kBlock) void k_mf_update(MfDev M, double* __restrict__ poses, int it) {
  if (M.flags[0]) return;
  if (M.flags[2]) {   // (set by the launches before this one: every workgroup sees the same value)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      M.flags[1] = it;
      M.flags[0] = 2;
    }
    return;
  }
  for (int p = blockIdx.x * kBlock + threadIdx.x; p < M.n; p += gridDim.x * kBlock) {
    const size_t v = 3 * (size_t)M.elim_vertex[p], o = 3 * (size_t)p;
    poses[v] += M.x[o];
    poses[v + 1] += M.x[o + 1];
    poses[v + 2] = norm_theta(poses[v + 2] + M.x[o + 2]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) M.flags[3] = it + 1;
}

__global__ void k_mf_finish(MfDev M, int iters, DirectResult* __restrict__ res) {
  const unsigned long long now = (unsigned long long)wall_clock64();
  res->done = M.flags[3];
  res->fail = M.flags[0];
  res->fail_iter = M.flags[1];
  res->cycles = 0;
  if (M.flags[0]) res->stamp[2 * M.flags[3] + 2] = now;
  else res->stamp[2 * iters + 1] = now;
}

}  // namespace

struct Mfront {
  MfPlan plan;
  MfrontInfo info;
  MfDev dev;
  void* buf = nullptr;
  std::vector<int> level_lds;   // dynamic LDS of the factor launch of every level
};

const MfrontInfo& mfront_info(const Mfront* m) { return m->info; }

void mfront_destroy(Mfront* m) {
  if (!m) return;
  if (m->buf) hipFree(m->buf);
  delete m;
}

Mfront* mfront_create(hipStream_t s, int V, int n, const int* free_id, const double* poses, int E, const int* ei, const int* ej,
                      int max_rows, std::string* why, std::string* err) {
  MfLimits lim;
  lim.max_rows = max_rows;
  if (const char* e = std::getenv("SGO_MFRONT_LEAF")) lim.leaf = std::max(4, std::atoi(e));
  if (const char* e = std::getenv("SGO_MFRONT_CRIT_MFLOP")) lim.max_crit_flops = 1e6 * std::atof(e);
  std::unique_ptr<Mfront> M(new Mfront);
  if (!mfront_analyze(V, n, free_id, poses, E, ei, ej, lim, &M->plan, why)) return nullptr;
  const MfPlan& P = M->plan;
  M->info.n = n;
  M->info.fronts = (int)P.fronts.size();
  M->info.height = P.height;
  M->info.max_dim = P.max_dim;
  M->info.max_own = P.max_own;
  M->info.max_bnd = P.max_bnd;
  M->info.order_kind = P.order_kind;
  M->info.crit_panels = P.crit_panels;
  M->info.flops = P.flops;
  M->info.crit_flops = P.crit_flops;
  M->info.arena_bytes = (size_t)P.arena_doubles * 8;
  const int nf = (int)P.fronts.size();
  std::vector<MfFrontDev> fd(nf);
  for (int f = 0; f < nf; ++f) {
    const MfFront& F = P.fronts[f];
    MfFrontDev& D = fd[f];
    D.e0 = F.e0;
    D.own3 = 3 * F.own;
    D.m = 3 * (F.own + F.nb);
    D.ld = F.ld;
    D.off = F.off;
    D.nb = F.nb;
    D.bnd_off = F.bnd_off;
    D.kid[0] = F.kid[0];
    D.kid[1] = F.kid[1];
    D.map_off[0] = F.map_off[0];
    D.map_off[1] = F.map_off[1];
    D.tgt0 = F.tgt0;
    D.tgt1 = F.tgt1;
  }
  M->level_lds.assign((size_t)P.height + 1, 0);
  for (int h = 0; h <= P.height; ++h) {
    int mm = 0;
    for (int q = P.level_ptr[h]; q < P.level_ptr[h + 1]; ++q) {
      const MfFront& F = P.fronts[P.level_front[q]];
      mm = std::max(mm, 3 * (F.own + F.nb));
    }
    M->level_lds[h] = (int)sizeof(double) * kMfPanel * ((mm + 2) | 1);
  }
  // one allocation, carved
  struct Part {
    const void* src;
    size_t bytes;
    size_t at;
  };
  std::vector<Part> parts;
  size_t total = 0;
  auto add = [&](const void* src, size_t bytes) {
    total = (total + 255) & ~(size_t)255;
    parts.push_back({src, bytes, total});
    total += bytes;
    return parts.size() - 1;
  };
  const size_t i_fr = add(fd.data(), sizeof(MfFrontDev) * fd.size());
  const size_t i_lf = add(P.level_front.data(), sizeof(int) * P.level_front.size());
  const size_t i_bn = add(P.bnd.data(), sizeof(int) * std::max<size_t>(P.bnd.size(), 1));
  const size_t i_cm = add(P.cmap.data(), sizeof(int) * std::max<size_t>(P.cmap.size(), 1));
  const size_t i_tg = add(P.targets.data(), sizeof(MfTarget) * std::max<size_t>(P.targets.size(), 1));
  const size_t i_ct = add(P.contrib.data(), sizeof(int) * std::max<size_t>(P.contrib.size(), 1));
  const size_t i_ev = add(P.elim_vertex.data(), sizeof(int) * P.elim_vertex.size());
  const size_t i_el = add(nullptr, sizeof(double) * kElemStride * (size_t)std::max(E, 1));
  const size_t i_x = add(nullptr, sizeof(double) * 3 * (size_t)n);
  const size_t i_pt = add(nullptr, sizeof(double) * 2 * kMaxPartials);
  const size_t i_fl = add(nullptr, sizeof(int) * 8);
  const bool debug = std::getenv("SGO_MFRONT_DEBUG") != nullptr;
  const size_t i_db = add(nullptr, debug ? sizeof(long long) * 8 * (size_t)nf : 0);
  const size_t i_ar = add(nullptr, sizeof(double) * (size_t)P.arena_doubles);
  hipError_t he = hipMalloc(&M->buf, total);
  if (he != hipSuccess) {
    M->buf = nullptr;
    if (why) *why = "frontal matrices do not fit the device (" + std::to_string(total >> 20) + " MiB)";
    (void)hipGetLastError();
    return nullptr;
  }
  char* base = (char*)M->buf;
  for (const Part& p : parts) {
    if (!p.src || p.bytes == 0) continue;
    const bool empty = (p.src == P.bnd.data() && P.bnd.empty()) || (p.src == P.cmap.data() && P.cmap.empty()) ||
                       (p.src == P.targets.data() && P.targets.empty()) || (p.src == P.contrib.data() && P.contrib.empty());
    if (empty) continue;
    he = hipMemcpyAsync(base + p.at, p.src, p.bytes, hipMemcpyHostToDevice, s);
    if (he != hipSuccess) {
      if (err) *err = std::string("multifrontal plan upload: ") + hipGetErrorString(he);
      return nullptr;
    }
  }
  he = hipMemsetAsync(base + parts[i_fl].at, 0, sizeof(int) * 8, s);
  if (he == hipSuccess) he = hipStreamSynchronize(s);   // (the host vectors above go out of scope)
  if (he != hipSuccess) {
    if (err) *err = std::string("multifrontal plan upload: ") + hipGetErrorString(he);
    return nullptr;
  }
  MfDev& D = M->dev;
  D.n = n;
  D.E = E;
  D.nfront = nf;
  D.fronts = (const MfFrontDev*)(base + parts[i_fr].at);
  D.level_front = (const int*)(base + parts[i_lf].at);
  D.bnd = (const int*)(base + parts[i_bn].at);
  D.cmap = (const int*)(base + parts[i_cm].at);
  D.targets = (const MfTarget*)(base + parts[i_tg].at);
  D.contrib = (const int*)(base + parts[i_ct].at);
  D.elim_vertex = (const int*)(base + parts[i_ev].at);
  D.elem = (double*)(base + parts[i_el].at);
  D.x = (double*)(base + parts[i_x].at);
  D.partials = (double*)(base + parts[i_pt].at);
  D.flags = (int*)(base + parts[i_fl].at);
  D.arena = (double*)(base + parts[i_ar].at);
  D.dbg = debug ? (long long*)(base + parts[i_db].at) : nullptr;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mf_factor), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)sizeof(double) * kMfPanel * ((kMfMaxDim + 2) | 1));
    attr_set = true;
  }
  return M.release();
}

hipError_t mfront_optimize(Mfront* m, hipStream_t s, const EdgeListDev& el, double* d_poses, int iters, double* d_hist,
                           DirectResult* d_res) {
  const MfPlan& P = m->plan;
  const MfDev& D = m->dev;
  hipError_t he = hipMemsetAsync(D.flags, 0, sizeof(int) * 8, s);
  if (he != hipSuccess) return he;
  const int egrid = std::max(1, std::min((D.E + kBlock - 1) / kBlock, kMaxPartials));
  const int ugrid = std::max(1, std::min((D.n + kBlock - 1) / kBlock, 1024));
  for (int it = 0; it <= iters; ++it) {
    const bool last = it == iters;
    hipLaunchKernelGGL(k_mf_edges, dim3(egrid), dim3(kBlock), 0, s, D, el, (const double*)d_poses, it, last ? 1 : 0, d_hist, d_res);
    if (last) break;
    for (int h = 0; h <= P.height; ++h) {
      const int cnt = P.level_ptr[h + 1] - P.level_ptr[h];
      hipLaunchKernelGGL(k_mf_factor, dim3(cnt), dim3(kMfThreads), (size_t)m->level_lds[h], s, D, P.level_ptr[h], it, h == 0 ? 2 * it + 1 : -1, d_res);
    }
    for (int h = P.height; h >= 0; --h) {
      const int cnt = P.level_ptr[h + 1] - P.level_ptr[h];
      hipLaunchKernelGGL(k_mf_solve, dim3(cnt), dim3(kMfThreads), 0, s, D, P.level_ptr[h]);
    }
    hipLaunchKernelGGL(k_mf_update, dim3(ugrid), dim3(kBlock), 0, s, D, d_poses, it);
  }
  hipLaunchKernelGGL(k_mf_finish, dim3(1), dim3(1), 0, s, D, iters, d_res);
  if (D.dbg && iters > 0) {   // diagnostic: phases of the LAST factorisation, per level the front with the longest total
    std::vector<long long> h(8 * P.fronts.size());
    if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h.data(), D.dbg, sizeof(long long) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
      std::fprintf(stderr, "[sgo] multifrontal phases [cycles] per level (slowest front): clear, edges, extend-add, then per front summed over panels: update, chol16, trsm, store, schur\n");
      for (int lv = 0; lv <= P.height; ++lv) {
        int bf = -1;
        long long bt = -1;
        for (int q = P.level_ptr[lv]; q < P.level_ptr[lv + 1]; ++q) {
          long long t = 0;
          for (int k = 0; k < 8; ++k) t += h[8 * (size_t)P.level_front[q] + k];
          if (t > bt) {
            bt = t;
            bf = P.level_front[q];
          }
        }
        const MfFront& F = P.fronts[bf];
        std::fprintf(stderr, "[sgo]   level %2d: %4d fronts; front %4d own %3d bnd %3d:", lv, P.level_ptr[lv + 1] - P.level_ptr[lv], bf, F.own, F.nb);
        for (int k = 0; k < 8; ++k) std::fprintf(stderr, " %7lld", h[8 * (size_t)bf + k]);
        std::fprintf(stderr, "  total %lld\n", bt);
      }
    }
  }
  return hipGetLastError();
}

double mfront_bytes(const Mfront* m, int E, int iters) {
  // per iteration: the edge list once, every front written and read once
  return (double)iters * (100.0 * E + 2.0 * 8.0 * (double)m->plan.arena_doubles);
}

}  // namespace sgo
