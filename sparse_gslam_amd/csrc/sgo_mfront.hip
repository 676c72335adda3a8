// sgo_mfront.hip -- numeric phase of the multifrontal path (sgo_mfront.h): per Gauss-Newton iteration
//   k_mf_edges    EdgeSE2::computeError + linearizeOplus + robust weighting, the 6x6 element of every edge (D_ii, D_jj, H_ij,
//                 b_i, b_j) and chi2 / robust chi2 (g2o: OptimizationAlgorithmGaussNewton::solve -> computeActiveErrors,
//                 linearizeOplus, constructQuadraticForm; src/sparse_gslam/src/graphs.cpp:9-37 chooses the algorithm)
//   k_mf_merge    one launch per level >= 1, thousands of waves: a parent's matrix gathered per 16 x 16 tile from its children,
//                 whose Schur complements are formed on the fly (fp64 matrix cores)                  } LinearSolverEigen /
//   k_mf_panels   one launch per level, one workgroup per front: the edges' contributions, then the own  } CHOLMOD's numeric
//                 columns factorised in panels of 16                                                  } factorisation, graphs.cpp:19
//   k_mf_solve    one launch per level, top-down: backward substitution
//   k_mf_update   SparseOptimizer::update -> VertexSE2::oplusImpl
// A front's matrix is column-major with leading dimension ld, lower triangle, rows 0 .. m-1 = its poses' scalar rows (own
// first, then boundary, both in elimination order) and row m = the right-hand side: the Cholesky factor of [[H, b], [b^T, .]]
// carries L^-1 b in its last row, and the Schur complement's last row is the children's contribution to the parent's
// right-hand side.  Bounds: the merge is L2 operand traffic spread over the chip; the panel loop is a chain of
// barrier-separated steps (update on the matrix cores, operands from L2 at ~15 B per cycle of the CU's address path -> 16 x 16
// Cholesky + inverse on four waves' registers, bound by the pivot chain's latency -> panel solve on the matrix cores);
// the substitution is launch floor + two memory round trips per level.  Measurements: DESIGN.md section 5c, NOTES.md section 10.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
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
  int pinv_off[2];   // child k: pinv[pinv_off[k] + local pose] = its index among the child's boundary poses, -1: not there
  int tgt0, tgt1;
  int parent;
};

struct MfDev {
  int n = 0, E = 0, nfront = 0;
  const MfFrontDev* fronts = nullptr;
  const int* level_front = nullptr;
  const int* bnd = nullptr;
  const int* pinv = nullptr;
  const int2* mtile = nullptr;   // k_mf_merge's work list: (front, tile row | tile column << 16), level by level
  const MfTarget* targets = nullptr;
  const int* contrib = nullptr;
  const int* elim_vertex = nullptr;
  double* arena = nullptr;
  double* elem = nullptr;      // [E][kElemStride]
  double* x = nullptr;         // [3 n] by elimination position
  double* invd = nullptr;      // [3 n] 1 / L[c][c] of every eliminated scalar row (k_mf_panels), for the substitution
  double* yinv = nullptr;      // [3 n][16] row i of the inverse of its 16 x 16 diagonal block's factor (zeros right of the diagonal)
  double* partials = nullptr;  // [2][kMaxPartials]
  long long* dbg = nullptr;    // diagnostic runs (SGO_MFRONT_DEBUG): [nfront][8] s_memtime cycles of the factor kernel's phases
  int* flags = nullptr;        // [0] fail (1 not positive definite, 2 non-finite update)  [1] iteration of the failure
                               // [2] a back-substitution produced a non-finite value  [3] updates applied
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
  for (; kk + 64 <= K; kk += 64) {
    double av[16], bv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      av[u] = pa[(size_t)(kk + 4 * u) * ld];
      bv[u] = pb[(size_t)(kk + 4 * u) * ld];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
  }
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
  // per-workgroup partial sums; the next launch on the stream (the first level's panel kernel, or k_mf_finish after the closing
  // pass) adds them in a fixed order: mf_chi2_sum
  block_sum_store<2>(acc, M.partials, kMaxPartials);
}

// one wave: hist[2 it], hist[2 it + 1] = the sums of k_mf_edges' nparts partial sums, always in the same order
__device__ __forceinline__ void mf_chi2_sum(const double* __restrict__ partials, int nparts, double* __restrict__ hist, int it, int lane) {
  double c0 = 0.0, c1 = 0.0;
  for (int q = lane; q < nparts; q += 64) {
    c0 += partials[q];
    c1 += partials[kMaxPartials + q];
  }
  c0 = wave_sum(c0);
  c1 = wave_sum(c1);
  if (lane == 0) {
    hist[2 * it] = c0;
    hist[2 * it + 1] = c1;
  }
}

// ---------------------------------------------------------------------------- k_mf_merge
// A front's matrix from its children, GATHERED per 16 x 16 tile of the parent by many workgroups (the upper levels have few
// fronts; one workgroup per front moved every element of a 200-row front through one CU's L2 port twice: 30-60 us per
// level, measured): element (row, col) = sum over the children that hold both poses of
//     F22_child[row', col'] - L21_child[row', :] . L21_child[col', :]
// -- the child's Schur complement is formed HERE, on the matrix cores, tile by tile in the parent's index space (a lane's
// operand row is the child row its parent row maps to), so no update matrix is ever written or read back, and elements no
// child reaches are written as zeros (no clearing pass).  The edges' own contributions are added by k_mf_panels afterwards.
// One wave per tile; fixed order of the two children: bitwise reproducible.
__global__ __launch_bounds__(kBlock) void k_mf_merge(MfDev M, int t0, int t1) {
  if (M.flags[0]) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  for (int t = t0 + (int)blockIdx.x * kWavesPerBlock + wave; t < t1; t += (int)gridDim.x * kWavesPerBlock) {
    const int2 te = M.mtile[t];
    const MfFrontDev F = M.fronts[te.x];
    const int R0 = 16 * (te.y & 0xffff), C0 = 16 * (te.y >> 16);
    const int m = F.m, ld = F.ld;
    double* __restrict__ A = M.arena + F.off;
    double tot[4] = {0.0, 0.0, 0.0, 0.0};
    const int prow = R0 + lr, pcol = C0 + lr;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (F.kid[k] < 0) continue;
      const MfFrontDev C = M.fronts[F.kid[k]];
      const int* __restrict__ pinv = M.pinv + F.pinv_off[k];
      auto cmap = [&](int r) -> int {   // parent scalar row -> child scalar row, -1: the child does not hold that pose
        if (r >= m) return r == m ? C.m : -1;
        const int b = pinv[r / 3];
        return b < 0 ? -1 : C.own3 + 3 * b + r % 3;
      };
      const int ca = cmap(prow);
      const int cb = pcol < m ? cmap(pcol) : -1;
      int orow[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) orow[q] = cmap(R0 + lk + 4 * q);
      if (__ballot(ca >= 0) == 0 || __ballot(cb >= 0) == 0) continue;   // the child holds no row or no column of this tile
      const double* __restrict__ Cm = M.arena + C.off;
      double cur[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) cur[q] = (orow[q] >= 0 && cb >= 0) ? Cm[(size_t)cb * C.ld + orow[q]] : 0.0;
      mf_d4 acc = {0.0, 0.0, 0.0, 0.0};
      mf_tile_dot(Cm + (size_t)lk * C.ld + (ca >= 0 ? ca : C.m), Cm + (size_t)lk * C.ld + (cb >= 0 ? cb : C.m), (size_t)C.ld, C.own3, lk, acc);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (orow[q] >= 0 && cb >= 0) tot[q] += cur[q] - acc[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = R0 + lk + 4 * q;
      if (row <= m && pcol < m && row >= pcol) A[(size_t)pcol * ld + row] = tot[q];
    }
  }
}

// ---------------------------------------------------------------------------- k_mf_panels
// One workgroup per front: the edges' contributions, then the own columns in panels of 16:
//   D1  P = A[k0.., k0 .. k0+16) - L[k0.., 0 .. k0) L[k0 .. k0+16, 0 .. k0)^T to LDS (matrix cores, operands from L2);
//   D2  the 16 x 16 Cholesky on wave 0 and the inverse of its factor on wave 1, one pivot behind (rows on the lanes, columns in
//       registers, broadcasts by v_readlane; wave 0 hands every finished column and pivot to wave 1 through LDS);
//   D3  L21 = P21 L11^-T on the matrix cores, straight to the front's matrix.
__global__ __launch_bounds__(kMfThreads) void k_mf_panels(MfDev M, int lvl0, int it, int stamp_slot, int nparts, double* __restrict__ hist,
                                                         DirectResult* __restrict__ res) {
  extern __shared__ double Pn[];               // panel: column c at Pn + c * ldp, rows relative to k0 
  __shared__ double Yt[kMfPanel * kMfPanel];   // Yt[t * 16 + c] = (L11^-1)[c][t]
  __shared__ double LX[kMfPanel][kMfPanel];    // LX[j][i] = L11[i][j]: column j as wave 0 finishes it
  __shared__ double LI[kMfPanel];              // 1 / L11[j][j]
  __shared__ double HX[kMfPanel / 2][kMfPanel]; // columns 8 .. 15 after pivots 0 .. 7 (wave 2 -> wave 0)
  __shared__ int s_half, s_fail;
  if (stamp_slot >= 0 && blockIdx.x == 0 && threadIdx.x == 0) res->stamp[stamp_slot] = (unsigned long long)wall_clock64();
  if (M.flags[0]) return;
  if (nparts > 0 && blockIdx.x == 0 && threadIdx.x >= kMfThreads - 64)   // (the first level's launch: chi2 of this iteration; the last wave, idle in most phases)
    mf_chi2_sum(M.partials, nparts, hist, it, threadIdx.x & 63);
  const MfFrontDev F = M.fronts[M.level_front[lvl0 + blockIdx.x]];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int m = F.m, s3 = F.own3, ld = F.ld;
  if (m == 0) return;
  double* __restrict__ A = M.arena + F.off;
  if (tid == 0) {
    s_fail = 0;
    s_half = 0;
  }
  if (tid < kMfPanel) LI[tid] = 0.0;
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = M.dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
  auto lapse = [&](int k) {
    if (M.dbg && tid == 0) {
      const long long now = (long long)__builtin_amdgcn_s_memtime();
      ph[k] += now - tprev;
      tprev = now;
    }
  };
  // ---- A. a leaf has no k_mf_merge before it: clear
  if (F.kid[0] < 0 && F.kid[1] < 0) {
    const long long tot = (long long)ld * m / 2;   // ld is even
    double2* A2 = reinterpret_cast<double2*>(A);
    for (long long i = tid; i < tot; i += kMfThreads) A2[i] = make_double2(0.0, 0.0);
    __syncthreads();
  }
  lapse(0);
  // ---- B. the edges whose first-eliminated endpoint is a pose of this front: one thread per 3x3 target, contributions in edge order
  for (int t = F.tgt0 + tid; t < F.tgt1; t += kMfThreads) {
    const MfTarget T = M.targets[t];
    int cv4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cv4[q] = M.contrib[min(T.c0 + q, T.c1 - 1)];
    if (T.li == T.lj) {
      const size_t c0 = 3 * (size_t)T.li;
      double* d[9] = {A + c0 * ld + c0,           A + c0 * ld + c0 + 1,       A + c0 * ld + c0 + 2, A + (c0 + 1) * ld + c0 + 1, A + (c0 + 1) * ld + c0 + 2,
                      A + (c0 + 2) * ld + c0 + 2, A + c0 * ld + m,            A + (c0 + 1) * ld + m, A + (c0 + 2) * ld + m};
      double v[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) v[q] = *d[q];
      for (int c = T.c0; c < T.c1; ++c) {
        const int w = (c - T.c0 < 4) ? cv4[c - T.c0] : M.contrib[c];
        const double* el = M.elem + (size_t)kElemStride * (w >> 2);
        const int side = w & 1;
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] += el[6 * side + q];
#pragma unroll
        for (int q = 0; q < 3; ++q) v[6 + q] += el[21 + 3 * side + q];
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) *d[q] = v[q];
    } else {
      const size_t r0 = 3 * (size_t)T.li, c0 = 3 * (size_t)T.lj;
      double H[9];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) H[3 * a + b] = A[(c0 + b) * ld + r0 + a];
      for (int c = T.c0; c < T.c1; ++c) {
        const int w = (c - T.c0 < 4) ? cv4[c - T.c0] : M.contrib[c];
        const double* el = M.elem + (size_t)kElemStride * (w >> 2) + 12;
        if ((w & 3) == 2) {
#pragma unroll
          for (int q = 0; q < 9; ++q) H[q] += el[q];
        } else {
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) H[3 * a + b] += el[3 * b + a];
        }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) A[(c0 + b) * ld + r0 + a] = H[3 * a + b];
    }
  }
  __syncthreads();
  lapse(1);
  // D2: the 16 x 16 diagonal block of the panel at Pd (columns k0 .. k0 + wp): Cholesky on wave 0, the inverse of the factor on
  // wave 1, one pivot behind; L11 goes to the front's matrix, the inverse to Yt, the pivots' reciprocals to M.invd
  auto diag16 = [&](const double* Pd, int ldp, int k0, int wp) {
    // Four waves share the block's 16 pivots.  Wave 0 runs the pivot chain and keeps the columns it needs next (0 .. 7, then
    // 8 .. 15); wave 2 applies pivots 0 .. 7 to columns 8 .. 15 and hands them over; waves 1 and 3 build the inverse of the
    // factor (columns 0 .. 4 and 5 .. 15: equal shares of the 136 updates).  The followers run one pivot behind: wave 0
    // publishes column j (LX[j]) and THEN the pivot's reciprocal (LI[j], zero until then) -- a wave's LDS operations execute in
    // order, so a follower that reads LI[j] and LX[j] in that order and finds LI[j] non-zero has the column too: one LDS round
    // trip per pivot and follower, no flag, no barrier.  A pivot step is issue-bound (two v_readlane + one fma per broadcast
    // element at ~8 cycles each): the shares keep every wave at <= ~30 instructions per pivot.
    constexpr int kHalf = kMfPanel / 2, kInvCut = 5, kHand = 6;
    auto follow = [&](int j, int i, double& lij, double& inv) {   // column j and 1 / L[j][j] as soon as wave 0 has them
      do {
        inv = __hip_atomic_load(&LI[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (relaxed LDS atomics: plain ds_read / ds_write;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                              // `volatile` would become FLAT accesses with sc0 sc1 and a
        lij = __hip_atomic_load(&LX[j][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // vmcnt wait; the fence pins the order of the two reads for the compiler)
      } while (inv == 0.0);
    };
    if (wave == 0) {
      int i = lr;
      asm volatile("" : "+v"(i));   // (opaque: or the unit-matrix selects of all 16 columns are hoisted out of the panel loop and held in registers)
      double a[kMfPanel];
#pragma unroll
      for (int c = 0; c < kHalf; ++c) a[c] = (i < wp && c < wp) ? Pd[c * ldp + i] : (c == i ? 1.0 : 0.0);
#pragma unroll
      for (int c = kHalf; c < kMfPanel; ++c) a[c] = 0.0;
      bool ok = true;
#pragma unroll
      for (int j = 0; j < kMfPanel; ++j) {
        if (j == kHalf) {
          // columns 8 .. 15 with pivots 0 .. kHand-1 applied, from wave 2 -- which finished them two pivots ago (a hand-over
          // after pivot 7 stalled this wave for 1.5 k cycles per panel: the follower is always a pivot behind) --, then the
          // pivots kHand .. 7 this wave has made since (their columns are still in a[])
          while (__hip_atomic_load(&s_half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int c = kHalf; c < kMfPanel; ++c) a[c] = HX[c - kHalf][i];
#pragma unroll
          for (int jj = kHand; jj < kHalf; ++jj)
#pragma unroll
            for (int c = kHalf; c < kMfPanel; ++c) a[c] -= a[jj] * mf_readlane(a[jj], c);
        }
        const double d = mf_readlane(a[j], j);
        ok = ok && d > 0.0 && isfinite(d);
        double inv = mf_rsqrt(d);
        if (!(inv > 0.0) || !isfinite(inv)) inv = 1.0;   // (a failed pivot: the followers wait for a NON-ZERO reciprocal; the call fails after the barrier)
        const double lij = a[j] * inv;   // lane j: sqrt(d)
        a[j] = lij;
        if (lane < kMfPanel) LX[j][i] = lij;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (compiler ordering only: the hardware keeps a wave's LDS stores in order)
        if (lane == 0) __hip_atomic_store(&LI[j], inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int c = j + 1; c < (j < kHalf ? kHalf : kMfPanel); ++c) a[c] -= lij * mf_readlane(lij, c);
        __builtin_amdgcn_sched_barrier(0);   // (pivots are sequential anyway; without it the scheduler hoists the broadcasts of several steps and spills scalar registers)
      }
      if (!ok && lane == 0) s_fail = 1;   // (the factor block goes to memory from LX, by wave 2)
    } else if (wave == 1 || wave == 3) {
      // the inverse of the factor: z_i = e_i - sum_{t < i} L[i][t] / L[t][t] z_t (row t final after pivot t), Y[i] = z_i / L[i][i]
      int i = lr;
      asm volatile("" : "+v"(i));
      const bool lowc = wave == 1;          // this wave's columns: [0, kInvCut) or [kInvCut, 16)
      double y[kMfPanel];
#pragma unroll
      for (int c = 0; c < kMfPanel; ++c) y[c] = (c == i) ? 1.0 : 0.0;
      double myinv = 1.0;
#pragma unroll
      for (int j = 0; j < kMfPanel; ++j) {
        double lij, inv;
        follow(j, i, lij, inv);
        const double lm = (i > j) ? lij * inv : 0.0;
        if (i == j) myinv = inv;
        if (lowc) {
#pragma unroll
          for (int c = 0; c <= (j < kInvCut - 1 ? j : kInvCut - 1); ++c) y[c] -= lm * mf_readlane(y[c], j);
        } else {
#pragma unroll
          for (int c = kInvCut; c <= j; ++c) y[c] -= lm * mf_readlane(y[c], j);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (lane < kMfPanel) {
        if (lowc) {
#pragma unroll
          for (int c = 0; c < kInvCut; ++c) {
            const double v = (c <= i) ? y[c] * myinv : 0.0;
            Yt[c * kMfPanel + i] = v;
            if (i < wp) M.yinv[(3 * (size_t)F.e0 + k0 + i) * kMfPanel + c] = v;
          }
          if (i < wp) M.invd[3 * (size_t)F.e0 + k0 + i] = myinv;
        } else {
#pragma unroll
          for (int c = kInvCut; c < kMfPanel; ++c) {
            const double v = (c <= i) ? y[c] * myinv : 0.0;
            Yt[c * kMfPanel + i] = v;
            if (i < wp) M.yinv[(3 * (size_t)F.e0 + k0 + i) * kMfPanel + c] = v;
          }
        }
      }
    } else if (wave == 2) {
      int i = lr;
      asm volatile("" : "+v"(i));
      double bq[kHalf];
#pragma unroll
      for (int c = 0; c < kHalf; ++c) bq[c] = (i < wp && c + kHalf < wp) ? Pd[(c + kHalf) * ldp + i] : (c + kHalf == i ? 1.0 : 0.0);
#pragma unroll
      for (int j = 0; j < kHand; ++j) {
        double lij, inv;
        follow(j, i, lij, inv);
#pragma unroll
        for (int c = 0; c < kHalf; ++c) bq[c] -= lij * mf_readlane(lij, c + kHalf);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (lane < kMfPanel) {
#pragma unroll
        for (int c = 0; c < kHalf; ++c) HX[c][i] = bq[c];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __hip_atomic_store(&s_half, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // the finished factor block to the front's matrix, column by column as wave 0 publishes them (off wave 0's path)
#pragma unroll
      for (int c = 0; c < kMfPanel; ++c) {
        double lic, inv;
        follow(c, i, lic, inv);
        if (lane < wp && c <= i) A[(size_t)(k0 + c) * ld + k0 + i] = lic;
      }
    }
  };
  // ---- D. own columns in panels of 16
  const int ldp = (m + 2) | 1;
  for (int k0 = 0; k0 < s3; k0 += kMfPanel) {
    const int wp = min(kMfPanel, s3 - k0), R = m + 1 - k0;
    const bool cv = lr < wp;
    // D1
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
    lapse(2);
    diag16(Pn, ldp, k0, wp);
    __syncthreads();
    lapse(3);
    if (s_fail) {
      if (tid == 0) {
        M.flags[1] = it;
        M.flags[0] = 1;
      }
      return;
    }
    if (tid == 0) s_half = 0;   // (both are read again only after the barrier that ends the next D1)
    if (tid < kMfPanel) LI[tid] = 0.0;
    // D3: rows below the diagonal block, straight to the front's matrix
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
          if (row < R && cv) A[(size_t)(k0 + lr) * ld + k0 + row] = acc[q];
        }
      }
    }
    __syncthreads();
    lapse(4);
  }

  if (M.dbg && tid == 0)
    for (int k = 0; k < 8; ++k) M.dbg[8 * (size_t)M.level_front[lvl0 + blockIdx.x] + k] = ph[k];
}

// ---------------------------------------------------------------------------- k_mf_solve
// Backward substitution, one workgroup per front.  Everything the front reads of its factor is requested up front -- the
// triangle L11 to LDS (packed by rows), the boundary block L21 into registers (one wave per column: columns are contiguous) --
// and the sequential part (16 x 16 triangular solves, then the finished block's contribution to the columns before it) runs
// on ONE wave from LDS, without barriers.  Fronts too large for that take the generic branch (blocks of 16 columns, the
// products with everything below a block by one wave per column).
// One level per launch, top-down.  (The whole tree in ONE launch, a front waiting for its parent's flag, was measured in
// round 4 and removed in round 5: the ten launch boundaries go, but every device-scope release writes the XCD's L2 back -- full
// of the factor just made -- and the chain of ten of them costs more than the launches did: C3s 0.78 against 0.67 ms per
// Gauss-Newton iteration.)
constexpr int kMfSolveOwn = 144;   // own scalar rows up to which L11 is held in LDS (packed: 83.5 KB)
constexpr int kMfSolveBnd = 256;   // ... and boundary scalar rows up to which the whole block L21 is requested in one go
constexpr int kMfSolveCols = (kMfSolveOwn + kMfNW - 1) / kMfNW;   // columns per wave
__global__ __launch_bounds__(kMfThreads) void k_mf_solve(MfDev M, int lvl0) {
  extern __shared__ double Ls[];               // row r of L11 at r (r + 1) / 2
  __shared__ double xs[kMfMaxDim + 1];
  __shared__ double tt[kMfSolveOwn];
  __shared__ double Yl[kMfSolveOwn * kMfPanel];   // the inverses of the diagonal blocks' factors, row by row
  __shared__ double Ld[kMfPanel * (kMfPanel + 1)];
  const int f = M.level_front[lvl0 + blockIdx.x];
  const MfFrontDev F = M.fronts[f];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = F.m, s3 = F.own3, ld = F.ld, nb3 = m - s3;
  if (M.flags[0] || s3 == 0) return;   // nothing to solve here
  const double* __restrict__ A = M.arena + F.off;
  if (s3 <= kMfSolveOwn && nb3 <= kMfSolveBnd) {
    // the block L21 (this wave's columns c = wave, wave + 8, ...: up to 18 columns x 4 strips of 64 rows per lane), the triangle
    double v[kMfSolveCols][kMfSolveBnd / 64], yc[kMfSolveCols];
#pragma unroll
    for (int q = 0; q < kMfSolveCols; ++q) {
      const int c = wave + kMfNW * q;
      const double* col = A + (size_t)min(c, s3 - 1) * ld;
#pragma unroll
      for (int ch = 0; ch < kMfSolveBnd / 64; ++ch) {
        const int r = s3 + 64 * ch + lane;
        v[q][ch] = (c < s3 && r < m) ? col[r] : 0.0;
      }
      yc[q] = (c < s3 && lane == 0) ? col[m] : 0.0;
    }
    for (int e = tid; e < s3 * kMfPanel; e += kMfThreads) Yl[e] = M.yinv[3 * (size_t)F.e0 * kMfPanel + e];
    for (int c = wave; c < s3; c += kMfNW)
      for (int r = c + lane; r < s3; r += 64) Ls[r * (r + 1) / 2 + c] = A[(size_t)c * ld + r];
    if (tid < nb3) xs[s3 + tid] = M.x[3 * (size_t)M.bnd[F.bnd_off + tid / 3] + tid % 3];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kMfSolveCols; ++q) {
      const int c = wave + kMfNW * q;
      if (c < s3) {   // uniform per wave
        double sum = 0.0;
#pragma unroll
        for (int ch = 0; ch < kMfSolveBnd / 64; ++ch) {
          const int r = s3 + 64 * ch + lane;
          if (r < m) sum += v[q][ch] * xs[r];
        }
        sum = wave_sum(sum);
        if (lane == 0) tt[c] = yc[q] - sum;
      }
    }
    __syncthreads();
    if (wave != 0) return;
    const int i = lane & 15;
    for (int c0 = ((s3 - 1) / kMfPanel) * kMfPanel; c0 >= 0; c0 -= kMfPanel) {
      const int wp = min(kMfPanel, s3 - c0);
      // x_blk = L_dd^-T t_blk with the block's inverse from the factorisation: 16 independent broadcast-fma pairs per lane
      // instead of a chain of 16 dependent steps
      const double tv = (i < wp) ? tt[c0 + i] : 0.0;
      double t = 0.0;
#pragma unroll
      for (int r = 0; r < kMfPanel; ++r)
        if (r < wp) t += Yl[(c0 + r) * kMfPanel + i] * mf_readlane(tv, r);   // (L^-1)[r][i] is zero for i > r
      if (lane < wp) {
        xs[c0 + lane] = t;
        M.x[3 * (size_t)F.e0 + c0 + lane] = t;
        if (!isfinite(t)) M.flags[2] = 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      // the finished block's contribution to the columns before it
      for (int c = lane; c < c0; c += 64) {
        double s = tt[c];
#pragma unroll 4
        for (int r = 0; r < wp; ++r) {
          const int rr = c0 + r;
          s -= Ls[rr * (rr + 1) / 2 + c] * xs[rr];
        }
        tt[c] = s;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    return;
  }
  // ---- generic branch
  for (int i = tid; i < nb3; i += kMfThreads) xs[s3 + i] = M.x[3 * (size_t)M.bnd[F.bnd_off + i / 3] + i % 3];
  __syncthreads();
  for (int c0 = ((s3 - 1) / kMfPanel) * kMfPanel; c0 >= 0; c0 -= kMfPanel) {
    const int wp = min(kMfPanel, s3 - c0);
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
          const double xr = mf_readlane(t, r) * M.invd[3 * (size_t)F.e0 + c0 + r];
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

__global__ __launch_bounds__(64) void k_mf_finish(MfDev M, int iters, int nparts, double* __restrict__ hist, DirectResult* __restrict__ res) {
  if (!M.flags[0]) mf_chi2_sum(M.partials, nparts, hist, iters, threadIdx.x);   // the closing pass
  if (threadIdx.x != 0) return;
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
  std::vector<int> level_lds;        // dynamic LDS of the panel launch of every level
  std::vector<int> level_solve_lds;  // ... of the substitution launch
  std::vector<int> mtile_ptr;        // k_mf_merge's tiles of level h: [mtile_ptr[h], mtile_ptr[h + 1])
};

const MfrontInfo& mfront_info(const Mfront* m) { return m->info; }

void mfront_destroy(Mfront* m) {
  delete m;   // (the device arrays belong to the caller's arena)
}

Mfront* mfront_create(hipStream_t s, DevArena* arena, int V, int n, const int* free_id, const double* poses, int E, const int* ei,
                      const int* ej, int max_rows, int* order_hint, std::string* why, std::string* err) {
  MfLimits lim;
  lim.max_rows = max_rows;
  // order_hint: kind | poses << 1 of the context's previous graph
  if (order_hint && *order_hint >= 0) {
    const int prev_n = *order_hint >> 1;
    if (std::abs(prev_n - n) * 10 <= prev_n) lim.only_kind = *order_hint & 1;
  }
  if (const char* e = std::getenv("SGO_MFRONT_LEAF")) lim.leaf = std::max(4, std::atoi(e));
  if (const char* e = std::getenv("SGO_MFRONT_CRIT_MFLOP")) lim.max_crit_flops = 1e6 * std::atof(e);
  if (const char* e = std::getenv("SGO_MFRONT_DEGREE")) lim.max_degree = std::atof(e);
  std::unique_ptr<Mfront, void (*)(Mfront*)> M(new Mfront, &mfront_destroy);   // (frees the device buffer on every early return)
  if (!mfront_analyze(V, n, free_id, poses, E, ei, ej, lim, &M->plan, why)) {
    if (lim.only_kind < 0) return nullptr;
    lim.only_kind = -1;   // the hinted order no longer qualifies: the full analysis has the last word
    if (!mfront_analyze(V, n, free_id, poses, E, ei, ej, lim, &M->plan, why)) return nullptr;
  }
  if (order_hint) *order_hint = M->plan.order_kind | (n << 1);
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
    D.pinv_off[0] = D.pinv_off[1] = 0;
    D.tgt0 = F.tgt0;
    D.tgt1 = F.tgt1;
    D.parent = F.parent;
  }
  // inverse extend-add maps and the merge kernel's tiles
  std::vector<int> pinv;
  std::vector<int2> mtile;
  M->mtile_ptr.assign((size_t)P.height + 2, 0);
  for (int f = 0; f < nf; ++f) {
    const MfFront& F = P.fronts[f];
    for (int k = 0; k < 2; ++k) {
      if (F.kid[k] < 0) continue;
      fd[f].pinv_off[k] = (int)pinv.size();
      pinv.resize(pinv.size() + (size_t)(F.own + F.nb), -1);
      const MfFront& C = P.fronts[F.kid[k]];
      for (int b = 0; b < C.nb; ++b) pinv[(size_t)fd[f].pinv_off[k] + P.cmap[(size_t)F.map_off[k] + b]] = b;
    }
  }
  for (int h = 1; h <= P.height; ++h) {
    M->mtile_ptr[h] = (int)mtile.size();
    for (int q = P.level_ptr[h]; q < P.level_ptr[h + 1]; ++q) {
      const int f = P.level_front[q];
      const int mm = 3 * (P.fronts[f].own + P.fronts[f].nb);
      const int nrt = (mm + 1 + 15) / 16, nct = (mm + 15) / 16;
      for (int tc = 0; tc < nct; ++tc)
        for (int tr = tc; tr < nrt; ++tr) mtile.push_back(make_int2(f, tr | (tc << 16)));
    }
  }
  M->mtile_ptr[0] = 0;
  M->mtile_ptr[(size_t)P.height + 1] = (int)mtile.size();
  if (P.height >= 1) M->mtile_ptr[1] = 0;
  M->level_lds.assign((size_t)P.height + 1, 0);
  M->level_solve_lds.assign((size_t)P.height + 1, 0);
  for (int h = 0; h <= P.height; ++h) {
    int mm = 0;
    for (int q = P.level_ptr[h]; q < P.level_ptr[h + 1]; ++q) {
      const MfFront& F = P.fronts[P.level_front[q]];
      mm = std::max(mm, 3 * (F.own + F.nb));
    }
    M->level_lds[h] = (int)sizeof(double) * kMfPanel * ((mm + 2) | 1);
    int so = 0;
    for (int q = P.level_ptr[h]; q < P.level_ptr[h + 1]; ++q) {
      const int s3 = 3 * P.fronts[P.level_front[q]].own;
      if (s3 <= kMfSolveOwn && 3 * P.fronts[P.level_front[q]].nb <= kMfSolveBnd) so = std::max(so, s3);
    }
    M->level_solve_lds[h] = (int)sizeof(double) * std::max(1, so * (so + 1) / 2);
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
  const size_t i_cm = add(pinv.data(), sizeof(int) * std::max<size_t>(pinv.size(), 1));
  const size_t i_mt = add(mtile.data(), sizeof(int2) * std::max<size_t>(mtile.size(), 1));
  const size_t i_tg = add(P.targets.data(), sizeof(MfTarget) * std::max<size_t>(P.targets.size(), 1));
  const size_t i_ct = add(P.contrib.data(), sizeof(int) * std::max<size_t>(P.contrib.size(), 1));
  const size_t i_ev = add(P.elim_vertex.data(), sizeof(int) * P.elim_vertex.size());
  const size_t i_el = add(nullptr, sizeof(double) * kElemStride * (size_t)std::max(E, 1));
  const size_t i_x = add(nullptr, sizeof(double) * 3 * (size_t)n);
  const size_t i_id = add(nullptr, sizeof(double) * 3 * (size_t)n);
  const size_t i_yi = add(nullptr, sizeof(double) * 3 * (size_t)n * kMfPanel);
  const size_t i_pt = add(nullptr, sizeof(double) * 2 * kMaxPartials);
  const size_t i_fl = add(nullptr, sizeof(int) * 8);
  const bool debug = std::getenv("SGO_MFRONT_DEBUG") != nullptr;
  const size_t i_db = add(nullptr, debug ? sizeof(long long) * 8 * (size_t)nf : 0);
  const size_t i_ar = add(nullptr, sizeof(double) * (size_t)P.arena_doubles);
  hipError_t he = hipSuccess;
  M->buf = arena->take(total);
  if (!M->buf) {
    if (why) *why = "frontal matrices do not fit the device (" + std::to_string(total >> 20) + " MiB)";
    (void)hipGetLastError();
    return nullptr;
  }
  char* base = (char*)M->buf;
  for (const Part& p : parts) {
    if (!p.src || p.bytes == 0) continue;
    const bool empty = (p.src == P.bnd.data() && P.bnd.empty()) || (p.src == pinv.data() && pinv.empty()) || (p.src == mtile.data() && mtile.empty()) ||
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
  D.pinv = (const int*)(base + parts[i_cm].at);
  D.mtile = (const int2*)(base + parts[i_mt].at);
  D.targets = (const MfTarget*)(base + parts[i_tg].at);
  D.contrib = (const int*)(base + parts[i_ct].at);
  D.elim_vertex = (const int*)(base + parts[i_ev].at);
  D.elem = (double*)(base + parts[i_el].at);
  D.x = (double*)(base + parts[i_x].at);
  D.invd = (double*)(base + parts[i_id].at);
  D.yinv = (double*)(base + parts[i_yi].at);
  D.partials = (double*)(base + parts[i_pt].at);
  D.flags = (int*)(base + parts[i_fl].at);
  D.arena = (double*)(base + parts[i_ar].at);
  D.dbg = debug ? (long long*)(base + parts[i_db].at) : nullptr;
  // (per call: the attribute belongs to the current device's copy of the kernel, and a process may hold contexts on several)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mf_panels), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)sizeof(double) * kMfPanel * ((kMfMaxDim + 2) | 1)) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mf_solve), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)sizeof(double) * kMfSolveOwn * (kMfSolveOwn + 1) / 2) != hipSuccess) {
    (void)hipGetLastError();
    if (why) *why = "the device does not grant the kernels' dynamic LDS";
    return nullptr;
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
      if (h > 0) {
        const int nt = m->mtile_ptr[h + 1] - m->mtile_ptr[h];
        const int grid = std::max(1, std::min((nt + kWavesPerBlock - 1) / kWavesPerBlock, 8192));
        hipLaunchKernelGGL(k_mf_merge, dim3(grid), dim3(kBlock), 0, s, D, m->mtile_ptr[h], m->mtile_ptr[h + 1]);
      }
      hipLaunchKernelGGL(k_mf_panels, dim3(cnt), dim3(kMfThreads), (size_t)m->level_lds[h], s, D, P.level_ptr[h], it, h == 0 ? 2 * it + 1 : -1,
                         h == 0 ? egrid : 0, d_hist, d_res);
    }
    for (int h = P.height; h >= 0; --h) {
      const int cnt = P.level_ptr[h + 1] - P.level_ptr[h];
      hipLaunchKernelGGL(k_mf_solve, dim3(cnt), dim3(kMfThreads), (size_t)m->level_solve_lds[h], s, D, P.level_ptr[h]);
    }
    hipLaunchKernelGGL(k_mf_update, dim3(ugrid), dim3(kBlock), 0, s, D, d_poses, it);
  }
  hipLaunchKernelGGL(k_mf_finish, dim3(1), dim3(64), 0, s, D, iters, egrid, d_hist, d_res);
  if (D.dbg && iters > 0) {   // diagnostic: phases of the LAST factorisation, per level the front with the longest total
    std::vector<long long> h(8 * P.fronts.size());
    if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h.data(), D.dbg, sizeof(long long) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
      std::fprintf(stderr, "[sgo] multifrontal phases [cycles] per level (slowest front): clear, edges, then summed over the panels: update, chol16 + inverse, trsm + store\n");
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
