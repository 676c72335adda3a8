// sgo_overlay.hip -- kernels and host side of the incremental re-initialisation (sgo_overlay.h): the appended edges'
// linearisation, the block-tridiagonal elimination of the appended chain, the operator's rank-3|T| term and the chain's
// back-substitution.  gfx950, wave64.  All sums in a fixed order (one thread per target, recompute instead of scatter).
#include <algorithm>
#include <cstring>

#include "sgo_device.h"
#include "sgo_overlay.h"

namespace sgo {
namespace {

constexpr int kOvTile = 8;           // chain rows per LDS tile of the elimination (3 x 8 x 193 doubles: 37 KB)
constexpr int kOvThreads = 256;
constexpr int kOvCols = 3 * kOvMaxTouched + 1;
static_assert(kOvCols <= kOvThreads, "one right-hand-side column per thread");
static_assert(3 * kOvMaxHubs * (3 * kOvMaxTouched + 1) <= 3 * kOvTile * kOvCols, "the hubs' Gauss-Jordan tableau lives in the elimination's LDS tile");

// What edge e contributes to the row on its `side` (0: vertices()[0], Jacobian A; 1: vertices()[1], Jacobian B):
// D += R^T Ow R (symmetric packing), b -= R^T Ow e, blk = R^T Ow C (block towards the other endpoint).  The arithmetic of
// k_linearize (EdgeSE2::computeError, linearizeOplus, RobustKernelDCS::robustify, constructQuadraticForm).
__device__ __forceinline__ void ov_edge_terms(const EdgeListDev& el, int e, int side, const double* __restrict__ poses,
                                              double (&D)[6], double (&b)[3], double (&blk)[9]) {
  const size_t ns = (size_t)el.E;
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
  double oe0 = o00 * er[0] + o01 * er[1] + o02 * er[2];
  double oe1 = o01 * er[0] + o11 * er[1] + o12 * er[2];
  double oe2 = o02 * er[0] + o12 * er[1] + o22 * er[2];
  const double e2 = er[0] * oe0 + er[1] * oe1 + er[2] * oe2;
  double r0_, w;
  dcs(e2, el.phi[e], &r0_, &w);
  const double w00 = w * o00, w01 = w * o01, w02 = w * o02, w11 = w * o11, w12 = w * o12, w22 = w * o22;
  oe0 *= w; oe1 *= w; oe2 *= w;
  double si, ci;
  sincos(ti, &si, &ci);
  const double ddx = xj - xi, ddy = yj - yi;
  const double a02 = -si * ddx + ci * ddy, a12 = -ci * ddx - si * ddy;
  const double A00 = cz * (-ci) - sz * si, A01 = cz * (-si) - sz * (-ci), A02 = cz * a02 - sz * a12;
  const double A10 = sz * (-ci) + cz * si, A11 = sz * (-si) + cz * (-ci), A12 = sz * a02 + cz * a12;
  const double B00 = cz * ci - sz * (-si), B01 = cz * si - sz * ci;
  const double B10 = sz * ci + cz * (-si), B11 = sz * si + cz * ci;
  const bool dir = side != 0;
  const double R00 = dir ? B00 : A00, R01 = dir ? B01 : A01, R02 = dir ? 0.0 : A02;
  const double R10 = dir ? B10 : A10, R11 = dir ? B11 : A11, R12 = dir ? 0.0 : A12;
  const double R22 = dir ? 1.0 : -1.0;
  const double T00 = w00 * R00 + w01 * R10, T01 = w00 * R01 + w01 * R11, T02 = w00 * R02 + w01 * R12 + w02 * R22;
  const double T10 = w01 * R00 + w11 * R10, T11 = w01 * R01 + w11 * R11, T12 = w01 * R02 + w11 * R12 + w12 * R22;
  const double T20 = w02 * R00 + w12 * R10, T21 = w02 * R01 + w12 * R11, T22 = w02 * R02 + w12 * R12 + w22 * R22;
  D[0] += R00 * T00 + R10 * T10;
  D[1] += R00 * T01 + R10 * T11;
  D[2] += R00 * T02 + R10 * T12;
  D[3] += R01 * T01 + R11 * T11;
  D[4] += R01 * T02 + R11 * T12;
  D[5] += R02 * T02 + R12 * T12 + R22 * T22;
  b[0] -= R00 * oe0 + R10 * oe1;
  b[1] -= R01 * oe0 + R11 * oe1;
  b[2] -= R02 * oe0 + R12 * oe1 + R22 * oe2;
  const double C00 = dir ? A00 : B00, C01 = dir ? A01 : B01, C02 = dir ? A02 : 0.0;
  const double C10 = dir ? A10 : B10, C11 = dir ? A11 : B11, C12 = dir ? A12 : 0.0;
  const double C22 = dir ? -1.0 : 1.0;
  blk[0] = T00 * C00 + T10 * C10; blk[1] = T00 * C01 + T10 * C11; blk[2] = T00 * C02 + T10 * C12 + T20 * C22;
  blk[3] = T01 * C00 + T11 * C10; blk[4] = T01 * C01 + T11 * C11; blk[5] = T01 * C02 + T11 * C12 + T21 * C22;
  blk[6] = T02 * C00 + T12 * C10; blk[7] = T02 * C01 + T12 * C11; blk[8] = T02 * C02 + T12 * C12 + T22 * C22;
}

// ---------------------------------------------------------------------------- k_ov_lin
// One thread per overlay row (new rows, then touched rows): the row's appended edges in entry order.
//   new row i:     Dn[i], b_N (last column of H0), Un[i] = H_{i,i+1}, H0 blocks towards touched rows
//   touched row t: M0 rows 3t..3t+2 (its own diagonal contribution and blocks towards other touched rows), bt
__global__ __launch_bounds__(kOvThreads) void k_ov_lin(OverlayDev O, const double* __restrict__ poses) {
  const int r = blockIdx.x * kOvThreads + threadIdx.x;
  const int k = O.k, nt = O.nt + O.nx, nc = O.ncol, nt3 = 3 * nt;   // (nt: the kept rows -- touched base rows, then hubs)
  if (r >= k + nt) return;
  double D[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  if (r < k) {
    double* h = O.H0 + (size_t)3 * r * nc;
    for (int q = 0; q < 3 * nc; ++q) h[q] = 0.0;
    double U[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = O.rp[r]; t < O.rp[r + 1]; ++t) {
      double blk[9];
      ov_edge_terms(O.el, O.ent_edge[t], O.ent_side[t], poses, D, b, blk);
      const int oth = O.ent_other[t];
      if (oth == kOvOtherFixed) continue;
      if (oth >= 0) {
        if (oth == r + 1) {
#pragma unroll
          for (int q = 0; q < 9; ++q) U[q] += blk[q];
        }
      } else {
        const int tc = 3 * (-1 - oth);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c2 = 0; c2 < 3; ++c2) h[a * nc + tc + c2] += blk[3 * a + c2];
      }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) O.Dn[6 * (size_t)r + q] = D[q];
#pragma unroll
    for (int q = 0; q < 9; ++q) O.Un[9 * (size_t)r + q] = U[q];
#pragma unroll
    for (int a = 0; a < 3; ++a) h[a * nc + nt3] = b[a];
  } else {
    const int t0 = r - k;
    double* m = O.M0 + (size_t)3 * t0 * nt3;
    for (int q = 0; q < 3 * nt3; ++q) m[q] = 0.0;
    for (int t = O.rp[r]; t < O.rp[r + 1]; ++t) {
      double blk[9];
      ov_edge_terms(O.el, O.ent_edge[t], O.ent_side[t], poses, D, b, blk);
      const int oth = O.ent_other[t];
      if (oth == kOvOtherFixed || oth >= 0) continue;   // (blocks towards new rows come from the new rows' side)
      const int tc = 3 * (-1 - oth);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c2 = 0; c2 < 3; ++c2) m[a * nt3 + tc + c2] += blk[3 * a + c2];
    }
    const int dc = 3 * t0;
    m[0 * nt3 + dc] += D[0]; m[0 * nt3 + dc + 1] += D[1]; m[0 * nt3 + dc + 2] += D[2];
    m[1 * nt3 + dc] += D[1]; m[1 * nt3 + dc + 1] += D[3]; m[1 * nt3 + dc + 2] += D[4];
    m[2 * nt3 + dc] += D[2]; m[2 * nt3 + dc + 1] += D[4]; m[2 * nt3 + dc + 2] += D[5];
    O.bt[dc] = b[0]; O.bt[dc + 1] = b[1]; O.bt[dc + 2] = b[2];
  }
}

// symmetric 3x3 inverse (packing 00 01 02 11 12 22); false when the block is not positive definite
__device__ __forceinline__ bool sym3_inverse(const double (&d)[6], double (&v)[6]) {
  const double c00 = d[3] * d[5] - d[4] * d[4], c01 = d[2] * d[4] - d[1] * d[5], c02 = d[1] * d[4] - d[2] * d[3];
  const double c11 = d[0] * d[5] - d[2] * d[2], c12 = d[1] * d[2] - d[0] * d[4], c22 = d[0] * d[3] - d[1] * d[1];
  const double det = d[0] * c00 + d[1] * c01 + d[2] * c02;
  const bool ok = d[0] > 0.0 && c22 > 0.0 && det > 0.0 && isfinite(det);
  const double id = ok ? 1.0 / det : 0.0;
  v[0] = c00 * id; v[1] = c01 * id; v[2] = c02 * id; v[3] = c11 * id; v[4] = c12 * id; v[5] = c22 * id;
  return ok;
}

// ---------------------------------------------------------------------------- k_ov_solve
// One workgroup.  Block-tridiagonal LDL^T of H_NN along the chain with the ncol right-hand sides [H_NT | b_N] on the THREADS
// of the workgroup (round 6; one wave's lanes before: 16 kept rows at most), one column each; every thread repeats the 3x3
// pivot arithmetic (uniform: same inputs from LDS, same registers, same order -- no hand-over between the waves inside a tile:
// a column is private to its thread), the chain walked in LDS tiles of kOvTile rows that all threads load and store; then M = M0 - H_TN Y (symmetrised) and g = bt - H_TN y_b, which is added to the touched rows'
// right-hand sides in dgb.  A pivot block that is not positive definite (an appended chain that hangs in the air) makes
// g non-finite: the PCG start then reports a breakdown, as for any Hessian that is not positive definite.
__global__ __launch_bounds__(kOvThreads) void k_ov_solve(OverlayDev O, double* __restrict__ dgb) {
  __shared__ double Yt[3 * kOvTile * kOvCols];
  __shared__ double Dt[kOvTile * 6], Ut[kOvTile * 9], Sv[kOvTile * 6];
  __shared__ double fcol[3 * kOvMaxHubs];
  __shared__ int fail, bad_x;
  const int tid = threadIdx.x;
  const int k = O.k, nc = O.ncol, nk3 = 3 * (O.nt + O.nx);
  if (tid == 0) fail = bad_x = 0;
  __syncthreads();
  // ---- forward elimination
  double sp[6] = {0, 0, 0, 0, 0, 0}, up[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, rpv[3] = {0, 0, 0};
  for (int t0 = 0; t0 < k; t0 += kOvTile) {
    const int rows = min(kOvTile, k - t0);
    for (int i = tid; i < 3 * rows * nc; i += kOvThreads) Yt[i] = O.H0[(size_t)3 * t0 * nc + i];
    for (int i = tid; i < 6 * rows; i += kOvThreads) Dt[i] = O.Dn[6 * (size_t)t0 + i];
    for (int i = tid; i < 9 * rows; i += kOvThreads) Ut[i] = O.Un[9 * (size_t)t0 + i];
    __syncthreads();
    {
      for (int i = 0; i < rows; ++i) {
        double S[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) S[q] = Dt[6 * i + q];
        double r[3] = {0, 0, 0};
        if (tid < nc) {
          r[0] = Yt[(3 * i) * nc + tid]; r[1] = Yt[(3 * i + 1) * nc + tid]; r[2] = Yt[(3 * i + 2) * nc + tid];
        }
        if (t0 + i > 0) {
          // L = U_prev^T Sinv_prev ; S -= L U_prev ; r -= L r_prev
          const double P[9] = {sp[0], sp[1], sp[2], sp[1], sp[3], sp[4], sp[2], sp[4], sp[5]};
          double L[9];
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c2 = 0; c2 < 3; ++c2) L[3 * a + c2] = up[a] * P[c2] + up[3 + a] * P[3 + c2] + up[6 + a] * P[6 + c2];
          S[0] -= L[0] * up[0] + L[1] * up[3] + L[2] * up[6];
          S[1] -= L[0] * up[1] + L[1] * up[4] + L[2] * up[7];
          S[2] -= L[0] * up[2] + L[1] * up[5] + L[2] * up[8];
          S[3] -= L[3] * up[1] + L[4] * up[4] + L[5] * up[7];
          S[4] -= L[3] * up[2] + L[4] * up[5] + L[5] * up[8];
          S[5] -= L[6] * up[2] + L[7] * up[5] + L[8] * up[8];
#pragma unroll
          for (int a = 0; a < 3; ++a) r[a] -= L[3 * a] * rpv[0] + L[3 * a + 1] * rpv[1] + L[3 * a + 2] * rpv[2];
        }
        double si[6];
        if (!sym3_inverse(S, si) && tid == 0) fail = 1;
        if (tid < 6) Sv[6 * i + tid] = si[tid];
        if (tid < nc) {
          Yt[(3 * i) * nc + tid] = r[0]; Yt[(3 * i + 1) * nc + tid] = r[1]; Yt[(3 * i + 2) * nc + tid] = r[2];
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) sp[q] = si[q];
#pragma unroll
        for (int q = 0; q < 9; ++q) up[q] = Ut[9 * i + q];
        rpv[0] = r[0]; rpv[1] = r[1]; rpv[2] = r[2];
      }
    }
    __syncthreads();
    for (int i = tid; i < 3 * rows * nc; i += kOvThreads) O.Y[(size_t)3 * t0 * nc + i] = Yt[i];
    for (int i = tid; i < 6 * rows; i += kOvThreads) O.Sinv[6 * (size_t)t0 + i] = Sv[i];
    __syncthreads();
  }
  // ---- back substitution: y_i = Sinv_i (r_i - U_i y_{i+1})
  double yn[3] = {0, 0, 0};
  const int ntile = (k + kOvTile - 1) / kOvTile;
  for (int tt = ntile - 1; tt >= 0; --tt) {
    const int t0 = tt * kOvTile, rows = min(kOvTile, k - t0);
    for (int i = tid; i < 3 * rows * nc; i += kOvThreads) Yt[i] = O.Y[(size_t)3 * t0 * nc + i];
    for (int i = tid; i < 6 * rows; i += kOvThreads) Sv[i] = O.Sinv[6 * (size_t)t0 + i];
    for (int i = tid; i < 9 * rows; i += kOvThreads) Ut[i] = O.Un[9 * (size_t)t0 + i];
    __syncthreads();
    if (tid < nc) {
      for (int i = rows - 1; i >= 0; --i) {
        double r[3] = {Yt[(3 * i) * nc + tid], Yt[(3 * i + 1) * nc + tid], Yt[(3 * i + 2) * nc + tid]};
        if (t0 + i < k - 1) {
          const double* U = Ut + 9 * i;
#pragma unroll
          for (int a = 0; a < 3; ++a) r[a] -= U[3 * a] * yn[0] + U[3 * a + 1] * yn[1] + U[3 * a + 2] * yn[2];
        }
        const double* v = Sv + 6 * i;
        yn[0] = v[0] * r[0] + v[1] * r[1] + v[2] * r[2];
        yn[1] = v[1] * r[0] + v[3] * r[1] + v[4] * r[2];
        yn[2] = v[2] * r[0] + v[4] * r[1] + v[5] * r[2];
        Yt[(3 * i) * nc + tid] = yn[0]; Yt[(3 * i + 1) * nc + tid] = yn[1]; Yt[(3 * i + 2) * nc + tid] = yn[2];
      }
    }
    __syncthreads();
    for (int i = tid; i < 3 * rows * nc; i += kOvThreads) O.Y[(size_t)3 * t0 * nc + i] = Yt[i];
    __syncthreads();
  }
  // ---- S = sym(M0 - H_KN Y), gk = bt - H_KN y_b over the kept rows K (H_KN = H_NK^T: only the rows listed in nz carry blocks)
  const bool bad = fail != 0;
  for (int idx = tid; idx < nk3 * (nk3 + 1); idx += kOvThreads) {
    const int a = idx / (nk3 + 1), bcol = idx % (nk3 + 1);
    double s1 = 0.0, s2 = 0.0;
    for (int z = 0; z < O.nnz; ++z) {
      const size_t base = (size_t)3 * O.nz[z] * nc;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const double* h = O.H0 + base + (size_t)q * nc;
        const double* y = O.Y + base + (size_t)q * nc;
        s1 += h[a] * y[bcol];
        if (bcol < nk3) s2 += h[bcol] * y[a];
      }
    }
    if (bcol < nk3) {
      const double mab = O.M0[a * nk3 + bcol] - s1, mba = O.M0[bcol * nk3 + a] - s2;
      O.S[a * nk3 + bcol] = 0.5 * (mab + mba);
    } else {
      O.gk[a] = O.bt[a] - s1;
    }
  }
  __syncthreads();
  // ---- the hubs: G = [S_XX | S_XT | g_X] -> S_XX^-1 [...] by Gauss-Jordan (SPD, no pivoting) in LDS, one lane per column
  const int nt3 = 3 * O.nt, nx3 = 3 * O.nx, gw = nx3 + nt3 + 1;
  double* G = Yt;   // (the chain's tiles are done with it)
  if (nx3 > 0) {
    for (int e = tid; e < nx3 * gw; e += kOvThreads) {
      const int r2 = e / gw, c2 = e % gw;
      G[e] = c2 < nx3 ? O.S[(nt3 + r2) * nk3 + nt3 + c2] : (c2 < nx3 + nt3 ? O.S[(nt3 + r2) * nk3 + (c2 - nx3)] : O.gk[nt3 + r2]);
    }
    __syncthreads();
    // reduced row echelon form of [S_XX | S_XT | g_X]: thread c owns column c (gw <= 3 x 64 + 1 columns: several waves).  Per pivot:
    // the pivot column's factors are copied aside and the scaled pivot row taken into registers BEFORE anything is overwritten
    // (a barrier in between: the waves are not in lockstep), then every thread updates its own column.
    for (int pv = 0; pv < nx3; ++pv) {
      const double piv = G[pv * gw + pv];
      if (tid == 0 && (!(piv > 0.0) || !isfinite(piv))) bad_x = 1;
      const double ip = piv != 0.0 ? 1.0 / piv : 0.0;
      const double prow = tid < gw ? G[pv * gw + tid] * ip : 0.0;
      if (tid < nx3) fcol[tid] = G[tid * gw + pv];
      __syncthreads();
      if (tid < gw) {
        for (int r2 = 0; r2 < nx3; ++r2) {
          if (r2 == pv) continue;
          G[r2 * gw + tid] -= fcol[r2] * prow;
        }
        G[pv * gw + tid] = prow;
      }
      __syncthreads();
    }
    // W = S_XX^-1 [S_XT | g_X] sits in the columns nx3 .. gw-1
    for (int e = tid; e < nx3 * (nt3 + 1); e += kOvThreads) O.Wx[e] = G[(e / (nt3 + 1)) * gw + nx3 + e % (nt3 + 1)];
    __syncthreads();
  }
  const bool failed = bad || bad_x != 0;
  // ---- M = sym(S_TT - S_TX W_T) on the touched rows, g_T = gk_T - S_TX W_g into the right-hand sides of dgb
  for (int idx = tid; idx < nt3 * (nt3 + 1); idx += kOvThreads) {
    const int a = idx / (nt3 + 1), bcol = idx % (nt3 + 1);
    double s1 = 0.0, s2 = 0.0;
    for (int r2 = 0; r2 < nx3; ++r2) {
      s1 += O.S[a * nk3 + nt3 + r2] * G[r2 * gw + nx3 + bcol];
      if (bcol < nt3) s2 += O.S[bcol * nk3 + nt3 + r2] * G[r2 * gw + nx3 + a];
    }
    if (bcol < nt3) {
      O.M[a * nt3 + bcol] = 0.5 * ((O.S[a * nk3 + bcol] - s1) + (O.S[bcol * nk3 + a] - s2));
    } else {
      const double g = failed ? __builtin_nan("") : O.gk[a] - s1;
      dgb[9 * (size_t)O.trow[a / 3] + 6 + a % 3] += g;
    }
  }
  if (failed && tid == 0) dgb[6] = __builtin_nan("");   // (also when no resident row is touched: the solve must report the failure)
}

// ---------------------------------------------------------------------------- k_ov_ax
// q_T += M p_T after the base product H_base p; the dot product p . q of the PCG recurrence gets its share added to the
// product's first partial sum (single writer, after the product kernel: fixed order).  One wave.  The sizes come from the
// device-resident header so that a captured hipGraph stays valid across updates.
__global__ __launch_bounds__(kOvThreads) void k_ov_ax(const int* __restrict__ hdr, const int* __restrict__ trow, const double* __restrict__ M,
                                                      const double* __restrict__ p, double* __restrict__ q, double* __restrict__ partials0,
                                                      const PcgScalars* S) {
  __shared__ double pt[3 * kOvMaxTouched];
  __shared__ double ws[kOvThreads / 64];
  if (S && S->stop) return;
  const int nt3 = 3 * hdr[1], r = threadIdx.x;
  if (nt3 == 0) return;
  size_t at = 0;
  if (r < nt3) {
    at = 3 * (size_t)trow[r / 3] + r % 3;
    pt[r] = p[at];
  }
  __syncthreads();
  double d = 0.0;
  if (r < nt3) {
    for (int c2 = 0; c2 < nt3; ++c2) d += M[r * nt3 + c2] * pt[c2];
    q[at] += d;
    d *= pt[r];
  }
  if (partials0) {   // the waves' sums, added in wave order by one thread (fixed order)
    const double s = wave_sum(d);
    if ((r & 63) == 0) ws[r >> 6] = s;
    __syncthreads();
    if (r == 0) {
      double t = ws[0];
      for (int w = 1; w < kOvThreads / 64; ++w) t += ws[w];
      partials0[0] += t;
    }
  }
}

// ---------------------------------------------------------------------------- k_ov_finish
// x_X = W_g - W_T x_T (hubs), x_N = y_b - Y_K [x_T; x_X] (chain rows), then VertexSE2::oplusImpl on the appended poses.
__global__ __launch_bounds__(kOvThreads) void k_ov_finish(OverlayDev O, const double* __restrict__ x, double* __restrict__ poses) {
  __shared__ double xk[3 * kOvMaxTouched];
  const int nt3 = 3 * O.nt, nx3 = 3 * O.nx, nk3 = nt3 + nx3, nc = O.ncol, t = threadIdx.x;
  if (t < nt3) xk[t] = x[3 * (size_t)O.trow[t / 3] + t % 3];
  __syncthreads();
  if (t < nx3) {
    const double* w = O.Wx + (size_t)t * (nt3 + 1);
    double v = w[nt3];
    for (int c2 = 0; c2 < nt3; ++c2) v -= w[c2] * xk[c2];
    xk[nt3 + t] = v;
  }
  __syncthreads();
  if (blockIdx.x == 0 && t < O.nx) {   // the hubs' poses
    const size_t v3 = 3 * (size_t)O.vtx[O.k + t];
    poses[v3] += xk[nt3 + 3 * t];
    poses[v3 + 1] += xk[nt3 + 3 * t + 1];
    poses[v3 + 2] = norm_theta(poses[v3 + 2] + xk[nt3 + 3 * t + 2]);
  }
  for (int i = blockIdx.x * kOvThreads + t; i < O.k; i += gridDim.x * kOvThreads) {
    double xn[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double* y = O.Y + (size_t)(3 * i + a) * nc;
      double v = y[nk3];
      for (int c2 = 0; c2 < nk3; ++c2) v -= y[c2] * xk[c2];
      xn[a] = v;
    }
    const size_t v3 = 3 * (size_t)O.vtx[i];
    poses[v3] += xn[0];
    poses[v3 + 1] += xn[1];
    poses[v3 + 2] = norm_theta(poses[v3 + 2] + xn[2]);
  }
}

}  // namespace

void launch_ov_lin(hipStream_t s, const OverlayDev& O, const double* poses) {
  const int rows = O.k + O.nt + O.nx;
  if (rows > 0) SGO_LAUNCH(k_ov_lin, dim3((rows + kOvThreads - 1) / kOvThreads), dim3(kOvThreads), 0, s, O, poses);
}
void launch_ov_solve(hipStream_t s, const OverlayDev& O, double* dgb) {
  if (O.k + O.nt + O.nx > 0) SGO_LAUNCH(k_ov_solve, dim3(1), dim3(kOvThreads), 0, s, O, dgb);
}
void launch_ov_ax(hipStream_t s, const OverlayDev& O, const double* p, double* q, double* partials0, const PcgScalars* S) {
  SGO_LAUNCH(k_ov_ax, dim3(1), dim3(kOvThreads), 0, s, O.hdr, O.trow, (const double*)O.M, p, q, partials0, S);
}
void launch_ov_finish(hipStream_t s, const OverlayDev& O, const double* x, double* poses) {
  if (O.k + O.nx > 0) SGO_LAUNCH(k_ov_finish, dim3(std::max(1, (O.k + kOvThreads - 1) / kOvThreads)), dim3(kOvThreads), 0, s, O, x, poses);
}

// ---------------------------------------------------------------------------- host side
// Device buffers of an overlay, carved out of one allocation made at first use (capacities of sgo_overlay.h):
//   ints:    header[4] = {k, nt, ncol, nnz} | rp | ent_edge | ent_other | vtx | trow | nz | el.vi | el.vj
//   bytes:   ent_side
//   doubles: el.phi, el.zinv[3], el.info[6], raw meas / info staging, Dn, Un, H0, Y, Sinv, M0, bt, M
namespace {
struct Layout {
  size_t i_hdr, i_rp, i_edge, i_other, i_vtx, i_trow, i_nz, i_vi, i_vj, n_int;
  size_t d_phi, d_zinv, d_info, d_raw, d_Dn, d_Un, d_H0, d_Y, d_Sinv, d_M0, d_bt, d_M, d_S, d_gk, d_Wx, n_dbl;
  Layout() {
    size_t o = 0;
    i_hdr = o; o += 8;
    i_rp = o; o += kOvMaxRows + kOvMaxTouched + 1;
    i_edge = o; o += 2 * kOvMaxEdges;
    i_other = o; o += 2 * kOvMaxEdges;
    i_vtx = o; o += kOvMaxRows + kOvMaxTouched;
    i_trow = o; o += kOvMaxTouched;
    i_nz = o; o += kOvMaxRows;
    n_int = o;                       // the structure part (uploaded per update) ends here
    i_vi = o; o += kOvMaxEdges;
    i_vj = o; o += kOvMaxEdges;
    const size_t ints = o;
    (void)ints;
    o = 0;
    d_phi = o; o += kOvMaxEdges;
    d_zinv = o; o += 3 * (size_t)kOvMaxEdges;
    d_info = o; o += 6 * (size_t)kOvMaxEdges;
    d_raw = o; o += 9 * (size_t)kOvMaxEdges;
    d_Dn = o; o += 6 * (size_t)kOvMaxRows;
    d_Un = o; o += 9 * (size_t)kOvMaxRows;
    d_H0 = o; o += 3 * (size_t)kOvMaxRows * kOvCols;
    d_Y = o; o += 3 * (size_t)kOvMaxRows * kOvCols;
    d_Sinv = o; o += 6 * (size_t)kOvMaxRows;
    d_M0 = o; o += (size_t)(kOvCols - 1) * (kOvCols - 1);
    d_bt = o; o += kOvCols;
    d_M = o; o += (size_t)(kOvCols - 1) * (kOvCols - 1);
    d_S = o; o += (size_t)(kOvCols - 1) * (kOvCols - 1);
    d_gk = o; o += kOvCols;
    d_Wx = o; o += (size_t)(kOvCols - 1) * kOvCols;
    n_dbl = o;
  }
  size_t int_total() const { return i_vj + kOvMaxEdges; }
};
const Layout& layout() {
  static const Layout L;
  return L;
}
}  // namespace

static bool overlay_alloc(Overlay& ov, std::string* err) {
  if (ov.buf) return true;
  const Layout& L = layout();
  const size_t bytes = sizeof(double) * L.n_dbl + sizeof(int) * L.int_total() + 2 * (size_t)kOvMaxEdges + 256;
  if (hipMalloc(&ov.buf, bytes) != hipSuccess) {
    ov.buf = nullptr;
    if (err) *err = "out of device memory (incremental set-up buffers)";
    return false;
  }
  if (hipHostMalloc((void**)&ov.h_int, sizeof(int) * L.n_int + 2 * (size_t)kOvMaxEdges) != hipSuccess ||
      hipHostMalloc((void**)&ov.h_edge, (sizeof(double) * 10 + sizeof(int) * 2) * (size_t)kOvMaxEdges) != hipSuccess) {
    if (ov.h_int) hipHostFree(ov.h_int);
    hipFree(ov.buf);
    ov.buf = nullptr;
    ov.h_int = nullptr;
    if (err) *err = "out of pinned host memory (incremental set-up buffers)";
    return false;
  }
  ov.h_side = (unsigned char*)(ov.h_int + L.n_int);
  double* d = (double*)ov.buf;
  ov.d_int = (int*)(d + L.n_dbl);
  ov.d_side = (unsigned char*)(ov.d_int + L.int_total());
  OverlayDev& O = ov.dev;
  O.el.E = kOvMaxEdges;
  O.el.cnt = 0;
  O.el.vi = ov.d_int + L.i_vi;
  O.el.vj = ov.d_int + L.i_vj;
  O.el.phi = d + L.d_phi;
  O.el.zinv = d + L.d_zinv;
  O.el.info = d + L.d_info;
  O.hdr = ov.d_int + L.i_hdr;
  O.rp = ov.d_int + L.i_rp;
  O.ent_edge = ov.d_int + L.i_edge;
  O.ent_other = ov.d_int + L.i_other;
  O.ent_side = ov.d_side;
  O.vtx = ov.d_int + L.i_vtx;
  O.trow = ov.d_int + L.i_trow;
  O.nz = ov.d_int + L.i_nz;
  O.Dn = d + L.d_Dn;
  O.Un = d + L.d_Un;
  O.H0 = d + L.d_H0;
  O.Y = d + L.d_Y;
  O.Sinv = d + L.d_Sinv;
  O.M0 = d + L.d_M0;
  O.bt = d + L.d_bt;
  O.M = d + L.d_M;
  O.S = d + L.d_S;
  O.gk = d + L.d_gk;
  O.Wx = d + L.d_Wx;
  return true;
}

void overlay_release(Overlay& ov) {
  if (ov.buf) hipFree(ov.buf);
  if (ov.h_int) hipHostFree(ov.h_int);
  if (ov.h_edge) hipHostFree(ov.h_edge);
  ov = Overlay();
}

bool overlay_upload_edges(Overlay& ov, hipStream_t s, int at, int cnt, const int32_t* ei, const int32_t* ej, const double* meas,
                          const double* info, const double* phi, std::string* err) {
  if (!overlay_alloc(ov, err)) return false;
  if (cnt <= 0) return true;
  const Layout& L = layout();
  double* d = (double*)ov.buf;
  double* raw = d + L.d_raw;
  // through pinned staging (the caller has drained the stream: the previous update's copies have left it)
  double* hm = ov.h_edge;                              // meas [3 cnt] | info [6 cnt] | phi [cnt] | ei, ej
  double* hi6 = hm + 3 * (size_t)kOvMaxEdges;
  double* hp = hi6 + 6 * (size_t)kOvMaxEdges;
  int* hv = (int*)(hp + kOvMaxEdges);
  std::memcpy(hm, meas, sizeof(double) * 3 * (size_t)cnt);
  std::memcpy(hi6, info, sizeof(double) * 6 * (size_t)cnt);
  std::memcpy(hp, phi, sizeof(double) * (size_t)cnt);
  std::memcpy(hv, ei, sizeof(int32_t) * (size_t)cnt);
  std::memcpy(hv + kOvMaxEdges, ej, sizeof(int32_t) * (size_t)cnt);
  hipError_t e = hipMemcpyAsync(ov.dev.el.vi + at, hv, sizeof(int32_t) * (size_t)cnt, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(ov.dev.el.vj + at, hv + kOvMaxEdges, sizeof(int32_t) * (size_t)cnt, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(ov.dev.el.phi + at, hp, sizeof(double) * (size_t)cnt, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(raw, hm, sizeof(double) * 3 * (size_t)cnt, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(raw + 3 * (size_t)kOvMaxEdges, hi6, sizeof(double) * 6 * (size_t)cnt, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) {
    if (err) *err = std::string("incremental set-up: edge upload: ") + hipGetErrorString(e);
    return false;
  }
  launch_edge_prepare(s, cnt, raw, raw + 3 * (size_t)kOvMaxEdges, ov.dev.el.zinv, ov.dev.el.info, (size_t)kOvMaxEdges, (size_t)at);
  return true;
}

bool overlay_build(Overlay& ov, int V, hipStream_t s, std::string* why, std::string* err) {
  const int ne = (int)ov.ei.size();
  if (ne > kOvMaxEdges) { *why = "more appended edges than the overlay holds"; return false; }
  if (!overlay_alloc(ov, err)) { *why = "no device memory"; return false; }
  // classes of the endpoints: base row (touched), fixed, or a new row
  auto base_row = [&](int v) { return v < ov.base_V ? ov.hpos[v] : -1; };
  std::vector<int> nv, tr;
  for (int e = 0; e < ne; ++e)
    for (int v : {ov.ei[e], ov.ej[e]}) {
      if (ov.fixed[v]) continue;
      const int br = base_row(v);
      if (br >= 0) tr.push_back(br);
      else nv.push_back(v);
    }
  std::sort(nv.begin(), nv.end());
  nv.erase(std::unique(nv.begin(), nv.end()), nv.end());
  std::sort(tr.begin(), tr.end());
  tr.erase(std::unique(tr.begin(), tr.end()), tr.end());
  const int nnew = (int)nv.size(), nt = (int)tr.size();
  auto new_idx = [&](int v) { return (int)(std::lower_bound(nv.begin(), nv.end(), v) - nv.begin()); };
  // hubs: an edge between two appended poses that are not neighbours in the chain takes its later endpoint out of the chain
  std::vector<char> is_hub((size_t)nnew, 0);
  for (int e = 0; e < ne; ++e) {
    const int a2 = ov.ei[e], b2 = ov.ej[e];
    if (ov.fixed[a2] || ov.fixed[b2] || base_row(a2) >= 0 || base_row(b2) >= 0) continue;
    const int ia = new_idx(a2), ib = new_idx(b2);
    if (std::abs(ia - ib) != 1 && !is_hub[ia] && !is_hub[ib]) is_hub[std::max(ia, ib)] = 1;
  }
  std::vector<int> chain_pos((size_t)nnew, -1), hub_pos((size_t)nnew, -1), chain_v, hub_v;
  for (int i = 0; i < nnew; ++i) {
    if (is_hub[i]) {
      hub_pos[i] = (int)hub_v.size();
      hub_v.push_back(nv[i]);
    } else {
      chain_pos[i] = (int)chain_v.size();
      chain_v.push_back(nv[i]);
    }
  }
  const int k = (int)chain_v.size(), nx = (int)hub_v.size(), nk = nt + nx;
  if (nnew > kOvMaxRows) { *why = "more appended poses than the overlay holds"; return false; }
  if (nk > kOvMaxTouched) { *why = "the appended edges end in more resident rows (and hub poses) than the overlay holds"; return false; }
  if (nx > kOvMaxHubs) { *why = "more hub poses among the appended ones than the overlay holds"; return false; }
  auto code = [&](int v) -> int {   // >= 0 chain row, -1 - t kept row t (touched base rows, then hubs), kOvOtherFixed
    if (ov.fixed[v]) return kOvOtherFixed;
    const int br = base_row(v);
    if (br >= 0) return -1 - (int)(std::lower_bound(tr.begin(), tr.end(), br) - tr.begin());
    const int i = new_idx(v);
    return is_hub[i] ? -1 - (nt + hub_pos[i]) : chain_pos[i];
  };
  const Layout& L = layout();
  int* hi = ov.h_int;              // (the stream was drained by the caller: the previous update's copy has long finished)
  unsigned char* hs = ov.h_side;
  int* rp = hi + L.i_rp;
  std::vector<int> ca(ne), cb(ne);
  std::vector<int> cntr((size_t)k + nk + 1, 0);
  auto rowof = [&](int c) { return c >= 0 ? c : k + (-1 - c); };
  for (int e = 0; e < ne; ++e) {
    ca[e] = code(ov.ei[e]);
    cb[e] = code(ov.ej[e]);
    if (ca[e] >= 0 && cb[e] >= 0 && std::abs(ca[e] - cb[e]) != 1) {
      *why = "appended edges among the new poses do not form chain segments";   // (cannot happen: such an edge has a hub endpoint)
      return false;
    }
    if (ca[e] == kOvOtherFixed && cb[e] == kOvOtherFixed) continue;   // (never active: the marshal layer drops such edges)
    if (ca[e] != kOvOtherFixed) cntr[rowof(ca[e]) + 1]++;
    if (cb[e] != kOvOtherFixed) cntr[rowof(cb[e]) + 1]++;
  }
  for (int r = 0; r < k + nk; ++r) cntr[r + 1] += cntr[r];
  std::copy(cntr.begin(), cntr.end(), rp);
  std::vector<int> fill(cntr.begin(), cntr.end() - 1);
  int* ent_edge = hi + L.i_edge;
  int* ent_other = hi + L.i_other;
  for (int e = 0; e < ne; ++e) {   // entries in edge order: every row's sums are taken in that order
    if (ca[e] != kOvOtherFixed) {
      const int t = fill[rowof(ca[e])]++;
      ent_edge[t] = e; ent_other[t] = cb[e]; hs[t] = 0;
    }
    if (cb[e] != kOvOtherFixed) {
      const int t = fill[rowof(cb[e])]++;
      ent_edge[t] = e; ent_other[t] = ca[e]; hs[t] = 1;
    }
  }
  int* vtx = hi + L.i_vtx;
  for (int i = 0; i < k; ++i) vtx[i] = chain_v[i];
  for (int j = 0; j < nx; ++j) vtx[k + j] = hub_v[j];
  // (vertex ids of the touched base rows are not needed: their poses are read through the edges' endpoints)
  int* trow = hi + L.i_trow;
  for (int t = 0; t < nt; ++t) trow[t] = tr[t];
  int nnz = 0;
  int* nz = hi + L.i_nz;
  for (int i = 0; i < k; ++i) {
    bool any = false;
    for (int t = rp[i]; t < rp[i + 1]; ++t) any = any || (ent_other[t] < 0 && ent_other[t] != kOvOtherFixed);
    if (any) nz[nnz++] = i;
  }
  hi[L.i_hdr] = k; hi[L.i_hdr + 1] = nt; hi[L.i_hdr + 2] = 3 * nk + 1; hi[L.i_hdr + 3] = nnz; hi[L.i_hdr + 4] = nx;
  // only what is used travels: header + row pointers, the entries, vertex / row lists
  const size_t nent = (size_t)cntr[(size_t)k + nk];
  hipError_t e1 = hipMemcpyAsync(ov.d_int, hi, sizeof(int) * (L.i_rp + (size_t)k + nk + 1), hipMemcpyHostToDevice, s);
  auto up = [&](size_t at, size_t cnt) {
    if (e1 == hipSuccess && cnt > 0) e1 = hipMemcpyAsync(ov.d_int + at, hi + at, sizeof(int) * cnt, hipMemcpyHostToDevice, s);
  };
  up(L.i_edge, nent);
  up(L.i_other, nent);
  up(L.i_vtx, (size_t)k + nx);
  up(L.i_trow, (size_t)nt);
  up(L.i_nz, (size_t)nnz);
  if (e1 == hipSuccess && nent > 0) e1 = hipMemcpyAsync(ov.d_side, hs, nent, hipMemcpyHostToDevice, s);
  if (e1 != hipSuccess) {
    if (err) *err = std::string("incremental set-up: structure upload: ") + hipGetErrorString(e1);
    *why = "upload failed";
    return false;
  }
  OverlayDev& O = ov.dev;
  O.k = k; O.nt = nt; O.nx = nx; O.ncol = 3 * nk + 1; O.nnz = nnz;
  O.el.cnt = ne;
  ov.new_vertex = nv;
  (void)V;
  return true;
}

}  // namespace sgo
