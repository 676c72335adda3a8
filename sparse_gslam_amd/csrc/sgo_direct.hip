// sgo_direct.hip -- the small-graph path (sgo_direct.h): optimize(iters) in one launch of one workgroup.
//
// Replaces, for graphs of the reference's own size, the whole loop of SparseOptimizer::optimize(20)
// (slc.cpp:287): computeActiveErrors + buildSystem (EdgeSE2::computeError / linearizeOplus /
// constructQuadraticForm with RobustKernelDCS), LinearSolverEigen::solve (graphs.cpp:19: a sparse LDL^T) and
// SparseOptimizer::update, iters times, + the closing computeActiveErrors (slc.cpp:288).
//
// Elimination order (host, direct_create).  A pose graph is a trajectory chain + loop closures.  A greedy
// vertex cover of the non-chain edges gives the SEPARATORS (every closure has an endpoint there); what is
// left is a set of chain segments, eliminated by cyclic reduction (every other vertex of a segment per level:
// log2 of the segment length levels, all columns of a level independent), then the separators as one dense
// block.  The symbolic factorisation is generic (elimination tree, column structures, fill) -- the chain
// only makes the tree shallow.
//
// Numeric phase (device, k_direct; 1024 threads, one CU).  Sparse columns keep W_ik = (updated) H_ik and
// the updated diagonal block D_k; L_ik = W_ik D_k^-1 is never stored: every use recomputes the 3x3 inverse.
//   level l, forward : every TARGET block of the level's columns' Schur updates (and every right-hand-side
//                      entry) is owned by one thread that sums its contributions W_ik D_k^-1 W_jk^T in a
//                      fixed order (gather lists from the host: no atomics, bitwise reproducible); the
//                      targets live in higher levels, the sources in this one: ONE barrier per level.
//   separators       : packed lower triangle in LDS, right-looking block LDL^T with 3x3 pivots, one
//                      barrier per pivot; back substitution by one wave (DPP reductions, no barriers).
//   level l, backward: one lane per stored block, x_k = D_k^-1 (b_k - sum_i W_ik^T x_i) by a wavefront
//                      segmented scan; x and b live in LDS.
// Sequential depth per Gauss-Newton iteration = 2 (levels + separators) steps of ~0.3-2 us instead of
// ~150 launches.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

#include "sgo_device.h"
#include "sgo_direct.h"
#include "sgo_internal.h"

namespace sgo {

namespace {

constexpr int kDT = 1024;               // threads of the workgroup
constexpr int kMaxSep = 50;             // separators: 150 x 150 packed triangle = 91 KB of LDS
constexpr int kMaxLevels = 24;
constexpr int kMaxEdges = 1 << 15;
constexpr int kMaxColumn = 63;          // stored blocks of one sparse column (a wave holds a column's segment)
constexpr int kMaxContrib = 1 << 20;
constexpr size_t kLdsBudget = 150 * 1024;

enum : unsigned { T_OFF = 0, T_DIAG = 1, T_DENSE = 2, T_RHS = 3 };

struct DirectDev {
  int n, nI, ns, NL, E, NB, tri, nss;
  const int* vertex_pos;      // [V] elimination position of a vertex, -1 fixed
  const int* pos_vertex;      // [n]
  const int* vptr;            // [n + 1] incident edges of a position ...
  const int* vlist;           // ... (edge << 1 | side)
  const int* slot_row;        // [NB] row position of a stored block (-1: dummy slot of an empty column)
  const int* slot_col;        // [NB] column position (-1: padding)
  const int* slot_edge;       // [NB] first edge of the pair (edge << 1 | transposed), -1 fill
  const int* enext;           // [E] next edge of the same pair, same encoding
  const int* ss_pair;         // [nss] separator pairs with edges: (si << 12 | sj), si > sj
  const int* ss_edge;         // [nss]
  const int* lslot;           // [NL + 1] slot range of a level (multiples of 64)
  const int* ltask;           // [NL + 1] task range of a level
  const unsigned* tk_target;  // kind << 28 | index
  const int* tk_cptr;         // contributions of a task
  const int* ca;              // slot of W_ik
  const int* cb;              // slot of W_jk
  const unsigned short* dpair;  // dense block pairs (bi << 8 | bj), bi >= bj, sorted by bj descending
  double* Wd;                 // [nI][9] diagonal blocks of the sparse columns
  double* Wo;                 // [NB][9] stored blocks
  double* escr;               // [E][27] per-edge terms of the current linearisation
};

// inverse of a symmetric 3x3 (d00 d01 d02 d11 d12 d22); false when the block is not positive definite
__device__ __forceinline__ bool inv_sym3(double d00, double d01, double d02, double d11, double d12, double d22,
                                         double (&iv)[6]) {
  const double c00 = d11 * d22 - d12 * d12, c01 = d02 * d12 - d01 * d22, c02 = d01 * d12 - d02 * d11;
  const double c11 = d00 * d22 - d02 * d02, c12 = d01 * d02 - d00 * d12, c22 = d00 * d11 - d01 * d01;
  const double det = d00 * c00 + d01 * c01 + d02 * c02;
  const double id = 1.0 / det;
  iv[0] = c00 * id; iv[1] = c01 * id; iv[2] = c02 * id; iv[3] = c11 * id; iv[4] = c12 * id; iv[5] = c22 * id;
  return d00 > 0.0 && c22 > 0.0 && det > 0.0 && isfinite(det);
}
// T = W * S, W row-major 3x3, S symmetric (6)
__device__ __forceinline__ void mul_sym(const double (&W)[9], const double (&S)[6], double (&T)[9]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double a = W[3 * r], b = W[3 * r + 1], c = W[3 * r + 2];
    T[3 * r] = a * S[0] + b * S[1] + c * S[2];
    T[3 * r + 1] = a * S[1] + b * S[3] + c * S[4];
    T[3 * r + 2] = a * S[2] + b * S[4] + c * S[5];
  }
}
__device__ __forceinline__ void load9g(const double* __restrict__ p, double (&v)[9]) {
#pragma unroll
  for (int c = 0; c < 9; ++c) v[c] = p[c];
}
__device__ __forceinline__ double readlane63(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int tri_at(int i, int j) { return i * (i + 1) / 2 + j; }   // i >= j

// Everything an edge contributes at the current poses: rec = Hii(6) bi(3) Hjj(6) bj(3) Hij(9)
// (EdgeSE2::computeError, linearizeOplus, RobustKernelDCS::robustify, constructQuadraticForm; same arithmetic as
// k_linearize).  Returns e2 and rho0 for the chi2 sums.
__device__ __forceinline__ void edge_terms(const EdgeListDev& el, int k, const double* __restrict__ poses, bool jac,
                                           double* __restrict__ rec, double* e2_out, double* rho_out) {
  const size_t E = (size_t)el.E;
  const int vi = el.vi[k], vj = el.vj[k];
  const double xi = poses[3 * (size_t)vi], yi = poses[3 * (size_t)vi + 1], ti = poses[3 * (size_t)vi + 2];
  const double xj = poses[3 * (size_t)vj], yj = poses[3 * (size_t)vj + 1], tj = poses[3 * (size_t)vj + 2];
  const double zx = el.zinv[k], zy = el.zinv[E + k], zt = el.zinv[2 * E + k];
  double sz, cz;
  sincos(zt, &sz, &cz);
  double e[3];
  edge_error(xi, yi, ti, xj, yj, tj, zx, zy, zt, sz, cz, e);
  const double o00 = el.info[k], o01 = el.info[E + k], o02 = el.info[2 * E + k];
  const double o11 = el.info[3 * E + k], o12 = el.info[4 * E + k], o22 = el.info[5 * E + k];
  double oe0 = o00 * e[0] + o01 * e[1] + o02 * e[2];
  double oe1 = o01 * e[0] + o11 * e[1] + o12 * e[2];
  double oe2 = o02 * e[0] + o12 * e[1] + o22 * e[2];
  const double e2 = e[0] * oe0 + e[1] * oe1 + e[2] * oe2;
  double r0, w;
  dcs(e2, el.phi[k], &r0, &w);
  *e2_out = e2;
  *rho_out = r0;
  if (!jac) return;
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
  // TA = Ow A (A's third row is (0, 0, -1)), TB = Ow B (third row (0, 0, 1), B02 = B12 = 0)
  const double TA00 = w00 * A00 + w01 * A10, TA01 = w00 * A01 + w01 * A11, TA02 = w00 * A02 + w01 * A12 - w02;
  const double TA10 = w01 * A00 + w11 * A10, TA11 = w01 * A01 + w11 * A11, TA12 = w01 * A02 + w11 * A12 - w12;
  const double TA20 = w02 * A00 + w12 * A10, TA21 = w02 * A01 + w12 * A11, TA22 = w02 * A02 + w12 * A12 - w22;
  const double TB00 = w00 * B00 + w01 * B10, TB01 = w00 * B01 + w01 * B11, TB02 = w02;
  const double TB10 = w01 * B00 + w11 * B10, TB11 = w01 * B01 + w11 * B11, TB12 = w12;
  const double TB22 = w22;
  // Hii = A^T TA
  rec[0] = A00 * TA00 + A10 * TA10;
  rec[1] = A00 * TA01 + A10 * TA11;
  rec[2] = A00 * TA02 + A10 * TA12;
  rec[3] = A01 * TA01 + A11 * TA11;
  rec[4] = A01 * TA02 + A11 * TA12;
  rec[5] = A02 * TA02 + A12 * TA12 - TA22;
  // bi = -A^T (Ow e)
  rec[6] = -(A00 * oe0 + A10 * oe1);
  rec[7] = -(A01 * oe0 + A11 * oe1);
  rec[8] = -(A02 * oe0 + A12 * oe1 - oe2);
  // Hjj = B^T TB
  rec[9] = B00 * TB00 + B10 * TB10;
  rec[10] = B00 * TB01 + B10 * TB11;
  rec[11] = B00 * TB02 + B10 * TB12;
  rec[12] = B01 * TB01 + B11 * TB11;
  rec[13] = B01 * TB02 + B11 * TB12;
  rec[14] = TB22;
  // bj = -B^T (Ow e)
  rec[15] = -(B00 * oe0 + B10 * oe1);
  rec[16] = -(B01 * oe0 + B11 * oe1);
  rec[17] = -oe2;
  // Hij = A^T Ow B = TA^T B  (row vi, column vj)
  rec[18] = TA00 * B00 + TA10 * B10;
  rec[19] = TA00 * B01 + TA10 * B11;
  rec[20] = TA20;
  rec[21] = TA01 * B00 + TA11 * B10;
  rec[22] = TA01 * B01 + TA11 * B11;
  rec[23] = TA21;
  rec[24] = TA02 * B00 + TA12 * B10;
  rec[25] = TA02 * B01 + TA12 * B11;
  rec[26] = TA22;
}

// sum over the edges of a pair (linked through enext) of H[row][col], row-major
__device__ __forceinline__ void pair_block(const DirectDev& D, int first, double (&b)[9]) {
#pragma unroll
  for (int c = 0; c < 9; ++c) b[c] = 0.0;
  for (int t = first; t >= 0; t = D.enext[t >> 1]) {
    const double* h = D.escr + 27 * (size_t)(t >> 1) + 18;
    if (t & 1) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) b[3 * r + c] += h[3 * c + r];
    } else {
#pragma unroll
      for (int c = 0; c < 9; ++c) b[c] += h[c];
    }
  }
}

__global__ __launch_bounds__(kDT) void k_direct(DirectDev D, EdgeListDev el, double* __restrict__ poses, int iters,
                                                double* __restrict__ hist, DirectResult* __restrict__ res) {
  extern __shared__ double lds[];
  double* xb = lds;                       // [3 n] right-hand side, then the solution, by elimination position
  double* Sd = lds + 3 * (size_t)D.n;     // [tri] packed lower triangle of the separator block
  double* red = Sd + D.tri;               // [2][16] chi2 partials
  __shared__ int fail_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = D.n, nI = D.nI, ns = D.ns;
  if (tid == 0) fail_flag = 0;
  int done = 0, fail = 0;
  for (int it = 0; it <= iters; ++it) {
    if (tid == 0) res->stamp[2 * it] = (unsigned long long)wall_clock64();
    // ---- edges: chi2 sums, and (unless this is the closing pass) the terms of the linearisation
    const bool jac = it < iters;
    double acc[2] = {0.0, 0.0};
    for (int k = tid; k < D.E; k += kDT) {
      double e2, r0;
      edge_terms(el, k, poses, jac, D.escr + 27 * (size_t)k, &e2, &r0);
      acc[0] += e2;
      acc[1] += r0;
    }
    seg_scan<2>(0, acc, lane);
    if (lane == 63) {
      red[wave] = acc[0];
      red[16 + wave] = acc[1];
    }
    __syncthreads();   // escr visible; red complete
    if (tid == 0) {
      double a = 0.0, b = 0.0;
      for (int w = 0; w < kDT / 64; ++w) {
        a += red[w];
        b += red[16 + w];
      }
      hist[2 * it] = a;
      hist[2 * it + 1] = b;
    }
    if (!jac) break;
    // ---- assembly
    for (int k = tid; k < D.tri; k += kDT) Sd[k] = 0.0;
    __syncthreads();
    for (int f = tid; f < n; f += kDT) {
      double d[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int q = D.vptr[f]; q < D.vptr[f + 1]; ++q) {
        const int t = D.vlist[q];
        const double* h = D.escr + 27 * (size_t)(t >> 1) + ((t & 1) ? 9 : 0);
#pragma unroll
        for (int c = 0; c < 9; ++c) d[c] += h[c];
      }
      xb[3 * f] = d[6];
      xb[3 * f + 1] = d[7];
      xb[3 * f + 2] = d[8];
      if (f < nI) {
        double* w = D.Wd + 9 * (size_t)f;
        w[0] = d[0]; w[1] = d[1]; w[2] = d[2];
        w[3] = d[1]; w[4] = d[3]; w[5] = d[4];
        w[6] = d[2]; w[7] = d[4]; w[8] = d[5];
      } else {
        const int r = 3 * (f - nI);
        Sd[tri_at(r, r)] = d[0];
        Sd[tri_at(r + 1, r)] = d[1];
        Sd[tri_at(r + 1, r + 1)] = d[3];
        Sd[tri_at(r + 2, r)] = d[2];
        Sd[tri_at(r + 2, r + 1)] = d[4];
        Sd[tri_at(r + 2, r + 2)] = d[5];
      }
    }
    for (int s = tid; s < D.NB; s += kDT) {
      if (D.slot_col[s] < 0 || D.slot_row[s] < 0) continue;
      double b[9];
      pair_block(D, D.slot_edge[s], b);
      double* w = D.Wo + 9 * (size_t)s;
#pragma unroll
      for (int c = 0; c < 9; ++c) w[c] = b[c];
    }
    for (int q = tid; q < D.nss; q += kDT) {
      const int pr = D.ss_pair[q], si = pr >> 12, sj = pr & 4095;
      double b[9];
      pair_block(D, D.ss_edge[q], b);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Sd[tri_at(3 * si + r, 3 * sj + c)] = b[3 * r + c];
    }
    __syncthreads();
    if (tid == 0) res->stamp[2 * it + 1] = (unsigned long long)wall_clock64();
    // ---- sparse levels, forward: Schur updates and right-hand sides of the level's columns
    for (int l = 0; l < D.NL; ++l) {
      const int t0 = D.ltask[l], t1 = D.ltask[l + 1];
      for (int T = t0 + tid; T < t1; T += kDT) {
        const unsigned tg = D.tk_target[T];
        const unsigned kind = tg >> 28;
        const int idx = (int)(tg & 0x0fffffffu);
        double a9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = D.tk_cptr[T]; c < D.tk_cptr[T + 1]; ++c) {
          const int sa = D.ca[c];
          const int k = D.slot_col[sa];
          double Wa[9], Dk[9], iv[6], Tm[9];
          load9g(D.Wo + 9 * (size_t)sa, Wa);
          load9g(D.Wd + 9 * (size_t)k, Dk);
          inv_sym3(Dk[0], Dk[1], Dk[2], Dk[4], Dk[5], Dk[8], iv);
          mul_sym(Wa, iv, Tm);
          if (kind == T_RHS) {
            const double b0 = xb[3 * k], b1 = xb[3 * k + 1], b2 = xb[3 * k + 2];
#pragma unroll
            for (int r = 0; r < 3; ++r) a9[r] += Tm[3 * r] * b0 + Tm[3 * r + 1] * b1 + Tm[3 * r + 2] * b2;
          } else {
            double Wb[9];
            load9g(D.Wo + 9 * (size_t)D.cb[c], Wb);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int q = 0; q < 3; ++q)
                a9[3 * r + q] += Tm[3 * r] * Wb[3 * q] + Tm[3 * r + 1] * Wb[3 * q + 1] + Tm[3 * r + 2] * Wb[3 * q + 2];
          }
        }
        if (kind == T_RHS) {
          xb[3 * idx] -= a9[0];
          xb[3 * idx + 1] -= a9[1];
          xb[3 * idx + 2] -= a9[2];
        } else if (kind == T_DENSE) {
          const int si = idx >> 12, sj = idx & 4095;
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q)
              if (si != sj || r >= q) Sd[tri_at(3 * si + r, 3 * sj + q)] -= a9[3 * r + q];
        } else {
          double* w = (kind == T_DIAG ? D.Wd : D.Wo) + 9 * (size_t)idx;
#pragma unroll
          for (int c = 0; c < 9; ++c) w[c] -= a9[c];
        }
      }
      __syncthreads();
    }
    // ---- separators: right-looking block LDL^T on the packed triangle, pivot block p; W form (panel kept)
    double* bs = xb + 3 * (size_t)nI;
    for (int p = 0; p + 1 < ns; ++p) {
      const int m = ns - 1 - p, cnt = m * (m + 1) / 2;
      const int P = 3 * p;
      double iv[6];
      inv_sym3(Sd[tri_at(P, P)], Sd[tri_at(P + 1, P)], Sd[tri_at(P + 2, P)], Sd[tri_at(P + 1, P + 1)],
               Sd[tri_at(P + 2, P + 1)], Sd[tri_at(P + 2, P + 2)], iv);
      for (int t = tid; t < cnt + m; t += kDT) {
        if (t < cnt) {
          const int pr = D.dpair[t], bi = pr >> 8, bj = pr & 255;
          double Wa[9], Wb[9], Tm[9];
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              Wa[3 * r + q] = Sd[tri_at(3 * bi + r, P + q)];
              Wb[3 * r + q] = Sd[tri_at(3 * bj + r, P + q)];
            }
          mul_sym(Wa, iv, Tm);
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q)
              if (bi != bj || r >= q)
                Sd[tri_at(3 * bi + r, 3 * bj + q)] -= Tm[3 * r] * Wb[3 * q] + Tm[3 * r + 1] * Wb[3 * q + 1] + Tm[3 * r + 2] * Wb[3 * q + 2];
        } else {
          const int bi = p + 1 + (t - cnt);
          double Wa[9], Tm[9];
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) Wa[3 * r + q] = Sd[tri_at(3 * bi + r, P + q)];
          mul_sym(Wa, iv, Tm);
          const double b0 = bs[P], b1 = bs[P + 1], b2 = bs[P + 2];
#pragma unroll
          for (int r = 0; r < 3; ++r) bs[3 * bi + r] -= Tm[3 * r] * b0 + Tm[3 * r + 1] * b1 + Tm[3 * r + 2] * b2;
        }
      }
      __syncthreads();
    }
    // back substitution of the separators by ONE wave: x_p = D_p^-1 (b_p - sum_{i > p} W_ip^T x_i)
    if (wave == 0) {
      for (int p = ns - 1; p >= 0; --p) {
        const int P = 3 * p;
        double v[3] = {0.0, 0.0, 0.0};
        for (int bi = p + 1 + lane; bi < ns; bi += 64) {
          const double x0 = bs[3 * bi], x1 = bs[3 * bi + 1], x2 = bs[3 * bi + 2];
#pragma unroll
          for (int q = 0; q < 3; ++q)
            v[q] += Sd[tri_at(3 * bi, P + q)] * x0 + Sd[tri_at(3 * bi + 1, P + q)] * x1 + Sd[tri_at(3 * bi + 2, P + q)] * x2;
        }
        seg_scan<3>(0, v, lane);
        const double s0 = readlane63(v[0]), s1 = readlane63(v[1]), s2 = readlane63(v[2]);
        double iv[6];
        const bool ok = inv_sym3(Sd[tri_at(P, P)], Sd[tri_at(P + 1, P)], Sd[tri_at(P + 2, P)], Sd[tri_at(P + 1, P + 1)],
                                 Sd[tri_at(P + 2, P + 1)], Sd[tri_at(P + 2, P + 2)], iv);
        const double r0 = bs[P] - s0, r1 = bs[P + 1] - s1, r2 = bs[P + 2] - s2;
        // (the LDS unit serves one wave's operations in issue order: every lane's read of b_p precedes this store,
        // and the store precedes the next step's reads)
        if (lane == 0) {
          bs[P] = iv[0] * r0 + iv[1] * r1 + iv[2] * r2;
          bs[P + 1] = iv[1] * r0 + iv[3] * r1 + iv[4] * r2;
          bs[P + 2] = iv[2] * r0 + iv[4] * r1 + iv[5] * r2;
          if (!ok) atomicOr(&fail_flag, 1);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    // ---- sparse levels, backward
    for (int l = D.NL - 1; l >= 0; --l) {
      const int s0 = D.lslot[l], s1 = D.lslot[l + 1];
      for (int s = s0 + tid; s < s1; s += kDT) {
        const int col = D.slot_col[s], row = D.slot_row[s];
        double v[3] = {0.0, 0.0, 0.0};
        double Dk[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (col >= 0) load9g(D.Wd + 9 * (size_t)col, Dk);
        if (col >= 0 && row >= 0) {
          double W[9];
          load9g(D.Wo + 9 * (size_t)s, W);
          const double x0 = xb[3 * row], x1 = xb[3 * row + 1], x2 = xb[3 * row + 2];
#pragma unroll
          for (int q = 0; q < 3; ++q) v[q] = W[q] * x0 + W[3 + q] * x1 + W[6 + q] * x2;
        }
        const int key = col >= 0 ? col : -1 - lane;
        seg_scan<3>(key, v, lane);
        const int nk = __shfl_down(key, 1);
        if (col >= 0 && (lane == 63 || nk != key)) {
          double iv[6];
          const bool ok = inv_sym3(Dk[0], Dk[1], Dk[2], Dk[4], Dk[5], Dk[8], iv);
          const double r0 = xb[3 * col] - v[0], r1 = xb[3 * col + 1] - v[1], r2 = xb[3 * col + 2] - v[2];
          xb[3 * col] = iv[0] * r0 + iv[1] * r1 + iv[2] * r2;
          xb[3 * col + 1] = iv[1] * r0 + iv[3] * r1 + iv[4] * r2;
          xb[3 * col + 2] = iv[2] * r0 + iv[4] * r1 + iv[5] * r2;
          if (!ok) atomicOr(&fail_flag, 1);
        }
      }
      __syncthreads();
    }
    // ---- update (VertexSE2::oplusImpl) unless the factorisation failed or the step is not finite
    bool bad = false;
    for (int f = tid; f < n; f += kDT) bad |= !(isfinite(xb[3 * f]) && isfinite(xb[3 * f + 1]) && isfinite(xb[3 * f + 2]));
    if (bad) atomicOr(&fail_flag, 2);
    __syncthreads();
    fail = fail_flag;
    if (fail) break;
    for (int f = tid; f < n; f += kDT) {
      const size_t v = 3 * (size_t)D.pos_vertex[f];
      poses[v] += xb[3 * f];
      poses[v + 1] += xb[3 * f + 1];
      poses[v + 2] = norm_theta(poses[v + 2] + xb[3 * f + 2]);
    }
    ++done;
    __syncthreads();
  }
  if (tid == 0) {
    if (fail) {
      // hist[2 done] already holds the chi2 at the poses that stay (the failed iteration's own start)
      res->fail_iter = done;
    }
    res->done = done;
    res->fail = (fail & 1) ? 1 : (fail ? 2 : 0);
    res->stamp[fail ? 2 * done + 2 : 2 * iters + 1] = (unsigned long long)wall_clock64();   // end of the call
  }
}

// ------------------------------------------------------------------------------------------------ host
template <typename T>
T* up(hipStream_t s, DevArena* ar, const std::vector<T>& v, hipError_t* e) {
  T* p = (T*)ar->take(sizeof(T) * std::max<size_t>(v.size(), 1));
  if (!p) {
    *e = hipErrorOutOfMemory;
    return nullptr;
  }
  if (!v.empty() && *e == hipSuccess) *e = hipMemcpyAsync(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, s);
  return p;
}

}  // namespace

struct Direct {
  DirectDev dev{};
  DirectInfo info{};
  // host copies of the uploaded lists must outlive the asynchronous copies
  std::vector<int> h_vertex_pos, h_pos_vertex, h_vptr, h_vlist, h_slot_row, h_slot_col, h_slot_edge, h_enext, h_ss_pair,
      h_ss_edge, h_lslot, h_ltask, h_tk_cptr, h_ca, h_cb;
  std::vector<unsigned> h_tk_target;
  std::vector<unsigned short> h_dpair;
};

const DirectInfo& direct_info(const Direct* d) { return d->info; }
void direct_destroy(Direct* d) { delete d; }

double direct_bytes(const Direct* d, int E, int iters) {
  // per iteration: edge operands + poses, the per-edge terms written and read, every stored block written and read
  return (double)iters * (96.0 * E + 2.0 * 216.0 * E + 2.0 * 72.0 * (d->info.slots + d->info.n_chain) + 72.0 * d->info.n);
}

Direct* direct_create(hipStream_t s, DevArena* arena, int V, int n, const int* free_id, int E, const int* ei, const int* ej,
                      int max_rows, std::string* why, std::string* err) {
  auto no = [&](const char* w) -> Direct* {
    *why = w;
    return nullptr;
  };
  if (n <= 0) return no("no free pose");
  if (n > max_rows) return no("more free poses than direct_rows");
  if (E > kMaxEdges) return no("too many edges");
  std::vector<int> fidx((size_t)V, -1);
  for (int f = 0; f < n; ++f) fidx[free_id[f]] = f;
  // unique free-free pairs (lo < hi in free index) and the chain / closure split
  std::vector<std::pair<int, int>> pairs;
  pairs.reserve((size_t)E);
  for (int e = 0; e < E; ++e) {
    if (ei[e] == ej[e]) return no("self loop");
    const int a = fidx[ei[e]], b = fidx[ej[e]];
    if (a >= 0 && b >= 0) pairs.emplace_back(std::min(a, b), std::max(a, b));
  }
  std::sort(pairs.begin(), pairs.end());
  pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
  // ---- separators: greedy vertex cover of the non-chain pairs
  std::vector<std::vector<int>> cadj((size_t)n);
  std::vector<char> chain_next((size_t)n, 0);   // chain pair (f, f + 1) present
  for (auto& pr : pairs) {
    if (pr.second == pr.first + 1) {
      chain_next[pr.first] = 1;
    } else {
      cadj[pr.first].push_back(pr.second);
      cadj[pr.second].push_back(pr.first);
    }
  }
  std::vector<char> is_sep((size_t)n, 0);
  std::vector<int> cdeg((size_t)n);
  for (int f = 0; f < n; ++f) cdeg[f] = (int)cadj[f].size();
  int ns = 0;
  for (;;) {
    int best = -1, bd = 0;
    for (int f = 0; f < n; ++f)
      if (cdeg[f] > bd) {
        bd = cdeg[f];
        best = f;
      }
    if (best < 0) break;
    if (++ns > kMaxSep) return no("more separators than the dense block holds");
    is_sep[best] = 1;
    cdeg[best] = 0;
    for (int g : cadj[best])
      if (!is_sep[g]) --cdeg[g];
  }
  const int nI = n - ns;
  // ---- elimination positions: chain segments by cyclic-reduction level, then the separators
  std::vector<int> crl((size_t)n, 0);
  for (int f = 0; f < n;) {
    if (is_sep[f]) {
      ++f;
      continue;
    }
    int g = f;
    while (g + 1 < n && !is_sep[g + 1] && chain_next[g]) ++g;
    for (int q = f; q <= g; ++q) crl[q] = __builtin_ctz((unsigned)(q - f + 1));   // 1-based position: trailing zeros
    f = g + 1;
  }
  std::vector<int> order;   // position -> free index
  order.reserve((size_t)n);
  for (int f = 0; f < n; ++f)
    if (!is_sep[f]) order.push_back(f);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return crl[a] < crl[b]; });
  for (int f = 0; f < n; ++f)
    if (is_sep[f]) order.push_back(f);
  std::vector<int> pos((size_t)n);
  for (int p = 0; p < n; ++p) pos[order[p]] = p;
  // ---- symbolic factorisation of the sparse columns
  std::vector<std::vector<int>> st((size_t)nI), kids((size_t)nI);
  {
    std::vector<std::vector<int>> later((size_t)nI);
    for (auto& pr : pairs) {
      const int a = std::min(pos[pr.first], pos[pr.second]), b = std::max(pos[pr.first], pos[pr.second]);
      if (a < nI) later[a].push_back(b);
    }
    std::vector<int> tmp;
    for (int k = 0; k < nI; ++k) {
      std::vector<int>& v = later[k];
      for (int c : kids[k])
        for (int i : st[c])
          if (i != k) v.push_back(i);
      std::sort(v.begin(), v.end());
      v.erase(std::unique(v.begin(), v.end()), v.end());
      if ((int)v.size() > kMaxColumn) return no("a column has more blocks than a wave holds");
      st[k] = v;
      if (!v.empty() && v[0] < nI) kids[v[0]].push_back(k);
    }
  }
  std::vector<int> level((size_t)nI, 0);
  int NL = 0;
  for (int k = 0; k < nI; ++k) {
    for (int c : kids[k]) level[k] = std::max(level[k], level[c] + 1);
    NL = std::max(NL, level[k] + 1);
  }
  if (NL > kMaxLevels) return no("elimination tree too deep");
  const int tri = 3 * ns * (3 * ns + 1) / 2;
  const size_t lds_bytes = sizeof(double) * (3 * (size_t)n + (size_t)tri + 32);
  if (lds_bytes > kLdsBudget) return no("right-hand side + separator block exceed the LDS");

  Direct* d = new Direct();
  std::vector<std::vector<int>> lcols((size_t)NL);
  for (int k = 0; k < nI; ++k) lcols[level[k]].push_back(k);
  // ---- slots: per level, per column (ascending), a column's blocks inside one wave
  std::vector<int> colbase((size_t)nI, 0);
  auto& srow = d->h_slot_row;
  auto& scol = d->h_slot_col;
  d->h_lslot.assign((size_t)NL + 1, 0);
  for (int l = 0; l < NL; ++l) {
    d->h_lslot[l] = (int)srow.size();
    for (int k : lcols[l]) {
      const int len = std::max<int>(1, (int)st[k].size());
      const int cur = (int)srow.size();
      if ((cur & 63) + len > 64)
        for (int q = cur; q < ((cur + 63) & ~63); ++q) {
          srow.push_back(-1);
          scol.push_back(-1);
        }
      colbase[k] = (int)srow.size();
      if (st[k].empty()) {
        srow.push_back(-1);
        scol.push_back(k);
      }
      for (int i : st[k]) {
        srow.push_back(i);
        scol.push_back(k);
      }
    }
    while (srow.size() & 63) {
      srow.push_back(-1);
      scol.push_back(-1);
    }
  }
  d->h_lslot[NL] = (int)srow.size();
  const int NB = (int)srow.size();
  auto slot_of = [&](int i, int k) -> int {   // stored block (row i, column k), k < nI
    const std::vector<int>& v = st[k];
    const auto it = std::lower_bound(v.begin(), v.end(), i);
    return colbase[k] + (int)(it - v.begin());
  };
  // ---- assembly lists
  d->h_vertex_pos.assign((size_t)V, -1);
  d->h_pos_vertex.resize((size_t)n);
  for (int p = 0; p < n; ++p) {
    d->h_pos_vertex[p] = free_id[order[p]];
    d->h_vertex_pos[free_id[order[p]]] = p;
  }
  d->h_vptr.assign((size_t)n + 1, 0);
  for (int e = 0; e < E; ++e) {
    const int a = d->h_vertex_pos[ei[e]], b = d->h_vertex_pos[ej[e]];
    if (a >= 0) ++d->h_vptr[a + 1];
    if (b >= 0) ++d->h_vptr[b + 1];
  }
  for (int p = 0; p < n; ++p) d->h_vptr[p + 1] += d->h_vptr[p];
  d->h_vlist.resize((size_t)d->h_vptr[n]);
  {
    std::vector<int> cur(d->h_vptr.begin(), d->h_vptr.end() - 1);
    for (int e = 0; e < E; ++e) {
      const int a = d->h_vertex_pos[ei[e]], b = d->h_vertex_pos[ej[e]];
      if (a >= 0) d->h_vlist[cur[a]++] = e << 1;
      if (b >= 0) d->h_vlist[cur[b]++] = e << 1 | 1;
    }
  }
  d->h_slot_edge.assign((size_t)NB, -1);
  d->h_enext.assign((size_t)std::max(E, 1), -1);
  std::vector<int> ss_first((size_t)ns * ns, -1);
  for (int e = E - 1; e >= 0; --e) {   // descending: the lists come out in ascending edge order
    const int a = d->h_vertex_pos[ei[e]], b = d->h_vertex_pos[ej[e]];
    if (a < 0 || b < 0) continue;
    // the stored block is H[row = later position][col = earlier position]; the edge holds H[vi][vj]
    const int row = std::max(a, b), col = std::min(a, b);
    const int enc = e << 1 | (row == a ? 0 : 1);
    int* head = col < nI ? &d->h_slot_edge[slot_of(row, col)] : &ss_first[(size_t)(row - nI) * ns + (col - nI)];
    d->h_enext[e] = *head;
    *head = enc;
  }
  for (int si = 0; si < ns; ++si)
    for (int sj = 0; sj < si; ++sj)
      if (ss_first[(size_t)si * ns + sj] >= 0) {
        d->h_ss_pair.push_back(si << 12 | sj);
        d->h_ss_edge.push_back(ss_first[(size_t)si * ns + sj]);
      }
  // ---- forward tasks per level: (target, slot a, slot b) sorted by target
  struct Contrib {
    unsigned target;
    int a, b;
  };
  std::vector<Contrib> cl;
  d->h_ltask.assign((size_t)NL + 1, 0);
  long long total_contrib = 0;
  for (int l = 0; l < NL; ++l) {
    cl.clear();
    for (int k : lcols[l]) {
      const std::vector<int>& v = st[k];
      for (size_t x = 0; x < v.size(); ++x) {
        const int i = v[x], sa = colbase[k] + (int)x;
        cl.push_back({T_RHS << 28 | (unsigned)i, sa, sa});
        for (size_t y = 0; y <= x; ++y) {
          const int j = v[y], sb = colbase[k] + (int)y;   // i >= j
          unsigned tg;
          if (j >= nI) tg = T_DENSE << 28 | (unsigned)((i - nI) << 12 | (j - nI));
          else if (i == j) tg = T_DIAG << 28 | (unsigned)j;
          else tg = T_OFF << 28 | (unsigned)slot_of(i, j);
          cl.push_back({tg, sa, sb});
        }
      }
    }
    total_contrib += (long long)cl.size();
    if (total_contrib > kMaxContrib) {
      delete d;
      return no("too much fill");
    }
    std::stable_sort(cl.begin(), cl.end(), [](const Contrib& x, const Contrib& y) { return x.target < y.target; });
    d->h_ltask[l] = (int)d->h_tk_target.size();
    for (size_t q = 0; q < cl.size(); ++q) {
      if (q == 0 || cl[q].target != cl[q - 1].target) {
        d->h_tk_target.push_back(cl[q].target);
        d->h_tk_cptr.push_back((int)d->h_ca.size());
      }
      d->h_ca.push_back(cl[q].a);
      d->h_cb.push_back(cl[q].b);
    }
  }
  d->h_ltask[NL] = (int)d->h_tk_target.size();
  d->h_tk_cptr.push_back((int)d->h_ca.size());
  // ---- dense pair table: bj descending so that the trailing blocks of pivot p are a prefix
  for (int bj = ns - 1; bj >= 0; --bj)
    for (int bi = bj; bi < ns; ++bi) d->h_dpair.push_back((unsigned short)(bi << 8 | bj));

  hipError_t e = hipSuccess;
  DirectDev& D = d->dev;
  D.n = n; D.nI = nI; D.ns = ns; D.NL = NL; D.E = E; D.NB = NB; D.tri = tri; D.nss = (int)d->h_ss_pair.size();
  D.vertex_pos = up(s, arena, d->h_vertex_pos, &e);
  D.pos_vertex = up(s, arena, d->h_pos_vertex, &e);
  D.vptr = up(s, arena, d->h_vptr, &e);
  D.vlist = up(s, arena, d->h_vlist, &e);
  D.slot_row = up(s, arena, d->h_slot_row, &e);
  D.slot_col = up(s, arena, d->h_slot_col, &e);
  D.slot_edge = up(s, arena, d->h_slot_edge, &e);
  D.enext = up(s, arena, d->h_enext, &e);
  D.ss_pair = up(s, arena, d->h_ss_pair, &e);
  D.ss_edge = up(s, arena, d->h_ss_edge, &e);
  D.lslot = up(s, arena, d->h_lslot, &e);
  D.ltask = up(s, arena, d->h_ltask, &e);
  D.tk_target = up(s, arena, d->h_tk_target, &e);
  D.tk_cptr = up(s, arena, d->h_tk_cptr, &e);
  D.ca = up(s, arena, d->h_ca, &e);
  D.cb = up(s, arena, d->h_cb, &e);
  D.dpair = up(s, arena, d->h_dpair, &e);
  D.Wd = (double*)arena->take(sizeof(double) * 9 * (size_t)std::max(nI, 1));
  D.Wo = (double*)arena->take(sizeof(double) * 9 * (size_t)std::max(NB, 1));
  D.escr = (double*)arena->take(sizeof(double) * 27 * (size_t)std::max(E, 1));
  if (e == hipSuccess && (!D.Wd || !D.Wo || !D.escr)) e = hipErrorOutOfMemory;
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_direct), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
  if (e != hipSuccess) {
    *err = std::string("direct_create: ") + hipGetErrorString(e);
    delete d;
    return nullptr;
  }
  d->info.n = n;
  d->info.n_chain = nI;
  d->info.n_sep = ns;
  d->info.levels = NL;
  d->info.slots = NB;
  d->info.contributions = (int)total_contrib;
  d->info.lds_bytes = lds_bytes;
  return d;
}

hipError_t direct_optimize(Direct* d, hipStream_t s, const EdgeListDev& el, double* d_poses, int iters, double* d_hist,
                           DirectResult* d_res) {
  SGO_LAUNCH(k_direct, dim3(1), dim3(kDT), d->info.lds_bytes, s, d->dev, el, d_poses, iters, d_hist, d_res);
  return hipGetLastError();
}

}  // namespace sgo
