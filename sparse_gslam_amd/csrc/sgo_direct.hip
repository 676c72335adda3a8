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
#include <memory>
#include <atomic>
#include <cstring>
#include <numeric>
#include <vector>

#include "sgo_device.h"
#include "sgo_direct.h"
#include "sgo_internal.h"

namespace sgo {

namespace {

constexpr int kDT = 512;                // threads of the workgroup (eight waves: the separator phase assigns jobs by wave)
static_assert(kDT == 512, "k_direct's separator phase lays its jobs out over exactly eight waves");
constexpr int kMaxSep = 60;             // separators: 180 x 180 packed triangle = 130 KB of LDS
constexpr int kMaxLevels = 24;
constexpr int kMaxEdges = 1 << 18;
constexpr int kMaxColumn = 63;          // stored blocks of one sparse column (a wave holds a column's segment)
constexpr int kMaxContrib = 1 << 20;
constexpr size_t kLdsBudget = 150 * 1024;

enum : unsigned { T_OFF = 0, T_DIAG = 1, T_DENSE = 2, T_RHS = 3 };

struct DirectDev {
  int n, nI, ns, NL, E, NB, tri, nzero, nmulti;
  const int* pos_vertex;      // [n] vertex id of an elimination position
  const int4* vrec;           // [n] incident edges of a position, (edge << 1 | side), -1 none; w <= -2: more at vover[-2 - w]
  const int* vover;           // overflow lists: count, entries
  const unsigned* edge_tgt;   // [E] where the edge's off-diagonal block goes: kind << 30 | transposed << 29 | index
                              //     kind 0 nowhere (a fixed endpoint), 1 stored block, 2 separator block (si << 12 | sj),
                              //     3 the pair has several edges (summed afterwards: `multi`)
  const int* zero_slots;      // [nzero] stored blocks that are pure fill
  const int2* multi;          // [nmulti] pairs with several edges: {target as above (kind 1 / 2), first edge << 1 | transposed}
  const int* enext;           // [E] next edge of the same pair, same encoding
  const int2* slot_rc;        // [NB] {row position (-1: dummy slot of an empty column), column position (-1: padding)}
  const int* lmeta;           // [2 (NL + 1)] slot range (multiples of 64) and task range of every level
  const int4* tk;             // forward tasks: {kind << 28 | index, first contribution, end, 0}
  const int4* ctr;            // contributions: {slot of W_ik, slot of W_jk, k, 0}
  const unsigned short* dpair;  // dense block pairs (bi << 8 | bj), bi >= bj, sorted by bj descending
  double* Wd;                 // [nI][9] diagonal blocks of the sparse columns
  double* Wo;                 // [NB][9] stored blocks
  double* xbg;                // [3 n] right-hand side / solution when it does not fit the LDS
  double* escr;               // [E][27] per-edge terms of the current linearisation: Hii(6) bi(3) Hjj(6) bj(3) [Hij(9)]
  double* zsc;                // [2][E] sin / cos of the inverse measurement's angle (constant per edge)
};

// 1 / x by the hardware reciprocal + two Newton steps (the IEEE division sequence is 3x longer and sits on the
// critical path of every pivot)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
}
// Pivot blocks are applied through their 3x3 LDL^T factors, not through an explicit inverse: the pivots of the upper
// elimination levels are Schur complements of long chain segments (kappa ~ 1e8 and more) and a factor solve is
// backward stable whatever the block's conditioning, at the cost of two more reciprocals per pivot (5 % of the
// kernel).  Its three pivots also ARE the positive-definiteness test (LinearSolverEigen's LDL^T failing).
// f = {l10, l20, l21, 1/p0, 1/p1, 1/p2} of D = L diag(p) L^T (d00 d01 d02 d11 d12 d22); false when a pivot is not positive.
__device__ __forceinline__ bool inv_sym3(double d00, double d01, double d02, double d11, double d12, double d22,
                                         double (&f)[6]) {
  const double r0 = fast_rcp(d00);
  const double l10 = d01 * r0, l20 = d02 * r0;
  const double p1 = d11 - l10 * d01;
  const double r1 = fast_rcp(p1);
  const double u21 = d12 - l20 * d01;
  const double l21 = u21 * r1;
  const double p2 = d22 - l20 * d02 - l21 * u21;
  const double r2 = fast_rcp(p2);
  f[0] = l10; f[1] = l20; f[2] = l21; f[3] = r0; f[4] = r1; f[5] = r2;
  return d00 > 0.0 && p1 > 0.0 && p2 > 0.0 && isfinite(p2);
}
// x = D^-1 y through the factors
__device__ __forceinline__ void solve3(const double* f, double y0, double y1, double y2, double& x0, double& x1, double& x2) {
  const double z1 = y1 - f[0] * y0;
  const double z2 = y2 - f[1] * y0 - f[2] * z1;
  x2 = z2 * f[5];
  x1 = z1 * f[4] - f[2] * x2;
  x0 = y0 * f[3] - f[0] * x1 - f[1] * x2;
}
// T = W * D^-1 (D symmetric: row r of T solves D t = w_r)
__device__ __forceinline__ void mul_sym(const double (&W)[9], const double (&F)[6], double (&T)[9]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) solve3(F, W[3 * r], W[3 * r + 1], W[3 * r + 2], T[3 * r], T[3 * r + 1], T[3 * r + 2]);
}
// Stored 3x3 blocks are 80-byte records (row-major, one double of padding): 16-byte aligned, so a gathered
// block is four 16-byte loads + one 8-byte load per lane instead of nine 8-byte ones.
constexpr int kBS = 10;
__device__ __forceinline__ void load9g(const double* __restrict__ p, double (&v)[9]) {
  const double2* __restrict__ q = reinterpret_cast<const double2*>(p);
  const double2 a = q[0], b = q[1], c = q[2], d = q[3];
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y; v[6] = d.x; v[7] = d.y;
  v[8] = p[8];
}
__device__ __forceinline__ void store9g(double* __restrict__ p, const double (&v)[9]) {
  double2* __restrict__ q = reinterpret_cast<double2*>(p);
  q[0] = make_double2(v[0], v[1]);
  q[1] = make_double2(v[2], v[3]);
  q[2] = make_double2(v[4], v[5]);
  q[3] = make_double2(v[6], v[7]);
  p[8] = v[8];
}
__device__ __forceinline__ double readlane_d(double v, int l) {   // l wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int tri_at(int i, int j) { return i * (i + 1) / 2 + j; }   // i >= j

// Everything an edge contributes at the current poses: rec = Hii(6) bi(3) Hjj(6) bj(3) Hij(9)
// (EdgeSE2::computeError, linearizeOplus, RobustKernelDCS::robustify, constructQuadraticForm; the arithmetic of
// k_linearize, with sin / cos of the inverse measurement's angle taken from zsc and sin(-t_i) = -sin(t_i)).
// Returns e2 and rho0 for the chi2 sums.
__device__ __forceinline__ void edge_terms(const EdgeListDev& el, const double* __restrict__ zsc, int k,
                                           const double* __restrict__ poses, bool jac, double* __restrict__ rec,
                                           double (&hij)[9], double* e2_out, double* rho_out) {
  const size_t E = (size_t)el.E;
  const int vi = el.vi[k], vj = el.vj[k];
  const double zx = el.zinv[k], zy = el.zinv[E + k], zt = el.zinv[2 * E + k];
  const double sz = zsc[k], cz = zsc[E + k];
  const double o00 = el.info[k], o01 = el.info[E + k], o02 = el.info[2 * E + k];
  const double o11 = el.info[3 * E + k], o12 = el.info[4 * E + k], o22 = el.info[5 * E + k];
  const double ph = el.phi[k];
  const double xi = poses[3 * (size_t)vi], yi = poses[3 * (size_t)vi + 1], ti = poses[3 * (size_t)vi + 2];
  const double xj = poses[3 * (size_t)vj], yj = poses[3 * (size_t)vj + 1], tj = poses[3 * (size_t)vj + 2];
  double si, ci;
  sincos(ti, &si, &ci);
  // e = toVector(Zi * (Xi^-1 * Xj)), SE2 algebra as edge_error (sgo_device.h): Xi^-1 = (R(-ti), -R(-ti) t_i)
  const double tin = norm_theta(-ti);
  const double s1 = -si, c1 = ci;
  const double ix = c1 * (-xi) - s1 * (-yi), iy = s1 * (-xi) + c1 * (-yi);
  const double dx = ix + c1 * xj - s1 * yj, dy = iy + s1 * xj + c1 * yj;
  const double dth = norm_theta(tin + tj);
  double e[3];
  e[0] = zx + cz * dx - sz * dy;
  e[1] = zy + sz * dx + cz * dy;
  e[2] = norm_theta(zt + dth);
  double oe0 = o00 * e[0] + o01 * e[1] + o02 * e[2];
  double oe1 = o01 * e[0] + o11 * e[1] + o12 * e[2];
  double oe2 = o02 * e[0] + o12 * e[1] + o22 * e[2];
  const double e2 = e[0] * oe0 + e[1] * oe1 + e[2] * oe2;
  double r0, w;
  dcs(e2, ph, &r0, &w);
  *e2_out = e2;
  *rho_out = r0;
  if (!jac) return;
  const double w00 = w * o00, w01 = w * o01, w02 = w * o02, w11 = w * o11, w12 = w * o12, w22 = w * o22;
  oe0 *= w; oe1 *= w; oe2 *= w;
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
  rec[0 * E] = A00 * TA00 + A10 * TA10;
  rec[1 * E] = A00 * TA01 + A10 * TA11;
  rec[2 * E] = A00 * TA02 + A10 * TA12;
  rec[3 * E] = A01 * TA01 + A11 * TA11;
  rec[4 * E] = A01 * TA02 + A11 * TA12;
  rec[5 * E] = A02 * TA02 + A12 * TA12 - TA22;
  // bi = -A^T (Ow e)
  rec[6 * E] = -(A00 * oe0 + A10 * oe1);
  rec[7 * E] = -(A01 * oe0 + A11 * oe1);
  rec[8 * E] = -(A02 * oe0 + A12 * oe1 - oe2);
  // Hjj = B^T TB
  rec[9 * E] = B00 * TB00 + B10 * TB10;
  rec[10 * E] = B00 * TB01 + B10 * TB11;
  rec[11 * E] = B00 * TB02 + B10 * TB12;
  rec[12 * E] = B01 * TB01 + B11 * TB11;
  rec[13 * E] = B01 * TB02 + B11 * TB12;
  rec[14 * E] = TB22;
  // bj = -B^T (Ow e)
  rec[15 * E] = -(B00 * oe0 + B10 * oe1);
  rec[16 * E] = -(B01 * oe0 + B11 * oe1);
  rec[17 * E] = -oe2;
  // Hij = A^T Ow B = TA^T B  (row vi, column vj)
  hij[0] = TA00 * B00 + TA10 * B10;
  hij[1] = TA00 * B01 + TA10 * B11;
  hij[2] = TA20;
  hij[3] = TA01 * B00 + TA11 * B10;
  hij[4] = TA01 * B01 + TA11 * B11;
  hij[5] = TA21;
  hij[6] = TA02 * B00 + TA12 * B10;
  hij[7] = TA02 * B01 + TA12 * B11;
  hij[8] = TA22;
}

// sum over the edges of a pair (linked through enext) of H[row][col], row-major
__device__ __forceinline__ void pair_block(const DirectDev& D, int first, double (&b)[9]) {
#pragma unroll
  for (int c = 0; c < 9; ++c) b[c] = 0.0;
  for (int t = first; t >= 0; t = D.enext[t >> 1]) {
    const size_t E = (size_t)D.E;
    const double* h = D.escr + (size_t)(t >> 1) + 18 * E;
    if (t & 1) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) b[3 * r + c] += h[(3 * c + r) * E];
    } else {
#pragma unroll
      for (int c = 0; c < 9; ++c) b[c] += h[c * E];
    }
  }
}
// store an off-diagonal block h = H[row][col] (row-major; `tr`: h is the transpose of what the target wants)
__device__ __forceinline__ void store_offdiag(const DirectDev& D, double* __restrict__ Sd, unsigned tgt, const double* h) {
  const unsigned kind = tgt >> 30;
  const bool tr = (tgt >> 29) & 1u;
  const int idx = (int)(tgt & 0x1fffffffu);
  if (kind == 1) {
    double v[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) v[3 * r + c] = tr ? h[3 * c + r] : h[3 * r + c];
    store9g(D.Wo + kBS * (size_t)idx, v);
  } else if (kind == 2) {
    const int si = idx >> 12, sj = idx & 4095;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Sd[tri_at(3 * si + r, 3 * sj + c)] = tr ? h[3 * c + r] : h[3 * r + c];
  }
}

// One forward task: a target minus its contributions W_ik D_k^-1 W_jk^T.  A DIAGONAL target (sparse column or
// separator) also owns its pose's right-hand side: both take a contribution from exactly the same columns k, so
// W_ik and D_k are fetched once for the two.
__device__ __forceinline__ void run_task(const DirectDev& D, double* __restrict__ xb, double* __restrict__ Sd, int4 tk, int4 c0) {
  const unsigned kind = (unsigned)tk.x >> 28;
  const int idx = tk.x & 0x0fffffff;
  const int si = idx >> 12, sj = idx & 4095;   // T_DENSE
  const bool diag = kind == T_DIAG || (kind == T_DENSE && si == sj);
  // a9 = (old value) - sum of contributions: a stored target is fetched together with the sources
  double a9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double r3[3] = {0, 0, 0};
  double* wt = (kind == T_DIAG ? D.Wd : D.Wo) + kBS * (size_t)idx;
  if (kind == T_OFF || kind == T_DIAG) load9g(wt, a9);
  for (int c = tk.y; c < tk.z; ++c) {
    const int4 cr = c == tk.y ? c0 : D.ctr[c];
    const int k = cr.z;
    double Wa[9], iv[6], Tm[9];
    load9g(D.Wo + kBS * (size_t)cr.x, Wa);
    const double2* dk = reinterpret_cast<const double2*>(D.Wd + kBS * (size_t)k);
    const double2 q0 = dk[0], q1 = dk[1], q2 = dk[2];
    const double d22 = D.Wd[kBS * (size_t)k + 8];
    inv_sym3(q0.x, q0.y, q1.x, q2.x, q2.y, d22, iv);
    mul_sym(Wa, iv, Tm);
    if (diag) {
      const double b0 = xb[3 * k], b1 = xb[3 * k + 1], b2 = xb[3 * k + 2];
#pragma unroll
      for (int r = 0; r < 3; ++r) r3[r] -= Tm[3 * r] * b0 + Tm[3 * r + 1] * b1 + Tm[3 * r + 2] * b2;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          a9[3 * r + q] -= Tm[3 * r] * Wa[3 * q] + Tm[3 * r + 1] * Wa[3 * q + 1] + Tm[3 * r + 2] * Wa[3 * q + 2];
    } else {
      double Wb[9];
      load9g(D.Wo + kBS * (size_t)cr.y, Wb);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          a9[3 * r + q] -= Tm[3 * r] * Wb[3 * q] + Tm[3 * r + 1] * Wb[3 * q + 1] + Tm[3 * r + 2] * Wb[3 * q + 2];
    }
  }
  if (diag) {
    const int pos = kind == T_DIAG ? idx : D.nI + si;
    xb[3 * pos] += r3[0];
    xb[3 * pos + 1] += r3[1];
    xb[3 * pos + 2] += r3[2];
  }
  if (kind == T_DENSE) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (si != sj || r >= q) Sd[tri_at(3 * si + r, 3 * sj + q)] += a9[3 * r + q];
  } else {
    store9g(wt, a9);
  }
}

constexpr int kPF = 1;   // forward tasks per thread whose records are fetched one level ahead

// XG: the right-hand side / solution vector lives in global memory (graphs whose 24 n bytes do not fit the LDS
// next to the separator block) instead of LDS.
template <bool XG>
__global__ __launch_bounds__(kDT) void k_direct(DirectDev D, EdgeListDev el, double* __restrict__ poses, int iters,
                                                double* __restrict__ hist, DirectResult* __restrict__ res) {
  extern __shared__ double lds[];
  double* xb;                             // [3 n] right-hand side, then the solution, by elimination position
  double* Sd;                             // [tri] packed lower triangle of the separator block
  if constexpr (XG) {
    xb = D.xbg;
    Sd = lds;
  } else {
    xb = lds;
    Sd = lds + 3 * (size_t)D.n;
  }
  double* pinv = Sd + D.tri;              // [ns][6] inverses of the separators' pivot blocks
  double* bs = pinv + 6 * (size_t)D.ns;   // [3 ns] the separators' right-hand side / solution during the dense phase
  double* red = bs + 3 * (size_t)D.ns;    // [2][16] chi2 partials
  int* lmeta = reinterpret_cast<int*>(red + 32);   // [2 (NL + 1)]
  unsigned short* dpl = reinterpret_cast<unsigned short*>(lmeta + 2 * (D.NL + 1));   // [ns (ns + 1) / 2] dense block pairs
  __shared__ int fail_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = D.n, nI = D.nI, ns = D.ns, NL = D.NL;
  const size_t E = (size_t)D.E;
  if (tid == 0) fail_flag = 0;
  for (int k = tid; k < 2 * (NL + 1); k += kDT) lmeta[k] = D.lmeta[k];
  for (int k = tid; k < D.tri; k += kDT) Sd[k] = 0.0;
  for (int k = tid; k < D.ns * (D.ns + 1) / 2; k += kDT) dpl[k] = D.dpair[k];
  for (int k = tid; k < D.E; k += kDT) {
    double sz, cz;
    sincos(el.zinv[2 * E + k], &sz, &cz);
    D.zsc[k] = sz;
    D.zsc[E + k] = cz;
  }
  __syncthreads();
  const int* lslot = lmeta;
  const int* ltask = lmeta + NL + 1;
  // records of this thread's first tasks of the next forward level (static data: fetched while the previous
  // level computes, so that a level costs one round trip to the blocks instead of three)
  int4 ntk[kPF], nc[kPF];
#pragma unroll
  for (int q = 0; q < kPF; ++q) ntk[q] = nc[q] = make_int4(0, 0, 0, 0);
  auto prefetch = [&](int l) {
    if (NL == 0) return;
    const int t0 = ltask[l], t1 = ltask[l + 1];
#pragma unroll
    for (int q = 0; q < kPF; ++q) {
      const int T = t0 + tid + q * kDT;
      if (T < t1) {
        ntk[q] = D.tk[T];
        nc[q] = D.ctr[ntk[q].y];
      }
    }
  };
  prefetch(0);
  // dense phase: the block-update threads (waves 0-2 and 4-6) and the indices of each one's first three work items
  constexpr int kBlockWorkers = 6 * 64;
  const int npairs = ns * (ns + 1) / 2;
  const int wid = (wave < 3 ? wave : wave - 1) * 64 + lane;   // meaningful on waves 0-2, 4-6
  int sbi0 = 0, sbj0 = 0, sra0 = 0, srb0 = 0, sbi1 = 0, sbj1 = 0, sra1 = 0, srb1 = 0, sbi2 = 0, sbj2 = 0, sra2 = 0, srb2 = 0;
  {
    auto decode = [&](int g, int& bi, int& bj, int& ra, int& rb) {
      const int pr = D.dpair[g < npairs ? g : 0];
      bi = pr >> 8;
      bj = pr & 255;
      ra = tri_at(3 * bi, 0);
      rb = tri_at(3 * bj, 0);
    };
    decode(wid, sbi0, sbj0, sra0, srb0);
    decode(wid + kBlockWorkers, sbi1, sbj1, sra1, srb1);
    decode(wid + 2 * kBlockWorkers, sbi2, sbj2, sra2, srb2);
  }
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
  int done = 0, fail = 0;
  for (int it = 0; it <= iters; ++it) {
    // per-thread addresses of 30-odd arrays are loop invariant; hoisted out of the GN loop they cost 50 spilled
    // registers: an opaque copy of the thread index keeps them inside the phase that uses them
    int tq = tid;
    asm volatile("" : "+v"(tq));
    if (tid == 0) {
      res->stamp[2 * it] = (unsigned long long)wall_clock64();
      if (it < iters) res->phase[0] = res->stamp[2 * it];
    }
    // ---- edges: chi2 sums, and (unless this is the closing pass) the terms of the linearisation; an edge
    //      that is alone on its pair of poses writes its off-diagonal block straight to where it is stored
    const bool jac = it < iters;
    double acc[2] = {0.0, 0.0};
    for (int k = tq; k < D.E; k += kDT) {
      double e2, r0, hij[9];
      double* o = D.escr + k;   // SoA: component c of edge k at escr[c E + k] (coalesced stores)
      edge_terms(el, D.zsc, k, poses, jac, o, hij, &e2, &r0);
      acc[0] += e2;
      acc[1] += r0;
      if (jac) {
        const unsigned tgt = D.edge_tgt[k];
        if ((tgt >> 30) == 3) {
#pragma unroll
          for (int c = 0; c < 9; ++c) o[(18 + c) * E] = hij[c];
        } else {
          store_offdiag(D, Sd, tgt, hij);
        }
      }
    }
    if (jac)
      for (int q = tq; q < D.nzero; q += kDT) {
        const double z9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        store9g(D.Wo + kBS * (size_t)D.zero_slots[q], z9);
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
    if (tid == 0) res->phase[1] = (unsigned long long)wall_clock64();
    // ---- assembly: diagonal blocks and right-hand sides (sums over a pose's incident edges), multi-edge pairs
    for (int f = tq; f < n; f += kDT) {
      const int4 vr = D.vrec[f];
      double d[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      auto add = [&](int t) {
        const double* h = D.escr + (size_t)(t >> 1) + ((t & 1) ? 9 * E : 0);
#pragma unroll
        for (int c = 0; c < 9; ++c) d[c] += h[c * E];
      };
      if (vr.x >= 0) add(vr.x);
      if (vr.y >= 0) add(vr.y);
      if (vr.z >= 0) add(vr.z);
      if (vr.w >= 0) add(vr.w);
      if (vr.w <= -2) {
        const int* ov = D.vover + (-2 - vr.w);
        const int cnt = ov[0];
        for (int q = 1; q <= cnt; ++q) add(ov[q]);
      }
      xb[3 * f] = d[6];
      xb[3 * f + 1] = d[7];
      xb[3 * f + 2] = d[8];
      if (f < nI) {
        const double w9[9] = {d[0], d[1], d[2], d[1], d[3], d[4], d[2], d[4], d[5]};
        store9g(D.Wd + kBS * (size_t)f, w9);
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
    for (int q = tq; q < D.nmulti; q += kDT) {
      const int2 m = D.multi[q];
      double b[9];
      pair_block(D, m.y, b);
      store_offdiag(D, Sd, (unsigned)m.x, b);
    }
    __syncthreads();
    if (tid == 0) res->phase[2] = res->stamp[2 * it + 1] = (unsigned long long)wall_clock64();
    // ---- sparse levels, forward: Schur updates and right-hand sides of the level's columns
    for (int l = 0; l < NL; ++l) {
      const int t0 = ltask[l], t1 = ltask[l + 1];
      int4 ctk[kPF], cc[kPF];
#pragma unroll
      for (int q = 0; q < kPF; ++q) {
        ctk[q] = ntk[q];
        cc[q] = nc[q];
      }
      prefetch(l + 1 < NL ? l + 1 : 0);
#pragma unroll
      for (int q = 0; q < kPF; ++q)
        if (t0 + tq + q * kDT < t1) run_task(D, xb, Sd, ctk[q], cc[q]);
      for (int T = t0 + tq + kPF * kDT; T < t1; T += kDT) {
        const int4 tk = D.tk[T];
        run_task(D, xb, Sd, tk, D.ctr[tk.y]);
      }
      __syncthreads();
    }
    if (tid == 0) res->phase[3] = (unsigned long long)wall_clock64();
    // ---- separators: right-looking block LDL^T on the packed triangle, 3x3 pivots, W form (the panel stays);
    //      the thread that finishes the next pivot block inverts it for everybody
    if (tid == 0 && ns > 0) {
      double iv[6];
      const bool ok = inv_sym3(Sd[tri_at(0, 0)], Sd[tri_at(1, 0)], Sd[tri_at(2, 0)], Sd[tri_at(1, 1)], Sd[tri_at(2, 1)],
                               Sd[tri_at(2, 2)], iv);
#pragma unroll
      for (int c = 0; c < 6; ++c) pinv[c] = iv[c];
      if (!ok) atomicOr(&fail_flag, 1);
    }
    for (int k = tq; k < 3 * ns; k += kDT) bs[k] = xb[3 * (size_t)nI + k];
    __syncthreads();
    double nextF[6] = {0, 0, 0, 0, 0, 0};   // wave 7, lane 0: the factor it made in the previous step (no LDS round trip for itself)
    for (int p = 0; p + 1 < ns; ++p) {
      const int m = ns - 1 - p, cnt = m * (m + 1) / 2;
      const int P = 3 * p;
      double iv[6];
      if (wave == 7 && p > 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) iv[c] = nextF[c];
      } else {
#pragma unroll
        for (int c = 0; c < 6; ++c) iv[c] = pinv[6 * p + c];
      }
      // A step is bound by VALU issue (a wave64 instruction holds its SIMD for four cycles and only a few waves are
      // busy), so the jobs are laid out by SIMD (wave w runs on SIMD w % 4): waves 0-2 and 4-6 update the trailing
      // blocks, one thread per block with its indices in registers; wave 3 updates the right-hand side; wave 7's
      // first lane forms the NEXT pivot block and factors it for everybody, sharing its SIMD only with wave 3.
      if (wave == 7) {
        if (lane == 0) {
          const int rb0 = tri_at(3 * (p + 1), 0), rb1 = rb0 + 3 * (p + 1) + 1, rb2 = rb1 + 3 * (p + 1) + 2;
          double W[9], Tm[9];
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            W[q] = Sd[rb0 + P + q];
            W[3 + q] = Sd[rb1 + P + q];
            W[6 + q] = Sd[rb2 + P + q];
          }
          const int d0 = rb0 + 3 * (p + 1), d1 = rb1 + 3 * (p + 1), d2 = rb2 + 3 * (p + 1);
          const double o00 = Sd[d0], o10 = Sd[d1], o11 = Sd[d1 + 1], o20 = Sd[d2], o21 = Sd[d2 + 1], o22 = Sd[d2 + 2];
          mul_sym(W, iv, Tm);
          const double e00 = o00 - (Tm[0] * W[0] + Tm[1] * W[1] + Tm[2] * W[2]);
          const double e10 = o10 - (Tm[3] * W[0] + Tm[4] * W[1] + Tm[5] * W[2]);
          const double e11 = o11 - (Tm[3] * W[3] + Tm[4] * W[4] + Tm[5] * W[5]);
          const double e20 = o20 - (Tm[6] * W[0] + Tm[7] * W[1] + Tm[8] * W[2]);
          const double e21 = o21 - (Tm[6] * W[3] + Tm[7] * W[4] + Tm[8] * W[5]);
          const double e22 = o22 - (Tm[6] * W[6] + Tm[7] * W[7] + Tm[8] * W[8]);
          // (the block itself is not stored back: from here on only its factor is used)
          const bool ok = inv_sym3(e00, e10, e20, e11, e21, e22, nextF);
#pragma unroll
          for (int c = 0; c < 6; ++c) pinv[6 * (p + 1) + c] = nextF[c];
          if (!ok) atomicOr(&fail_flag, 1);
        }
      } else if (wave == 3) {
        const double p0 = bs[P], p1 = bs[P + 1], p2 = bs[P + 2];
        for (int bi = p + 1 + lane; bi < ns; bi += 64) {   // right-hand side of block row bi
          const int ra0 = tri_at(3 * bi, 0), ra1 = ra0 + 3 * bi + 1, ra2 = ra1 + 3 * bi + 2;
          double W[9], Tm[9];
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            W[q] = Sd[ra0 + P + q];
            W[3 + q] = Sd[ra1 + P + q];
            W[6 + q] = Sd[ra2 + P + q];
          }
          const double o0 = bs[3 * bi], o1 = bs[3 * bi + 1], o2 = bs[3 * bi + 2];
          mul_sym(W, iv, Tm);
          bs[3 * bi] = o0 - (Tm[0] * p0 + Tm[1] * p1 + Tm[2] * p2);
          bs[3 * bi + 1] = o1 - (Tm[3] * p0 + Tm[4] * p1 + Tm[5] * p2);
          bs[3 * bi + 2] = o2 - (Tm[6] * p0 + Tm[7] * p1 + Tm[8] * p2);
        }
      } else {
        for (int g = wid, trip = 0; g < cnt; g += kBlockWorkers, ++trip) {
          // which pair a work item is does not depend on the step (the trailing pairs of a step are a prefix of the
          // table): a thread's first three are decoded once, in registers
          int bi, bj, ra0, rb0;
          if (trip == 0) { bi = sbi0; bj = sbj0; ra0 = sra0; rb0 = srb0; }
          else if (trip == 1) { bi = sbi1; bj = sbj1; ra0 = sra1; rb0 = srb1; }
          else if (trip == 2) { bi = sbi2; bj = sbj2; ra0 = sra2; rb0 = srb2; }
          else {
            const int pr = dpl[g];
            bi = pr >> 8; bj = pr & 255; ra0 = tri_at(3 * bi, 0); rb0 = tri_at(3 * bj, 0);
          }
          if (bi == bj && bi == p + 1) continue;   // the next pivot block: wave 7's job
          const int ra1 = ra0 + 3 * bi + 1, ra2 = ra1 + 3 * bi + 2, rb1 = rb0 + 3 * bj + 1, rb2 = rb1 + 3 * bj + 2;
          double Wa[9], Wb[9], Tm[9], o[9];
          // every load before any store: Sd is one array, so a store would fence the loads behind it
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            Wa[q] = Sd[ra0 + P + q]; Wa[3 + q] = Sd[ra1 + P + q]; Wa[6 + q] = Sd[ra2 + P + q];
            Wb[q] = Sd[rb0 + P + q]; Wb[3 + q] = Sd[rb1 + P + q]; Wb[6 + q] = Sd[rb2 + P + q];
          }
          const bool dg = bi == bj;
          double* o0p = Sd + ra0 + 3 * bj;
          double* o1p = Sd + ra1 + 3 * bj;
          double* o2p = Sd + ra2 + 3 * bj;
          o[0] = o0p[0]; o[3] = o1p[0]; o[4] = o1p[1]; o[6] = o2p[0]; o[7] = o2p[1]; o[8] = o2p[2];
          o[1] = dg ? 0.0 : o0p[1]; o[2] = dg ? 0.0 : o0p[2]; o[5] = dg ? 0.0 : o1p[2];
          mul_sym(Wa, iv, Tm);
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q)
              o[3 * r + q] -= Tm[3 * r] * Wb[3 * q] + Tm[3 * r + 1] * Wb[3 * q + 1] + Tm[3 * r + 2] * Wb[3 * q + 2];
          o0p[0] = o[0]; o1p[0] = o[3]; o1p[1] = o[4]; o2p[0] = o[6]; o2p[1] = o[7]; o2p[2] = o[8];
          if (!dg) {
            o0p[1] = o[1]; o0p[2] = o[2]; o1p[2] = o[5];
          }
        }
      }
      __syncthreads();
    }
    if (tid == 0) res->phase[4] = (unsigned long long)wall_clock64();
    // back substitution of the separators by ONE wave, column oriented: lane p owns b_p; step i: x_i = D_i^-1 b_i is
    // broadcast from lane i and every lane p < i takes W_ip^T x_i off its b_p.  No barriers, no reductions.
    if (wave == 0 && ns > 0) {
      double b0 = 0.0, b1 = 0.0, b2 = 0.0;
      if (lane < ns) {
        b0 = bs[3 * lane];
        b1 = bs[3 * lane + 1];
        b2 = bs[3 * lane + 2];
      }
      const int lp = lane < ns ? lane : 0;
      for (int i = ns - 1; i >= 0; --i) {
        double W[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (lane < i) {
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) W[3 * r + q] = Sd[tri_at(3 * i + r, 3 * lp + q)];
        }
        const double* iv = pinv + 6 * i;
        const double y0 = readlane_d(b0, i), y1 = readlane_d(b1, i), y2 = readlane_d(b2, i);
        double x0, x1, x2;
        solve3(iv, y0, y1, y2, x0, x1, x2);
        if (lane == i) {
          double* xo = xb + 3 * (size_t)(nI + i);
          xo[0] = x0;
          xo[1] = x1;
          xo[2] = x2;
        }
        if (lane < i) {
          b0 -= W[0] * x0 + W[3] * x1 + W[6] * x2;
          b1 -= W[1] * x0 + W[4] * x1 + W[7] * x2;
          b2 -= W[2] * x0 + W[5] * x1 + W[8] * x2;
        }
      }
    }
    __syncthreads();
    if (tid == 0) res->phase[5] = (unsigned long long)wall_clock64();
    // the separator block is dead until the next linearisation writes into it: clear it now (the barriers of the
    // backward levels / the update separate this from those writes)
    for (int k = tq; k < D.tri; k += kDT) Sd[k] = 0.0;
    // ---- sparse levels, backward
    for (int l = NL - 1; l >= 0; --l) {
      const int s0 = lslot[l], s1 = lslot[l + 1];
      for (int s = s0 + tq; s < s1; s += kDT) {
        const int2 rc = D.slot_rc[s];
        const int row = rc.x, col = rc.y;
        double v[3] = {0.0, 0.0, 0.0};
        double d00 = 1, d01 = 0, d02 = 0, d11 = 1, d12 = 0, d22 = 1;
        if (col >= 0) {
          const double2* dk = reinterpret_cast<const double2*>(D.Wd + kBS * (size_t)col);
          const double2 q0 = dk[0], q1 = dk[1], q2 = dk[2];
          d00 = q0.x; d01 = q0.y; d02 = q1.x; d11 = q2.x; d12 = q2.y;
          d22 = D.Wd[kBS * (size_t)col + 8];
        }
        if (col >= 0 && row >= 0) {
          double W[9];
          load9g(D.Wo + kBS * (size_t)s, W);
          const double x0 = xb[3 * row], x1 = xb[3 * row + 1], x2 = xb[3 * row + 2];
#pragma unroll
          for (int q = 0; q < 3; ++q) v[q] = W[q] * x0 + W[3 + q] * x1 + W[6 + q] * x2;
        }
        const int key = col >= 0 ? col : -1 - lane;
        seg_scan<3>(key, v, lane);
        const int nk = next_lane_key(key);
        if (col >= 0 && (lane == 63 || nk != key)) {
          double iv[6];
          const bool ok = inv_sym3(d00, d01, d02, d11, d12, d22, iv);
          const double r0 = xb[3 * col] - v[0], r1 = xb[3 * col + 1] - v[1], r2 = xb[3 * col + 2] - v[2];
          double x0, x1, x2;
          solve3(iv, r0, r1, r2, x0, x1, x2);
          xb[3 * col] = x0;
          xb[3 * col + 1] = x1;
          xb[3 * col + 2] = x2;
          if (!ok) atomicOr(&fail_flag, 1);
        }
      }
      __syncthreads();
    }
    if (tid == 0) res->phase[6] = (unsigned long long)wall_clock64();
    // ---- update (VertexSE2::oplusImpl) unless the factorisation failed or the step is not finite
    bool bad = false;
    for (int f = tq; f < n; f += kDT) bad |= !(isfinite(xb[3 * f]) && isfinite(xb[3 * f + 1]) && isfinite(xb[3 * f + 2]));
    if (bad) atomicOr(&fail_flag, 2);
    __syncthreads();
    fail = fail_flag;
    if (fail) break;
    for (int f = tq; f < n; f += kDT) {
      const size_t v = 3 * (size_t)D.pos_vertex[f];
      poses[v] += xb[3 * f];
      poses[v + 1] += xb[3 * f + 1];
      poses[v + 2] = norm_theta(poses[v + 2] + xb[3 * f + 2]);
    }
    ++done;
    __syncthreads();
    if (tid == 0) res->phase[7] = (unsigned long long)wall_clock64();
  }
  if (tid == 0) {
    if (fail) res->fail_iter = done;   // hist[2 done] holds the chi2 at the poses that stay
    res->done = done;
    res->cycles = __builtin_amdgcn_s_memtime() - clk0;
    res->fail = (fail & 1) ? 1 : (fail ? 2 : 0);
    res->stamp[fail ? 2 * done + 2 : 2 * iters + 1] = (unsigned long long)wall_clock64();   // end of the call
  }
}

// ------------------------------------------------------------------------------------------------ host
}  // namespace

struct Direct {
  DirectDev dev{};
  DirectInfo info{};
  // host copies of the uploaded lists must outlive the asynchronous copies
  std::vector<int> h_pos_vertex, h_vover, h_zero, h_enext, h_lmeta;
  std::vector<int4> h_vrec, h_tk, h_ctr;
  std::vector<int2> h_multi, h_slot_rc;
  std::vector<unsigned> h_edge_tgt;
  std::vector<unsigned short> h_dpair;
  std::vector<char> h_blob;   // what is uploaded: must outlive the asynchronous copy
  bool xb_global = false;
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
  // ---- symbolic factorisation of the sparse columns (flat storage: this runs before every optimize(20) of the
  //      reference's loop, slc.cpp:286, and a thousand small vectors cost more than the arithmetic)
  struct Span {
    const int *b, *e;
    size_t size() const { return (size_t)(e - b); }
    bool empty() const { return b == e; }
    const int* begin() const { return b; }
    const int* end() const { return e; }
    int operator[](size_t q) const { return b[q]; }
  };
  std::vector<int> st_ptr((size_t)nI + 1, 0), st_idx, first_kid((size_t)std::max(nI, 1), -1), next_sib((size_t)std::max(nI, 1), -1),
      level((size_t)std::max(nI, 1), 0);
  st_idx.reserve(6 * (size_t)nI + 64);
  int NL = 0;
  {
    std::vector<int> lptr((size_t)nI + 1, 0), lidx;
    for (auto& pr : pairs) {
      const int a = std::min(pos[pr.first], pos[pr.second]);
      if (a < nI) ++lptr[a + 1];
    }
    for (int k = 0; k < nI; ++k) lptr[k + 1] += lptr[k];
    lidx.resize((size_t)lptr[nI]);
    std::vector<int> cur(lptr.begin(), lptr.end() - 1);
    for (auto& pr : pairs) {
      const int a = std::min(pos[pr.first], pos[pr.second]), b = std::max(pos[pr.first], pos[pr.second]);
      if (a < nI) lidx[cur[a]++] = b;
    }
    std::vector<int> v;
    v.reserve(256);
    for (int k = 0; k < nI; ++k) {
      v.assign(lidx.begin() + lptr[k], lidx.begin() + lptr[k + 1]);
      for (int c = first_kid[k]; c >= 0; c = next_sib[c]) {
        for (int q = st_ptr[c]; q < st_ptr[c + 1]; ++q)
          if (st_idx[q] != k) v.push_back(st_idx[q]);
        level[k] = std::max(level[k], level[c] + 1);
      }
      std::sort(v.begin(), v.end());
      v.erase(std::unique(v.begin(), v.end()), v.end());
      if ((int)v.size() > kMaxColumn) return no("a column has more blocks than a wave holds");
      st_idx.insert(st_idx.end(), v.begin(), v.end());
      st_ptr[k + 1] = (int)st_idx.size();
      if (!v.empty() && v[0] < nI) {
        next_sib[k] = first_kid[v[0]];
        first_kid[v[0]] = k;
      }
      NL = std::max(NL, level[k] + 1);
    }
  }
  auto ST = [&](int k) { return Span{st_idx.data() + st_ptr[k], st_idx.data() + st_ptr[k + 1]}; };
  if (NL > kMaxLevels) return no("elimination tree too deep");
  const int tri = 3 * ns * (3 * ns + 1) / 2;
  const size_t lds_fixed = sizeof(double) * ((size_t)tri + 9 * (size_t)ns + 32) + sizeof(int) * 2 * ((size_t)NL + 1) +
                           sizeof(unsigned short) * ((size_t)ns * (ns + 1) / 2 + 4);
  const bool xb_global = lds_fixed + sizeof(double) * 3 * (size_t)n > kLdsBudget;   // the vector then lives in global memory
  const size_t lds_bytes = lds_fixed + (xb_global ? 0 : sizeof(double) * 3 * (size_t)n);
  if (lds_bytes > kLdsBudget) return no("separator block exceeds the LDS");

  std::unique_ptr<Direct> owner(new Direct());   // released to the caller at the end: a throwing std::vector below frees it
  Direct* d = owner.get();
  std::vector<std::vector<int>> lcols((size_t)NL);
  for (int k = 0; k < nI; ++k) lcols[level[k]].push_back(k);
  // ---- slots: per level, per column (ascending), a column's blocks inside one wave
  std::vector<int> colbase((size_t)nI, 0);
  std::vector<int> srow, scol, lslot((size_t)NL + 1, 0), ltask((size_t)NL + 1, 0);
  for (int l = 0; l < NL; ++l) {
    lslot[l] = (int)srow.size();
    for (int k : lcols[l]) {
      const int len = std::max<int>(1, (int)ST(k).size());
      const int cur = (int)srow.size();
      if ((cur & 63) + len > 64)
        for (int q = cur; q < ((cur + 63) & ~63); ++q) {
          srow.push_back(-1);
          scol.push_back(-1);
        }
      colbase[k] = (int)srow.size();
      if (ST(k).empty()) {
        srow.push_back(-1);
        scol.push_back(k);
      }
      for (int i : ST(k)) {
        srow.push_back(i);
        scol.push_back(k);
      }
    }
    while (srow.size() & 63) {
      srow.push_back(-1);
      scol.push_back(-1);
    }
  }
  lslot[NL] = (int)srow.size();
  const int NB = (int)srow.size();
  d->h_slot_rc.resize((size_t)NB);
  for (int q = 0; q < NB; ++q) d->h_slot_rc[q] = make_int2(srow[q], scol[q]);
  auto slot_of = [&](int i, int k) -> int {   // stored block (row i, column k), k < nI
    const Span v = ST(k);
    const auto it = std::lower_bound(v.begin(), v.end(), i);
    return colbase[k] + (int)(it - v.begin());
  };
  // ---- assembly lists
  std::vector<int> vertex_pos((size_t)V, -1);
  d->h_pos_vertex.resize((size_t)n);
  for (int p = 0; p < n; ++p) {
    d->h_pos_vertex[p] = free_id[order[p]];
    vertex_pos[free_id[order[p]]] = p;
  }
  {   // incident edges of every position: four inline, the rest in an overflow list
    std::vector<int> iptr((size_t)n + 1, 0), iidx;
    for (int e = 0; e < E; ++e) {
      const int a = vertex_pos[ei[e]], b = vertex_pos[ej[e]];
      if (a >= 0) ++iptr[a + 1];
      if (b >= 0) ++iptr[b + 1];
    }
    for (int p = 0; p < n; ++p) iptr[p + 1] += iptr[p];
    iidx.resize((size_t)iptr[n]);
    std::vector<int> cur(iptr.begin(), iptr.end() - 1);
    for (int e = 0; e < E; ++e) {
      const int a = vertex_pos[ei[e]], b = vertex_pos[ej[e]];
      if (a >= 0) iidx[cur[a]++] = e << 1;
      if (b >= 0) iidx[cur[b]++] = e << 1 | 1;
    }
    d->h_vrec.resize((size_t)n);
    for (int p = 0; p < n; ++p) {
      const int* v = iidx.data() + iptr[p];
      const int cnt = iptr[p + 1] - iptr[p];
      int4 r = make_int4(-1, -1, -1, -1);
      const int inl = cnt <= 4 ? cnt : 3;
      if (inl > 0) r.x = v[0];
      if (inl > 1) r.y = v[1];
      if (inl > 2) r.z = v[2];
      if (inl > 3) r.w = v[3];
      if (cnt > inl) {
        r.w = -2 - (int)d->h_vover.size();
        d->h_vover.push_back(cnt - inl);
        for (int q = inl; q < cnt; ++q) d->h_vover.push_back(v[q]);
      }
      d->h_vrec[p] = r;
    }
  }
  // off-diagonal blocks: the stored block is H[row = later position][col = earlier position]; an edge holds H[vi][vj]
  std::vector<int> slot_first((size_t)NB, -1), slot_cnt((size_t)NB, 0), ss_first((size_t)ns * ns, -1), ss_cnt((size_t)ns * ns, 0);
  d->h_enext.assign((size_t)std::max(E, 1), -1);
  d->h_edge_tgt.assign((size_t)std::max(E, 1), 0u);
  std::vector<unsigned> tgt_of((size_t)std::max(E, 1), 0u);
  for (int e = E - 1; e >= 0; --e) {   // descending: the lists come out in ascending edge order
    const int a = vertex_pos[ei[e]], b = vertex_pos[ej[e]];
    if (a < 0 || b < 0) continue;
    const int row = std::max(a, b), col = std::min(a, b);
    const unsigned tr = row == a ? 0u : 1u;
    int *head, *cnt;
    unsigned tgt;
    if (col < nI) {
      const int sl = slot_of(row, col);
      head = &slot_first[sl];
      cnt = &slot_cnt[sl];
      tgt = 1u << 30 | (unsigned)sl;
    } else {
      const size_t q = (size_t)(row - nI) * ns + (col - nI);
      head = &ss_first[q];
      cnt = &ss_cnt[q];
      tgt = 2u << 30 | (unsigned)((row - nI) << 12 | (col - nI));
    }
    d->h_enext[e] = *head;
    *head = (int)(e << 1 | (int)tr);
    ++*cnt;
    tgt_of[e] = tgt;
    d->h_edge_tgt[e] = tgt | tr << 29;
  }
  for (int e = 0; e < E; ++e) {
    const unsigned tgt = tgt_of[e];
    if (!tgt) continue;
    const bool dense = (tgt >> 30) == 2;
    const int idx = (int)(tgt & 0x1fffffffu);
    const size_t q = dense ? (size_t)(idx >> 12) * ns + (idx & 4095) : (size_t)idx;
    const int cnt = dense ? ss_cnt[q] : slot_cnt[q];
    if (cnt > 1) {
      d->h_edge_tgt[e] = 3u << 30;
      const int first = dense ? ss_first[q] : slot_first[q];
      if ((first >> 1) == e) d->h_multi.push_back(make_int2((int)tgt, first));   // once per pair
    }
  }
  for (int q = 0; q < NB; ++q)
    if (srow[q] >= 0 && scol[q] >= 0 && slot_cnt[q] == 0) d->h_zero.push_back(q);
  // ---- forward tasks per level: (target, slot a, slot b) sorted by target
  struct Contrib {
    unsigned target;
    int a, b, k;
  };
  std::vector<Contrib> cl;
  long long total_contrib = 0;
  for (int l = 0; l < NL; ++l) {
    cl.clear();
    for (int k : lcols[l]) {
      const Span v = ST(k);
      for (size_t x = 0; x < v.size(); ++x) {
        const int i = v[x], sa = colbase[k] + (int)x;
        for (size_t y = 0; y <= x; ++y) {
          const int j = v[y], sb = colbase[k] + (int)y;   // i >= j
          unsigned tg;
          if (j >= nI) tg = T_DENSE << 28 | (unsigned)((i - nI) << 12 | (j - nI));
          else if (i == j) tg = T_DIAG << 28 | (unsigned)j;
          else tg = T_OFF << 28 | (unsigned)slot_of(i, j);
          cl.push_back({tg, sa, sb, k});
        }
      }
    }
    total_contrib += (long long)cl.size();
    if (total_contrib > kMaxContrib) {
      return no("too much fill");
    }
    std::stable_sort(cl.begin(), cl.end(), [](const Contrib& x, const Contrib& y) { return x.target < y.target; });
    ltask[l] = (int)d->h_tk.size();
    for (size_t q = 0; q < cl.size(); ++q) {
      if (q == 0 || cl[q].target != cl[q - 1].target) {
        if (!d->h_tk.empty() && (int)d->h_tk.size() > ltask[l]) d->h_tk.back().z = (int)d->h_ctr.size();
        d->h_tk.push_back(make_int4((int)cl[q].target, (int)d->h_ctr.size(), 0, 0));
      }
      d->h_ctr.push_back(make_int4(cl[q].a, cl[q].b, cl[q].k, 0));
    }
    if ((int)d->h_tk.size() > ltask[l]) d->h_tk.back().z = (int)d->h_ctr.size();
  }
  ltask[NL] = (int)d->h_tk.size();
  d->h_lmeta = lslot;
  d->h_lmeta.insert(d->h_lmeta.end(), ltask.begin(), ltask.end());
  // ---- dense pair table: bj descending so that the trailing blocks of pivot p are a prefix
  for (int bj = ns - 1; bj >= 0; --bj)
    for (int bi = bj; bi < ns; ++bi) d->h_dpair.push_back((unsigned short)(bi << 8 | bj));

  hipError_t e = hipSuccess;
  DirectDev& D = d->dev;
  D.n = n; D.nI = nI; D.ns = ns; D.NL = NL; D.E = E; D.NB = NB; D.tri = tri;
  D.nzero = (int)d->h_zero.size();
  D.nmulti = (int)d->h_multi.size();
  {   // all lists in ONE host blob, one allocation, one copy (a dozen small pageable copies cost 0.1 ms)
    std::vector<char>& blob = d->h_blob;
    size_t off[12], total = 0;
    int q = 0;
    auto put = [&](const void* src, size_t bytes) {
      total = (total + 15) & ~(size_t)15;
      off[q++] = total;
      blob.resize(total + bytes);
      if (bytes) std::memcpy(blob.data() + total, src, bytes);
      total += bytes;
    };
    put(d->h_pos_vertex.data(), sizeof(int) * d->h_pos_vertex.size());
    put(d->h_vrec.data(), sizeof(int4) * d->h_vrec.size());
    put(d->h_vover.data(), sizeof(int) * d->h_vover.size());
    put(d->h_edge_tgt.data(), sizeof(unsigned) * d->h_edge_tgt.size());
    put(d->h_zero.data(), sizeof(int) * d->h_zero.size());
    put(d->h_multi.data(), sizeof(int2) * d->h_multi.size());
    put(d->h_enext.data(), sizeof(int) * d->h_enext.size());
    put(d->h_slot_rc.data(), sizeof(int2) * d->h_slot_rc.size());
    put(d->h_lmeta.data(), sizeof(int) * d->h_lmeta.size());
    put(d->h_tk.data(), sizeof(int4) * d->h_tk.size());
    put(d->h_ctr.data(), sizeof(int4) * d->h_ctr.size());
    put(d->h_dpair.data(), sizeof(unsigned short) * d->h_dpair.size());
    char* base = (char*)arena->take(std::max<size_t>(total, 16));
    if (!base) e = hipErrorOutOfMemory;
    else e = hipMemcpyAsync(base, blob.data(), total, hipMemcpyHostToDevice, s);
    if (base) {
      D.pos_vertex = (const int*)(base + off[0]);
      D.vrec = (const int4*)(base + off[1]);
      D.vover = (const int*)(base + off[2]);
      D.edge_tgt = (const unsigned*)(base + off[3]);
      D.zero_slots = (const int*)(base + off[4]);
      D.multi = (const int2*)(base + off[5]);
      D.enext = (const int*)(base + off[6]);
      D.slot_rc = (const int2*)(base + off[7]);
      D.lmeta = (const int*)(base + off[8]);
      D.tk = (const int4*)(base + off[9]);
      D.ctr = (const int4*)(base + off[10]);
      D.dpair = (const unsigned short*)(base + off[11]);
    }
  }
  D.Wd = (double*)arena->take(sizeof(double) * kBS * (size_t)std::max(nI, 1));
  D.Wo = (double*)arena->take(sizeof(double) * kBS * (size_t)std::max(NB, 1));
  D.escr = (double*)arena->take(sizeof(double) * 27 * (size_t)std::max(E, 1));
  D.zsc = (double*)arena->take(sizeof(double) * 2 * (size_t)std::max(E, 1));
  D.xbg = xb_global ? (double*)arena->take(sizeof(double) * 3 * (size_t)n) : nullptr;
  if (xb_global && !D.xbg && e == hipSuccess) e = hipErrorOutOfMemory;
  d->xb_global = xb_global;
  if (e == hipSuccess && (!D.Wd || !D.Wo || !D.escr || !D.zsc)) e = hipErrorOutOfMemory;
  static std::atomic<unsigned long long> attr_devices{0};   // devices on which the LDS limit of the two kernels is raised
  int dev_id = 0;
  hipGetDevice(&dev_id);
  const unsigned long long dev_bit = 1ull << (dev_id & 63);
  const bool attr_set_already = (attr_devices.load() & dev_bit) != 0;
  bool attr_set = attr_set_already;
  if (e == hipSuccess && !attr_set) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_direct<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_direct<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
    attr_set = e == hipSuccess;
    if (attr_set) attr_devices.fetch_or(dev_bit);
  }
  if (e != hipSuccess) {
    *err = std::string("direct_create: ") + hipGetErrorString(e);
    return nullptr;
  }
  d->info.n = n;
  d->info.n_chain = nI;
  d->info.n_sep = ns;
  d->info.levels = NL;
  d->info.slots = NB;
  d->info.contributions = (int)total_contrib;
  d->info.lds_bytes = lds_bytes;
  return owner.release();
}

hipError_t direct_optimize(Direct* d, hipStream_t s, const EdgeListDev& el, double* d_poses, int iters, double* d_hist,
                           DirectResult* d_res) {
  if (d->xb_global) SGO_LAUNCH(k_direct<true>, dim3(1), dim3(kDT), d->info.lds_bytes, s, d->dev, el, d_poses, iters, d_hist, d_res);
  else SGO_LAUNCH(k_direct<false>, dim3(1), dim3(kDT), d->info.lds_bytes, s, d->dev, el, d_poses, iters, d_hist, d_res);
  return hipGetLastError();
}

}  // namespace sgo
