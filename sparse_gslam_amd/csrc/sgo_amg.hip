// sgo_amg.hip -- rigid-body smoothed-aggregation multigrid preconditioner for the GN Hessian.
//
// Why it exists: the block-Jacobi PCG that BASELINE.json's north_star names needs 8 000+
// iterations per GN iteration on the 10k-pose graph and does not reach 1e-8 in 20 000 on the
// 100k-pose graph (profiles/r01_bj_c4_kernel_stats.csv), because the pose-graph Hessian is a
// vector-Laplacian-like operator whose low-energy modes are near-rigid motions of large parts
// of the trajectory.  This preconditioner keeps block-Jacobi as the smoother and adds a
// coarse-grid hierarchy that represents exactly those modes:
//   * nodes are aggregated by strength of connection (greedy root-node aggregation on
//     Frobenius norms of the 3x3 blocks, threshold theta) -- structure built on the host once
//     per sgo_set_graph_se2 (like g2o's symbolic analysis, once per optimize());
//   * the tentative prolongator T is the rigid-body one: an aggregate moves as a rigid body
//     (u_x, u_y, w) about its centre c, so node i at p_i gets T(p_i - c) u with
//     T(d) = [[1,0,-d_y],[0,1,d_x],[0,0,1]] -- the three rigid motions of SE(2) are represented
//     exactly on every level (they are the null space of every edge's Jacobian pair);
//   * the prolongator used is T smoothed by one damped block-Jacobi step, P = (I - w D^-1 H) T
//     (smoothed aggregation); a level whose smoothed coarse operator would be too dense keeps T;
//   * coarse operators are Galerkin products P^T H P, recomputed on the device every GN
//     iteration (values change, structure does not) from product lists the host made at set-up:
//     one lane per block product in target order, wavefront segmented scan per target block --
//     no atomics, bitwise reproducible;
//   * cycle: V-cycle with one damped block-Jacobi sweep before and after (levels that keep T: K-cycle,
//     two flexible-CG steps per level, Notay's AGMG scheme, with the prolongation and the FCG vector
//     updates fused into the SpMV-type launches k_spmv<4..6>); the coarsest level (<= 400 nodes) is
//     solved with an explicit dense inverse recomputed every GN iteration (blocked Gauss-Jordan).
// All launches go to the caller's stream with fixed pointers, so a whole PCG iteration
// including the cycle is captured into one hipGraph.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "sgo_amg.h"
#include "sgo_amg_host.h"
#include "sgo_sort.h"
#include "sgo_comm.h"
#include "sgo_device.h"

namespace sgo {

namespace {

// --------------------------------------------------------------------------------- kernels
// Frobenius norm of every slot's block (strength of connection input).
__global__ __launch_bounds__(kBlock) void k_block_norms(BsrDev A, double* __restrict__ w) {
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < A.nslot; k += gridDim.x * kBlock) {
    double b[9];
    load_block(A, (size_t)k, b);
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) s += b[c] * b[c];
    w[k] = sqrt(s);
  }
}

// Level-0 node positions from the pose array: pos[h] = poses[free_id[h]].xy
__global__ __launch_bounds__(kBlock) void k_positions0(int n, const int* __restrict__ free_id,
                                                       const double* __restrict__ poses, double* __restrict__ pos) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const size_t v = 3 * (size_t)free_id[i];
    pos[2 * (size_t)i] = poses[v];
    pos[2 * (size_t)i + 1] = poses[v + 1];
  }
}

// Aggregate centres and lever arms: one wave per aggregate (lanes stride over the members, wave-wide sums in a fixed
// order).  An aggregate may hold hundreds of members -- the last level of a stalled hierarchy collapses into one --
// which one thread per aggregate walked serially (57 us per launch on C4 with random closures, 24 us on C4).
__global__ __launch_bounds__(kBlock) void k_centres(int nc, const int* __restrict__ mem_ptr, const int* __restrict__ mem,
                                                    const double* __restrict__ pos, double* __restrict__ cpos,
                                                    double* __restrict__ d) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), nwaves = gridDim.x * kWavesPerBlock;
  for (int a = wave; a < nc; a += nwaves) {
    const int lo = mem_ptr[a], hi = mem_ptr[a + 1];
    double sx = 0.0, sy = 0.0;
    for (int t = lo + lane; t < hi; t += 64) {
      const int i = mem[t];
      sx += pos[2 * (size_t)i];
      sy += pos[2 * (size_t)i + 1];
    }
    sx = wave_sum(sx);   // (the total, in every lane)
    sy = wave_sum(sy);
    const double inv = 1.0 / (double)(hi - lo);
    const double cx = sx * inv, cy = sy * inv;
    if (lane == 0) {
      cpos[2 * (size_t)a] = cx;
      cpos[2 * (size_t)a + 1] = cy;
    }
    for (int t = lo + lane; t < hi; t += 64) {
      const int i = mem[t];
      d[2 * (size_t)i] = pos[2 * (size_t)i] - cx;
      d[2 * (size_t)i + 1] = pos[2 * (size_t)i + 1] - cy;
    }
  }
}

// Galerkin product: coarse slot s gets sum over its fine slots k=(i,j) of T_i^T B_k T_j.
// One lane per contribution, contributions sorted by coarse slot, segmented scan by coarse slot.
struct GalerkinMap {
  int n = 0;              // contributions (= fine slots)
  int ngrp = 0;
  const int* src = nullptr;  // fine slot of contribution t
  const int* tgt = nullptr;  // coarse slot of contribution t
  const int* grp = nullptr;  // wave groups aligned to coarse-slot boundaries
};
// Multi-GPU (row-owner mode): only the fine slots of the rows [row0, row1) contribute (row1 == 0: all); the ranks'
// partial coarse operators are summed by an all-reduce.
__global__ __launch_bounds__(kBlock) void k_galerkin(BsrDev F, BsrDev C, GalerkinMap g, const double* __restrict__ d, int row0, int row1) {
  const int lane = threadIdx.x & 63;
  const size_t ncs = (size_t)C.nslot;
  int gi, gend, gstride;
  group_walk(g.ngrp, &gi, &gend, &gstride);
  for (; gi < gend; gi += gstride) {
    const int gb = g.grp[gi], ge = g.grp[gi + 1];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int key = -1 - lane;
    for (int t = gb + lane; t < ge; t += 64) {
      key = g.tgt[t];
      const int k = g.src[t];
      const int i = F.row[k], j = F.col[k];
      if (row1 > 0 && (i < row0 || i >= row1)) continue;
      const double dxi = d[2 * (size_t)i], dyi = d[2 * (size_t)i + 1];
      const double dxj = d[2 * (size_t)j], dyj = d[2 * (size_t)j + 1];
      double b[9];
      load_block(F, (size_t)k, b);
      // M = B T_j : third column = -dy_j * col0 + dx_j * col1 + col2
      const double m02 = -dyj * b[0] + dxj * b[1] + b[2];
      const double m12 = -dyj * b[3] + dxj * b[4] + b[5];
      const double m22 = -dyj * b[6] + dxj * b[7] + b[8];
      // C = T_i^T M : third row = -dy_i * row0 + dx_i * row1 + row2
      acc[0] += b[0];
      acc[1] += b[1];
      acc[2] += m02;
      acc[3] += b[3];
      acc[4] += b[4];
      acc[5] += m12;
      acc[6] += -dyi * b[0] + dxi * b[3] + b[6];
      acc[7] += -dyi * b[1] + dxi * b[4] + b[7];
      acc[8] += -dyi * m02 + dxi * m12 + m22;
    }
    seg_scan<9>(key, acc, lane);
    const int kn = next_lane_key(key);
    if (key >= 0 && (lane == 63 || kn != key)) {
#pragma unroll
      for (int c = 0; c < 9; ++c) C.blk[blk_at(c, key, ncs)] = acc[c];
    }
  }
}

// ------------------------------------------------------------------ smoothed aggregation
// Prolongator smoothed by one damped block-Jacobi step, P = (I - w D^-1 A) T, and the Galerkin
// operator A_c = P^T A P formed in two sparse products (AP = A P, then P^T AP).  The sparsity of
// P, AP and A_c and the list of block products behind every entry are fixed by the graph and the
// aggregation: the host lists them once (amg_create), sorted by target entry, and the numeric
// phase of every GN iteration is three segmented-sum kernels of the k_galerkin kind.
struct ProdMap {
  int n = 0;                 // products
  int ngrp = 0;
  const int* a = nullptr;    // left operand of product t
  const int* b = nullptr;    // right operand
  const int* tgt = nullptr;  // target entry (non-decreasing in t)
  const int* grp = nullptr;  // wave groups aligned to target boundaries
};
struct PDev {
  int np = 0;                // blocks of P
  int stream_nt = 1;         // the transfer kernels read r_blk / t_blk with non-temporal loads (large levels: the stream must
                             // not displace the level-0 matrix from the Infinity Cache); small levels (0) stay in L2 between cycles
  int* rowptr = nullptr;     // [n + 1] entries of fine row i: [rowptr[i], rowptr[i+1]), coarse cols ascending
  int* row = nullptr;        // [np]
  int* col = nullptr;        // [np]
  double* blk = nullptr;     // [np][9]: the copy the block products GATHER from (one 72-byte record each)
  // The two streamed copies are indexed by GLOBAL entry / position numbers; in multi-GPU row-owner mode (level 0) they hold
  // this rank's rows' entries only: allocated for the rank's range, base pointers shifted, pair strides r_n / t_n and
  // component-8 base pointers of their own (as Sym0Dev::ublk).
  // They are fp32 (the transfers are inside the preconditioner; the Galerkin products gather the fp64 records of blk):
  // quads (0..3), (4..7) as two float4 per entry + component 8, i.e. two 16-byte loads and one 4-byte load per lane.
  int r_n = 0;               // quad stride of r_blk (= np; the rank's entries in row-owner mode)
  float* r_blk = nullptr;    // quad-SoA: the copy the prolongation STREAMS (row order,
                             // non-temporal loads: read once per cycle, must not push the level-0 matrix out of the MALL)
  float* r_blk8 = nullptr;
  int* r_grp = nullptr;      // wave groups over the entries aligned to fine rows (prolongation)
  int r_ngrp = 0;
  ProdMap val;               // a = fine slot k = (i, j), tgt = entry (i, agg(j))
  // the same entries grouped by coarse column, with their own copy of the blocks, so that the
  // restriction streams too: position t holds entry t_idx[t]
  int* t_pos = nullptr;      // [np] position of entry e in column order
  int* t_row = nullptr;      // [np] fine row at position t
  int* t_col = nullptr;      // [np] coarse column at position t
  int t_n = 0;               // entries in the column-ordered copy = its quad stride (= np; the rank's rows' entries in row-owner mode)
  float* t_blk = nullptr;    // quad-SoA: column order, streamed by the restriction (non-temporal loads)
  float* t_blk8 = nullptr;
  int* t_grp = nullptr;      // wave groups over the positions aligned to columns
  int t_ngrp = 0;
  int* t_long = nullptr;     // [t_nlong][2] position ranges of the columns longer than kLongColumn entries (the last level of a
  int t_nlong = 0;           // stalled hierarchy: thousands of fine rows in two aggregates): one WORKGROUP each (k_restrict_p_long)
  int nap = 0;               // blocks of AP
  double* apblk = nullptr;   // [nap][9]
  ProdMap ap;                // a = fine slot (i, j), b = P entry (j, c), tgt = AP entry (i, c)
  ProdMap rap;               // a = P entry (i, a), b = AP entry (i, c), tgt = coarse slot (a, c), c >= a only
  int* rap_mirror = nullptr; // [coarse slots] slot (c, a) of an upper slot (a, c): gets the transposed block (-1: none)
  bool local_lists = false;  // row-owner mode: val / ap / rap list THIS rank's rows' products only (targets of P^T A P without
                             // any are not visited: the caller zeroes the coarse blocks first)
  // FILTERED smoothing (SaHost::filtered): P = (I - w D_F^-1 A_F) T with the operator of the strong connections.  val lists
  // the kept slots only; the diagonal block and its inverse come from here instead of from the level's operator
  const unsigned char* strong = nullptr;   // [nslot]
  double* dF = nullptr;                    // [n][9] D_F, row-major (k_filtered_diag; NOT symmetric on the coarse levels, see there)
  double* dinvF = nullptr;                 // [n][9] its inverse
  int* f_grp = nullptr;                    // wave groups over the level's slots aligned to rows (level 0's logical view has none)
  int f_ngrp = 0;
};

// entry e of a streamed fp32 copy of P (quad-SoA): two 16-byte loads and one 4-byte load per lane; non-temporal on the
// large levels (read once per cycle: the stream must not push the level-0 matrix out of the Infinity Cache)
typedef double sgo_d2 __attribute__((ext_vector_type(2)));
typedef float sgo_f4 __attribute__((ext_vector_type(4)));
constexpr int kLongColumn = 512;   // entries of a P column from which a workgroup instead of a wave restricts it
__device__ __forceinline__ void load9_pairs(const float* __restrict__ base, const float* __restrict__ base8, size_t e, size_t n,
                                            double (&v)[9], bool nt) {
  const sgo_f4* __restrict__ bp = reinterpret_cast<const sgo_f4*>(base);
  sgo_f4 q0, q1;
  float f8;
  if (nt) {
    q0 = __builtin_nontemporal_load(bp + e);
    q1 = __builtin_nontemporal_load(bp + n + e);
    f8 = __builtin_nontemporal_load(base8 + e);
  } else {
    q0 = bp[e];
    q1 = bp[n + e];
    f8 = base8[e];
  }
  v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w; v[8] = f8;
}
__device__ __forceinline__ void load9(const double* __restrict__ base, size_t e, double (&v)[9]) {
#pragma unroll
  for (int c = 0; c < 9; ++c) v[c] = base[9 * e + c];
}

// Diagonal blocks of the FILTERED operator A_F (smoothing the tentative transfer with the strong connections only, SaHost in
// sgo_amg_host.h).  A weak block A_ij is dropped TOGETHER WITH its share of the diagonal block: every edge's contribution to row
// i annihilates the rigid motions, A_ii^e T(p_i) + A_ij^e T(p_j) = 0 with T(p) = [[1,0,-p_y],[0,1,p_x],[0,0,1]], hence
// A_ii^e = -A_ij^e G_ji with G_ji = T(p_j) T(p_i)^-1 = T(p_j - p_i) -- which holds for the assembled blocks and, because the
// transfers reproduce the rigid motions exactly, for the Galerkin operators of the coarse levels too.  So
//     D_F,i = D_i + sum over the weak slots k = (i, j) of A_k T(p_j - p_i)
// is the diagonal block the strong connections alone would have assembled (anchored rows keep their anchor: an edge to a fixed
// vertex has no off-diagonal block to be weak), and A_F T = 0 in the interior exactly as A T = 0: the smoothed columns keep
// representing the rigid motions.  D_F is NOT symmetrised: on level 0 every term A_k T(p_j - p_i) = -A_ii^e is symmetric, but a
// coarse level's operator is a sum of elements over up to four nodes each (a fine edge seen through the smoothed transfer), only
// the SUM over an element's nodes is symmetric, and the symmetric part alone does not keep the row sums: A_F T = O(1e-5 |D|)
// instead of 0, the next level's transfer then misses the rigid motions by as much -- five orders of magnitude above the weak
// connections' own energy -- and the cycle stalls (measured on a numpy prototype of the whole hierarchy, 20k poses from a
// dead-reckoned start: 141 PCG iterations with the symmetrised block, 36 without; the tentative transfer: 266).  D_F is only a
// recipe for P; the coarse operator P^T A P is symmetric whatever it is.
// One lane per slot, rows in wave groups, segmented scan per row; the row's last lane adds D_i and inverts (general 3x3).
// Rows [row0, row1) only when row1 > 0.
__global__ __launch_bounds__(kBlock) void k_filtered_diag(BsrDev F, const int* __restrict__ grp, int ngrp, const unsigned char* __restrict__ strong,
                                                          const double* __restrict__ pos, double* __restrict__ dF, double* __restrict__ dinvF,
                                                          int row0, int row1) {
  const int lane = threadIdx.x & 63;
  int gi, gend, gstride;
  group_walk(ngrp, &gi, &gend, &gstride);
  for (; gi < gend; gi += gstride) {
    const int gb = grp[gi], ge = grp[gi + 1];
    double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // [9]: the row's kept off-diagonal slots (counted)
    int key = -1 - lane;
    for (int k = gb + lane; k < ge; k += 64) {
      const int i = F.row[k];
      if (row1 > 0 && (i < row0 || i >= row1)) {
        key = -1 - lane;
        continue;
      }
      key = i;
      const int j = F.col[k];
      if (j == i) continue;
      if (strong[k]) {
        acc[9] += 1.0;
        continue;
      }
      const double dx = pos[2 * (size_t)j] - pos[2 * (size_t)i], dy = pos[2 * (size_t)j + 1] - pos[2 * (size_t)i + 1];
      double b[9];
      load_block(F, (size_t)k, b);
      acc[0] += b[0]; acc[1] += b[1]; acc[2] += -dy * b[0] + dx * b[1] + b[2];
      acc[3] += b[3]; acc[4] += b[4]; acc[5] += -dy * b[3] + dx * b[4] + b[5];
      acc[6] += b[6]; acc[7] += b[7]; acc[8] += -dy * b[6] + dx * b[7] + b[8];
    }
    seg_scan<10>(key, acc, lane);
    const int kn = next_lane_key(key);
    if (key >= 0 && (lane == 63 || kn != key)) {
      double d[9];
      load_block(F, (size_t)F.rowptr[key], d);   // the diagonal slot comes first in its row
      // (a row without any kept connection keeps its whole diagonal block: with everything lumped D_F would be the row sum of
      // the rigid motions -- zero up to rounding in the interior -- and its "inverse" noise.  Its inverse is written as ZERO, like
      // an unsafe row's below: k_p_values still lists the row's diagonal slot, and with D_F = D that term alone would make
      // P_i = (1 - omega_p) T_i -- a row that no longer reproduces the rigid motions (both neighbours heavy hubs, or every
      // closure switched off by DCS: exactly the start the filter exists for).  With a zero inverse the row stays T_i.)
      const double keep = acc[9] > 0.0 ? 1.0 : 0.0;
      // How much of the row's stiffness the dropped connections carried: L = -acc is the sum of their (positive semi-definite)
      // shares of the diagonal block, and trace(D^-1 L) bounds the largest eigenvalue of D^-1 L.  "Negligible" is judged by
      // Frobenius norms; with full information matrices a block's norm can be dominated by one direction (the rotation's, with
      // its lever arms) while the dropped connections carry most of another -- D_F is then nearly singular there and the
      // smoothed row blows up (seen: 150 k poses / 900 k edges, full information, from a dead-reckoned start: PCG broke down
      // in the sixth Gauss-Newton iteration).  Such a row is not smoothed at all: its inverse is written as zero, which
      // k_p_values turns into the tentative row T_i -- the rigid motions stay reproduced exactly.
      const double* dv = F.dinv + 6 * (size_t)key;
      const double tr = -(dv[0] * acc[0] + dv[1] * (acc[1] + acc[3]) + dv[2] * (acc[2] + acc[6]) + dv[3] * acc[4] + dv[4] * (acc[5] + acc[7]) +
                          dv[5] * acc[8]);
      const bool unsafe = !(tr <= 0.5);   // (also when tr is not finite)
#pragma unroll
      for (int c = 0; c < 9; ++c) d[c] += keep * acc[c];
      // inverse by cofactors: inv[r][c] = cof[c][r] / det
      const double c00 = d[4] * d[8] - d[5] * d[7], c01 = d[5] * d[6] - d[3] * d[8], c02 = d[3] * d[7] - d[4] * d[6];
      const double c10 = d[2] * d[7] - d[1] * d[8], c11 = d[0] * d[8] - d[2] * d[6], c12 = d[1] * d[6] - d[0] * d[7];
      const double c20 = d[1] * d[5] - d[2] * d[4], c21 = d[2] * d[3] - d[0] * d[5], c22 = d[0] * d[4] - d[1] * d[3];
      const double det = d[0] * c00 + d[1] * c01 + d[2] * c02;
      const double id = (det != 0.0 && isfinite(det) && keep != 0.0 && !unsafe) ? 1.0 / det : 0.0;
      double* o = dF + 9 * (size_t)key;
#pragma unroll
      for (int c = 0; c < 9; ++c) o[c] = d[c];
      double* di = dinvF + 9 * (size_t)key;
      di[0] = c00 * id; di[1] = c10 * id; di[2] = c20 * id;
      di[3] = c01 * id; di[4] = c11 * id; di[5] = c21 * id;
      di[6] = c02 * id; di[7] = c12 * id; di[8] = c22 * id;
    }
  }
}

// P_e = [e is the own-aggregate entry] T_i - w D_i^-1 sum_{k in e} A_k T_col(k)
// (filtered smoothing, P.dF != nullptr: the listed slots are the kept ones, D_F in place of D_i -- also for the diagonal slot's
// own term)
// (multi-GPU, row-owner mode: the entries of the rows [row0, row1) only; row1 == 0: all)
__global__ __launch_bounds__(kBlock) void k_p_values(BsrDev F, PDev P, const int* __restrict__ agg,
                                                     const double* __restrict__ d, double omega_p, int row0, int row1) {
  const int lane = threadIdx.x & 63;
  int gi, gend, gstride;
  group_walk(P.val.ngrp, &gi, &gend, &gstride);
  for (; gi < gend; gi += gstride) {
    const int gb = P.val.grp[gi], ge = P.val.grp[gi + 1];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int key = -1 - lane;
    for (int t = gb + lane; t < ge; t += 64) {
      key = P.val.tgt[t];
      const int k = P.val.a[t];
      if (row1 > 0) {
        const int i = F.row[k];
        if (i < row0 || i >= row1) {
          key = -1 - lane;
          continue;
        }
      }
      const int j = F.col[k];
      const double dxj = d[2 * (size_t)j], dyj = d[2 * (size_t)j + 1];
      double b[9];
      load_block(F, (size_t)k, b);
      if (P.dF && k == F.rowptr[j]) {   // the diagonal slot of a filtered level (column j = its own row)
        const double* f = P.dF + 9 * (size_t)j;
#pragma unroll
        for (int c = 0; c < 9; ++c) b[c] = f[c];
      }
      acc[0] += b[0]; acc[1] += b[1]; acc[2] += -dyj * b[0] + dxj * b[1] + b[2];
      acc[3] += b[3]; acc[4] += b[4]; acc[5] += -dyj * b[3] + dxj * b[4] + b[5];
      acc[6] += b[6]; acc[7] += b[7]; acc[8] += -dyj * b[6] + dxj * b[7] + b[8];
    }
    seg_scan<9>(key, acc, lane);
    const int kn = next_lane_key(key);
    if (key >= 0 && (lane == 63 || kn != key)) {
      const size_t i = (size_t)P.row[key];
      double o[9];
      if (P.dinvF) {   // filtered smoothing: the general 3x3 inverse of D_F
        const double* di = P.dinvF + 9 * i;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          o[c] = -omega_p * (di[0] * acc[c] + di[1] * acc[3 + c] + di[2] * acc[6 + c]);
          o[3 + c] = -omega_p * (di[3] * acc[c] + di[4] * acc[3 + c] + di[5] * acc[6 + c]);
          o[6 + c] = -omega_p * (di[6] * acc[c] + di[7] * acc[3 + c] + di[8] * acc[6 + c]);
        }
      } else {
        const double* di = F.dinv + 6 * i;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          o[c] = -omega_p * (di[0] * acc[c] + di[1] * acc[3 + c] + di[2] * acc[6 + c]);
          o[3 + c] = -omega_p * (di[1] * acc[c] + di[3] * acc[3 + c] + di[4] * acc[6 + c]);
          o[6 + c] = -omega_p * (di[2] * acc[c] + di[4] * acc[3 + c] + di[5] * acc[6 + c]);
        }
      }
      if (P.col[key] == agg[i]) {
        o[0] += 1.0; o[4] += 1.0; o[8] += 1.0;
        o[2] += -d[2 * i + 1];
        o[5] += d[2 * i];
      }
      const size_t tp = (size_t)P.t_pos[key];
#pragma unroll
      for (int c = 0; c < 9; ++c) P.blk[9 * (size_t)key + c] = o[c];
      {
        const sgo_f4 q0 = {(float)o[0], (float)o[1], (float)o[2], (float)o[3]}, q1 = {(float)o[4], (float)o[5], (float)o[6], (float)o[7]};
        sgo_f4* rq = reinterpret_cast<sgo_f4*>(P.r_blk);
        sgo_f4* tq = reinterpret_cast<sgo_f4*>(P.t_blk);
        rq[key] = q0;
        rq[(size_t)P.r_n + key] = q1;
        tq[tp] = q0;
        tq[(size_t)P.t_n + tp] = q1;
        P.r_blk8[key] = (float)o[8];
        P.t_blk8[tp] = (float)o[8];
      }
    }
  }
}

// out_t = sum over the products of target t of  X(a) * Y(b)   (TRANSPOSE_X: X(a)^T * Y(b))
// X blocks: the logical slots of a level's operator XA (X_BSR; load_block) or plain records X[nx][9];
// Y plain [ny][9]; out plain or pair-SoA (a BsrDev's blk).
// (C4 level 0: 13 M products of A P in 445 us, 3 M of P^T A P in 170 us.  Measured in round 3 and dropped: a second copy of
// the level-0 blocks as 72-byte records for the gathers of A -- 434 us, and the linearisation pays 19 us for writing it:
// lanes of a target share their A blocks, the pair-SoA image is not what costs; records of P / A P padded to 80 bytes so
// that a record is five 16-byte-aligned loads instead of nine 8-byte ones -- A P 474 us, P^T A P 165 us: bytes, not load
// instructions, are what the gathers pay for.)
template <bool X_BSR, bool TRANSPOSE_X, bool OUT_BSR>
__global__ __launch_bounds__(kBlock) void k_block_products(ProdMap mp, BsrDev XA, const double* __restrict__ X,
                                                           const double* __restrict__ Y,
                                                           double* __restrict__ out, size_t nout, const int* __restrict__ mirror,
                                                           const int* __restrict__ rowof, int row0, int row1, int keep_key) {
  const int lane = threadIdx.x & 63;
  int gi, gend, gstride;
  group_walk(mp.ngrp, &gi, &gend, &gstride);
  int ngb = 0, nge = 0;   // the next group's bounds are requested while the current group is worked on
  if (gi < gend) {
    ngb = mp.grp[gi];
    nge = mp.grp[gi + 1];
  }
  for (; gi < gend; gi += gstride) {
    const int gb = ngb, ge = nge;
    if (gi + gstride < gend) {
      ngb = mp.grp[gi + gstride];
      nge = mp.grp[gi + gstride + 1];
    }
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int key = -1 - lane;
    for (int t = gb + lane; t < ge; t += 64) {
      key = mp.tgt[t];
      const size_t ia = (size_t)mp.a[t], ib = (size_t)mp.b[t];
      // multi-GPU, row-owner mode: only the products whose left operand lives in one of this rank's fine rows (rowof[a]).
      // keep_key (P^T A P, summed over ranks afterwards): a target without any owned product still stores the zero it
      // owes the all-reduce; otherwise (A P: a target's products all sit in one row) it is another rank's and is skipped
      if (row1 > 0) {
        const int i = rowof[ia];
        if (i < row0 || i >= row1) {
          if (!keep_key) key = -1 - lane;
          continue;
        }
      }
      // (the right operand first: its address is known from the index triple, while the left operand of A P hangs on a
      // chain of its own -- slot -> reference -> block; requested behind that chain it was one dependent round trip more)
      double x[9], y[9];
      load9(Y, ib, y);
      if (X_BSR) load_block(XA, ia, x);
      else load9(X, ia, x);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (TRANSPOSE_X) acc[3 * r + c] += x[r] * y[c] + x[3 + r] * y[3 + c] + x[6 + r] * y[6 + c];
          else acc[3 * r + c] += x[3 * r] * y[c] + x[3 * r + 1] * y[3 + c] + x[3 * r + 2] * y[6 + c];
        }
    }
    seg_scan<9>(key, acc, lane);
    const int kn = next_lane_key(key);
    if (key >= 0 && (lane == 63 || kn != key)) {
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        if (OUT_BSR) out[blk_at(c, (size_t)key, nout)] = acc[c];
        else out[9 * (size_t)key + c] = acc[c];
      }
      if (OUT_BSR && mirror) {   // symmetric target matrix: the block's transpose goes to the mirrored slot
        const int mk = mirror[key];
        if (mk >= 0) {
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) out[blk_at(3 * c + r, (size_t)mk, nout)] = acc[3 * r + c];
        }
      }
    }
  }
}

// rc[a] = sum over the entries e of column a of P_e^T r[row(e)]   (column-ordered copy of P)
// Multi-GPU: only the fine rows [row0, row1) contribute (row1 == 0: all) -- the partial coarse right-hand sides of
// the ranks are then summed by an all-reduce of 3 n_c doubles instead of all-reducing the fine residual.
// The first nb_main workgroups walk the wave groups; columns longer than kLongColumn entries are left to the workgroups
// behind them, ONE WORKGROUP each in the same launch (every thread a few entries, all of them requested at once; wave sums,
// then the wave totals added in a fixed order): a single wave walking such a column stride by stride is a chain of ~70
// dependent round trips (C5's last level: 4 350 entries per column, 109 us), a launch of its own costs 4 us per cycle.
// (block: this workgroup's index within the restriction's part of the launch)
__device__ __forceinline__ void restrict_groups(const PDev& P, const double* __restrict__ r, double* __restrict__ rc,
                                                const PcgScalars* S, int row0, int row1, int nb_main, int block) {
  const int lane = threadIdx.x & 63;
  const size_t np = (size_t)P.t_n;
  if (block >= nb_main) {   // a long column
    if (S && S->stop) return;
    __shared__ double sm[kWavesPerBlock][3];
    const int lc = block - nb_main;
    const int gb = P.t_long[2 * lc], ge = P.t_long[2 * lc + 1];
    double acc[3] = {0.0, 0.0, 0.0};
    for (int t = gb + (int)threadIdx.x; t < ge; t += kBlock) {
      const size_t i = (size_t)P.t_row[t];
      if (row1 > 0 && ((int)i < row0 || (int)i >= row1)) continue;
      const double r0 = r[3 * i], r1 = r[3 * i + 1], r2 = r[3 * i + 2];
      double b[9];
      load9_pairs(P.t_blk, P.t_blk8, (size_t)t, np, b, P.stream_nt != 0);
      acc[0] += b[0] * r0 + b[3] * r1 + b[6] * r2;
      acc[1] += b[1] * r0 + b[4] * r1 + b[7] * r2;
      acc[2] += b[2] * r0 + b[5] * r1 + b[8] * r2;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double v = wave_sum(acc[q]);
      if (lane == 0) sm[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
      double v = sm[0][threadIdx.x];
      for (int w = 1; w < kWavesPerBlock; ++w) v += sm[w][threadIdx.x];
      rc[3 * (size_t)P.t_col[gb] + threadIdx.x] = v;
    }
    return;
  }
  // (as in k_spmv: the first group's bounds are requested before the stop flag is waited for -- one dependent
  // round trip less in a launch that is a chain of four)
  int g, gend, gstride;
  group_walk_b(P.t_ngrp, nb_main, block, &g, &gend, &gstride);
  int gb0 = 0, ge0 = 0;
  if (g < gend) {
    gb0 = P.t_grp[g];
    ge0 = P.t_grp[g + 1];
  }
  if (S && S->stop) return;
  for (bool first = true; g < gend; g += gstride, first = false) {
    const int gb = first ? gb0 : P.t_grp[g], ge = first ? ge0 : P.t_grp[g + 1];
    if (P.t_nlong > 0 && ge - gb > kLongColumn) continue;   // a long column: the workgroups at the end of the grid
    double acc[3] = {0.0, 0.0, 0.0};
    int key = -1 - lane;
    for (int t = gb + lane; t < ge; t += 64) {
      key = P.t_col[t];
      const size_t i = (size_t)P.t_row[t];
      if (row1 > 0 && ((int)i < row0 || (int)i >= row1)) continue;
      const double r0 = r[3 * i], r1 = r[3 * i + 1], r2 = r[3 * i + 2];
      double b[9];
      load9_pairs(P.t_blk, P.t_blk8, (size_t)t, np, b, P.stream_nt != 0);
      acc[0] += b[0] * r0 + b[3] * r1 + b[6] * r2;
      acc[1] += b[1] * r0 + b[4] * r1 + b[7] * r2;
      acc[2] += b[2] * r0 + b[5] * r1 + b[8] * r2;
    }
    seg_scan<3>(key, acc, lane);
    const int kn = next_lane_key(key);
    if (key >= 0 && (lane == 63 || kn != key)) {
      rc[3 * (size_t)key] = acc[0];
      rc[3 * (size_t)key + 1] = acc[1];
      rc[3 * (size_t)key + 2] = acc[2];
    }
  }
}
__global__ __launch_bounds__(kBlock) void k_restrict_p(PDev P, const double* __restrict__ r, double* __restrict__ rc,
                                                       const PcgScalars* S, int row0, int row1, int nb_main) {
  restrict_groups(P, r, rc, S, row0, row1, nb_main, (int)blockIdx.x);
}
// The folded cycle's two level-0 launches that read the same right-hand side and do not depend on each other -- the
// Jacobi pass of the wave-group kernel (M2 r, workgroups [0, nb_spmv)) and the restriction with P~^T (the workgroups
// behind them) -- as ONE launch: on the graphs that fold level 0 a launch costs more than what it moves.
__global__ __launch_bounds__(kBlock, 4) void k_jacobi0_restrict(Sym0Dev A, Spmv0Args a, int nb_spmv, PDev P, double* __restrict__ rc,
                                                                 int nb_main) {
  if ((int)blockIdx.x < nb_spmv) spmv0_groups<S0_JACOBI>(A, a, nb_spmv);
  else restrict_groups(P, a.b, rc, a.S, 0, 0, nb_main, (int)blockIdx.x - nb_spmv);
}
// r_c = P^T r on the stream
void launch_restrict_p(hipStream_t s, const PDev& P, const double* r, double* rc, const PcgScalars* S, int row0, int row1) {
  const int nb = grid_for(P.t_ngrp, kWavesPerBlock);
  SGO_LAUNCH(k_restrict_p, dim3(nb + P.t_nlong), dim3(kBlock), 0, s, P, r, rc, S, row0, row1, nb);
}

// x_i += sum over the entries e of row i of P_e (c1 u1 + c2 u2)[col(e)]  (+ xadd_i)
__global__ __launch_bounds__(kBlock) void k_prolong_p(int n, PDev P, const double* __restrict__ u1, SpmvRatio r1,
                                                      const double* __restrict__ u2, SpmvRatio r2,
                                                      double* __restrict__ x, const PcgScalars* S,
                                                      const double* __restrict__ xadd, int row0, int row1) {
  int g, gend, gstride;
  group_walk(P.r_ngrp, &g, &gend, &gstride);
  int gb0 = 0, ge0 = 0;   // requested before the stop flag is waited for (see k_restrict_p)
  if (g < gend) {
    gb0 = P.r_grp[g];
    ge0 = P.r_grp[g + 1];
  }
  if (S && S->stop) return;
  double c1 = 1.0, c2 = 0.0;
  if (r1.num || u2) {   // (the plain V-cycle prolongates an unscaled correction: no partial sums to reduce)
    const double* const parts[4] = {r1.num ? r1.den : nullptr, r1.num, (u2 && r2.num) ? r2.den : nullptr,
                                    u2 ? r2.num : nullptr};
    const int cnt[4] = {r1.n_den, r1.n_num, r2.n_den, r2.n_num};
    double v[4];
    block_reduce_parts_n<4>(parts, cnt, v);
    if (r1.num) c1 = (v[0] > 0.0 && isfinite(v[0]) && isfinite(v[1])) ? v[1] / v[0] : 0.0;
    if (u2) c2 = (v[2] > 0.0 && isfinite(v[2]) && isfinite(v[3])) ? v[3] / v[2] : 0.0;
  }
  // one lane per entry (entries are sorted by fine row), wavefront segmented sum per row
  const int lane = threadIdx.x & 63;
  const size_t np = (size_t)P.r_n;
  for (bool first = true; g < gend; g += gstride, first = false) {
    const int gb = first ? gb0 : P.r_grp[g], ge = first ? ge0 : P.r_grp[g + 1];
    double acc[3] = {0.0, 0.0, 0.0};
    int key = -1 - lane;
    for (int e = gb + lane; e < ge; e += 64) {
      key = P.row[e];
      if (row1 > 0 && (key < row0 || key >= row1)) {   // multi-GPU: another rank's row
        key = -1 - lane;
        continue;
      }
      const size_t a = 3 * (size_t)P.col[e];
      double w0 = c1 * u1[a], w1 = c1 * u1[a + 1], w2 = c1 * u1[a + 2];
      if (u2) {
        w0 += c2 * u2[a]; w1 += c2 * u2[a + 1]; w2 += c2 * u2[a + 2];
      }
      double b[9];
      load9_pairs(P.r_blk, P.r_blk8, (size_t)e, np, b, P.stream_nt != 0);
      acc[0] += b[0] * w0 + b[1] * w1 + b[2] * w2;
      acc[1] += b[3] * w0 + b[4] * w1 + b[5] * w2;
      acc[2] += b[6] * w0 + b[7] * w1 + b[8] * w2;
    }
    seg_scan<3>(key, acc, lane);
    const int kn = next_lane_key(key);
    if (key >= 0 && (lane == 63 || kn != key)) {
      const size_t o = 3 * (size_t)key;
      if (xadd) {
        acc[0] += xadd[o]; acc[1] += xadd[o + 1]; acc[2] += xadd[o + 2];
      }
      x[o] += acc[0]; x[o + 1] += acc[1]; x[o + 2] += acc[2];
    }
  }
}

__device__ __forceinline__ int find_sorted(const int* __restrict__ v, int lo, int hi, int key) {   // position of key in v[lo, hi) or -1
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int x = v[mid];
    if (x == key) return mid;
    if (x < key) lo = mid + 1;
    else hi = mid;
  }
  return -1;
}

// ------------------------------------------------------------------ folded V-cycle
// One damped block-Jacobi sweep before and after the coarse correction, S = I - w D^-1 A, first sweep from zero:
//   x1 = w D^-1 r;  r1 = r - A x1 = S^T r;  e = P B_c P^T r1;  z = S (x1 + e) + w D^-1 r
//      = [x1 + w D^-1 r1]  +  (S P) B_c (S P)^T r
// i.e. the multiplicative cycle IS an additive one with the transfer operator P~ = S P = P - w D^-1 (A P) -- whose
// blocks follow from A P, which the Galerkin product makes anyway -- and the two-sweep term M2 r = x1 + w D^-1 (r - A x1),
// which does not depend on the coarse correction.  A level then costs two launches on the critical path of the cycle
// (restriction with P~^T, prolongation with P~) instead of four (sweep + residual, restriction, prolongation, sweep):
// on level 0 (small graphs only, see amg_create) the M2 term is the level-0 Jacobi pass -- one pass instead of residual pass +
// sweep --; on the coarser levels it is folded into the prolongation launch (k_up_fold: the row's slots of A and its entries of P~ in one list).
// Same preconditioner in exact arithmetic (B_c: the coarser levels' cycle, recursively the same), symmetric as before
// (restriction and prolongation read the same rounded fp32 values of P~).
struct FoldDev {   // pattern of P~ (= that of A P); arrays addressed by GLOBAL A P entry numbers f (row-owner mode: shifted)
  int f_lo = 0, f_hi = 0;      // entries held (this rank's rows' in row-owner mode)
  const int* row = nullptr;    // fine row of entry f
  const int* col = nullptr;    // coarse column of entry f
  int* ap2p = nullptr;         // entry of P at the same (row, column), -1: none
  int* st_pos = nullptr;       // position of f in column order, 0-based within the entries held
};
struct UpDev {     // levels >= 1: per row its slots of A, then its entries of P~, as ONE row-major list
  int n = 0, ngrp = 0;
  const int* row = nullptr;
  const int* idx = nullptr;    // slot k >= 0, or ~f for entry f of P~
  const int* col = nullptr;    // column of the slot / coarse column of the entry
  const int* grp = nullptr;    // wave groups aligned to rows
};

__global__ __launch_bounds__(kBlock) void k_fold_match(FoldDev F, const int* __restrict__ p_rowptr, const int* __restrict__ p_col,
                                                       unsigned long long* __restrict__ keys) {
  for (int f = F.f_lo + blockIdx.x * kBlock + threadIdx.x; f < F.f_hi; f += gridDim.x * kBlock) {
    const int i = F.row[f], c = F.col[f];
    F.ap2p[f] = find_sorted(p_col, p_rowptr[i], p_rowptr[i + 1], c);
    keys[f - F.f_lo] = ((unsigned long long)(unsigned)c << 32) | (unsigned)(f - F.f_lo);   // (column, row-major rank): sorted = column order
  }
}
__global__ __launch_bounds__(kBlock) void k_fold_unpack(FoldDev F, const unsigned long long* __restrict__ sorted, int* __restrict__ t_row,
                                                        int* __restrict__ t_col) {
  const int nf = F.f_hi - F.f_lo;
  for (int t = blockIdx.x * kBlock + threadIdx.x; t < nf; t += gridDim.x * kBlock) {
    const unsigned long long k = sorted[t];
    const int f = F.f_lo + (int)(unsigned)(k & 0xffffffffull);
    F.st_pos[f] = t;
    t_col[t] = (int)(k >> 32);
    t_row[t] = F.row[f];
  }
}
// ptr[c] = first position of column c in the sorted keys (c = 0 .. nc; ptr[nc] = nf)
__global__ __launch_bounds__(kBlock) void k_fold_colptr(const unsigned long long* __restrict__ sorted, int nf, int nc, int* __restrict__ ptr) {
  for (int c = blockIdx.x * kBlock + threadIdx.x; c <= nc; c += gridDim.x * kBlock) {
    const unsigned long long key = (unsigned long long)(unsigned)c << 32;
    int lo = 0, hi = nf;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sorted[mid] < key) lo = mid + 1;
      else hi = mid;
    }
    ptr[c] = lo;
  }
}
// the wave groups longer than kLongColumn (single long columns): their ranges, in any order (one workgroup each later on)
__global__ __launch_bounds__(kBlock) void k_fold_long(const int* __restrict__ grp, int ngrp, int* __restrict__ count, int* __restrict__ ranges,
                                                      int cap) {
  for (int g = blockIdx.x * kBlock + threadIdx.x; g < ngrp; g += gridDim.x * kBlock) {
    const int b = grp[g], e = grp[g + 1];
    if (e - b > kLongColumn) {
      const int q = atomicAdd(count, 1);
      if (q < cap) {
        ranges[2 * q] = b;
        ranges[2 * q + 1] = e;
      }
    }
  }
}

// P~_f = P_(row, col) - w D_row^-1 (A P)_f : the row-ordered and the column-ordered fp32 copies the folded cycle streams
__global__ __launch_bounds__(kBlock) void k_ptilde_values(FoldDev F, const double* __restrict__ apblk, const double* __restrict__ pblk,
                                                          const double* __restrict__ dinv, double omega, PDev PS) {
  for (int f = F.f_lo + blockIdx.x * kBlock + threadIdx.x; f < F.f_hi; f += gridDim.x * kBlock) {
    const size_t i = (size_t)F.row[f];
    const int e = F.ap2p[f];
    const size_t tp = (size_t)F.st_pos[f];
    const double* di = dinv + 6 * i;
    double a[9], o[9];
    load9(apblk, (size_t)f, a);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      o[c] = -omega * (di[0] * a[c] + di[1] * a[3 + c] + di[2] * a[6 + c]);
      o[3 + c] = -omega * (di[1] * a[c] + di[3] * a[3 + c] + di[4] * a[6 + c]);
      o[6 + c] = -omega * (di[2] * a[c] + di[4] * a[3 + c] + di[5] * a[6 + c]);
    }
    if (e >= 0) {
#pragma unroll
      for (int c = 0; c < 9; ++c) o[c] += pblk[9 * (size_t)e + c];
    }
    const sgo_f4 q0 = {(float)o[0], (float)o[1], (float)o[2], (float)o[3]}, q1 = {(float)o[4], (float)o[5], (float)o[6], (float)o[7]};
    sgo_f4* rq = reinterpret_cast<sgo_f4*>(PS.r_blk);
    sgo_f4* tq = reinterpret_cast<sgo_f4*>(PS.t_blk);
    rq[f] = q0;
    rq[(size_t)PS.r_n + f] = q1;
    tq[tp] = q0;
    tq[(size_t)PS.t_n + tp] = q1;
    PS.r_blk8[f] = (float)o[8];
    PS.t_blk8[tp] = (float)o[8];
  }
}

// Level 0: z_i = y_i + sum over the entries f of row i of P~_f (c1 u1 + c2 u2)[col(f)]; optional partials of dotA . z, dotA2 . z
// (P: the view of P~ -- row / col / r_grp / r_blk of the folded operator; single GPU: multi-GPU runs keep level 0 unfolded).
// Workgroups of 1024 threads: the consumer of the dot products re-reduces one partial sum per WORKGROUP in every one of its
// own workgroups, so there should be a few hundred of them.
constexpr int kFoldThreads = 1024;
__global__ __launch_bounds__(kFoldThreads) void k_prolong_fold(PDev P, const double* __restrict__ u1, SpmvRatio r1,
                                                               const double* __restrict__ u2, SpmvRatio r2, const double* __restrict__ y,
                                                               double* __restrict__ out, const PcgScalars* S, const double* __restrict__ dotA,
                                                               const double* __restrict__ dotA2, double* __restrict__ partials) {
  constexpr int NW = kFoldThreads / 64;
  __shared__ double sm[4][NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD-aware walk (group_walk's rule) for NW waves per workgroup
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int glo = (int)(((long long)P.r_ngrp * xcd) >> 3), gend = (int)(((long long)P.r_ngrp * (xcd + 1)) >> 3);
  const int gstride = per_xcd * NW;
  int g = glo + slot * NW + wave;
  int gb0 = 0, ge0 = 0;   // requested before the stop flag is waited for (see k_restrict_p)
  if (g < gend) {
    gb0 = P.r_grp[g];
    ge0 = P.r_grp[g + 1];
  }
  if (S && S->stop) return;
  double c1 = 1.0, c2 = 0.0;
  if (r1.num || u2) {   // ratios of partial sums (K-cycle below): every workgroup reduces them in the same fixed order
    const double* const parts[4] = {r1.num ? r1.den : nullptr, r1.num, (u2 && r2.num) ? r2.den : nullptr, u2 ? r2.num : nullptr};
    const int cnt[4] = {r1.n_den, r1.n_num, r2.n_den, r2.n_num};
    double v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double t = 0.0;
      if (parts[c])
        for (int i = threadIdx.x; i < cnt[c]; i += kFoldThreads) t += parts[c][i];
      t = wave_sum(t);
      if (lane == 0) sm[c][wave] = t;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double t = sm[c][0];
#pragma unroll 1
      for (int k = 1; k < NW; ++k) t += sm[c][k];   // (not unrolled: 64 LDS reads in flight would set the kernel's register count)
      v[c] = t;
    }
    __syncthreads();
    if (r1.num) c1 = (v[0] > 0.0 && isfinite(v[0]) && isfinite(v[1])) ? v[1] / v[0] : 0.0;
    if (u2) c2 = (v[2] > 0.0 && isfinite(v[2]) && isfinite(v[3])) ? v[3] / v[2] : 0.0;
  }
  const size_t np = (size_t)P.r_n;
  double dotacc[2] = {0.0, 0.0};
  for (bool first = true; g < gend; g += gstride, first = false) {
    const int gb = first ? gb0 : P.r_grp[g], ge = first ? ge0 : P.r_grp[g + 1];
    double acc[3] = {0.0, 0.0, 0.0};
    int key = -1 - lane;
    for (int e = gb + lane; e < ge; e += 64) {
      key = P.row[e];
      const size_t a = 3 * (size_t)P.col[e];
      double w0 = c1 * u1[a], w1 = c1 * u1[a + 1], w2 = c1 * u1[a + 2];
      if (u2) {
        w0 += c2 * u2[a]; w1 += c2 * u2[a + 1]; w2 += c2 * u2[a + 2];
      }
      double b[9];
      load9_pairs(P.r_blk, P.r_blk8, (size_t)e, np, b, P.stream_nt != 0);
      acc[0] += b[0] * w0 + b[1] * w1 + b[2] * w2;
      acc[1] += b[3] * w0 + b[4] * w1 + b[5] * w2;
      acc[2] += b[6] * w0 + b[7] * w1 + b[8] * w2;
    }
    // the row's own term is requested before the scan (one dependent round trip less)
    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
    if (key >= 0) {
      const size_t o = 3 * (size_t)key;
      y0 = y[o]; y1 = y[o + 1]; y2 = y[o + 2];
    }
    seg_scan<3>(key, acc, lane);
    const int kn = next_lane_key(key);
    if (key >= 0 && (lane == 63 || kn != key)) {
      const size_t o = 3 * (size_t)key;
      const double o0 = y0 + acc[0], o1 = y1 + acc[1], o2 = y2 + acc[2];
      out[o] = o0; out[o + 1] = o1; out[o + 2] = o2;
      if (dotA) dotacc[0] += dotA[o] * o0 + dotA[o + 1] * o1 + dotA[o + 2] * o2;
      if (dotA2) dotacc[1] += dotA2[o] * o0 + dotA2[o + 1] * o1 + dotA2[o + 2] * o2;
    }
  }
  if (partials) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double t = wave_sum(dotacc[i]);
      if (lane == 0) sm[i][wave] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        double t = sm[i][0];
#pragma unroll
        for (int k = 1; k < NW; ++k) t += sm[i][k];
        partials[(size_t)i * kMaxPartials + blockIdx.x] = t;
      }
    }
  }
}

// Levels >= 1: out_i = M2 b |_i + sum_f P~_f (c1 u1 + c2 u2)[col(f)],  M2 b = x1 + w D^-1 (b - A x1), x1 = w D^-1 b
// (the operand of a slot (i, j) is made on the fly from b_j and D_j^-1: nothing but b and the coarse solution is read)
__global__ __launch_bounds__(kBlock) void k_up_fold(BsrDev A, UpDev U, PDev PS, const double* __restrict__ b, double omega,
                                                    const double* __restrict__ u1, SpmvRatio r1, const double* __restrict__ u2,
                                                    SpmvRatio r2, double* __restrict__ out, const PcgScalars* S,
                                                    const double* __restrict__ xadd) {
  int g, gend, gstride;
  group_walk(U.ngrp, &g, &gend, &gstride);
  int gb0 = 0, ge0 = 0;
  if (g < gend) {
    gb0 = U.grp[g];
    ge0 = U.grp[g + 1];
  }
  if (S && S->stop) return;
  double c1 = 1.0, c2 = 0.0;
  if (r1.num || u2) {
    const double* const parts[4] = {r1.num ? r1.den : nullptr, r1.num, (u2 && r2.num) ? r2.den : nullptr,
                                    u2 ? r2.num : nullptr};
    const int cnt[4] = {r1.n_den, r1.n_num, r2.n_den, r2.n_num};
    double v[4];
    block_reduce_parts_n<4>(parts, cnt, v);
    if (r1.num) c1 = (v[0] > 0.0 && isfinite(v[0]) && isfinite(v[1])) ? v[1] / v[0] : 0.0;
    if (u2) c2 = (v[2] > 0.0 && isfinite(v[2]) && isfinite(v[3])) ? v[3] / v[2] : 0.0;
  }
  const int lane = threadIdx.x & 63;
  const size_t ns = (size_t)A.nslot, np = (size_t)PS.r_n;
  for (bool first = true; g < gend; g += gstride, first = false) {
    const int gb = first ? gb0 : U.grp[g], ge = first ? ge0 : U.grp[g + 1];
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // [0..2] A x1, [3..5] P~ e
    int key = -1 - lane;
    for (int t = gb + lane; t < ge; t += 64) {
      key = U.row[t];
      const int idx = U.idx[t];
      const size_t j = (size_t)U.col[t];
      if (idx >= 0) {
        const size_t k = (size_t)idx;
        const double* dj = A.dinv + 6 * j;
        const double b0 = b[3 * j], b1 = b[3 * j + 1], b2 = b[3 * j + 2];
        const double x0 = omega * (dj[0] * b0 + dj[1] * b1 + dj[2] * b2);
        const double x1 = omega * (dj[1] * b0 + dj[3] * b1 + dj[4] * b2);
        const double x2 = omega * (dj[2] * b0 + dj[4] * b1 + dj[5] * b2);
        const double2* __restrict__ bp = reinterpret_cast<const double2*>(A.blk);
        const double2 p0 = bp[k], p1 = bp[ns + k], p2 = bp[2 * ns + k], p3 = bp[3 * ns + k];
        const double b8 = A.blk[8 * ns + k];
        acc[0] += p0.x * x0 + p0.y * x1 + p1.x * x2;
        acc[1] += p1.y * x0 + p2.x * x1 + p2.y * x2;
        acc[2] += p3.x * x0 + p3.y * x1 + b8 * x2;
      } else {
        const size_t f = (size_t)(~idx);
        double w0 = c1 * u1[3 * j], w1 = c1 * u1[3 * j + 1], w2 = c1 * u1[3 * j + 2];
        if (u2) {
          w0 += c2 * u2[3 * j]; w1 += c2 * u2[3 * j + 1]; w2 += c2 * u2[3 * j + 2];
        }
        double q[9];
        load9_pairs(PS.r_blk, PS.r_blk8, f, np, q, false);
        acc[3] += q[0] * w0 + q[1] * w1 + q[2] * w2;
        acc[4] += q[3] * w0 + q[4] * w1 + q[5] * w2;
        acc[5] += q[6] * w0 + q[7] * w1 + q[8] * w2;
      }
    }
    // the row's own right-hand side and block-diagonal inverse, requested before the scan
    double bi0 = 0.0, bi1 = 0.0, bi2 = 0.0, d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0, d4 = 0.0, d5 = 0.0;
    if (key >= 0) {
      const size_t i = (size_t)key;
      bi0 = b[3 * i]; bi1 = b[3 * i + 1]; bi2 = b[3 * i + 2];
      const double* di = A.dinv + 6 * i;
      d0 = di[0]; d1 = di[1]; d2 = di[2]; d3 = di[3]; d4 = di[4]; d5 = di[5];
    }
    seg_scan<6>(key, acc, lane);
    const int kn = next_lane_key(key);
    if (key >= 0 && (lane == 63 || kn != key)) {
      const size_t o = 3 * (size_t)key;
      // x1 + w D^-1 (b - A x1) = w D^-1 (2 b - A x1) evaluated as the cycle does: x1 first, then the residual's sweep
      const double x0 = omega * (d0 * bi0 + d1 * bi1 + d2 * bi2), x1 = omega * (d1 * bi0 + d3 * bi1 + d4 * bi2),
                   x2 = omega * (d2 * bi0 + d4 * bi1 + d5 * bi2);
      const double r0 = bi0 - acc[0], rr1 = bi1 - acc[1], rr2 = bi2 - acc[2];
      double o0 = x0 + omega * (d0 * r0 + d1 * rr1 + d2 * rr2) + acc[3];
      double o1 = x1 + omega * (d1 * r0 + d3 * rr1 + d4 * rr2) + acc[4];
      double o2 = x2 + omega * (d2 * r0 + d4 * rr1 + d5 * rr2) + acc[5];
      if (xadd) {   // (an outer sweep's iterate the folded cycle corrects)
        o0 += xadd[o]; o1 += xadd[o + 1]; o2 += xadd[o + 2];
      }
      out[o] = o0; out[o + 1] = o1; out[o + 2] = o2;
    }
  }
}

// dinv of a coarse level from its diagonal slots (first slot of each row)
__global__ __launch_bounds__(kBlock) void k_level_dinv(BsrDev A) {
  const size_t ns = (size_t)A.nslot;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < A.n; i += gridDim.x * kBlock) {
    const int k0 = A.rowptr[i];
    // symmetrise (the Galerkin sum is symmetric up to rounding)
    const double d00 = A.blk[blk_at(0, k0, ns)], d01 = 0.5 * (A.blk[blk_at(1, k0, ns)] + A.blk[blk_at(3, k0, ns)]);
    const double d02 = 0.5 * (A.blk[blk_at(2, k0, ns)] + A.blk[blk_at(6, k0, ns)]), d11 = A.blk[blk_at(4, k0, ns)];
    const double d12 = 0.5 * (A.blk[blk_at(5, k0, ns)] + A.blk[blk_at(7, k0, ns)]), d22 = A.blk[blk_at(8, k0, ns)];
    const double c00 = d11 * d22 - d12 * d12, c01 = d02 * d12 - d01 * d22, c02 = d01 * d12 - d02 * d11;
    const double c11 = d00 * d22 - d02 * d02, c12 = d01 * d02 - d00 * d12, c22 = d00 * d11 - d01 * d01;
    const double det = d00 * c00 + d01 * c01 + d02 * c02;
    const double id = (det != 0.0 && isfinite(det)) ? 1.0 / det : 0.0;
    double* di = A.dinv + 6 * (size_t)i;
    di[0] = c00 * id; di[1] = c01 * id; di[2] = c02 * id; di[3] = c11 * id; di[4] = c12 * id; di[5] = c22 * id;
  }
}

// rc[a] = sum_{i in a} T_i^T r_i : one lane per member (members sorted by aggregate), wavefront
// segmented scan per aggregate -- one dependent load chain instead of a serial member loop.
__global__ __launch_bounds__(kBlock) void k_restrict(int ngrp, const int* __restrict__ grp, const int* __restrict__ mem,
                                                     const int* __restrict__ agg, const double* __restrict__ d,
                                                     const double* __restrict__ r, double* __restrict__ rc,
                                                     const PcgScalars* S, int row0, int row1) {
  const int lane = threadIdx.x & 63;
  int g, gend, gstride;
  group_walk(ngrp, &g, &gend, &gstride);
  int gb0 = 0, ge0 = 0;   // requested before the stop flag is waited for (see k_restrict_p)
  if (g < gend) {
    gb0 = grp[g];
    ge0 = grp[g + 1];
  }
  if (S && S->stop) return;
  for (bool first = true; g < gend; g += gstride, first = false) {
    const int gb = first ? gb0 : grp[g], ge = first ? ge0 : grp[g + 1];
    double acc[3] = {0.0, 0.0, 0.0};
    int key = -1 - lane;
    for (int t = gb + lane; t < ge; t += 64) {
      const int i = mem[t];
      key = agg[i];
      if (row1 > 0 && (i < row0 || i >= row1)) continue;
      const double r0 = r[3 * (size_t)i], r1 = r[3 * (size_t)i + 1], r2 = r[3 * (size_t)i + 2];
      acc[0] += r0;
      acc[1] += r1;
      acc[2] += -d[2 * (size_t)i + 1] * r0 + d[2 * (size_t)i] * r1 + r2;
    }
    seg_scan<3>(key, acc, lane);
    const int kn = next_lane_key(key);
    if (key >= 0 && (lane == 63 || kn != key)) {
      rc[3 * (size_t)key] = acc[0];
      rc[3 * (size_t)key + 1] = acc[1];
      rc[3 * (size_t)key + 2] = acc[2];
    }
  }
}

// x_i += T_i (c1 u1 + c2 u2)[agg(i)]   (u2 may be null; scalars as in k_spmv's fused modes)
__global__ __launch_bounds__(kBlock) void k_prolong_add(int n, const int* __restrict__ agg, const double* __restrict__ d,
                                                        const double* __restrict__ u1, SpmvRatio r1,
                                                        const double* __restrict__ u2, SpmvRatio r2,
                                                        double* __restrict__ x, const PcgScalars* S,
                                                        const double* __restrict__ xadd, int row0, int row1) {
  if (S && S->stop) return;
  double c1 = 1.0, c2 = 0.0;
  {
    const double* const parts[4] = {r1.num ? r1.den : nullptr, r1.num, (u2 && r2.num) ? r2.den : nullptr,
                                    u2 ? r2.num : nullptr};
    const int cnt[4] = {r1.n_den, r1.n_num, r2.n_den, r2.n_num};
    double v[4];
    block_reduce_parts_n<4>(parts, cnt, v);
    if (r1.num) c1 = (v[0] > 0.0 && isfinite(v[0]) && isfinite(v[1])) ? v[1] / v[0] : 0.0;
    if (u2) c2 = (v[2] > 0.0 && isfinite(v[2]) && isfinite(v[3])) ? v[3] / v[2] : 0.0;
  }
  const int ilo = row1 > 0 ? row0 : 0, ihi = row1 > 0 ? row1 : n;
  for (int i = ilo + blockIdx.x * kBlock + threadIdx.x; i < ihi; i += gridDim.x * kBlock) {
    const size_t a = 3 * (size_t)agg[i], o = 3 * (size_t)i;
    double w0 = c1 * u1[a], w1 = c1 * u1[a + 1], w = c1 * u1[a + 2];
    if (u2) {
      w0 += c2 * u2[a]; w1 += c2 * u2[a + 1]; w += c2 * u2[a + 2];
    }
    if (xadd) {
      w0 += xadd[o]; w1 += xadd[o + 1];
      x[o + 2] += xadd[o + 2];
    }
    x[o] += w0 - d[2 * (size_t)i + 1] * w;
    x[o + 1] += w1 + d[2 * (size_t)i] * w;
    x[o + 2] += w;
  }
}

// Row-owner mode: the prolongated coarse correction on a LIST of rows -- the copies of the neighbours' boundary rows this rank
// keeps -- so that the post-smoothing sweep finds x1 + P e there without an exchange: the coarse solution is replicated, the
// boundary rows' entries of P travel once per Gauss-Newton iteration (the gathered records P.blk), and x1 = omega Dinv r on these
// rows is kept current by the repeated recurrences (k_update_xr_rows).  The records are rounded to the fp32 values the owner's
// k_prolong_p streams, so only the order of a row's few additions differs from the owner's result.  P.np == 0: tentative
// transfer (aggregate + lever arm, as k_prolong_add).
__global__ __launch_bounds__(kBlock) void k_prolong_rows(int nrows, const int* __restrict__ rows, PDev P, const int* __restrict__ agg,
                                                         const double* __restrict__ d, const double* __restrict__ u1, SpmvRatio r1,
                                                         const double* __restrict__ u2, SpmvRatio r2, double* __restrict__ x,
                                                         const PcgScalars* S) {
  if (S && S->stop) return;
  double c1 = 1.0, c2 = 0.0;
  if (r1.num || u2) {
    const double* const parts[4] = {r1.num ? r1.den : nullptr, r1.num, (u2 && r2.num) ? r2.den : nullptr, u2 ? r2.num : nullptr};
    const int cnt[4] = {r1.n_den, r1.n_num, r2.n_den, r2.n_num};
    double v[4];
    block_reduce_parts_n<4>(parts, cnt, v);
    if (r1.num) c1 = (v[0] > 0.0 && isfinite(v[0]) && isfinite(v[1])) ? v[1] / v[0] : 0.0;
    if (u2) c2 = (v[2] > 0.0 && isfinite(v[2]) && isfinite(v[3])) ? v[3] / v[2] : 0.0;
  }
  for (int t = blockIdx.x * kBlock + threadIdx.x; t < nrows; t += gridDim.x * kBlock) {
    const size_t i = (size_t)rows[t], o = 3 * i;
    if (P.np > 0) {
      double acc[3] = {0.0, 0.0, 0.0};
      for (int e = P.rowptr[i]; e < P.rowptr[i + 1]; ++e) {
        const size_t a = 3 * (size_t)P.col[e];
        double w0 = c1 * u1[a], w1 = c1 * u1[a + 1], w2 = c1 * u1[a + 2];
        if (u2) {
          w0 += c2 * u2[a]; w1 += c2 * u2[a + 1]; w2 += c2 * u2[a + 2];
        }
        const double* bl = P.blk + 9 * (size_t)e;
        double b[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) b[q] = (double)(float)bl[q];
        acc[0] += b[0] * w0 + b[1] * w1 + b[2] * w2;
        acc[1] += b[3] * w0 + b[4] * w1 + b[5] * w2;
        acc[2] += b[6] * w0 + b[7] * w1 + b[8] * w2;
      }
      x[o] += acc[0]; x[o + 1] += acc[1]; x[o + 2] += acc[2];
    } else {
      const size_t a = 3 * (size_t)agg[i];
      double w0 = c1 * u1[a], w1 = c1 * u1[a + 1], w = c1 * u1[a + 2];
      if (u2) {
        w0 += c2 * u2[a]; w1 += c2 * u2[a + 1]; w += c2 * u2[a + 2];
      }
      x[o] += w0 - d[2 * i + 1] * w;
      x[o + 1] += w1 + d[2 * i] * w;
      x[o + 2] += w;
    }
  }
}

// Coarsest level: explicit dense inverse, recomputed every GN iteration by a blocked in-place
// Gauss-Jordan (SPD, no pivoting) spread over many workgroups: for pivot block K (32 x 32)
//   P = inv(A_KK);  A_Kj <- P A_Kj (j != K);  A_ij <- A_ij - A_iK A_Kj (i, j != K);
//   A_iK <- -A_iK P (i != K);  A_KK <- P.
// The matrix is stored row-major with leading dimension Np = N rounded up to 32 and an identity
// on the padding, so every tile is full.  One launch per pivot block (k_gj_step).
// (64 x 64 blocks with 1024 threads -- half the sequential block steps -- were measured in round 3: much slower, C2 1.71 ->
// 2.77 ms and C4 4.84 -> 5.86 ms per GN iteration: the 64 elimination steps of a pivot block with sixteen waves at every
// barrier cost more than the launches they save.)
constexpr int kGjB = 32;

// The coarse levels' slots are unique per (row, col) (the host lists them so): one thread per slot stores its
// 3x3 block, no accumulation.  M was zeroed by the caller.
__global__ __launch_bounds__(kBlock) void k_dense_fill_unique(BsrDev A, int Np, double* __restrict__ M) {
  const int N = 3 * A.n;
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < A.nslot; k += gridDim.x * kBlock) {
    const int r = A.row[k], c = A.col[k];
    double b[9];
    load_block(A, (size_t)k, b);
#pragma unroll
    for (int e = 0; e < 9; ++e) M[(size_t)(3 * r + e / 3) * Np + 3 * c + e % 3] = b[e];
  }
  for (int i = N + blockIdx.x * kBlock + threadIdx.x; i < Np; i += gridDim.x * kBlock) M[(size_t)i * Np + i] = 1.0;
}
__global__ __launch_bounds__(kBlock) void k_dense_fill(BsrDev A, int Np, double* __restrict__ M) {
  // one thread per block row, slots added in order: level 0 can hold several slots for the same
  // (row, col) -- duplicate edges, and the zero blocks of fixed-column slots that alias the
  // diagonal -- so the fill must accumulate, and a row-exclusive sequential sum keeps it
  // deterministic.  M was zeroed by the caller.
  const int N = 3 * A.n;
  for (int r = blockIdx.x * kBlock + threadIdx.x; r < A.n; r += gridDim.x * kBlock) {
    for (int k = A.rowptr[r]; k < A.rowptr[r + 1]; ++k) {
      const int c = A.col[k];
      double b[9];
      load_block(A, (size_t)k, b);
#pragma unroll
      for (int e = 0; e < 9; ++e) M[(size_t)(3 * r + e / 3) * Np + 3 * c + e % 3] += b[e];
    }
  }
  for (int i = N + blockIdx.x * kBlock + threadIdx.x; i < Np; i += gridDim.x * kBlock) M[(size_t)i * Np + i] = 1.0;
}

// 1 / x by the hardware reciprocal and two Newton steps (the IEEE division is a chain of ~12 dependent instructions on
// the critical path of every elimination step of the pivot-block inverse)
__device__ __forceinline__ double gj_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// P = inv(A_KK) by scalar Gauss-Jordan (one workgroup, one thread per matrix element: the element
// lives in a register, only the pivot row / column travel through LDS; 2 barriers per step)
__global__ __launch_bounds__(kGjB * kGjB) void k_gj_pivot(double* __restrict__ M, int Np, int kb, double* __restrict__ P,
                                                          int* __restrict__ fail) {
  __shared__ double a2[2][kGjB][kGjB + 1];   // ping-pong: one barrier per elimination step
  const int i = threadIdx.x / kGjB, j = threadIdx.x % kGjB, base = kb * kGjB;
  double v = M[(size_t)(base + i) * Np + base + j];
  a2[0][i][j] = v;
  __syncthreads();
  // TWO pivots per elimination step (the 2 x 2 pivot block of an SPD matrix is inverted in closed form): 16 barrier-separated
  // steps instead of 32 -- the steps, not their arithmetic, are what the inverse costs
  for (int k = 0; k < kGjB; k += 2) {
    double (*a)[kGjB + 1] = a2[(k >> 1) & 1];
    const double pa = a[k][k], pb = a[k][k + 1], pd = a[k + 1][k + 1];
    const double det = pa * pd - pb * pb;
    if (threadIdx.x == 0 && (!(pa > 0.0) || !(det > 0.0) || !isfinite(det))) *fail = 1;
    const double id = (det != 0.0) ? gj_rcp(det) : 0.0;
    const double p00 = pd * id, p01 = -pb * id, p11 = pa * id;
    const double a0j = a[k][j], a1j = a[k + 1][j], ai0 = a[i][k], ai1 = a[i][k + 1];
    const double t0 = ai0 * p00 + ai1 * p01, t1 = ai0 * p01 + ai1 * p11;
    if (i == k) v = (j == k) ? p00 : ((j == k + 1) ? p01 : p00 * a0j + p01 * a1j);
    else if (i == k + 1) v = (j == k) ? p01 : ((j == k + 1) ? p11 : p01 * a0j + p11 * a1j);
    else if (j == k) v = -t0;
    else if (j == k + 1) v = -t1;
    else v = v - (t0 * a0j + t1 * a1j);
    a2[((k >> 1) + 1) & 1][i][j] = v;
    __syncthreads();
  }
  P[threadIdx.x] = v;
}

// Z = X * Y for 32x32 fp64 tiles held in LDS, on the matrix cores: wave w of the workgroup's four makes the 16x16 quadrant
// (w >> 1, w & 1) in eight v_mfma_f64_16x16x4_f64 steps -- two LDS reads per lane and step, where the scalar loop (one row
// x four columns per thread) read five per k: the tile products of a block step were LDS-bandwidth-bound, ~5 us of its
// 17 with two or three workgroups per CU.  Operand lane maps: A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col =
// lane & 15]; result register q of a lane is row (lane >> 4) + 4 q, column lane & 15.
typedef double sgo_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void tile_mm(const double (*X)[kGjB + 1], const double (*Y)[kGjB + 1], double (*Z)[kGjB + 1]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i0 = 16 * (w >> 1), j0 = 16 * (w & 1), lr = lane & 15, lk = lane >> 4;
  sgo_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < kGjB / 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[i0 + lr][4 * s + lk], Y[4 * s + lk][j0 + lr], acc, 0, 0, 0);
#pragma unroll
  for (int q = 0; q < 4; ++q) Z[i0 + lk + 4 * q][j0 + lr] = acc[q];
}

// One Gauss-Jordan block step K = kb is ONE launch over all 32x32 tiles, reading the matrix of the previous
// step (Min) and writing the next one (Mout) -- ping-pong, so that no tile is read after another workgroup
// has overwritten it and no panel kernel has to run first:
//   (K, K)            <- P                       (P = inv(A_KK), made by the previous launch)
//   (K, j), j != K    <- P A_Kj
//   (i, K), i != K    <- -A_iK P
//   (i, j) elsewhere  <- A_ij - A_iK (P A_Kj)    (the row-panel product recomputed per tile: two tile
//                                                 products instead of one buy half the launches)
// and the workgroup that finishes the next diagonal block (K+1, K+1) inverts it in LDS straight away and
// leaves it in Pout.  The 32 elimination steps of that inverse are the critical path of the whole inversion
// (one sequential 32x32 inverse per block step), so they ping-pong between two LDS tiles: step k reads tile
// k & 1 and writes the other one, ONE barrier per step.
__global__ __launch_bounds__(kBlock) void k_gj_step(const double* __restrict__ Min, double* __restrict__ Mout, int Np, int kb,
                                                    const double* __restrict__ Pin, double* __restrict__ Pout,
                                                    int* __restrict__ fail) {
  __shared__ double X[kGjB][kGjB + 1], Y[kGjB][kGjB + 1], Z[kGjB][kGjB + 1];
  const int t = threadIdx.x, K0 = kb * kGjB;
  const int bj = blockIdx.x, bi = blockIdx.y;
  const int r = t >> 3, c0 = (t & 7) * 4;
  double v[4];
  double* dst = Mout + (size_t)(bi * kGjB + r) * Np + bj * kGjB + c0;
  if (bi == kb && bj == kb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = Pin[r * kGjB + c0 + q];
    return;
  }
  // Everything this tile reads from global memory is requested up front -- the operands of the first product, the
  // column-panel tile of the second and the tile itself -- so that a block step is ONE memory round trip deep instead
  // of three (the step's launch is a dependent chain on the critical path of the whole inversion).
  const bool general = bi != kb && bj != kb;
  double x1[4], y1[4], x2[4] = {0, 0, 0, 0}, sv[4] = {0, 0, 0, 0};
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int e = t + m * kBlock, i = e / kGjB, j = e % kGjB;
    if (bj == kb) {
      x1[m] = Min[(size_t)(bi * kGjB + i) * Np + K0 + j];
      y1[m] = Pin[e];
    } else {
      x1[m] = Pin[e];
      y1[m] = Min[(size_t)(K0 + i) * Np + bj * kGjB + j];
    }
    if (general) x2[m] = Min[(size_t)(bi * kGjB + i) * Np + K0 + j];
  }
  if (general) {
    const double* src = Min + (size_t)(bi * kGjB + r) * Np + bj * kGjB + c0;
#pragma unroll
    for (int q = 0; q < 4; ++q) sv[q] = src[q];
  }
  // first product: Z = P A_Kj (row panel and general tiles) or A_iK P (column panel)
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int e = t + m * kBlock;
    X[e / kGjB][e % kGjB] = x1[m];
    Y[e / kGjB][e % kGjB] = y1[m];
  }
  __syncthreads();
  tile_mm(X, Y, Z);
  __syncthreads();
  if (bi == kb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = Z[r][c0 + q];
    return;
  }
  if (bj == kb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = -Z[r][c0 + q];
    return;
  }
  // general tile: A_ij - A_iK (P A_Kj): the second product's right operand is Z as it lies; X takes A_iK, Y the product
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int e = t + m * kBlock;
    X[e / kGjB][e % kGjB] = x2[m];
  }
  __syncthreads();
  tile_mm(X, Z, Y);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) dst[q] = v[q] = sv[q] - Y[r][c0 + q];
  if (bi != kb + 1 || bj != kb + 1) return;
  // next pivot block: Pout = inv(A_K'K') by scalar Gauss-Jordan
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) X[r][c0 + q] = v[q];
  __syncthreads();
  for (int k = 0; k < kGjB; k += 2) {   // two pivots per step (see k_gj_pivot)
    double (*rd)[kGjB + 1] = ((k >> 1) & 1) ? Y : X;
    double (*wr)[kGjB + 1] = ((k >> 1) & 1) ? X : Y;
    const double pa = rd[k][k], pb = rd[k][k + 1], pd = rd[k + 1][k + 1];
    const double ar0 = rd[r][k], ar1 = rd[r][k + 1];
    double a0c[4], a1c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a0c[q] = rd[k][c0 + q];
      a1c[q] = rd[k + 1][c0 + q];
    }
    const double det = pa * pd - pb * pb;
    if (t == 0 && (!(pa > 0.0) || !(det > 0.0) || !isfinite(det))) *fail = 1;
    const double id = (det != 0.0) ? gj_rcp(det) : 0.0;
    const double p00 = pd * id, p01 = -pb * id, p11 = pa * id;
    const double t0 = ar0 * p00 + ar1 * p01, t1 = ar0 * p01 + ar1 * p11;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + q;
      if (r == k) v[q] = (c == k) ? p00 : ((c == k + 1) ? p01 : p00 * a0c[q] + p01 * a1c[q]);
      else if (r == k + 1) v[q] = (c == k) ? p01 : ((c == k + 1) ? p11 : p01 * a0c[q] + p11 * a1c[q]);
      else if (c == k) v[q] = -t0;
      else if (c == k + 1) v[q] = -t1;
      else v[q] = v[q] - (t0 * a0c[q] + t1 * a1c[q]);
      wr[r][c0 + q] = v[q];
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) Pout[r * kGjB + c0 + q] = v[q];
}

// x = inv * b : one wave per row (inv is symmetric; row reads are coalesced)
__global__ __launch_bounds__(kBlock) void k_dense_apply(int N, int Np, const double* __restrict__ inv,
                                                       const double* __restrict__ b, double* __restrict__ x,
                                                       const PcgScalars* S) {
  const int lane = threadIdx.x & 63;
  int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  // the first row's first stride is requested before the stop flag is waited for
  double a0 = 0.0, b0 = 0.0;
  if (i < N && lane < N) {
    a0 = inv[(size_t)i * Np + lane];
    b0 = b[lane];
  }
  if (S && S->stop) return;
  for (bool first = true; i < N; i += gridDim.x * kWavesPerBlock, first = false) {
    const double* row = inv + (size_t)i * Np;
    double s = first ? a0 * b0 : 0.0;
    // four strides per trip, all eight loads requested before the first product (as a plain loop the compiler waits for
    // every stride's pair of loads before it requests the next: eleven dependent round trips for N = 750)
    int j = lane + (first ? 64 : 0);
    for (; j + 192 < N; j += 256) {
      const double r0 = row[j], r1 = row[j + 64], r2 = row[j + 128], r3 = row[j + 192];
      const double c0 = b[j], c1 = b[j + 64], c2 = b[j + 128], c3 = b[j + 192];
      s += r0 * c0;
      s += r1 * c1;
      s += r2 * c2;
      s += r3 * c3;
    }
    {
      const bool h0 = j < N, h1 = j + 64 < N, h2 = j + 128 < N;
      const double r0 = h0 ? row[j] : 0.0, r1 = h1 ? row[j + 64] : 0.0, r2 = h2 ? row[j + 128] : 0.0;
      const double c0 = h0 ? b[j] : 0.0, c1 = h1 ? b[j + 64] : 0.0, c2 = h2 ? b[j + 128] : 0.0;
      s += r0 * c0;
      s += r1 * c1;
      s += r2 * c2;
    }
    s = wave_sum(s);
    if (lane == 0) x[i] = s;
  }
}

// partials[0][blk] = sum z.a ; partials[1][blk] = sum z.b (b optional)
__global__ __launch_bounds__(kBlock) void k_dots2(int n, const double* __restrict__ z, const double* __restrict__ a,
                                                  const double* __restrict__ b, double* __restrict__ partials,
                                                  const PcgScalars* S) {
  if (S && S->stop) return;
  double acc[2] = {0.0, 0.0};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    acc[0] += z[i] * a[i];
    if (b) acc[1] += z[i] * b[i];
  }
  block_sum_store<2>(acc, partials, kMaxPartials);
}

// --------------------------------------------------------------------------------- product lists on the device
// The patterns of P, A P and P^T A P come from the host (a per-row sort of a few distinct columns each); the lists of
// block products behind every entry -- 13 M + 3 M triples on C4 level 0, the bulk of the symbolic phase's time and of
// the set-up's upload -- are made here from the patterns: one thread per TARGET entry walks its candidates in the order
// the host lists them (A P: the slots k of row i in ascending order, each contributing the entry of P row col(k) in
// column c if it has one; P^T A P: the entries t of column a of P in ascending order, each contributing the entry of
// A P row row(t) in column c) -- first counting, then, after a prefix sum over the targets, writing.
struct ApPattern {
  int f_lo = 0;                    // targets [f_lo, nap) are listed (row-owner mode: this rank's rows' entries; the arrays
                                   // indexed by a target are then allocated for that range and shifted)
  int nap = 0;
  const int* ap_row = nullptr;     // [nap] fine row of target f
  const int* ap_col = nullptr;     // [nap] coarse column of target f (ascending within a row)
  const int* ap_rowptr = nullptr;  // [n + 1]
};
// Eight lanes per target: lane q of a target's group takes the candidates q, q + 8, ... of the walk, the matches of one
// round are ranked by a ballot (candidate order = lane order within a round), so the list keeps the host's order while
// the walk -- a chain of dependent index loads and a binary search per candidate -- is eight times shorter.
template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_ap_list(ApPattern ap, const int* __restrict__ a_rowptr, const int* __restrict__ a_col,
                                                    const int* __restrict__ p_rowptr, const int* __restrict__ p_col,
                                                    int* __restrict__ cnt_or_ptr, int* __restrict__ la, int* __restrict__ lb,
                                                    int* __restrict__ lt) {
  const int lane = threadIdx.x & 63, sub = lane & 7, g8 = lane >> 3;
  const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), nwaves = gridDim.x * kWavesPerBlock;
  for (int f0 = ap.f_lo + wave * 8; f0 < ap.nap; f0 += nwaves * 8) {
    const int f = f0 + g8;
    const bool valid = f < ap.nap;
    const int i = valid ? ap.ap_row[f] : 0, c = valid ? ap.ap_col[f] : 0;
    int k = valid ? a_rowptr[i] + sub : 0;
    const int kend = valid ? a_rowptr[i + 1] : 0;
    int out = (FILL && valid) ? cnt_or_ptr[f] : 0;
    while (__any(k < kend)) {
      int e = -1;
      if (k < kend) {
        const int j = a_col[k];
        e = find_sorted(p_col, p_rowptr[j], p_rowptr[j + 1], c);
      }
      const unsigned long long m = __ballot(e >= 0);
      const unsigned sm = (unsigned)(m >> (8 * g8)) & 0xFFu;
      if (FILL && e >= 0) {
        const int pos = out + __popc(sm & ((1u << sub) - 1u));
        la[pos] = k;
        lb[pos] = e;
        lt[pos] = f;
      }
      out += __popc(sm);
      k += 8;
    }
    if (!FILL && valid && sub == 0) cnt_or_ptr[f] = out;
  }
}
template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_rap_list(int nslot_c, const int* __restrict__ c_row, const int* __restrict__ c_col,
                                                     const int* __restrict__ t_ptr, const int* __restrict__ t_row,
                                                     const int* __restrict__ t_idx, ApPattern ap, int* __restrict__ cnt_or_ptr,
                                                     int* __restrict__ la, int* __restrict__ lb, int* __restrict__ lt) {
  const int lane = threadIdx.x & 63, sub = lane & 7, g8 = lane >> 3;
  const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), nwaves = gridDim.x * kWavesPerBlock;
  for (int s0 = wave * 8; s0 < nslot_c; s0 += nwaves * 8) {
    const int sl = s0 + g8;
    const bool valid = sl < nslot_c;
    const int a = valid ? c_row[sl] : 0, c = valid ? c_col[sl] : 0;
    const bool upper = valid && c >= a;   // upper triangle only: the lower one is mirrored by the numeric kernel
    int t = upper ? t_ptr[a] + sub : 0;
    const int tend = upper ? t_ptr[a + 1] : 0;
    int out = (FILL && valid) ? cnt_or_ptr[sl] : 0;
    while (__any(t < tend)) {
      int f = -1;
      if (t < tend) {
        const int i = t_row[t];
        f = find_sorted(ap.ap_col, ap.ap_rowptr[i], ap.ap_rowptr[i + 1], c);
      }
      const unsigned long long m = __ballot(f >= 0);
      const unsigned sm = (unsigned)(m >> (8 * g8)) & 0xFFu;
      if (FILL && f >= 0) {
        const int pos = out + __popc(sm & ((1u << sub) - 1u));
        la[pos] = t_idx[t];
        lb[pos] = f;
        lt[pos] = sl;
      }
      out += __popc(sm);
      t += 8;
    }
    if (!FILL && valid && sub == 0) cnt_or_ptr[sl] = out;
  }
}
// Exclusive prefix sum of n ints in place (v[n] receives the total): blocks of kScanChunk elements summed, the block
// sums scanned by one workgroup, the blocks rescanned with their offsets.
constexpr int kScanChunk = 4096;
__global__ __launch_bounds__(kBlock) void k_scan_sums(const int* __restrict__ v, int n, int* __restrict__ sums) {
  __shared__ int sm[kBlock];
  const int base = blockIdx.x * kScanChunk;
  int acc = 0;
  for (int q = threadIdx.x; q < kScanChunk && base + q < n; q += kBlock) acc += v[base + q];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = sm[0];
}
__global__ __launch_bounds__(kBlock) void k_scan_top(int* __restrict__ sums, int nb) {   // one workgroup; sums[nb] = total
  __shared__ int sm[kBlock];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += kBlock) {
    const int q = base + threadIdx.x;
    const int x = q < nb ? sums[q] : 0;
    sm[threadIdx.x] = x;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {   // Hillis-Steele inclusive scan
      const int y = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
      __syncthreads();
      sm[threadIdx.x] += y;
      __syncthreads();
    }
    if (q < nb) sums[q] = carry + sm[threadIdx.x] - x;
    __syncthreads();
    if (threadIdx.x == 0) carry += sm[kBlock - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[nb] = carry;
}
__global__ __launch_bounds__(kBlock) void k_scan_apply(int* __restrict__ v, int n, const int* __restrict__ sums, int nb) {
  // kScanChunk = 16 per thread: every thread scans its 16 consecutive elements, the threads' totals are scanned in LDS
  __shared__ int sm[kBlock];
  const int base = blockIdx.x * kScanChunk + threadIdx.x * (kScanChunk / kBlock);
  int x[kScanChunk / kBlock], tot = 0;
#pragma unroll
  for (int q = 0; q < kScanChunk / kBlock; ++q) {
    x[q] = base + q < n ? v[base + q] : 0;
    tot += x[q];
  }
  sm[threadIdx.x] = tot;
  __syncthreads();
  for (int off = 1; off < kBlock; off <<= 1) {
    const int y = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
    __syncthreads();
    sm[threadIdx.x] += y;
    __syncthreads();
  }
  int run = sums[blockIdx.x] + sm[threadIdx.x] - tot;
#pragma unroll
  for (int q = 0; q < kScanChunk / kBlock; ++q) {
    if (base + q < n) v[base + q] = run;
    run += x[q];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) v[n] = sums[nb];
}
// The same in ONE launch of one workgroup for short lists: every thread scans a contiguous piece, the pieces' totals are scanned in LDS.
constexpr int kScanSmall = 32768;
__global__ __launch_bounds__(1024) void k_scan_small(int* __restrict__ v, int n) {
  __shared__ int sm[1024];
  const int per = (n + 1023) / 1024, b = threadIdx.x * per, e = min(n, b + per);
  int tot = 0;
  for (int q = b; q < e; ++q) tot += v[q];
  sm[threadIdx.x] = tot;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int y = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
    __syncthreads();
    sm[threadIdx.x] += y;
    __syncthreads();
  }
  int run = sm[threadIdx.x] - tot;
  for (int q = b; q < e; ++q) {
    const int x = v[q];
    v[q] = run;
    run += x;
  }
  if (threadIdx.x == 1023) v[n] = sm[1023];
}
// Wave groups over the segments [ptr[f], ptr[f+1]): whole segments packed up to 64 items, a longer segment its own
// group -- make_groups' rule, applied independently to chunks of kGroupChunk segments (one thread each: short chunks keep the serial walk short; a chunk starts a new group; the
// grouping does not change a single sum).  Pass 1 counts a chunk's groups, pass 2 (after a prefix sum) writes them.
// (kGroupChunk: sgo_amg_host.h -- the host's make_groups applies the same rule to the same chunks)
template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_group_chunks(const int* __restrict__ ptr, int nseg, int* __restrict__ cnt_or_off,
                                                         int* __restrict__ grp) {
  const int ch = blockIdx.x * kBlock + threadIdx.x, s0 = ch * kGroupChunk;
  if (s0 >= nseg) return;
  const int s1 = min(nseg, s0 + kGroupChunk);
  int out = FILL ? cnt_or_off[ch] : 0, cur = 0, start = ptr[s0];
  // a group is recorded by its START position; the list is closed with the total by the caller
  bool open = false;
  for (int f = s0; f < s1; ++f) {
    const int b = ptr[f], e = ptr[f + 1], len = e - b;
    if (open && cur + len > 64) {
      if (FILL) grp[out] = start;
      ++out;
      open = false;
      cur = 0;
    }
    if (!open) {
      start = b;
      open = true;
    }
    cur += len;
    if (cur >= 64) {
      if (FILL) grp[out] = start;
      ++out;
      open = false;
      cur = 0;
    }
  }
  if (open) {
    if (FILL) grp[out] = start;
    ++out;
  }
  if (!FILL) cnt_or_off[ch] = out;
}

// --------------------------------------------------------------------------------- host
template <class T>
T* dev_alloc(DevArena* pool, size_t count) {
  return (T*)pool->take(std::max<size_t>(count, 1) * sizeof(T));
}
template <class T>
T* dev_upload(DevArena* pool, const std::vector<T>& v, hipStream_t s) {
  T* p = dev_alloc<T>(pool, v.size());
  if (p && !v.empty()) hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
  return p;
}

int* dev_upload(DevArena* pool, const UVec& v, hipStream_t s) {
  int* d = dev_alloc<int>(pool, v.n);
  if (d && v.n) hipMemcpyAsync(d, v.p, v.n * sizeof(int), hipMemcpyHostToDevice, s);
  return d;
}

}  // namespace

// one level of the hierarchy on the device
struct AmgLevel {
  BsrDev A;                 // operator of this level (level 0 aliases the context's matrix)
  int spmv_grid = 0;
  // transfer to the next level (absent on the coarsest)
  int nc = 0;
  int* agg = nullptr;
  int* mem_ptr = nullptr;
  int* mem = nullptr;
  int* mem_grp = nullptr;   // wave groups over the member list, aligned to aggregates
  int mem_ngrp = 0;
  double* pos = nullptr;    // [n][2]
  double* d = nullptr;      // [n][2] lever arms
  GalerkinMap gal;
  bool smoothed = false;    // transfer by the smoothed prolongator P (below) instead of the tentative one
  PDev P;
  // folded cycle (see "folded V-cycle" above): the transfer operator P~ = (I - w D^-1 A) P as a PDev VIEW (row / col /
  // r_* / t_* of P~'s pattern, the one of A P), its pattern bookkeeping, and on levels >= 1 the merged list of k_up_fold
  bool fold = false;
  PDev PS;
  FoldDev F;
  UpDev U;
  // work vectors [n][3]
  double *xs = nullptr, *rs = nullptr;                     // smoother state of cycle()
  double* tR = nullptr;                                    // second residual buffer of multi-sweep smoothing
  double *bk = nullptr, *xk = nullptr, *z1 = nullptr, *z2 = nullptr, *q = nullptr;  // K-cycle FCG (levels >= 1)
  double *bk2 = nullptr, *p2 = nullptr, *q2 = nullptr;     // residual after the first FCG step; second direction; A p2
  double *pA = nullptr, *pB = nullptr, *pC = nullptr;      // [2][kMaxPartials] each
};

struct Amg {
  AmgConfig cfg;
  AmgProf prof;
  Sym0Dev S0;               // level-0 operator in symmetric storage (lv[0].A is its logical view)
  Tile0Dev T0;              // ... and its tile view
  Comm* comm = nullptr;     // multi-GPU, all-reduce mode: level-0 products over the units [u0, u1) (= rows [row0, row1)) + all-reduce
  const HaloDev* halo = nullptr;   // multi-GPU, row-owner mode (sgo_internal.h): level-0 work on the owned rows, boundary exchanges
  const HaloDev* slices = nullptr; // multi-GPU, all-reduce mode with a communicator: product vectors all-gathered by rank slices
  int u0 = 0, u1 = 0, row0 = 0, row1 = 0;
  bool comm_failed = false;
  long long level0_bytes = 0;   // device bytes of the level-0 transfer data (P blocks, A P blocks, product lists)
  DevArena* pool = nullptr;   // the caller's arena (not owned)
  std::vector<AmgLevel> lv;
  const double* d_poses = nullptr;
  const int* d_free_id = nullptr;
  int kdepth = 1 << 20;  // levels <= kdepth use the K-cycle (two FCG steps), deeper ones a V-cycle
  // Levels <= this take two FCG steps, deeper K-cycle levels one.  Two steps everywhere visit level l 2^l times:
  // on hierarchies of six levels (C4 with 5 % random closures: 100k -> 11k -> 2.5k -> 897 -> 627 -> 1) that is ~150
  // coarse launches per PCG iteration.  Two steps on level 1 only keep the PCG counts within 3-9 % (C4r 580 -> 597
  // over optimize(20)) at a third of the launches: C4r 22.5 -> 10.8 ms per GN iteration, 30k-pose graphs 1.2-1.3 x
  // (scripts/kcycle_sweep.py, profiles/r02_kcycle_sweep.txt); one step everywhere is faster still on average but
  // needs 40-50 % more iterations and doubles the worst solve.
  int fcg2_depth = 1;
  // coarsest dense inverse (row-major, leading dimension Np = N rounded up to 32)
  int N = 0, Np = 0;
  double* inv = nullptr;
  double* inv0 = nullptr;  // ping-pong buffers of the block Gauss-Jordan; `inv` is the one the last step writes
  double* inv1 = nullptr;
  double* gjP[2] = {nullptr, nullptr};   // [32][32] inverse of the current / next pivot block
  int* d_fail = nullptr;
  std::string desc;
  AmgKeptAgg kept;   // host copies of every level's aggregates (a rebuild may keep them: AmgConfig::keep_agg)
};

namespace {

struct Scope {
  const AmgProf& p;
  Scope(const AmgProf& p_, int kid, double bytes) : p(p_) {
    if (p.begin) p.begin(p.user, kid, bytes);
  }
  ~Scope() {
    if (p.end) p.end(p.user);
  }
};

double bytes_spmv(const BsrDev& A) { return 80.0 * A.nslot + 48.0 * A.n; }

// Values of the operator of level l+1 from those of level l (lever arms L.d must be current):
// tentative prolongator: one Galerkin pass; smoothed: P values, AP = A P, A_c = P^T AP.
// Multi-GPU, row-owner mode (level 0 only; the coarser levels are replicated): every rank makes the entries of P and of
// A P of its own fine rows -- A P needs the P rows of the neighbours' boundary rows: one exchange of 72-byte records --
// and its rows' share of P^T A P; the ranks' partial coarse operators are summed by an all-reduce of the level-1 blocks.
void launch_coarse_operator(Amg* m, hipStream_t s, AmgLevel& L, AmgLevel& C, bool level0) {
  const HaloDev* H = level0 ? m->halo : nullptr;
  const int row0 = H ? H->row0 : 0, row1 = H ? H->row1 : 0;
  if (!L.smoothed) {
    {
      Scope sc(m->prof, level0 ? K_GALERKIN0 : K_GALERKIN, (72.0 + 16.0 + 32.0) * L.A.nslot + 72.0 * C.A.nslot);
      SGO_LAUNCH(k_galerkin, dim3(grid_for(L.gal.ngrp, kWavesPerBlock)), dim3(kBlock), 0, s, L.A, C.A, L.gal, L.d, row0, row1);
    }
    if (H && H->comm) {
      std::string e;
      if (!H->comm->allreduce_f64(C.A.blk, 9 * (size_t)C.A.nslot, s, &e)) m->comm_failed = true;
    }
    return;
  }
  PDev& P = L.P;
  if (P.dF) {   // filtered smoothing: the diagonal blocks of the strong connections' operator first
    Scope sc(m->prof, level0 ? K_SA_P0 : K_SA_P, (72.0 + 9.0 + 16.0) * L.A.nslot + 144.0 * L.A.n);
    SGO_LAUNCH(k_filtered_diag, dim3(grid_for(P.f_ngrp, kWavesPerBlock)), dim3(kBlock), 0, s, L.A, (const int*)P.f_grp, P.f_ngrp, P.strong,
               (const double*)L.pos, P.dF, P.dinvF, row0, row1);
  }
  {
    Scope sc(m->prof, level0 ? K_SA_P0 : K_SA_P, (72.0 + 12.0 + 16.0) * P.val.n + 80.0 * P.r_n);
    SGO_LAUNCH(k_p_values, dim3(grid_for(P.val.ngrp, kWavesPerBlock)), dim3(kBlock), 0, s, L.A, P, (const int*)L.agg,
               (const double*)L.d, m->cfg.omega_p, row0, row1);
  }
  if (H && H->comm) {
    std::string e;
    if (!halo_exchange(*H, s, P.blk, 9, H->pent, H->pemax, HaloScalars(), &e)) m->comm_failed = true;
  }
  {
    Scope sc(m->prof, level0 ? K_SA_AP0 : K_SA_AP, 156.0 * P.ap.n + 72.0 * P.nap);
    SGO_LAUNCH((k_block_products<true, false, false>), dim3(grid_for(P.ap.ngrp, kWavesPerBlock)), dim3(kBlock), 0, s,
               P.ap, L.A, (const double*)nullptr, (const double*)P.blk, P.apblk, (size_t)P.nap, (const int*)nullptr,
               (const int*)L.A.row, row0, P.local_lists ? 0 : row1, 0);   // (local lists hold this rank's products only: no filter)
  }
  if (L.fold) {
    Scope sc(m->prof, level0 ? K_PTILDE0 : K_PTILDE, (72.0 + 36.0 + 72.0 + 12.0 + 72.0) * (L.F.f_hi - L.F.f_lo));
    SGO_LAUNCH(k_ptilde_values, dim3(grid_for((long long)(L.F.f_hi - L.F.f_lo), kBlock)), dim3(kBlock), 0, s, L.F, (const double*)P.apblk,
               (const double*)P.blk, (const double*)L.A.dinv, m->cfg.omega, L.PS);
  }
  if (P.local_lists) hipMemsetAsync(C.A.blk, 0, sizeof(double) * 9 * (size_t)C.A.nslot, s);   // targets this rank has no product for
  {
    Scope sc(m->prof, level0 ? K_SA_RAP0 : K_SA_RAP, 156.0 * P.rap.n + 72.0 * C.A.nslot);
    SGO_LAUNCH((k_block_products<false, true, true>), dim3(grid_for(P.rap.ngrp, kWavesPerBlock)), dim3(kBlock), 0, s,
               P.rap, BsrDev(), (const double*)P.blk, (const double*)P.apblk, C.A.blk, (size_t)C.A.nslot, (const int*)P.rap_mirror,
               (const int*)P.row, row0, P.local_lists ? 0 : row1, 1);
  }
  if (H && H->comm) {
    std::string e;
    if (!H->comm->allreduce_f64(C.A.blk, 9 * (size_t)C.A.nslot, s, &e)) m->comm_failed = true;
  }
}

// The coarse solution of level l as seen by its parent: xk (dense level) or the flexible-CG
// combination c1 z1 + c2 p2 whose scalars are ratios of the partial sums the FCG SpMVs left
// in pA / pB (never materialised: the consumers apply it on the fly).
struct CoarseSol {
  const double* u1 = nullptr;
  const double* u2 = nullptr;
  SpmvRatio c1, c2;
};

// xs = omega Dinv rhs (first sweep from zero), rs = rhs - A xs on a coarse level.  One fused launch (the gathered operand is
// omega Dinv[col] rhs[col]: 72 B per slot) where a launch costs more than the level's data; on a LARGE level (C5's first coarse
// levels: millions of slots) the sweep from zero as a vector kernel of its own and a plain residual pass that gathers 24 B per slot.
constexpr int kUnfuseSlots = 400000;   // (swept on C5 / C4r: 10^6 87.0 / 64.4 M edge-Jacobians/s, 4 10^5 89.4 / 65.3, 1.5 10^5 89.9 / 64.8, 5 10^4 90.3 / 63.4)
static void pre_resid(Amg* m, hipStream_t s, AmgLevel& L, const double* rhs, const PcgScalars* S) {
  SpmvArgs a{};
  a.b = rhs; a.y = L.rs; a.omega = m->cfg.omega; a.S = S;
  if (L.A.nslot >= kUnfuseSlots) {
    {
      Scope sc(m->prof, K_DOT, 96.0 * L.A.n);
      launch_precond_bj(s, L.A.n, L.A.dinv, rhs, L.xs, m->cfg.omega);
    }
    a.x = L.xs;
    Scope sc(m->prof, K_SPMV_RESID, 80.0 * L.A.nslot + 72.0 * L.A.n);
    launch_spmv_ex(s, L.A, SPMV_RESID, a);
    return;
  }
  a.y2 = L.xs;
  Scope sc(m->prof, K_SPMV_PRE_RESID, 80.0 * L.A.nslot + 120.0 * L.A.n);
  launch_spmv_ex(s, L.A, SPMV_PRE_RESID, a);
}

int cycle(Amg* m, hipStream_t s, int l, const double* rhs, const double* rhs_sub, const SpmvRatio& rhs_c,
          double* rhs_out, double* out, const double* dotvec, double* dotparts, const PcgScalars* S,
          const double* dotvec2 = nullptr, int xs0_ready = 0);

// Two flexible-CG steps on level l for A x = bk (Notay's K-cycle), with the vector updates
// fused into the neighbouring SpMV-type launches:
//   z1 = cycle(bk);               q = A z1            -> pA = (z1.q, z1.bk)     a1 = pA[1]/pA[0]
//   z2 = cycle(bk - a1 q) [bk2];  q = A (z2 - b z1)   -> pC = (q.z2), b = pC[0]/pA[0]; p2 = z2 - b z1
//                                                        pB = (p2.q, p2.bk2)      a2 = pB[1]/pB[0]
//   x = a1 z1 + a2 p2   (returned as a CoarseSol)
CoarseSol fcg(Amg* m, hipStream_t s, int l, const PcgScalars* S) {
  AmgLevel& L = m->lv[l];
  SpmvRatio none;
  cycle(m, s, l, L.bk, nullptr, none, nullptr, L.z1, nullptr, nullptr, S);
  int gA;
  {
    SpmvArgs a{};
    a.x = L.z1; a.y = L.q; a.dotA = L.z1; a.dotB = L.z1; a.dotC = L.bk; a.partials = L.pA; a.S = S;
    Scope sc(m->prof, K_SPMV_AX, bytes_spmv(L.A));
    gA = launch_spmv_ex(s, L.A, SPMV_AX, a);
  }
  SpmvRatio a1{L.pA + kMaxPartials, gA, L.pA, gA};
  if (l > m->fcg2_depth) {  // one FCG step only (steepest descent in the cycle's direction)
    CoarseSol one;
    one.u1 = L.z1;
    one.c1 = a1;
    return one;
  }
  const int gC = cycle(m, s, l, L.bk, L.q, a1, L.bk2, L.z2, L.q, L.pC, S);
  int gB;
  {
    SpmvArgs a{};
    a.x = L.z2; a.x2 = L.z1; a.x_out = L.p2; a.y = L.q2; a.dotC = L.bk2; a.partials = L.pB; a.S = S;
    a.c1 = SpmvRatio{L.pC, gC, L.pA, gA};
    Scope sc(m->prof, K_SPMV_AX_C, 80.0 * L.A.nslot + 120.0 * L.A.n);
    gB = launch_spmv_ex(s, L.A, SPMV_AX_C, a);
  }
  CoarseSol cs;
  cs.u1 = L.z1;
  cs.c1 = a1;
  cs.u2 = L.p2;
  cs.c2 = SpmvRatio{L.pB + kMaxPartials, gB, L.pB, gB};
  return cs;
}

// out = cycle(l, rhs'): rhs' = rhs - c rhs_sub when rhs_sub != nullptr (stored to rhs_out).
// pre-smooth from zero + residual (one launch), restrict, coarse solve (dense inverse or two FCG
// steps), prolongation fused into the post-smoothing launch on levels >= 1 (separate launch on
// level 0, where the extra gathers would cost more than the launch).  Optional partials of
// dotvec . out (and dotvec2 . out).  Returns the grid of the last kernel.
// The coarse solve for the right-hand side C.bk of level l + 1, as the parent level sees it.
CoarseSol coarse_solve(Amg* m, hipStream_t s, int l, const PcgScalars* S) {
  AmgLevel& C = m->lv[l + 1];
  const int last = (int)m->lv.size() - 1;
  CoarseSol cs;
  if (l + 1 == last) {
    Scope sc(m->prof, K_DENSE_APPLY, 8.0 * m->N * m->N);
    SGO_LAUNCH(k_dense_apply, dim3(grid_for(m->N, kWavesPerBlock)), dim3(kBlock), 0, s, m->N, m->Np, m->inv, C.bk,
                       C.xk, S);
    cs.u1 = C.xk;
  } else if (l + 1 > m->kdepth && m->lv[l + 1].smoothed) {  // V-cycle below the K-cycle depth; a level whose own
                                                             // transfer is the tentative one always gets the K-cycle
    SpmvRatio none;
    cycle(m, s, l + 1, C.bk, nullptr, none, nullptr, C.xk, nullptr, nullptr, S);
    cs.u1 = C.xk;
  } else {
    cs = fcg(m, s, l + 1, S);
  }
  return cs;
}

// out = cycle(l, rhs) in the folded form (see "folded V-cycle"): restriction with P~^T of the right-hand side itself,
// coarse solve, prolongation with P~ onto the two-sweep term M2 rhs -- on level 0 the level-0 Jacobi pass; on the coarser
// levels part of the prolongation launch.
int cycle_fold(Amg* m, hipStream_t s, int l, const double* rhs, double* out, const double* dotvec, double* dotparts,
               const PcgScalars* S, const double* dotvec2, int xs0_ready, const double* xadd = nullptr) {
  AmgLevel& L = m->lv[l];
  AmgLevel& C = m->lv[l + 1];
  bool fused0 = false;
  if (l == 0) {   // (single GPU: multi-GPU runs keep level 0 unfolded)
    if (!xs0_ready) {   // xs = omega Dinv rhs (normally left by the producer of rhs)
      Scope sc(m->prof, K_DOT, 96.0 * L.A.n);
      launch_precond_bj(s, L.A.n, m->S0.dinv, rhs, L.xs, m->cfg.omega);
    }
    // M2 rhs = xs + omega Dinv (rhs - H xs): the level-0 Jacobi pass, into rs.  (It does not depend on the coarse levels;
    // running it BESIDE them on a second stream -- a parallel branch of the captured hipGraph -- was measured: the fork and
    // join cost 25 us per PCG iteration on this runtime, C2 1.45 -> 1.95 ms per GN iteration.)
    Spmv0Args b{};
    b.x = L.xs; b.b = rhs; b.y = L.rs; b.omega = m->cfg.omega; b.S = S;
    if (m->T0.ntile == 0) {   // wave-group kernel: the pass and the restriction in one launch
      const int nb_spmv = grid_for(m->S0.ngrp, kWavesPerBlock), nb_main = grid_for(L.PS.t_ngrp, kWavesPerBlock);
      Scope sc(m->prof, K_JACOBI0_RESTRICT, 76.0 * m->S0.npairs + 168.0 * m->S0.n + 44.0 * L.PS.t_n + 24.0 * L.A.n + 24.0 * L.nc);
      SGO_LAUNCH(k_jacobi0_restrict, dim3(nb_spmv + nb_main + L.PS.t_nlong), dim3(kBlock), 0, s, m->S0, b, nb_spmv, L.PS, C.bk, nb_main);
      fused0 = true;
    } else {
      const bool f32 = m->S0.fblk != nullptr;
      Scope sc(m->prof, f32 ? K_SPMV0T_JACOBI_F32 : K_SPMV0T_JACOBI, (f32 ? 40.0 : 76.0) * m->S0.npairs + 168.0 * m->S0.n);
      launch_spmv0_any(s, m->S0, m->T0, S0_JACOBI, b);
    }
  }
  if (!fused0) {
    Scope sc(m->prof, l == 0 ? K_RESTRICT_P0 : K_RESTRICT_P, 44.0 * L.PS.t_n + 24.0 * L.A.n + 24.0 * L.nc);
    launch_restrict_p(s, L.PS, rhs, C.bk, S, 0, 0);
  }
  const CoarseSol cs = coarse_solve(m, s, l, S);
  if (l > 0) {
    Scope sc(m->prof, K_UP_FOLD, 80.0 * L.A.nslot + 44.0 * L.PS.r_n + 100.0 * L.A.n);
    const int grid = grid_for(L.U.ngrp, kWavesPerBlock);
    SGO_LAUNCH(k_up_fold, dim3(grid), dim3(kBlock), 0, s, L.A, L.U, L.PS, rhs, m->cfg.omega, cs.u1, cs.c1, cs.u2, cs.c2, out, S, xadd);
    return grid;
  }
  // (the dot products' partial sums are re-reduced by every workgroup of the consumer: a few hundred of them, not thousands)
  const int grid = std::min(grid_for(L.PS.r_ngrp, kFoldThreads / 64), 512);
  Scope sc(m->prof, K_PROLONG_FOLD0, 44.0 * L.PS.r_n + 72.0 * L.A.n);
  SGO_LAUNCH(k_prolong_fold, dim3(grid), dim3(kFoldThreads), 0, s, L.PS, cs.u1, cs.c1, cs.u2, cs.c2, (const double*)L.rs, out, S, dotvec,
             dotvec2, dotvec ? dotparts : nullptr);
  return grid;
}

int cycle(Amg* m, hipStream_t s, int l, const double* rhs, const double* rhs_sub, const SpmvRatio& rhs_c,
          double* rhs_out, double* out, const double* dotvec, double* dotparts, const PcgScalars* S,
          const double* dotvec2, int xs0_ready) {
  AmgLevel& L = m->lv[l];
  AmgLevel& C = m->lv[l + 1];
  {
    const int nu_l = (l > 0 && L.tR && L.smoothed) ? std::max(1, m->cfg.nu_coarse) : 1;
    if (L.fold && nu_l == 1 && !rhs_sub && (l == 0 ? !(m->halo || m->comm) : !dotvec)) return cycle_fold(m, s, l, rhs, out, dotvec, dotparts, S, dotvec2, xs0_ready);
    if (L.fold && nu_l == 2 && !rhs_sub && l > 0 && !dotvec) {
      // two sweeps per side = one explicit sweep around the folded cycle: S E S with E the folded cycle's error propagator
      pre_resid(m, s, L, rhs, S);   // xs = omega Dinv rhs, rs = rhs - A xs
      cycle_fold(m, s, l, L.rs, L.tR, nullptr, nullptr, S, nullptr, 0, L.xs);   // tR = xs + cycle(rs)
      SpmvArgs a{};
      a.x = L.tR; a.b = rhs; a.y = out; a.omega = m->cfg.omega; a.S = S;
      Scope sc(m->prof, K_SPMV_JACOBI, 80.0 * L.A.nslot + 120.0 * L.A.n);
      return launch_spmv_ex(s, L.A, SPMV_JACOBI, a);
    }
  }
  const double* rhs_eff = rhs;
  const HaloDev* H = l == 0 ? m->halo : nullptr;           // multi-GPU, row-owner mode
  const bool sharded0 = l == 0 && m->comm != nullptr && !H;   // multi-GPU, all-reduce mode
  const int frow0 = H ? H->row0 : (sharded0 ? m->row0 : 0), frow1 = H ? H->row1 : (sharded0 ? m->row1 : 0);
  if (l == 0) {
    // finest level, symmetric storage: xs = omega Dinv rhs (first sweep from zero; normally left by the
    // producer of rhs), then the residual rs = rhs - H xs in one pass over the stored blocks
    if (!xs0_ready) {
      Scope sc(m->prof, K_DOT, 96.0 * L.A.n);
      if (H) launch_precond_bj(s, H->row1 - H->row0, m->S0.dinv + 6 * (size_t)H->row0, rhs + 3 * (size_t)H->row0, L.xs + 3 * (size_t)H->row0, m->cfg.omega);
      else launch_precond_bj(s, L.A.n, m->S0.dinv, rhs, L.xs, m->cfg.omega);
    }
    Spmv0Args a{};
    a.x = L.xs; a.b = rhs; a.y = L.rs; a.S = S;
    if (H) {   // the neighbours' boundary rows of xs (unless the caller keeps them current itself), then this rank's tiles
      std::string e;
      if (xs0_ready < 2 && !halo_exchange(*H, s, L.xs, 3, H->bnd, H->bmax, HaloScalars(), &e)) m->comm_failed = true;
      a.u0 = H->u0; a.u1 = H->u1;
    } else if (m->comm) {
      a.u0 = m->u0; a.u1 = m->u1;
      if (!m->slices) hipMemsetAsync(L.rs, 0, sizeof(double) * 3 * (size_t)L.A.n, s);   // (the restriction reads this rank's rows only)
    }
    if (!(H || m->comm) || a.u1 > a.u0) {
      const bool f32 = m->T0.ntile > 0 && m->S0.fblk != nullptr;   // (fp32 copy of the blocks: 36 + 4 B per pair)
      Scope sc(m->prof, m->T0.ntile > 0 ? (f32 ? K_SPMV0T_RESID_F32 : K_SPMV0T_RESID) : K_SPMV0_RESID,
               ((f32 ? 40.0 : 76.0) * m->S0.npairs + 120.0 * m->S0.n) / (H ? H->G : 1));
      launch_spmv0_any(s, m->S0, m->T0, S0_RESID, a);
    }
    // multi-GPU: the residual stays a per-rank partial (this rank's rows); the restriction below takes only those
    // rows and the coarse right-hand side is what gets all-reduced (3 n_c doubles instead of 3 n)
  } else {
    SpmvArgs a{};
    a.b = rhs; a.y = L.rs; a.y2 = L.xs; a.omega = m->cfg.omega; a.S = S;
    if (rhs_sub) {
      a.bsub = rhs_sub; a.c1 = rhs_c; a.b_out = rhs_out;
      rhs_eff = rhs_out;
      Scope sc(m->prof, K_SPMV_PRE_RESID_S, 80.0 * L.A.nslot + 168.0 * L.A.n);
      launch_spmv_ex(s, L.A, SPMV_PRE_RESID_S, a);
    } else {
      pre_resid(m, s, L, rhs, S);
    }
  }
  // further pre-smoothing sweeps (levels walked by the V-cycle only): sweep s applied to the residual
  // of sweep s-1 gives the next correction (accumulated into xs) and the next residual (rs <-> tR)
  const int nu = (l > 0 && L.tR && L.smoothed) ? std::max(1, m->cfg.nu_coarse) : 1;
  double* res = L.rs;
  for (int sw = 1; sw < nu; ++sw) {
    double* nxt = (res == L.rs) ? L.tR : L.rs;
    SpmvArgs a{};
    a.b = res; a.y = nxt; a.y2 = L.xs; a.omega = m->cfg.omega; a.S = S;
    Scope sc(m->prof, K_SPMV_PRE_RESID_ACC, 80.0 * L.A.nslot + 144.0 * L.A.n);
    launch_spmv_ex(s, L.A, SPMV_PRE_RESID_ACC, a);
    res = nxt;
  }
  if (H && L.smoothed && L.P.local_lists) hipMemsetAsync(C.bk, 0, sizeof(double) * 3 * (size_t)C.A.n, s);   // coarse rows none of this rank's rows reaches
  if (L.smoothed) {
    Scope sc(m->prof, l == 0 ? K_RESTRICT_P0 : K_RESTRICT_P, 44.0 * L.P.t_n + 24.0 * L.A.n + 24.0 * L.nc);
    launch_restrict_p(s, L.P, res, C.bk, S, frow0, frow1);
  } else {
    Scope sc(m->prof, l == 0 ? K_RESTRICT0 : K_RESTRICT, 40.0 * L.A.n + 24.0 * L.nc);
    SGO_LAUNCH(k_restrict, dim3(grid_for(L.mem_ngrp, kWavesPerBlock)), dim3(kBlock), 0, s, L.mem_ngrp, L.mem_grp,
                       L.mem, L.agg, L.d, res, C.bk, S, frow0, frow1);
  }
  if (sharded0 || H) {   // the ranks' partial coarse right-hand sides (each from its own fine rows) -> their sum
    std::string e;
    if (!(H ? H->comm : m->comm)->allreduce_f64(C.bk, 3 * (size_t)C.A.n, s, &e)) m->comm_failed = true;
  }
  const CoarseSol cs = coarse_solve(m, s, l, S);
  SpmvArgs a{};
  a.x = L.xs; a.b = rhs_eff; a.y = out; a.omega = m->cfg.omega; a.S = S;
  if (dotvec) {
    a.dotA = dotvec;
    a.dotA2 = dotvec2;
    a.partials = dotparts;
  }
  if (l == 0) {   // prolongation, then the post-smoothing sweep on the symmetric storage
    if (L.smoothed) {
      Scope sc(m->prof, K_PROLONG_P0, 44.0 * L.P.r_n + 52.0 * L.A.n);
      SGO_LAUNCH(k_prolong_p, dim3(grid_for(L.P.r_ngrp, kWavesPerBlock)), dim3(kBlock), 0, s, L.A.n, L.P, cs.u1, cs.c1, cs.u2, cs.c2,
                 L.xs, S, (const double*)nullptr, H ? H->row0 : 0, H ? H->row1 : 0);
    } else {
      Scope sc(m->prof, K_PROLONG0, 68.0 * L.A.n);
      SGO_LAUNCH(k_prolong_add, dim3(grid_for(H ? H->row1 - H->row0 : L.A.n, kBlock)), dim3(kBlock), 0, s, L.A.n, L.agg, L.d, cs.u1, cs.c1,
                         cs.u2, cs.c2, L.xs, S, (const double*)nullptr, H ? H->row0 : 0, H ? H->row1 : 0);
    }
    Spmv0Args b{};
    b.x = L.xs; b.b = rhs; b.y = out; b.omega = m->cfg.omega; b.S = S;
    if (dotvec) {
      b.dotA = dotvec;
      b.dotA2 = dotvec2;
      b.partials = dotparts;
    }
    if (H) {
      // row-owner mode: the corrected xs of the neighbours' boundary rows, then the sweep over this rank's tiles; the dot
      // products ride on the kernel as per-workgroup partials of the OWNED rows (the caller exchanges their sums)
      // (no exchange for that: this rank prolongates the replicated coarse solution on its copies of the neighbours' boundary
      // rows itself, k_prolong_rows)
      if (H->nhalo > 0) {
        PDev Ph = L.smoothed ? L.P : PDev();
        SGO_LAUNCH(k_prolong_rows, dim3(grid_for(H->nhalo, kBlock)), dim3(kBlock), 0, s, H->nhalo, H->halo_rows, Ph, (const int*)L.agg,
                   (const double*)L.d, cs.u1, cs.c1, cs.u2, cs.c2, L.xs, S);
      }
      b.u0 = H->u0; b.u1 = H->u1;
      Scope sc(m->prof, m->S0.fblk ? K_SPMV0T_JACOBI_F32 : K_SPMV0T_JACOBI, ((m->S0.fblk ? 40.0 : 76.0) * m->S0.npairs + 168.0 * m->S0.n) / H->G);
      return launch_spmv0_any(s, m->S0, m->T0, S0_JACOBI, b);
    }
    if (m->comm) {
      // this rank's rows, zeros elsewhere, all-reduce, then the dot products on the full vector (replicated)
      b.u0 = m->u0; b.u1 = m->u1;
      b.dotA = nullptr; b.dotA2 = nullptr; b.partials = nullptr;
      if (!m->slices) hipMemsetAsync(out, 0, sizeof(double) * 3 * (size_t)L.A.n, s);
      if (b.u1 > b.u0) {
        Scope sc(m->prof, m->T0.ntile > 0 ? K_SPMV0T_JACOBI : K_SPMV0_JACOBI, 76.0 * m->S0.npairs + 168.0 * m->S0.n);
        launch_spmv0_any(s, m->S0, m->T0, S0_JACOBI, b);
      }
      std::string e;
      if (m->slices) {
        if (!halo_gather_slices(*m->slices, s, out, 3, &e)) m->comm_failed = true;
      } else if (!m->comm->allreduce_f64(out, 3 * (size_t)L.A.n, s, &e)) {
        m->comm_failed = true;
      }
      if (!dotvec) return 0;
      const int grid = grid_for(3LL * L.A.n, kBlock);
      Scope sc(m->prof, K_DOT, 72.0 * L.A.n);
      SGO_LAUNCH(k_dots2, dim3(grid), dim3(kBlock), 0, s, 3 * L.A.n, (const double*)out, dotvec, dotvec2, dotparts, S);
      return grid;
    }
    const bool f32 = m->T0.ntile > 0 && m->S0.fblk != nullptr;
    Scope sc(m->prof, m->T0.ntile > 0 ? (f32 ? K_SPMV0T_JACOBI_F32 : K_SPMV0T_JACOBI) : K_SPMV0_JACOBI, (f32 ? 40.0 : 76.0) * m->S0.npairs + 168.0 * m->S0.n);
    return launch_spmv0_any(s, m->S0, m->T0, S0_JACOBI, b);
  }
  if (L.smoothed) {
    {
      Scope sc(m->prof, K_PROLONG_P, 80.0 * L.P.np + 52.0 * L.A.n);
      SGO_LAUNCH(k_prolong_p, dim3(grid_for(L.P.r_ngrp, kWavesPerBlock)), dim3(kBlock), 0, s, L.A.n, L.P, cs.u1, cs.c1, cs.u2, cs.c2,
                 L.xs, S, (const double*)nullptr, 0, 0);
    }
    // post-smoothing: nu sweeps, the first nu - 1 through the two residual buffers (free by now)
    for (int sw = 1; sw < nu; ++sw) {
      double* dst = (a.x == L.rs) ? L.tR : L.rs;
      SpmvArgs b = a;
      b.y = dst; b.dotA = nullptr; b.dotA2 = nullptr; b.partials = nullptr;
      {
        Scope sc(m->prof, K_SPMV_JACOBI, 80.0 * L.A.nslot + 120.0 * L.A.n);
        launch_spmv_ex(s, L.A, SPMV_JACOBI, b);
      }
      a.x = dst;
    }
    Scope sc(m->prof, K_SPMV_JACOBI, 80.0 * L.A.nslot + 120.0 * L.A.n);
    return launch_spmv_ex(s, L.A, SPMV_JACOBI, a);
  }
  if (L.A.nslot >= kUnfuseSlots && a.x == L.xs) {
    // a large level: the prolongation as a vector kernel of its own (same arithmetic, in place), then a plain sweep that gathers
    // 24 B per slot instead of x, the aggregate number, the coarse vectors and the lever arm (68-92 B)
    {
      Scope sc(m->prof, K_PROLONG, 52.0 * L.A.n + 48.0 * L.nc);
      SGO_LAUNCH(k_prolong_add, dim3(grid_for(L.A.n, kBlock)), dim3(kBlock), 0, s, L.A.n, L.agg, L.d, cs.u1, cs.c1, cs.u2, cs.c2, L.xs, S,
                 (const double*)nullptr, 0, 0);
    }
    Scope sc(m->prof, K_SPMV_JACOBI, 80.0 * L.A.nslot + 120.0 * L.A.n);
    return launch_spmv_ex(s, L.A, SPMV_JACOBI, a);
  }
  a.agg = L.agg; a.d = L.d; a.u1 = cs.u1; a.u2 = cs.u2; a.c1 = cs.c1; a.c2 = cs.c2;
  Scope sc(m->prof, K_SPMV_JACOBI_P, 80.0 * L.A.nslot + 140.0 * L.A.n + 48.0 * L.nc);
  return launch_spmv_ex(s, L.A, SPMV_JACOBI_P, a);
}

}  // namespace

int amg_num_levels(const Amg* m) { return m ? (int)m->lv.size() : 0; }
bool amg_has_filtered(const Amg* m) {
  if (m)
    for (const AmgLevel& L : m->lv)
      if (L.smoothed && L.P.dF) return true;
  return false;
}
long long amg_level0_bytes(const Amg* m) { return m ? m->level0_bytes : 0; }
bool amg_coarsest_not_spd(Amg* m, hipStream_t s) {
  int f = 0;
  if (!m || !m->d_fail) return false;
  if (hipMemcpyAsync(&f, m->d_fail, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return false;
  if (hipStreamSynchronize(s) != hipSuccess) return false;
  return f != 0;
}
void amg_describe(const Amg* m, std::string* out) { *out = m ? m->desc : ""; }
void amg_kept_aggregates(const Amg* m, AmgKeptAgg* out) {
  if (m && out) *out = m->kept;
}

void amg_destroy(Amg* m) {
  if (!m) return;
  delete m;   // the device memory belongs to the caller's arena
}


int amg_update(Amg* m, hipStream_t s, std::string* err) {
  const int last = (int)m->lv.size() - 1;
  {
    AmgLevel& L0 = m->lv[0];
    Scope sc(m->prof, K_POSITIONS0, 40.0 * L0.A.n);
    SGO_LAUNCH(k_positions0, dim3(grid_for(L0.A.n, kBlock)), dim3(kBlock), 0, s, L0.A.n, m->d_free_id, m->d_poses,
                       L0.pos);
  }
  for (int l = 0; l < last; ++l) {
    AmgLevel& L = m->lv[l];
    AmgLevel& C = m->lv[l + 1];
    {
      Scope sc(m->prof, K_CENTRES, 36.0 * L.A.n);
      SGO_LAUNCH(k_centres, dim3(grid_for(L.nc, kWavesPerBlock)), dim3(kBlock), 0, s, L.nc, L.mem_ptr, L.mem, L.pos, C.pos, L.d);
    }
    launch_coarse_operator(m, s, L, C, l == 0);
    {
      Scope sc(m->prof, K_LEVEL_DINV, 120.0 * C.A.n);
      SGO_LAUNCH(k_level_dinv, dim3(grid_for(C.A.n, kBlock)), dim3(kBlock), 0, s, C.A);
    }
  }
  {
    Scope sc(m->prof, K_DENSE_INVERT, 8.0 * m->Np * m->Np);
    const int nb = m->Np / kGjB;
    hipMemsetAsync(m->inv0, 0, sizeof(double) * (size_t)m->Np * m->Np, s);
    hipMemsetAsync(m->d_fail, 0, sizeof(int), s);
    if (last > 0)
      SGO_LAUNCH(k_dense_fill_unique, dim3(grid_for(m->lv[last].A.nslot, kBlock)), dim3(kBlock), 0, s, m->lv[last].A, m->Np, m->inv0);
    else   // level 0 can hold several slots per (row, col): duplicate edges
      SGO_LAUNCH(k_dense_fill, dim3(grid_for(m->lv[last].A.n, kBlock)), dim3(kBlock), 0, s, m->lv[last].A, m->Np, m->inv0);
    SGO_LAUNCH(k_gj_pivot, dim3(1), dim3(kGjB * kGjB), 0, s, m->inv0, m->Np, 0, m->gjP[0], m->d_fail);
    for (int kb = 0; kb < nb; ++kb) {   // the result of step kb lands in buffer (kb + 1) & 1: m->inv after the last one
      const double* src = (kb & 1) ? m->inv1 : m->inv0;
      double* dst = (kb & 1) ? m->inv0 : m->inv1;
      SGO_LAUNCH(k_gj_step, dim3(nb, nb), dim3(kBlock), 0, s, src, dst, m->Np, kb, (const double*)m->gjP[kb & 1],
                 m->gjP[(kb + 1) & 1], m->d_fail);
    }
  }
  const hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    if (err) *err = std::string("amg_update: a kernel launch failed: ") + hipGetErrorString(le);
    return SGO_EHIP;
  }
  return SGO_OK;
}

void amg_set_shard(Amg* m, Comm* comm, int u0, int u1, int row0, int row1, const HaloDev* slices) {
  m->slices = slices;
  m->comm = comm;
  m->u0 = u0;
  m->u1 = u1;
  m->row0 = row0;
  m->row1 = row1;
}
bool amg_comm_failed(const Amg* m) { return m && m->comm_failed; }
// Test hook: the coarse right-hand side the first half of a level-0 cycle produces from r (first sweep from zero,
// residual pass, restriction) -- with a shard set and no communicator, this rank's PARTIAL coarse right-hand side.
int amg_debug_coarse_rhs(Amg* m, hipStream_t s, const double* r, double* out_dev, int cap3) {
  if (!m || m->lv.size() < 2) return 0;
  AmgLevel& L = m->lv[0];
  AmgLevel& C = m->lv[1];
  const int n3c = 3 * C.A.n;
  if (cap3 < n3c) return -1;
  const bool sharded = m->comm != nullptr;
  if (L.fold) {   // folded cycle: the restriction with P~^T of r itself (the same vector in exact arithmetic)
    launch_restrict_p(s, L.PS, r, C.bk, nullptr, sharded ? m->row0 : 0, sharded ? m->row1 : 0);
    hipMemcpyAsync(out_dev, C.bk, sizeof(double) * n3c, hipMemcpyDeviceToDevice, s);
    return n3c;
  }
  launch_precond_bj(s, L.A.n, m->S0.dinv, r, L.xs, m->cfg.omega);
  Spmv0Args a{};
  a.x = L.xs; a.b = r; a.y = L.rs;
  if (sharded) {
    a.u0 = m->u0; a.u1 = m->u1;
    hipMemsetAsync(L.rs, 0, sizeof(double) * 3 * (size_t)L.A.n, s);
  }
  if (!sharded || a.u1 > a.u0) launch_spmv0_any(s, m->S0, m->T0, S0_RESID, a);
  if (L.smoothed)
    launch_restrict_p(s, L.P, L.rs, C.bk, nullptr, sharded ? m->row0 : 0, sharded ? m->row1 : 0);
  else
    SGO_LAUNCH(k_restrict, dim3(grid_for(L.mem_ngrp, kWavesPerBlock)), dim3(kBlock), 0, s, L.mem_ngrp, L.mem_grp, L.mem, L.agg,
               L.d, (const double*)L.rs, C.bk, (const PcgScalars*)nullptr, sharded ? m->row0 : 0, sharded ? m->row1 : 0);
  hipMemcpyAsync(out_dev, C.bk, sizeof(double) * n3c, hipMemcpyDeviceToDevice, s);
  return n3c;
}
double* amg_xs0(Amg* m) { return (m && m->lv.size() > 1) ? m->lv[0].xs : nullptr; }
double amg_omega(const Amg* m) { return m ? m->cfg.omega : 0.0; }

int amg_apply(Amg* m, hipStream_t s, const double* r, double* z, const double* dotvec, double* partials,
              const PcgScalars* S, const double* dotvec2, int xs0_ready) {
  if (m->lv.size() == 1) {  // single (dense) level: z = H^-1 r
    {
      Scope sc(m->prof, K_DENSE_APPLY, 8.0 * m->N * m->N);
      SGO_LAUNCH(k_dense_apply, dim3(grid_for(m->N, kWavesPerBlock)), dim3(kBlock), 0, s, m->N, m->Np, m->inv, r, z, S);
    }
    if (!dotvec) return 0;
    const int grid = grid_for(m->N, kBlock);
    Scope sc(m->prof, K_DOT, 24.0 * m->N);
    SGO_LAUNCH(k_dots2, dim3(grid), dim3(kBlock), 0, s, m->N, (const double*)z, dotvec, dotvec2, partials, S);
    return grid;
  }
  SpmvRatio none;
  return cycle(m, s, 0, r, nullptr, none, nullptr, z, dotvec, partials, S, dotvec2, xs0_ready);
}

// v[0..n) counts -> exclusive prefix sums in place, v[n] = total (device, on the stream); `sums` holds n / kScanChunk + 2 ints
void dev_scan_exclusive(hipStream_t s, int* v, int n, int* sums) {
  if (n <= kScanSmall) {   // one workgroup, one launch (the set-up's scans are mostly this small: three launches each were a third of its launches)
    SGO_LAUNCH(k_scan_small, dim3(1), dim3(1024), 0, s, v, n);
    return;
  }
  const int nb = (n + kScanChunk - 1) / kScanChunk;
  SGO_LAUNCH(k_scan_sums, dim3(std::max(nb, 1)), dim3(kBlock), 0, s, (const int*)v, n, sums);
  SGO_LAUNCH(k_scan_top, dim3(1), dim3(kBlock), 0, s, sums, nb);
  SGO_LAUNCH(k_scan_apply, dim3(std::max(nb, 1)), dim3(kBlock), 0, s, v, n, (const int*)sums, nb);
}
// Wave groups over the segments of ptr[0..nseg] on the device: returns the group list (ngrp + 1 starts, closed by the
// total) and ngrp; synchronises the stream once (the group count sizes the list).  nullptr on failure.
int* dev_make_groups(hipStream_t s, DevArena* pool, const int* ptr, int nseg, int total, int* ngrp_out) {
  const int nch = std::max(1, (nseg + kGroupChunk - 1) / kGroupChunk);
  int* cnt = dev_alloc<int>(pool, (size_t)nch + 1);
  int* sums = dev_alloc<int>(pool, (size_t)nch / kScanChunk + 3);
  if (!cnt || !sums) return nullptr;
  hipMemsetAsync(cnt, 0, sizeof(int) * ((size_t)nch + 1), s);
  SGO_LAUNCH((k_group_chunks<false>), dim3((nch + kBlock - 1) / kBlock), dim3(kBlock), 0, s, ptr, nseg, cnt, (int*)nullptr);
  dev_scan_exclusive(s, cnt, nch, sums);
  int ngrp = 0;
  if (hipMemcpyAsync(&ngrp, cnt + nch, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return nullptr;
  int* grp = dev_alloc<int>(pool, (size_t)ngrp + 1);
  if (!grp) return nullptr;
  SGO_LAUNCH((k_group_chunks<true>), dim3((nch + kBlock - 1) / kBlock), dim3(kBlock), 0, s, ptr, nseg, cnt, grp);
  hipMemcpyAsync(grp + ngrp, &total, sizeof(int), hipMemcpyHostToDevice, s);
  hipStreamSynchronize(s);   // `total` is a stack variable
  *ngrp_out = ngrp;
  return grp;
}

// The configuration amg_create works with for a level-0 operator of n rows / nslot logical slots: the caller's
// values, the environment overrides and the size-dependent choices.
AmgConfig amg_effective_config(const AmgConfig& cfg_in, int n, int nslot) {
  AmgConfig cfg = cfg_in;
  // A sweep on a coarse level is a 5-10 us launch whatever the graph; it pays when a PCG iteration is
  // dominated by level 0 (C4: 31 instead of 39 iterations, 9.7 instead of 10.6 ms) and costs a few
  // per cent on graphs whose level 0 is itself launch-bound (10k / 40k: 2.67 instead of 2.51 ms).
  cfg.nu_coarse = nslot >= 600000 ? 2 : 1;   // (with the folded cycle, scripts/nu_sweep.py: 30k / 300k 2.38 -> 2.31, 50k / 500k 3.00 -> 2.84 ms per GN
                                             // iteration with two sweeps; 20k / 200k 1.81 -> 1.91, C2 1.24 -> 1.36)
  if (const char* e = std::getenv("SGO_AMG_SMOOTH")) cfg.smooth = std::atoi(e) != 0;
  if (const char* e = std::getenv("SGO_AMG_LISTS")) cfg.lists_on_device = std::string(e) != "host";
  if (const char* e = std::getenv("SGO_AMG_THETA_FILTER")) cfg.theta_filter = std::atof(e);
  if (const char* e = std::getenv("SGO_AMG_FILTER")) cfg.filtered_smoothing = cfg.filtered_smoothing && std::atoi(e) != 0;   // (can only switch it off)
  // (experiment knobs of scripts/param_sweep.py.  Round 3, folded cycle, optimize(20) on C4 / C2 / C3s: omega 0.7 / 0.8 / 0.9 /
  // 1.0 -> 473 / 442 / 422 / 1015 PCG iterations on C4 (0.9 is 3-5 % better on every shape of scripts/robustness.py, 1.0 is
  // past the cliff: the default keeps its margin); omega_p 0.5 / 0.66 / 0.8 / 1.0 -> 486 / 442 / 557 / 809; theta 0.01 / 0.02 /
  // 0.04 -> C2 387 / 391 / 285, C3s 598 / 549 / 499, but on C4 theta 0.04 makes the smoothed level-0 operator too dense, the
  // hierarchy falls back to the tentative transfer with stagnating levels (100k -> 13k -> 4.6k -> 2.6k -> 2.0k) and needs
  // 175 iterations for the first solve instead of 30; 0.025 / 0.03 keep the smoothed transfer on C4 but its denser coarse
  // operators double the time (85 -> 159 / 167 ms), and 10k poses / 100k edges goes 354 -> 552 / 748 iterations: the
  // threshold is not a free parameter)
  if (const char* e = std::getenv("SGO_AMG_OMEGA")) cfg.omega = std::atof(e);
  if (const char* e = std::getenv("SGO_AMG_OMEGA_P")) cfg.omega_p = std::atof(e);
  if (const char* e = std::getenv("SGO_AMG_THETA")) cfg.theta = cfg.theta_coarse = std::atof(e);
  if (const char* e = std::getenv("SGO_AMG_FOLD")) cfg.fold = std::atoi(e) != 0;
  if (const char* e = std::getenv("SGO_AMG_FOLD0_ROWS")) cfg.fold0_rows = std::atoi(e);
  cfg.fold = cfg.fold && cfg.smooth && cfg.lists_on_device;
  // (the folded cycle folds ONE sweep per side into the transfers -- two would need the pattern of A A P; a second sweep is
  // one explicit sweep around it, see cycle().  One sweep everywhere was measured on C4: five coarse launches instead of
  // nine, but 27.9 instead of 22.1 PCG iterations, 5.07 against 4.59 ms per GN iteration)
  if (const char* e = std::getenv("SGO_AMG_NU")) cfg.nu_coarse = std::max(1, std::atoi(e));
  // larger graphs afford a larger dense coarsest level (its inverse costs O(N^3) once per GN
  // iteration, one K-cycle level less halves the coarse-level launches of every PCG iteration).  A SMALLER dense level
  // (one more sparse level) was measured in round 3: the dense inverse gets cheaper but the cycle weaker -- stopping at
  // <= 128 instead of <= 400 nodes: C2 19.6 -> 25.4 PCG iterations (1.76 -> 1.98 ms per GN iteration), C4 22.1 -> 26.5
  // (4.81 -> 5.82 ms); <= 48: C4 30.7 iterations (6.24 ms)
  cfg.coarsest_nodes = std::min(1000, std::max(cfg.coarsest_nodes, n / 1500));
  return cfg;
}

struct AmgHostL0 {
  HostCoarse hc;
  bool ready = false;
  bool agg_only = false;      // hc holds the aggregation only (agg, visit_c, nc; nc == 0: level 0 cannot be coarsened)
  double theta_used = 0.0;
};

#include "sgo_amg_dev.inc"

// Level 0's host analysis made AHEAD of amg_create: amg_host_l0_run is what a helper thread executes once the
// strength weights `w` of the level-0 slots (logical order of H0) are on the host; amg_create(..., pre0) then skips its
// own strength kernel, aggregation and symbolic phase for level 0.  `scratch` is used by the run (and must not be
// touched by anybody else meanwhile).
AmgHostL0* amg_host_l0_new() { return new AmgHostL0(); }
void amg_host_l0_free(AmgHostL0* p) { delete p; }
bool amg_host_l0_ready(const AmgHostL0* p) { return p && p->ready; }
bool amg_host_l0_agg_only(const AmgHostL0* p) { return p && p->ready && p->agg_only; }
void amg_host_l0_run(AmgHostL0* p, const HostLevel& H0, const std::vector<double>& w, const AmgConfig& cfg_in, ChunkArena* scratch, bool agg_only) {
  try {
    const AmgConfig cfg = amg_effective_config(cfg_in, H0.n, H0.nslot);
    if (H0.n <= cfg.coarsest_nodes || 1 >= cfg.max_levels) return;   // amg_create will not coarsen level 0 at all
    if (agg_only) {   // the patterns follow on the device (amg_create_dev)
      p->hc.nc = host_aggregate(H0, w, cfg, 0, scratch, p->hc.agg, p->hc.visit_c, &p->theta_used);
      p->agg_only = true;
      p->ready = true;
      return;
    }
    host_coarsen(H0, w, cfg, 0, scratch, p->hc);
    p->ready = p->hc.err.empty();
  } catch (...) {
    p->ready = false;   // amg_create does the work itself (and reports what fails there)
  }
}

Amg* amg_create(hipStream_t s, const BsrDev& A0, const Sym0Dev& S0, const Tile0Dev& T0, const HostLevel& H0, const double* d_poses,
                const int* d_free_id, const AmgConfig& cfg_in, const AmgProf& prof, std::string* err,
                ChunkArena* scratch, DevArena* arena, AmgHostL0* pre0, const AmgHalo* halo) {
  Amg* m = new Amg();
  if (halo) m->halo = halo->dev;
  m->pool = arena;
  m->cfg = cfg_in;
  m->S0 = S0;
  m->T0 = T0;
  m->cfg = amg_effective_config(cfg_in, A0.n, A0.nslot);
  // With the smoothed prolongator a V-cycle needs ~1.4x the PCG iterations of the K-cycle (C4: 39 vs
  // 27) at less than half the launches per iteration: V is the default there, K for the tentative one.
  if (m->cfg.smooth) m->kdepth = 0;
  if (const char* e = std::getenv("SGO_AMG_KDEPTH")) m->kdepth = std::atoi(e);
  if (const char* e = std::getenv("SGO_AMG_FCG2_DEPTH")) m->fcg2_depth = std::atoi(e);
  m->prof = prof;
  m->d_poses = d_poses;
  m->d_free_id = d_free_id;
  auto fail = [&](const std::string& msg) -> Amg* {
    if (err) *err = msg;
    amg_destroy(m);
    return nullptr;
  };

  HostLevel Hown;              // structure of the level being coarsened: H0 first, then the level built last
  const HostLevel* Hp = &H0;

  AmgLevel L0;
  L0.A = A0;
  m->lv.push_back(L0);
  char line[160];
  for (int l = 0;; ++l) {
    AmgLevel& L = m->lv[l];
    const HostLevel& H = *Hp;
    const int n = L.A.n, n3 = 3 * n;
    L.spmv_grid = grid_for(L.A.ngrp, kWavesPerBlock);
    std::snprintf(line, sizeof line, "L%d n=%d slots=%d; ", l, n, L.A.nslot);
    m->desc += line;
    L.xs = dev_alloc<double>(m->pool, n3);
    L.rs = dev_alloc<double>(m->pool, n3);
    L.tR = dev_alloc<double>(m->pool, n3);
    if (!L.pos) L.pos = dev_alloc<double>(m->pool, 2 * (size_t)n);
    if (l > 0) {
      L.bk = dev_alloc<double>(m->pool, n3);
      L.xk = dev_alloc<double>(m->pool, n3);
      L.z1 = dev_alloc<double>(m->pool, n3);
      L.z2 = dev_alloc<double>(m->pool, n3);
      L.q = dev_alloc<double>(m->pool, n3);
      L.bk2 = dev_alloc<double>(m->pool, n3);
      L.p2 = dev_alloc<double>(m->pool, n3);
      L.q2 = dev_alloc<double>(m->pool, n3);
      L.pA = dev_alloc<double>(m->pool, 2 * (size_t)kMaxPartials);
      L.pB = dev_alloc<double>(m->pool, 2 * (size_t)kMaxPartials);
      L.pC = dev_alloc<double>(m->pool, 2 * (size_t)kMaxPartials);
      if (!L.pC || !L.bk2 || !L.p2 || !L.q2) return fail("amg_create: out of device memory");
      hipMemsetAsync(L.pA, 0, sizeof(double) * 2 * kMaxPartials, s);
      hipMemsetAsync(L.pB, 0, sizeof(double) * 2 * kMaxPartials, s);
      hipMemsetAsync(L.pC, 0, sizeof(double) * 2 * kMaxPartials, s);
    }
    if (!L.xs || !L.rs || !L.pos) return fail("amg_create: out of device memory");
    if (n <= m->cfg.coarsest_nodes || l + 1 >= m->cfg.max_levels) break;

    // strength of connection from the current values of this level
    std::vector<double> w;
    if (l == 0 && halo && !(pre0 && pre0->ready)) {
      if (!halo->w0 || (int)halo->w0->size() != H.nslot) return fail("amg_create: row-owner mode needs the level-0 strength weights");
      w = *halo->w0;
    } else if (!(l == 0 && pre0 && pre0->ready)) {
      w.resize(H.nslot);
      double* d_w = dev_alloc<double>(m->pool, (size_t)std::max(H.nslot, 1));   // (stays in the arena until its rewind)
      if (!d_w) return fail("amg_create: out of device memory");
      SGO_LAUNCH(k_block_norms, dim3(grid_for(H.nslot, kBlock)), dim3(kBlock), 0, s, L.A, d_w);
      hipMemcpyAsync(w.data(), d_w, sizeof(double) * H.nslot, hipMemcpyDeviceToHost, s);
      const hipError_t e = hipStreamSynchronize(s);
      if (e != hipSuccess) return fail("amg_create: strength kernel failed");
    }
    HostCoarse hc_own;
    HostCoarse* hcp = &hc_own;
    if (m->cfg.keep_agg && l < (int)m->cfg.keep_agg->agg.size() && (int)m->cfg.keep_agg->agg[l].size() == n && !(l == 0 && pre0 && pre0->ready)) {
      hc_own.agg = m->cfg.keep_agg->agg[l];
      hc_own.visit_c = m->cfg.keep_agg->visit_c[l];
      hc_own.nc = m->cfg.keep_agg->nc[l];
      hc_own.reuse_agg = true;
    }
    if (l == 0 && pre0 && pre0->ready) {
      hcp = &pre0->hc;   // made ahead on the helper thread, from the same structure and the strengths at the same poses
    } else {
      host_coarsen(H, w, m->cfg, l, scratch, hc_own);
    }
    HostCoarse& hc = *hcp;
    if (!hc.err.empty()) return fail(hc.err);
    if (hc.stop) break;
    const int nc = hc.nc;
    const bool smooth = hc.smooth;
    std::vector<int>&agg = hc.agg, &mem_ptr = hc.mem_ptr, &mem = hc.mem, &order = hc.order, &tgt = hc.tgt, &grp_g = hc.grp_g,
                    &grp_c = hc.grp_c, &visit_c = hc.visit_c;
    SaHost& sa = hc.sa;
    HostLevel& Hc = hc.Hc;
    auto ms_since = [](std::chrono::steady_clock::time_point t) {
      return 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
    };
    if (std::getenv("SGO_VERBOSE"))
      std::fprintf(stderr, "[sgo] amg level %d: host aggregation + coarse structure %.1f ms (aggregate %.1f, sort %.1f; n=%d -> %d)%s\n",
                   l, hc.t_all, hc.t_agg, hc.t_sort, n, nc, hcp == &hc_own ? "" : " [made ahead on the helper thread]");

    // upload transfer data of level l and the structure of level l+1
    const auto tU = std::chrono::steady_clock::now();
    L.nc = nc;
    m->kept.agg.push_back(agg);
    m->kept.visit_c.push_back(visit_c);
    m->kept.nc.push_back(nc);
    L.agg = dev_upload(m->pool, agg, s);
    L.mem_ptr = dev_upload(m->pool, mem_ptr, s);
    L.mem = dev_upload(m->pool, mem, s);
    std::vector<int> grp_m = make_groups(mem_ptr);
    L.mem_grp = dev_upload(m->pool, grp_m, s);
    L.mem_ngrp = (int)grp_m.size() - 1;
    L.d = dev_alloc<double>(m->pool, 2 * (size_t)n);
    struct { int *ap_rowptr, *ap_col, *ap_row, *t_ptr, *t_idx; } l0_dev = {nullptr, nullptr, nullptr, nullptr, nullptr};
    struct { bool own; int F0, F1; } l0_own = {false, 0, 0};
    if (smooth) {
      PDev& P = L.P;
      L.smoothed = true;
      // Row-owner mode (level 0): the value lists, the streamed copies of P, the blocks of A P and the product lists are made
      // for this rank's rows only -- allocated for the rank's range of entries, base pointers shifted so that global entry
      // numbers address them; the patterns (a few integers per entry) and the gathered copy P.blk stay whole.
      const bool own = l == 0 && halo != nullptr;
      if (own && !sa.lists_on_device) return fail("amg_create: the row-owner mode needs the device-made product lists (SGO_AMG_LISTS=host is single-GPU only)");
      const int orow0 = own ? halo->dev->row0 : 0, orow1 = own ? halo->dev->row1 : n;
      const size_t E0 = (size_t)sa.p_rowptr[orow0], E1 = (size_t)sa.p_rowptr[orow1], npl = E1 - E0;   // entries of P held
      const size_t V0 = (size_t)sa.val_rowptr[orow0], V1 = (size_t)sa.val_rowptr[orow1];              // their value products (kept slots)
      auto up_range = [&](const int* host, size_t lo, size_t hi) -> int* {   // device copy of host[lo, hi), addressed by global numbers
        int* d = dev_alloc<int>(m->pool, hi - lo);
        if (d && hi > lo) hipMemcpyAsync(d, host + lo, (hi - lo) * sizeof(int), hipMemcpyHostToDevice, s);
        return d ? d - lo : nullptr;
      };
      P.local_lists = own;
      if (sa.filtered) {
        unsigned char* d_strong = (unsigned char*)m->pool->take(std::max<size_t>(sa.strong.size(), 1));
        P.dF = dev_alloc<double>(m->pool, 9 * (size_t)n);
        P.dinvF = dev_alloc<double>(m->pool, 9 * (size_t)n);
        if (!d_strong || !P.dF || !P.dinvF) return fail("amg_create: out of device memory");
        hipMemcpyAsync(d_strong, sa.strong.data(), sa.strong.size(), hipMemcpyHostToDevice, s);
        P.strong = d_strong;
        const std::vector<int> fg = make_groups(H.rowptr);
        P.f_grp = dev_upload(m->pool, fg, s);
        P.f_ngrp = (int)fg.size() - 1;
        if (!P.f_grp || hipStreamSynchronize(s) != hipSuccess) return fail("amg_create: out of device memory");   // (fg is a local)
      }
      P.np = (int)sa.p_row.size();
      P.stream_nt = npl >= 200000 ? 1 : 0;   // 2 x 72 B per block streamed per cycle: below ~30 MB it may stay cached
      P.rowptr = dev_upload(m->pool, sa.p_rowptr, s);
      P.row = dev_upload(m->pool, sa.p_row, s);
      P.col = dev_upload(m->pool, sa.p_col, s);
      P.blk = dev_alloc<double>(m->pool, 9 * (size_t)P.np);
      P.val.n = (int)(V1 - V0);
      P.val.a = up_range(sa.val_src.data(), V0, V1);
      P.val.tgt = up_range(sa.val_tgt.data(), V0, V1);
      std::vector<int> val_grp_l, r_grp_l, t_pos_l, t_row_l, t_col_l, t_idx_l, t_ptr_l, t_grp_l;   // (live until the stream is synchronised below)
      if (own) {
        std::vector<int> vptr;
        for (size_t q = V0; q < V1; ++q)
          if (q == V0 || sa.val_tgt[q] != sa.val_tgt[q - 1]) vptr.push_back((int)q);
        vptr.push_back((int)V1);
        val_grp_l = make_groups(vptr);
        r_grp_l = make_groups(std::vector<int>(sa.p_rowptr.begin() + orow0, sa.p_rowptr.begin() + orow1 + 1));
        // this rank's rows' entries in column order (what the restriction streams and what P^T A P is listed from)
        t_pos_l.assign(npl, 0);
        t_ptr_l.assign((size_t)nc + 1, 0);
        t_row_l.reserve(npl);
        t_col_l.reserve(npl);
        t_idx_l.reserve(npl);
        for (int a = 0; a < nc; ++a) {
          for (int t = sa.t_ptr[a]; t < sa.t_ptr[a + 1]; ++t) {
            const int i = sa.t_row[t];
            if (i < orow0 || i >= orow1) continue;
            const int e = sa.t_idx[t];
            t_pos_l[(size_t)e - E0] = (int)t_row_l.size();
            t_row_l.push_back(i);
            t_col_l.push_back(a);
            t_idx_l.push_back(e);
          }
          t_ptr_l[(size_t)a + 1] = (int)t_row_l.size();
        }
        t_grp_l = make_groups(t_ptr_l);
      }
      const std::vector<int>& val_grp = own ? val_grp_l : sa.val_grp;
      const std::vector<int>& r_grp = own ? r_grp_l : sa.r_grp;
      const std::vector<int>& t_grp = own ? t_grp_l : sa.t_grp;
      P.val.grp = dev_upload(m->pool, val_grp, s);
      P.val.ngrp = (int)val_grp.size() - 1;
      P.r_grp = dev_upload(m->pool, r_grp, s);
      P.r_ngrp = (int)r_grp.size() - 1;
      P.t_pos = own ? up_range(t_pos_l.data() - E0, E0, E1) : dev_upload(m->pool, sa.t_pos, s);
      P.t_row = dev_upload(m->pool, own ? t_row_l : sa.t_row, s);
      P.t_col = dev_upload(m->pool, own ? t_col_l : sa.t_col, s);
      P.r_n = P.t_n = (int)npl;
      {
        float* rb = dev_alloc<float>(m->pool, 9 * npl + 4);
        float* tb = dev_alloc<float>(m->pool, 9 * npl + 4);
        if (!rb || !tb) return fail("amg_create: out of device memory");
        P.r_blk = rb - 4 * E0;            // row order, addressed by global entry numbers
        P.r_blk8 = rb + 8 * npl - E0;
        P.t_blk = tb;                     // column order, addressed by the rank's own positions
        P.t_blk8 = tb + 8 * npl;
      }
      P.t_grp = dev_upload(m->pool, t_grp, s);
      P.t_ngrp = (int)t_grp.size() - 1;
      std::vector<int> t_long;   // (lives until the stream is synchronised below)
      for (size_t g = 0; g + 1 < t_grp.size(); ++g)
        if (t_grp[g + 1] - t_grp[g] > kLongColumn) {
          t_long.push_back(t_grp[g]);
          t_long.push_back(t_grp[g + 1]);
        }
      P.t_nlong = (int)t_long.size() / 2;
      P.t_long = P.t_nlong ? dev_upload(m->pool, t_long, s) : nullptr;
      if (P.t_nlong && !P.t_long) return fail("amg_create: out of device memory");
      if (P.t_nlong && hipStreamSynchronize(s) != hipSuccess) return fail("amg_create: upload failed");
      P.nap = sa.nap;
      size_t F0 = 0, F1 = (size_t)sa.nap;
      if (own) {
        F0 = (size_t)sa.ap_rowptr[orow0];
        F1 = (size_t)sa.ap_rowptr[orow1];
      }
      {
        double* ab = dev_alloc<double>(m->pool, 9 * (F1 - F0));
        if (!ab) return fail("amg_create: out of device memory");
        P.apblk = ab - 9 * F0;
      }
      int *d_ap_rowptr = nullptr, *d_ap_col = nullptr, *d_ap_row = nullptr, *d_t_ptr = nullptr, *d_t_idx = nullptr;
      if (sa.lists_on_device) {
        // the patterns go up (12 B per A P entry instead of 12 B per product); the lists are made below, once the
        // coarse structure is on the device too
        d_ap_rowptr = dev_upload(m->pool, sa.ap_rowptr, s);
        d_ap_col = up_range(sa.ap_col.data(), F0, F1);
        d_ap_row = up_range(sa.ap_row.data(), F0, F1);
        d_t_ptr = dev_upload(m->pool, own ? t_ptr_l : sa.t_ptr, s);
        d_t_idx = dev_upload(m->pool, own ? t_idx_l : sa.t_idx, s);
        if (!d_ap_rowptr || !d_ap_col || !d_ap_row || !d_t_ptr || !d_t_idx) return fail("amg_create: out of device memory");
        P.ap.n = (int)sa.n_ap_prod;
        P.rap.n = (int)sa.n_rap_prod;
      } else {
      P.ap.n = (int)sa.ap_a.size();
      P.ap.a = dev_upload(m->pool, sa.ap_a, s);
      P.ap.b = dev_upload(m->pool, sa.ap_b, s);
      P.ap.tgt = dev_upload(m->pool, sa.ap_tgt, s);
      P.ap.grp = dev_upload(m->pool, sa.ap_grp, s);
      P.ap.ngrp = (int)sa.ap_grp.size() - 1;
      P.rap.n = (int)sa.rap_a.size();
      P.rap.a = dev_upload(m->pool, sa.rap_a, s);
      P.rap.b = dev_upload(m->pool, sa.rap_b, s);
      P.rap.tgt = dev_upload(m->pool, sa.rap_tgt, s);
      P.rap.grp = dev_upload(m->pool, sa.rap_grp, s);
      P.rap.ngrp = (int)sa.rap_grp.size() - 1;
      }
      P.rap_mirror = dev_upload(m->pool, sa.rap_mirror, s);
      if (!P.rowptr || !P.row || !P.col || !P.blk || !P.val.a || !P.val.tgt || !P.val.grp || !P.r_grp || !P.t_pos || !P.t_row || !P.t_col || !P.t_grp ||
          (!sa.lists_on_device && (!P.ap.a || !P.ap.b || !P.ap.tgt || !P.ap.grp || !P.rap.a || !P.rap.b || !P.rap.tgt || !P.rap.grp)))
        return fail("amg_create: out of device memory");
      if (own && hipStreamSynchronize(s) != hipSuccess) return fail("amg_create: upload failed");   // (the rank-local host lists die with this scope)
      if (l == 0) m->level0_bytes += (long long)(72 * (size_t)P.np + 72 * npl + 72 * (F1 - F0) + 8 * (V1 - V0) + 12 * npl + 8 * (F1 - F0));
      l0_own = {own, (int)F0, (int)F1};
      l0_dev = {d_ap_rowptr, d_ap_col, d_ap_row, d_t_ptr, d_t_idx};
      // (Level 0 is folded only where it is itself launch-bound: P~ has the pattern of A P, twice the entries of P, and on
      // large graphs streaming it twice per cycle costs what the saved launch and pass bring -- C4: restriction + prolongation
      // 14 + 20 us with P~ against 11 + 9 us with P.  Multi-GPU runs keep level 0 unfolded as well.)
      if (m->cfg.fold && sa.lists_on_device && F1 > F0 && (l > 0 || (!halo && n <= m->cfg.fold0_rows))) {
        // ---- folded cycle: pattern bookkeeping of P~ (the pattern of A P) -- which entry of P sits at the same place, the
        // column order (a device radix sort of (column, row-major rank) keys instead of a host counting sort), its wave
        // groups and long columns -- and the two streamed fp32 copies
        const size_t nf = F1 - F0;
        FoldDev& Fd = L.F;
        Fd.f_lo = (int)F0;
        Fd.f_hi = (int)F1;
        Fd.row = d_ap_row;
        Fd.col = d_ap_col;
        int* a2p = dev_alloc<int>(m->pool, nf);
        int* stp = dev_alloc<int>(m->pool, nf);
        unsigned long long* keys = dev_alloc<unsigned long long>(m->pool, nf);
        unsigned long long* sorted = dev_alloc<unsigned long long>(m->pool, nf);
        int bits = 33;
        while (bits < 64 && (1ull << (bits - 32)) <= (unsigned long long)nc) ++bits;
        const size_t tmp_bytes = sort_u64_temp_bytes(nf, bits);
        void* tmp = tmp_bytes ? m->pool->take(tmp_bytes) : nullptr;
        int* st_row = dev_alloc<int>(m->pool, nf);
        int* st_col = dev_alloc<int>(m->pool, nf);
        int* st_ptr = dev_alloc<int>(m->pool, (size_t)nc + 1);
        constexpr int kLongCap = 4096;
        int* d_cnt = dev_alloc<int>(m->pool, 1);
        int* ranges = dev_alloc<int>(m->pool, 2 * (size_t)kLongCap);
        float* sb = dev_alloc<float>(m->pool, 9 * nf + 4);
        float* tb = dev_alloc<float>(m->pool, 9 * nf + 4);
        if (!a2p || !stp || !keys || !sorted || !tmp || !st_row || !st_col || !st_ptr || !d_cnt || !ranges || !sb || !tb)
          return fail("amg_create: out of device memory");
        Fd.ap2p = a2p - F0;
        Fd.st_pos = stp - F0;
        SGO_LAUNCH(k_fold_match, dim3(grid_for((long long)nf, kBlock)), dim3(kBlock), 0, s, Fd, (const int*)P.rowptr, (const int*)P.col, keys);
        if (!sort_u64(tmp, tmp_bytes, (const uint64_t*)keys, (uint64_t*)sorted, nf, bits, s)) return fail("amg_create: device sort failed");
        SGO_LAUNCH(k_fold_unpack, dim3(grid_for((long long)nf, kBlock)), dim3(kBlock), 0, s, Fd, (const unsigned long long*)sorted, st_row, st_col);
        SGO_LAUNCH(k_fold_colptr, dim3(grid_for((long long)nc + 1, kBlock)), dim3(kBlock), 0, s, (const unsigned long long*)sorted, (int)nf, nc, st_ptr);
        int st_ngrp = 0;
        int* st_grp = dev_make_groups(s, m->pool, st_ptr, nc, (int)nf, &st_ngrp);
        if (!st_grp) return fail("amg_create: out of device memory");
        hipMemsetAsync(d_cnt, 0, sizeof(int), s);
        SGO_LAUNCH(k_fold_long, dim3(grid_for((long long)st_ngrp, kBlock)), dim3(kBlock), 0, s, (const int*)st_grp, st_ngrp, d_cnt, ranges, kLongCap);
        int nlong = 0;
        if (hipMemcpyAsync(&nlong, d_cnt, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
          return fail("amg_create: folded-transfer kernels failed");
        std::vector<int> s_grp = make_groups(std::vector<int>(sa.ap_rowptr.begin() + orow0, sa.ap_rowptr.begin() + orow1 + 1));
        PDev& PS = L.PS;
        PS = PDev();
        PS.np = (int)nf;
        PS.stream_nt = nf >= 200000 ? 1 : 0;
        PS.row = d_ap_row;
        PS.col = d_ap_col;
        PS.r_n = PS.t_n = (int)nf;
        PS.r_blk = sb - 4 * F0;
        PS.r_blk8 = sb + 8 * nf - F0;
        PS.t_blk = tb;
        PS.t_blk8 = tb + 8 * nf;
        PS.r_grp = dev_upload(m->pool, s_grp, s);
        PS.r_ngrp = (int)s_grp.size() - 1;
        PS.t_row = st_row;
        PS.t_col = st_col;
        PS.t_grp = st_grp;
        PS.t_ngrp = st_ngrp;
        PS.t_long = ranges;
        PS.t_nlong = nlong;
        if (!PS.r_grp) return fail("amg_create: out of device memory");
        if (l > 0) {
          // levels >= 1: the slots of A and the entries of P~ of every row as one list (k_up_fold)
          std::vector<int> u_ptr((size_t)n + 1, 0), u_row, u_idx, u_col;
          for (int i = 0; i < n; ++i) u_ptr[(size_t)i + 1] = u_ptr[i] + (H.rowptr[i + 1] - H.rowptr[i]) + (sa.ap_rowptr[i + 1] - sa.ap_rowptr[i]);
          u_row.resize((size_t)u_ptr[n]);
          u_idx.resize((size_t)u_ptr[n]);
          u_col.resize((size_t)u_ptr[n]);
          for (int i = 0; i < n; ++i) {
            int q = u_ptr[i];
            for (int k = H.rowptr[i]; k < H.rowptr[i + 1]; ++k, ++q) {
              u_row[q] = i;
              u_idx[q] = k;
              u_col[q] = H.col[k];
            }
            for (int f = sa.ap_rowptr[i]; f < sa.ap_rowptr[i + 1]; ++f, ++q) {
              u_row[q] = i;
              u_idx[q] = ~f;
              u_col[q] = sa.ap_col[f];
            }
          }
          std::vector<int> u_grp = make_groups(u_ptr);
          UpDev& U = L.U;
          U.n = u_ptr[n];
          U.row = dev_upload(m->pool, u_row, s);
          U.idx = dev_upload(m->pool, u_idx, s);
          U.col = dev_upload(m->pool, u_col, s);
          U.grp = dev_upload(m->pool, u_grp, s);
          U.ngrp = (int)u_grp.size() - 1;
          if (!U.row || !U.idx || !U.col || !U.grp) return fail("amg_create: out of device memory");
        }
        if (hipStreamSynchronize(s) != hipSuccess) return fail("amg_create: upload failed");   // (host lists of this scope)
        L.fold = nlong <= kLongCap;
        if (l == 0) m->level0_bytes += (long long)(72 * nf + 28 * nf);
      }
      if (l == 0 && halo) {
        // the entries of P in every rank's boundary rows: what A P of a neighbour's rows gathers from this rank's P
        std::vector<std::vector<int>> pe((size_t)halo->G);
        int pemax = 1;
        for (int q = 0; q < halo->G; ++q) {
          for (int t = 0; t < halo->bmax; ++t) {
            const int r = halo->bnd_host[(size_t)q * halo->bmax + t];
            if (r < 0) break;
            for (int e = sa.p_rowptr[r]; e < sa.p_rowptr[r + 1]; ++e) pe[q].push_back(e);
          }
          pemax = std::max(pemax, (int)pe[q].size());
        }
        std::vector<int> flat((size_t)halo->G * pemax, -1);
        for (int q = 0; q < halo->G; ++q) std::copy(pe[q].begin(), pe[q].end(), flat.begin() + (size_t)q * pemax);
        int* d_pe = dev_upload(m->pool, flat, s);
        if (!d_pe || hipStreamSynchronize(s) != hipSuccess) return fail("amg_create: out of device memory");
        halo->dev->pemax = pemax;
        halo->dev->pent = d_pe;
        if (halo->reserve && !halo->reserve(halo->user, (size_t)kHaloScalars + 9 * (size_t)pemax)) return fail("amg_create: out of device memory (exchange buffers)");
      }
      std::snprintf(line, sizeof line, "(P %d%s, AP %d blocks; %d + %d products) ", P.np, sa.filtered ? " filtered" : "", P.nap, P.ap.n, P.rap.n);
      m->desc += line;
    } else {
      L.gal.n = H.nslot;
      L.gal.src = dev_upload(m->pool, order, s);
      L.gal.tgt = dev_upload(m->pool, tgt, s);
      L.gal.grp = dev_upload(m->pool, grp_g, s);
      L.gal.ngrp = (int)grp_g.size() - 1;
      if (!L.gal.src || !L.gal.tgt || !L.gal.grp) return fail("amg_create: out of device memory");
    }
    AmgLevel C;
    C.A.n = nc;
    C.A.nslot = Hc.nslot;
    C.A.ngrp = (int)grp_c.size() - 1;
    C.A.row = dev_upload(m->pool, Hc.row, s);
    C.A.col = dev_upload(m->pool, Hc.col, s);
    C.A.grp = dev_upload(m->pool, grp_c, s);
    C.A.rowptr = dev_upload(m->pool, Hc.rowptr, s);
    C.A.blk = dev_alloc<double>(m->pool, 9 * (size_t)Hc.nslot);
    C.A.dinv = dev_alloc<double>(m->pool, 6 * (size_t)nc);
    if (!L.agg || !L.mem_ptr || !L.mem || !L.mem_grp || !L.d || !C.A.row || !C.A.col ||
        !C.A.grp || !C.A.rowptr || !C.A.blk || !C.A.dinv)
      return fail("amg_create: out of device memory");
    const double t_up0 = ms_since(tU);
    if (smooth && sa.lists_on_device) {
      // product lists from the patterns: count per target, prefix sum, fill, wave groups (A P, then P^T A P)
      PDev& P = L.P;
      ApPattern ap;
      ap.nap = P.nap;
      ap.ap_row = l0_dev.ap_row;
      ap.ap_col = l0_dev.ap_col;
      ap.ap_rowptr = l0_dev.ap_rowptr;
      ap.f_lo = l0_own.own ? l0_own.F0 : 0;
      ap.nap = l0_own.own ? l0_own.F1 : P.nap;
      const int seg_lo[2] = {ap.f_lo, 0};
      const int nseg[2] = {ap.nap - ap.f_lo, Hc.nslot};
      const int nprod[2] = {P.ap.n, P.rap.n};
      ProdMap* maps[2] = {&P.ap, &P.rap};
      for (int w = 0; w < 2; ++w) {
        int* ptr0 = dev_alloc<int>(m->pool, (size_t)nseg[w] + 1);
        int* sums = dev_alloc<int>(m->pool, (size_t)nseg[w] / kScanChunk + 3);
        if (!ptr0 || !sums) return fail("amg_create: out of device memory");
        int* ptr = ptr0 - seg_lo[w];   // addressed by global target numbers
        int *la = nullptr, *lb = nullptr, *lt = nullptr;
        const dim3 grid(grid_for(8LL * nseg[w], kBlock)), block(kBlock);   // eight lanes per target
        if (w == 0)
          SGO_LAUNCH((k_ap_list<false>), grid, block, 0, s, ap, (const int*)L.A.rowptr, (const int*)L.A.col, (const int*)P.rowptr,
                     (const int*)P.col, ptr, la, lb, lt);
        else
          SGO_LAUNCH((k_rap_list<false>), grid, block, 0, s, Hc.nslot, (const int*)C.A.row, (const int*)C.A.col,
                     (const int*)l0_dev.t_ptr, (const int*)P.t_row, (const int*)l0_dev.t_idx, ap, ptr, la, lb, lt);
        dev_scan_exclusive(s, ptr0, nseg[w], sums);
        int ngrp = 0, total = -1;
        if (hipMemcpyAsync(&total, ptr0 + nseg[w], sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
          return fail("amg_create: product-list kernels failed");
        if (!l0_own.own && total != nprod[w])
          return fail("amg_create: internal error (device product lists: " + std::to_string(total) + " products, the host counted " +
                      std::to_string(nprod[w]) + ")");
        if (total < 0 || total > nprod[w]) return fail("amg_create: internal error (device product lists)");
        la = dev_alloc<int>(m->pool, (size_t)std::max(total, 1));
        lb = dev_alloc<int>(m->pool, (size_t)std::max(total, 1));
        lt = dev_alloc<int>(m->pool, (size_t)std::max(total, 1));
        if (!la || !lb || !lt) return fail("amg_create: out of device memory");
        if (w == 0)
          SGO_LAUNCH((k_ap_list<true>), grid, block, 0, s, ap, (const int*)L.A.rowptr, (const int*)L.A.col, (const int*)P.rowptr,
                     (const int*)P.col, ptr, la, lb, lt);
        else
          SGO_LAUNCH((k_rap_list<true>), grid, block, 0, s, Hc.nslot, (const int*)C.A.row, (const int*)C.A.col,
                     (const int*)l0_dev.t_ptr, (const int*)P.t_row, (const int*)l0_dev.t_idx, ap, ptr, la, lb, lt);
        int* grp = dev_make_groups(s, m->pool, ptr0, nseg[w], total, &ngrp);
        if (!grp) return fail("amg_create: out of device memory");
        maps[w]->n = total;
        maps[w]->a = la;
        maps[w]->b = lb;
        maps[w]->tgt = lt;
        maps[w]->grp = grp;
        maps[w]->ngrp = ngrp;
        if (l == 0) m->level0_bytes += 12LL * total;
      }
    }
    if (hipStreamSynchronize(s) != hipSuccess) return fail("amg_create: upload failed");  // host vectors die below
    const double t_up = ms_since(tU);
    m->lv.push_back(C);  // invalidates L
    // values of level l+1 (needed for the next level's strengths): positions, centres, Galerkin
    {
      AmgLevel& Lr = m->lv[l];
      AmgLevel& Cr = m->lv[l + 1];
      Cr.pos = dev_alloc<double>(m->pool, 2 * (size_t)nc);
      if (!Cr.pos) return fail("amg_create: out of device memory");
      if (l == 0)
        SGO_LAUNCH(k_positions0, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, d_free_id, d_poses, Lr.pos);
      SGO_LAUNCH(k_centres, dim3(grid_for(nc, kWavesPerBlock)), dim3(kBlock), 0, s, nc, Lr.mem_ptr, Lr.mem, Lr.pos, Cr.pos, Lr.d);
      launch_coarse_operator(m, s, Lr, Cr, l == 0);
      if (l + 1 < m->cfg.max_levels) SGO_LAUNCH(k_level_dinv, dim3(grid_for(Cr.A.n, kBlock)), dim3(kBlock), 0, s, Cr.A);
      if (hipStreamSynchronize(s) != hipSuccess) return fail("amg_create: Galerkin kernel failed");
      if (std::getenv("SGO_VERBOSE") && (n > 20000 || std::atoi(std::getenv("SGO_VERBOSE")) > 1))
        std::fprintf(stderr, "[sgo] amg level %d: alloc + upload %.1f ms (of which product lists on the device %.1f), first values %.1f ms\n", l, t_up,
                     t_up - t_up0, ms_since(tU) - t_up);
    }
    Hown = std::move(Hc);
    Hown.visit = std::move(visit_c);
    Hp = &Hown;
  }
  const int last = (int)m->lv.size() - 1;
  // last == 0: the whole graph is at most coarsest_nodes large (or cannot be coarsened) and is
  // "solved" by the dense inverse directly -- the preconditioner is then exact (1-2 PCG iterations)
  if (last == 0 && m->lv[0].A.n > 1024) return fail("amg_create: graph not coarsenable; use the block-Jacobi solver");
  m->N = 3 * m->lv[last].A.n;
  m->Np = (m->N + kGjB - 1) / kGjB * kGjB;
  if (m->N > 3072) return fail("amg_create: coarsest level too large (" + std::to_string(m->N) + " unknowns)");
  m->inv0 = dev_alloc<double>(m->pool, (size_t)m->Np * m->Np);
  m->inv1 = dev_alloc<double>(m->pool, (size_t)m->Np * m->Np);
  m->gjP[0] = dev_alloc<double>(m->pool, kGjB * kGjB);
  m->gjP[1] = dev_alloc<double>(m->pool, kGjB * kGjB);
  m->inv = ((m->Np / kGjB) & 1) ? m->inv1 : m->inv0;
  m->d_fail = dev_alloc<int>(m->pool, 1);
  if (!m->inv0 || !m->inv1 || !m->d_fail || !m->gjP[0] || !m->gjP[1]) return fail("amg_create: out of device memory");
  hipMemsetAsync(m->d_fail, 0, sizeof(int), s);
  // Two sweeps per coarse level (amg_effective_config) pay while the coarse levels are small next to level 0.  A hierarchy
  // whose level 1 holds more than a quarter of level 0's blocks -- filtered transfers along the trajectory: aggregates of three
  // poses whose rows keep all their closures, C4 from a dead-reckoned start: 1.15 M of 2.1 M -- pays four bandwidth-bound
  // passes per level for them: one sweep there (measured: 23.3 -> 20.9 ms per Gauss-Newton iteration at that start; C4's usual
  // hierarchy, level 1 at 6 %, keeps two: 4.06 against 4.45 ms with one).
  if (last >= 1 && !std::getenv("SGO_AMG_NU") && 4LL * m->lv[1].A.nslot > (long long)m->lv[0].A.nslot) m->cfg.nu_coarse = 1;
  m->cfg.keep_agg = nullptr;   // (the caller's object: only read during this set-up)
  std::snprintf(line, sizeof line, "coarsest dense N=%d; theta=%.3g omega=%.2f nu=%d", m->N,
                m->cfg.theta * m->cfg.theta_scale, m->cfg.omega, m->cfg.nu_coarse);
  m->desc += line;
  return m;
}

}  // namespace sgo
