// sgo_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Gauss-Newton inner loop.
//
// What each kernel replaces in the reference's g2o path:
//   k_chi2        computeActiveErrors + activeChi2/activeRobustChi2      (slc.cpp:288)
//   k_linearize   EdgeSE2::computeError + linearizeOplus +
//                 BaseBinaryEdge::constructQuadraticForm (+ RobustKernelDCS) over all edges
//                 = BlockSolver<3,3>::buildSystem                         (graphs.cpp:18-20)
//   k_finalize    block-diagonal inverse (preconditioner) + PCG start vectors
//   k_spmv        the Hessian product inside the linear solve that replaces
//                 LinearSolverEigen::solve                                (graphs.cpp:19)
//   k_update_*    PCG vector recurrences
//   k_pose_update SparseOptimizer::update -> VertexSE2::oplusImpl
//
// All of it is HBM-/cache-bound fp64 gather + stream work on 3x3 blocks: no MFMA (a 3x3 block is
// not a dense contraction).  Design rules used (cdna_hip_programming.md G2, G11, G12, G13,
// Appendix B scatter/gather): SoA slot arrays (blocks in component pairs) so every wave load is
// one contiguous 512-B or 1-KiB segment; per-row sums by a wavefront segmented scan over
// row-aligned slot groups (no atomics, bitwise reproducible); grids capped at 2048 blocks and mapped so that each XCD walks one
// contiguous band of rows (its L2 then holds that band's vector entries).
#include <algorithm>
#include <cstdlib>

#include "sgo_comm.h"
#include "sgo_device.h"
#include "sgo_internal.h"

namespace sgo {

thread_local LaunchEvents tl_launch_ev;

const char* const kKernelNames[K_COUNT] = {
    "k_chi2",        "k_reduce2",  "k_linearize",   "k_finalize",   "k_init_scalars", "k_spmv<0>",  "k_spmv<1>",
    "k_spmv<2>", "k_spmv<3>", "k_spmv<4>", "k_spmv<5>", "k_spmv<6>",      "k_alpha",       "k_update_xr",  "k_beta",         "k_update_p", "k_dot",
    "k_pose_update", "k_positions0", "k_centres",   "k_galerkin",   "k_level_dinv",   "k_restrict", "k_prolong_add",
    "k_gj_step (dense inverse, all block steps)", "k_dense_apply", "k_p_values", "k_block_products<1, 0, 0>",
    "k_block_products<0, 1, 1>", "k_restrict_p", "k_prolong_p", "k_spmv<7>",
    "k_spmv0<0>", "k_spmv0<1>", "k_spmv0<2>", "k_spmv0t<0, 1024, false>", "k_spmv0t<1, 1024, false>", "k_spmv0t<2, 1024, false>",
    "k_direct", "k_spmv0t<1, 1024, true>", "k_spmv0t<2, 1024, true>", "k_restrict_p @level0", "k_prolong_p @level0", "k_p_values @level0", "k_block_products<1, 0, 0> @level0",
    "k_block_products<0, 1, 1> @level0", "k_galerkin @level0", "k_restrict @level0", "k_prolong_add @level0",
    "k_ptilde_values", "k_ptilde_values @level0", "k_up_fold", "k_prolong_fold @level0", "k_jacobi0_restrict @level0",
    "k_mf_edges + k_mf_factor + k_mf_solve + k_mf_update (multifrontal optimize)"};

namespace {

struct EdgeLin {
  double e[3];
  double e2, rho0, rho1;
};

// ---------------------------------------------------------------------------- k_chi2
__device__ __forceinline__ void chi2_range(const EdgeListDev& el, int e0, int e1, const double* __restrict__ poses,
                                           double* __restrict__ e2_out, double (&acc)[2]) {
  const int E = el.E;
  for (int k = e0 + blockIdx.x * kBlock + threadIdx.x; k < e1; k += gridDim.x * kBlock) {
    const int vi = el.vi[k], vj = el.vj[k];
    const double xi = poses[3 * (size_t)vi], yi = poses[3 * (size_t)vi + 1], ti = poses[3 * (size_t)vi + 2];
    const double xj = poses[3 * (size_t)vj], yj = poses[3 * (size_t)vj + 1], tj = poses[3 * (size_t)vj + 2];
    const double zx = el.zinv[k], zy = el.zinv[(size_t)E + k], zt = el.zinv[2 * (size_t)E + k];
    double sz, cz;
    sincos(zt, &sz, &cz);
    double e[3];
    edge_error(xi, yi, ti, xj, yj, tj, zx, zy, zt, sz, cz, e);
    const double o00 = el.info[k], o01 = el.info[(size_t)E + k], o02 = el.info[2 * (size_t)E + k];
    const double o11 = el.info[3 * (size_t)E + k], o12 = el.info[4 * (size_t)E + k], o22 = el.info[5 * (size_t)E + k];
    const double oe0 = o00 * e[0] + o01 * e[1] + o02 * e[2];
    const double oe1 = o01 * e[0] + o11 * e[1] + o12 * e[2];
    const double oe2 = o02 * e[0] + o12 * e[1] + o22 * e[2];
    const double e2 = e[0] * oe0 + e[1] * oe1 + e[2] * oe2;
    double r0, r1;
    dcs(e2, el.phi[k], &r0, &r1);
    if (e2_out) e2_out[k] = e2;
    acc[0] += e2;
    acc[1] += r0;
  }
}
// edges [e0, e1) of el, then the first el2.cnt edges of el2 (an empty list when there is no overlay)
__global__ __launch_bounds__(kBlock) void k_chi2(EdgeListDev el, int e0, int e1, EdgeListDev el2, const double* __restrict__ poses,
                                                 double* __restrict__ e2_out, double* __restrict__ partials) {
  double acc[2] = {0.0, 0.0};
  chi2_range(el, e0, e1, poses, e2_out, acc);
  if (el2.cnt > 0) chi2_range(el2, 0, el2.cnt, poses, e2_out ? e2_out + el.E : nullptr, acc);
  block_sum_store<2>(acc, partials, kMaxPartials);
}

// out2[i] = sum(partials[i][0..nparts))
__global__ __launch_bounds__(kBlock) void k_reduce2(const double* __restrict__ partials, int nparts,
                                                    double* __restrict__ out2) {
  const double a = block_reduce_parts(partials, nparts);
  const double b = block_reduce_parts(partials + kMaxPartials, nparts);
  if (threadIdx.x == 0) {
    out2[0] = a;
    out2[1] = b;
  }
}

// ---------------------------------------------------------------------------- set-up kernels
// EdgeSE2::setMeasurement caches the inverse measurement (computed once per set_graph): zinv = Z^-1 in SoA,
// and the information's upper triangle from the caller's AoS rows to SoA.
__global__ __launch_bounds__(kBlock) void k_edge_prepare(int E, const double* __restrict__ meas, const double* __restrict__ info,
                                                         double* __restrict__ zinv, double* __restrict__ info_soa, size_t stride, size_t at) {
  for (int e = blockIdx.x * kBlock + threadIdx.x; e < E; e += gridDim.x * kBlock) {
    const double zx = meas[3 * (size_t)e], zy = meas[3 * (size_t)e + 1], zt = meas[3 * (size_t)e + 2];
    const double th = norm_theta(-zt);
    double sn, cs;
    sincos(th, &sn, &cs);
    const size_t o = at + (size_t)e;
    zinv[o] = cs * (-zx) - sn * (-zy);
    zinv[stride + o] = sn * (-zx) + cs * (-zy);
    zinv[2 * stride + o] = th;
#pragma unroll
    for (int q = 0; q < 6; ++q) info_soa[q * stride + o] = info[6 * (size_t)e + q];
  }
}
// The per-slot operand arrays of k_linearize (coalesced per slot) from the per-edge arrays: slot k came from
// edge eidx[k]; the host only lists that index and the side (4 + 1 B per slot instead of 96 B).
__global__ __launch_bounds__(kBlock) void k_slot_expand(int k0, int k1, const int* __restrict__ eidx, EdgeListDev el, EdgeSlotsDev es) {
  const size_t ns = (size_t)es.stride, E = (size_t)el.E;
  for (int k = k0 + blockIdx.x * kBlock + threadIdx.x; k < k1; k += gridDim.x * kBlock) {
    const size_t e = (size_t)eidx[k];
    es.vi[k] = el.vi[e];
    es.vj[k] = el.vj[e];
#pragma unroll
    for (int q = 0; q < 3; ++q) es.zinv[q * ns + k] = el.zinv[q * E + e];
#pragma unroll
    for (int q = 0; q < 6; ++q) es.info[q * ns + k] = el.info[q * E + e];
    es.phi[k] = el.phi[e];
  }
}

// ---------------------------------------------------------------------------- early strength weights
// The multigrid aggregation needs the Frobenius norms of the Hessian blocks at the initial poses and nothing else of
// the matrix.  This kernel produces them straight from the edge list and the per-row slot lists -- before the level-0
// storage (tiles, slot types) exists -- so that the host's aggregation and symbolic phase can run on a helper thread
// while the storage is still being laid out (sgo_structure.cpp, build_structure).  One thread per row walks the row's
// compact slots (edge, side) and evaluates each edge as k_linearize does: w of the row's logical slots -- the diagonal
// slot first (norm of the summed R^T Ow R), then ||R^T Ow C||_F for every slot whose column is free.
__global__ __launch_bounds__(kBlock) void k_row_strength(int n, const int* __restrict__ rowptr, const int* __restrict__ eidx,
                                                         const unsigned char* __restrict__ flags, const int* __restrict__ hrowptr,
                                                         EdgeListDev el, const double* __restrict__ poses, double* __restrict__ w) {
  const size_t E = (size_t)el.E;
  for (int r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) {
    double d[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int q = hrowptr[r] + 1;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      const int e = eidx[k], fl = flags[k];
      const int vi = el.vi[e], vj = el.vj[e];
      const double xi = poses[3 * (size_t)vi], yi = poses[3 * (size_t)vi + 1], ti = poses[3 * (size_t)vi + 2];
      const double xj = poses[3 * (size_t)vj], yj = poses[3 * (size_t)vj + 1], tj = poses[3 * (size_t)vj + 2];
      const double zx = el.zinv[e], zy = el.zinv[E + e], zt = el.zinv[2 * E + e];
      double sz, cz;
      sincos(zt, &sz, &cz);
      double er[3];
      edge_error(xi, yi, ti, xj, yj, tj, zx, zy, zt, sz, cz, er);
      double O[3][3];
      O[0][0] = el.info[e]; O[0][1] = O[1][0] = el.info[E + e]; O[0][2] = O[2][0] = el.info[2 * E + e];
      O[1][1] = el.info[3 * E + e]; O[1][2] = O[2][1] = el.info[4 * E + e]; O[2][2] = el.info[5 * E + e];
      double e2 = 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) e2 += er[a] * O[a][b] * er[b];
      double r0_, wt;
      dcs(e2, el.phi[e], &r0_, &wt);
      double si, ci;
      sincos(ti, &si, &ci);
      const double ddx = xj - xi, ddy = yj - yi;
      const double a02 = -si * ddx + ci * ddy, a12 = -ci * ddx - si * ddy;
      double A[3][3], B[3][3];
      A[0][0] = cz * (-ci) - sz * si; A[0][1] = cz * (-si) - sz * (-ci); A[0][2] = cz * a02 - sz * a12;
      A[1][0] = sz * (-ci) + cz * si; A[1][1] = sz * (-si) + cz * (-ci); A[1][2] = sz * a02 + cz * a12;
      A[2][0] = 0.0; A[2][1] = 0.0; A[2][2] = -1.0;
      B[0][0] = cz * ci - sz * (-si); B[0][1] = cz * si - sz * ci; B[0][2] = 0.0;
      B[1][0] = sz * ci + cz * (-si); B[1][1] = sz * si + cz * ci; B[1][2] = 0.0;
      B[2][0] = 0.0; B[2][1] = 0.0; B[2][2] = 1.0;
      const bool dir = fl & kSlotDir;   // the row's Jacobian is B
      double T[3][3];                   // Ow R
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const double r0b = dir ? B[0][b] : A[0][b], r1b = dir ? B[1][b] : A[1][b], r2b = dir ? B[2][b] : A[2][b];
          T[a][b] = wt * (O[a][0] * r0b + O[a][1] * r1b + O[a][2] * r2b);
        }
      double off = 0.0;
      int t = 0;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          // (R^T Ow C)[b][a] = sum_m T[m][b] C[m][a]  (T^T C: Ow is symmetric)
          const double c0 = dir ? A[0][a] : B[0][a], c1 = dir ? A[1][a] : B[1][a], c2 = dir ? A[2][a] : B[2][a];
          const double oc = T[0][b] * c0 + T[1][b] * c1 + T[2][b] * c2;
          off += oc * oc;
          if (b >= a) {
            const double ra0 = dir ? B[0][a] : A[0][a], ra1 = dir ? B[1][a] : A[1][a], ra2 = dir ? B[2][a] : A[2][a];
            d[t++] += ra0 * T[0][b] + ra1 * T[1][b] + ra2 * T[2][b];
          }
        }
      if (!(fl & kSlotFixedCol)) w[q++] = sqrt(off);
    }
    w[hrowptr[r]] = sqrt(d[0] * d[0] + d[3] * d[3] + d[5] * d[5] + 2.0 * (d[1] * d[1] + d[2] * d[2] + d[4] * d[4]));
  }
}

// ---------------------------------------------------------------------------- k_linearize
// One lane per compact slot of the level-0 matrix (Sym0Dev).  For a slot of row r that came from edge
// (i,j): row Jacobian Jr = A (dir 0) or B (dir 1), column Jacobian Jc the other one; the lane
// segment-sums Jr^T Ow Jr (6 unique) and -Jr^T Ow e (3) over the row -- the row's last lane stores them
// to dgb[r][0..8] -- and, when the slot OWNS the block, writes the off-diagonal block Jr^T Ow Jc into the
// symmetric storage (the transposed slot of the other endpoint's row evaluates the same edge for its
// own row sums and writes nothing: recompute instead of scatter).
__global__ __launch_bounds__(kBlock) void k_linearize(Sym0Dev A, int g0, int g1, EdgeSlotsDev es,
                                                      const double* __restrict__ poses, double* __restrict__ dgb) {
  const int lane = threadIdx.x & 63;
  const size_t ns = (size_t)es.stride, nu = (size_t)A.nus;
  int g, gend, gstride;
  group_walk(g1 - g0, &g, &gend, &gstride);   // this rank's band of row groups [g0, g1)
  g += g0;
  gend += g0;
  for (; g < gend; g += gstride) {
    const int gb = A.grp[g], ge = A.grp[g + 1], r0 = A.grow[g];
    int ob = A.gown[g];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int row = -1 - lane;  // inactive lanes: unique negative keys (never merge)
    for (int kb = gb; kb < ge; kb += 64) {
      const int k = kb + lane;
      const bool active = k < ge;
      const int m = active ? (int)A.meta[k] : (kSlotNoBlock << 6);
      const bool owned = active && (m >> 6) == kSlotOwned;
      const unsigned long long omask = __ballot(owned);
      const int idx = ob + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(omask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)omask, 0u));
      ob += __popcll(omask);
      if (!active) continue;
      row = r0 + (m & 63);
      const int fl = es.flags[k];
      if (fl & kSlotNoEdge) continue;
      const int vi = es.vi[k], vj = es.vj[k];
      const double xi = poses[3 * (size_t)vi], yi = poses[3 * (size_t)vi + 1], ti = poses[3 * (size_t)vi + 2];
      const double xj = poses[3 * (size_t)vj], yj = poses[3 * (size_t)vj + 1], tj = poses[3 * (size_t)vj + 2];
      const double zx = es.zinv[k], zy = es.zinv[ns + k], zt = es.zinv[2 * ns + k];
      double sz, cz;
      sincos(zt, &sz, &cz);
      double e[3];
      edge_error(xi, yi, ti, xj, yj, tj, zx, zy, zt, sz, cz, e);
      const double o00 = es.info[k], o01 = es.info[ns + k], o02 = es.info[2 * ns + k];
      const double o11 = es.info[3 * ns + k], o12 = es.info[4 * ns + k], o22 = es.info[5 * ns + k];
      double oe0 = o00 * e[0] + o01 * e[1] + o02 * e[2];
      double oe1 = o01 * e[0] + o11 * e[1] + o12 * e[2];
      double oe2 = o02 * e[0] + o12 * e[1] + o22 * e[2];
      const double e2 = e[0] * oe0 + e[1] * oe1 + e[2] * oe2;
      double r0_, w;
      dcs(e2, es.phi[k], &r0_, &w);
      // robustInformation: Ow = rho1 * Omega ; omega_r scaled by rho1
      const double w00 = w * o00, w01 = w * o01, w02 = w * o02, w11 = w * o11, w12 = w * o12, w22 = w * o22;
      oe0 *= w; oe1 *= w; oe2 *= w;
      // EdgeSE2::linearizeOplus: A = Rz a, B = Rz b  (third rows (0,0,-1) and (0,0,1))
      double si, ci;
      sincos(ti, &si, &ci);
      const double ddx = xj - xi, ddy = yj - yi;
      const double a02 = -si * ddx + ci * ddy, a12 = -ci * ddx - si * ddy;
      const double A00 = cz * (-ci) - sz * si, A01 = cz * (-si) - sz * (-ci), A02 = cz * a02 - sz * a12;
      const double A10 = sz * (-ci) + cz * si, A11 = sz * (-si) + cz * (-ci), A12 = sz * a02 + cz * a12;
      const double B00 = cz * ci - sz * (-si), B01 = cz * si - sz * ci;
      const double B10 = sz * ci + cz * (-si), B11 = sz * si + cz * ci;
      const bool dir = fl & kSlotDir;
      // row / column Jacobians (row-major), third row is (0,0,s3)
      const double R00 = dir ? B00 : A00, R01 = dir ? B01 : A01, R02 = dir ? 0.0 : A02;
      const double R10 = dir ? B10 : A10, R11 = dir ? B11 : A11, R12 = dir ? 0.0 : A12;
      const double R22 = dir ? 1.0 : -1.0;
      // T = Ow * R  (3x3), R has zero entries (2,0),(2,1)
      const double T00 = w00 * R00 + w01 * R10, T01 = w00 * R01 + w01 * R11, T02 = w00 * R02 + w01 * R12 + w02 * R22;
      const double T10 = w01 * R00 + w11 * R10, T11 = w01 * R01 + w11 * R11, T12 = w01 * R02 + w11 * R12 + w12 * R22;
      const double T20 = w02 * R00 + w12 * R10, T21 = w02 * R01 + w12 * R11, T22 = w02 * R02 + w12 * R12 + w22 * R22;
      // D = R^T T (symmetric): d00 d01 d02 d11 d12 d22
      acc[0] += R00 * T00 + R10 * T10;
      acc[1] += R00 * T01 + R10 * T11;
      acc[2] += R00 * T02 + R10 * T12;
      acc[3] += R01 * T01 + R11 * T11;
      acc[4] += R01 * T02 + R11 * T12;
      acc[5] += R02 * T02 + R12 * T12 + R22 * T22;
      // b = -R^T (Ow e)
      acc[6] -= R00 * oe0 + R10 * oe1;
      acc[7] -= R01 * oe0 + R11 * oe1;
      acc[8] -= R02 * oe0 + R12 * oe1 + R22 * oe2;
      if (owned) {
        // off-diagonal block  R^T Ow C = T^T C   (T^T because Ow is symmetric: R^T Ow = (Ow R)^T)
        const double C00 = dir ? A00 : B00, C01 = dir ? A01 : B01, C02 = dir ? A02 : 0.0;
        const double C10 = dir ? A10 : B10, C11 = dir ? A11 : B11, C12 = dir ? A12 : 0.0;
        const double C22 = dir ? -1.0 : 1.0;
        double2* __restrict__ bp = reinterpret_cast<double2*>(A.ublk);
        bp[idx] = make_double2(T00 * C00 + T10 * C10, T00 * C01 + T10 * C11);
        bp[nu + idx] = make_double2(T00 * C02 + T10 * C12 + T20 * C22, T01 * C00 + T11 * C10);
        bp[2 * nu + idx] = make_double2(T01 * C01 + T11 * C11, T01 * C02 + T11 * C12 + T21 * C22);
        bp[3 * nu + idx] = make_double2(T02 * C00 + T12 * C10, T02 * C01 + T12 * C11);
        const double b8 = T02 * C02 + T12 * C12 + T22 * C22;
        A.ublk8[idx] = b8;
        if (A.fblk) {   // fp32 copy for the preconditioner's passes
          float4* __restrict__ fp = reinterpret_cast<float4*>(A.fblk);
          fp[idx] = make_float4((float)(T00 * C00 + T10 * C10), (float)(T00 * C01 + T10 * C11),
                                (float)(T00 * C02 + T10 * C12 + T20 * C22), (float)(T01 * C00 + T11 * C10));
          fp[nu + idx] = make_float4((float)(T01 * C01 + T11 * C11), (float)(T01 * C02 + T11 * C12 + T21 * C22),
                                     (float)(T02 * C00 + T12 * C10), (float)(T02 * C01 + T12 * C11));
          A.fblk8[idx] = (float)b8;
        }
      }
    }
    seg_scan<9>(row, acc, lane);
    const int rn = next_lane_key(row);
    if (row >= 0 && (lane == 63 || rn != row)) {
      double* d = dgb + 9 * (size_t)row;
#pragma unroll
      for (int c = 0; c < 9; ++c) d[c] = acc[c];
    }
  }
}

// ---------------------------------------------------------------------------- k_finalize
// Per row: diagonal block (symmetric packing), block-diagonal inverse, and the PCG start state
//   x = 0, r = b, z = Dinv b, p = z ; partials[0][blk] = r.z, partials[1][blk] = b.b
// and, for the multigrid cycle's first smoothing sweep from zero, xs = omega Dinv b.
__global__ __launch_bounds__(kBlock) void k_finalize(Sym0Dev A, int row0, int row1, const double* __restrict__ dgb,
                                                     double* __restrict__ b, double* __restrict__ x,
                                                     double* __restrict__ r, double* __restrict__ z,
                                                     double* __restrict__ p, double* __restrict__ xs, double omega,
                                                     double* __restrict__ partials) {
  double acc[2] = {0.0, 0.0};
  for (int i = row0 + blockIdx.x * kBlock + threadIdx.x; i < row1; i += gridDim.x * kBlock) {   // (multi-GPU: this rank's rows)
    const double* d = dgb + 9 * (size_t)i;
    const double d00 = d[0], d01 = d[1], d02 = d[2], d11 = d[3], d12 = d[4], d22 = d[5];
    const double b0 = d[6], b1 = d[7], b2 = d[8];
    double* dd = A.dblk + 6 * (size_t)i;
    dd[0] = d00; dd[1] = d01; dd[2] = d02; dd[3] = d11; dd[4] = d12; dd[5] = d22;
    // symmetric 3x3 inverse by cofactors
    const double c00 = d11 * d22 - d12 * d12, c01 = d02 * d12 - d01 * d22, c02 = d01 * d12 - d02 * d11;
    const double c11 = d00 * d22 - d02 * d02, c12 = d01 * d02 - d00 * d12, c22 = d00 * d11 - d01 * d01;
    const double det = d00 * c00 + d01 * c01 + d02 * c02;
    const double id = (det != 0.0 && isfinite(det)) ? 1.0 / det : 0.0;
    const double i00 = c00 * id, i01 = c01 * id, i02 = c02 * id, i11 = c11 * id, i12 = c12 * id, i22 = c22 * id;
    double* di = A.dinv + 6 * (size_t)i;
    di[0] = i00; di[1] = i01; di[2] = i02; di[3] = i11; di[4] = i12; di[5] = i22;
    const double z0 = i00 * b0 + i01 * b1 + i02 * b2;
    const double z1 = i01 * b0 + i11 * b1 + i12 * b2;
    const double z2 = i02 * b0 + i12 * b1 + i22 * b2;
    const size_t o = 3 * (size_t)i;
    b[o] = b0; b[o + 1] = b1; b[o + 2] = b2;
    if (x) {
      x[o] = 0.0; x[o + 1] = 0.0; x[o + 2] = 0.0;
      r[o] = b0; r[o + 1] = b1; r[o + 2] = b2;
      z[o] = z0; z[o + 1] = z1; z[o + 2] = z2;
      p[o] = z0; p[o + 1] = z1; p[o + 2] = z2;
    }
    if (xs) {
      xs[o] = omega * z0; xs[o + 1] = omega * z1; xs[o + 2] = omega * z2;
    }
    acc[0] += b0 * z0 + b1 * z1 + b2 * z2;
    acc[1] += b0 * b0 + b1 * b1 + b2 * b2;
  }
  block_sum_store<2>(acc, partials, kMaxPartials);
}

// How far the level-0 diagonal blocks have moved since the multigrid hierarchy's coarse operators were last refreshed:
// partials [0] sum_i ||D_i - Dref_i||_F, [1] sum_i ||D_i||_F, [2] rows whose block moved by more than a quarter of its norm.
// (optimize_gn decides from these whether this Gauss-Newton iteration's solve keeps the coarse operators of the last one.)
__global__ __launch_bounds__(kBlock) void k_diag_change(int row0, int row1, const double* __restrict__ dblk, double* __restrict__ dref,
                                                        int store_ref, double* __restrict__ partials) {
  double acc[3] = {0.0, 0.0, 0.0};
  for (int i = row0 + blockIdx.x * kBlock + threadIdx.x; i < row1; i += gridDim.x * kBlock) {
    const double* d = dblk + 6 * (size_t)i;
    double* e = dref + 6 * (size_t)i;
    double dn = 0.0, nn = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double w = (k == 0 || k == 3 || k == 5) ? 1.0 : 2.0, v = d[k], u = v - e[k];
      dn += w * u * u;
      nn += w * v * v;
      if (store_ref) e[k] = v;   // (the caller already knows this solve refreshes: the blocks become the new reference)
    }
    dn = sqrt(dn);
    nn = sqrt(nn);
    acc[0] += dn;
    acc[1] += nn;
    acc[2] += (dn > 0.25 * nn) ? 1.0 : 0.0;
  }
  block_sum_store<3>(acc, partials, kMaxPartials);
}
// The host has seen no new minimum of r.r for thousands of iterations (run_pcg): the solve ends as one that ran out of iterations.
__global__ void k_force_stop(PcgScalars* S, PcgScalars* mirror) {
  if (S->stop == 0) S->stop = 2;
  if (mirror) *mirror = *S;
}
__global__ void k_set_probe(PcgScalars* S, int probe_k, double probe_max) {
  S->probe_k = probe_k;
  S->probe_rel = 0.0;
  S->probe_max = probe_max;
}
// PCG continues an interrupted solve with another preconditioner: the caller has set z = M^-1 r, p = z for the
// CURRENT residual; r.z is re-reduced, the recurrence scalars restart, x / r / iteration count / ||b|| / tolerance stay.
// keep_stop: a stop flag that is already set stays (a solve that init found trivially converged); otherwise the flag
// is cleared (a solve interrupted by its iteration cap carries on).
__global__ __launch_bounds__(kBlock) void k_restart_scalars(PcgScalars* S, const double* __restrict__ rz_parts, int n_rz, int maxit,
                                                            int keep_stop) {
  const double rz = block_reduce_parts(rz_parts, n_rz);
  if (threadIdx.x == 0) {
    S->rz = rz;
    S->rz_prev = rz;
    S->pq = 0.0;
    S->alpha = 0.0;
    S->beta = 0.0;
    S->iter_prev = S->iter;
    S->maxit = maxit;
    if (!(keep_stop && S->stop)) S->stop = (isfinite(rz) && rz >= 0.0) ? (rz == 0.0 ? 1 : 0) : 3;
  }
}

// PCG start from the previous Gauss-Newton step: x = gamma x_prev, r = b - gamma H x_prev with the energy-optimal
// gamma = (b . x_prev) / (x_prev . H x_prev) (every workgroup re-reduces the two partial sums in the same order);
// a useless gamma (non-finite, <= 0, > 4) gives the cold start x = 0, r = b.
__global__ __launch_bounds__(kBlock) void k_warm_start(int n3, const double* __restrict__ xp, const double* __restrict__ q,
                                                       const double* __restrict__ b, double* __restrict__ x, double* __restrict__ r,
                                                       const double* __restrict__ xq_parts, int n_xq,
                                                       const double* __restrict__ bx_parts, int n_bx) {
  const double xq = block_reduce_parts(xq_parts, n_xq);
  const double bx = block_reduce_parts(bx_parts, n_bx);
  double gamma = bx / xq;
  if (!(xq > 0.0) || !isfinite(gamma) || !(gamma > 0.0) || gamma > 4.0) gamma = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n3; i += gridDim.x * kBlock) {
    x[i] = gamma * xp[i];
    r[i] = b[i] - gamma * q[i];
  }
}

__global__ __launch_bounds__(kBlock) void k_init_scalars(PcgScalars* S, const double* __restrict__ rz_parts, int n_rz,
                                                         const double* __restrict__ bb_parts, int n_bb, double tol,
                                                         int maxit, double bb_ref, double tol_cap) {
  const double* const parts[2] = {rz_parts, bb_parts};
  const int cnt[2] = {n_rz, n_bb};
  double v[2];
  block_reduce_parts_n<2>(parts, cnt, v);
  const double rz = v[0], bb = v[1];
  if (threadIdx.x == 0) {
    S->rz = rz;
    S->bb = bb;
    S->rr = bb;
    S->pq = 0.0;
    S->alpha = 0.0;
    S->beta = 0.0;
    // bb_ref > 0: ABSOLUTE accuracy of the call's first solve, ||r|| <= tol ||b_first||, for the later Gauss-Newton
    // iterations whose right-hand side has shrunk (capped at the relative tolerance tol_cap)
    double t2 = tol * tol;
    if (bb_ref > 0.0 && bb > 0.0 && bb < bb_ref) t2 = fmin(tol_cap * tol_cap, t2 * (bb_ref / bb));
    S->tol2 = t2;
    S->rz_prev = rz;
    S->iter = 0;
    S->iter_prev = 0;
    S->maxit = maxit;
    S->stop = (bb == 0.0) ? 1 : (isfinite(bb) && isfinite(rz) ? 0 : 3);
  }
}

// ---------------------------------------------------------------------------- k_spmv
// One lane per slot, wavefront segmented scan per row.  MODE selects operand and row epilogue:
//   SPMV_AX           y = A x
//   SPMV_RESID        y = b - A x
//   SPMV_JACOBI       y = x + omega Dinv (b - A x)               (damped block-Jacobi sweep)
//   SPMV_PRE_RESID    y2 = omega Dinv b ; y = b - A y2            (first sweep from x = 0 fused with
//                     the residual: the gathered operand is omega Dinv[col] b[col])
// Fused variants used on the coarse multigrid levels, where a launch costs more than its data
// (NOTES.md section 7); scalars c1, c2 are ratios of per-block partial sums that every block
// reduces itself in a fixed order:
//   SPMV_JACOBI_P     as JACOBI with x' = x + P (c1 u1 + c2 u2)   (prolongation of the coarse
//                     correction, itself the FCG combination of the child level, fused in)
//   SPMV_PRE_RESID_S  as PRE_RESID with b' = b - c1 bsub ; b_out = b'   (FCG residual update fused in)
//   SPMV_PRE_RESID_ACC as PRE_RESID with y2 += omega Dinv b         (pre-smoothing sweeps after the first)
//   SPMV_AX_C         as AX with x' = x - c1 x2 ; x_out = x'       (FCG direction update fused in);
//                     dots: partials[0] = x'.y, partials[1] = x'.dotC
// Optional dot partials (row epilogue): partials[0] += dotA[row].out[row],
// partials[1] += dotA2[row].out[row]  or  dotB[row].dotC[row].
__device__ __forceinline__ double ratio_value(const SpmvRatio& r, double den, double num) {
  if (!r.num) return r.den ? 0.0 : 1.0;
  return (den > 0.0 && isfinite(den) && isfinite(num)) ? num / den : 0.0;
}
// c1 = ratio r1, c2 = ratio r2 (only when use2): all partial sums in one block reduction
__device__ __forceinline__ void ratios2(const SpmvRatio& r1, const SpmvRatio& r2, bool use2, double& c1, double& c2) {
  const double* const parts[4] = {r1.num ? r1.den : nullptr, r1.num, (use2 && r2.num) ? r2.den : nullptr,
                                  use2 ? r2.num : nullptr};
  const int cnt[4] = {r1.n_den, r1.n_num, r2.n_den, r2.n_num};
  double v[4];
  block_reduce_parts_n<4>(parts, cnt, v);
  c1 = ratio_value(r1, v[0], v[1]);
  if (use2) c2 = ratio_value(r2, v[2], v[3]);
}

// The two level-0 sweeps (JACOBI, PRE_RESID) would take 84 / 76 VGPRs = 5 / 6 waves per SIMD, which
// leaves part of the 2048-block grid waiting for a second round; bounding them to 8 blocks per CU
// (<= 64 VGPRs, no spills) keeps the whole grid resident: 37 -> 30 us per sweep on C4 (6.0 TB/s).
template <int MODE>
__global__ __launch_bounds__(kBlock, ((MODE == SPMV_JACOBI || MODE == SPMV_PRE_RESID) ? 8 : 1))
void k_spmv(BsrDev A, SpmvArgs a) {
  // the first group's bounds are requested BEFORE the stop flag is waited for: on the coarse levels a launch is a
  // chain of four or five dependent memory round trips, and this folds two of them into one
  int g, gend, gstride;
  group_walk(A.ngrp, &g, &gend, &gstride);
  int gb0 = 0, ge0 = 0;
  if (g < gend) {
    gb0 = A.grp[g];
    ge0 = A.grp[g + 1];
  }
  if (a.S && a.S->stop) return;
  const int lane = threadIdx.x & 63;
  const size_t ns = (size_t)A.nslot;
  double c1 = 1.0, c2 = 0.0;
  if (MODE == SPMV_JACOBI_P || MODE == SPMV_PRE_RESID_S || MODE == SPMV_AX_C) {
    ratios2(a.c1, a.c2, MODE == SPMV_JACOBI_P && a.u2 != nullptr, c1, c2);
  }
  // operand of vertex v (3 doubles) under the mode's transformation
  auto operand = [&](size_t v, double& x0, double& x1, double& x2) {
    if (MODE == SPMV_PRE_RESID || MODE == SPMV_PRE_RESID_S || MODE == SPMV_PRE_RESID_ACC) {
      const double* di = A.dinv + 6 * v;
      double b0 = a.b[3 * v], b1 = a.b[3 * v + 1], b2 = a.b[3 * v + 2];
      if (MODE == SPMV_PRE_RESID_S) {
        b0 -= c1 * a.bsub[3 * v]; b1 -= c1 * a.bsub[3 * v + 1]; b2 -= c1 * a.bsub[3 * v + 2];
      }
      x0 = a.omega * (di[0] * b0 + di[1] * b1 + di[2] * b2);
      x1 = a.omega * (di[1] * b0 + di[3] * b1 + di[4] * b2);
      x2 = a.omega * (di[2] * b0 + di[4] * b1 + di[5] * b2);
    } else {
      x0 = a.x[3 * v]; x1 = a.x[3 * v + 1]; x2 = a.x[3 * v + 2];
      if (MODE == SPMV_JACOBI_P) {
        const size_t ag = 3 * (size_t)a.agg[v];
        double w0 = c1 * a.u1[ag], w1 = c1 * a.u1[ag + 1], w2 = c1 * a.u1[ag + 2];
        if (a.u2) {
          w0 += c2 * a.u2[ag]; w1 += c2 * a.u2[ag + 1]; w2 += c2 * a.u2[ag + 2];
        }
        x0 += w0 - a.d[2 * v + 1] * w2;
        x1 += w1 + a.d[2 * v] * w2;
        x2 += w2;
      } else if (MODE == SPMV_AX_C) {
        x0 -= c1 * a.x2[3 * v]; x1 -= c1 * a.x2[3 * v + 1]; x2 -= c1 * a.x2[3 * v + 2];
      }
    }
  };
  double dotacc[2] = {0.0, 0.0};
  for (bool first = true; g < gend; g += gstride, first = false) {
    const int gb = first ? gb0 : A.grp[g], ge = first ? ge0 : A.grp[g + 1];
    double acc[3] = {0.0, 0.0, 0.0};
    int row = -1 - lane;
    for (int k = gb + lane; k < ge; k += 64) {
      row = A.row[k];
      double x0, x1, x2;
      operand((size_t)A.col[k], x0, x1, x2);
      // four 16-byte loads (component pairs) + one 8-byte load
      const double2* __restrict__ bp = reinterpret_cast<const double2*>(A.blk);
      const double2 p0 = bp[k], p1 = bp[ns + k], p2 = bp[2 * ns + k], p3 = bp[3 * ns + k];
      const double b0 = p0.x, b1 = p0.y, b2 = p1.x, b3 = p1.y, b4 = p2.x, b5 = p2.y, b6 = p3.x, b7 = p3.y;
      const double b8 = A.blk[8 * ns + k];
      acc[0] += b0 * x0 + b1 * x1 + b2 * x2;
      acc[1] += b3 * x0 + b4 * x1 + b5 * x2;
      acc[2] += b6 * x0 + b7 * x1 + b8 * x2;
    }
    seg_scan<3>(row, acc, lane);
    const int rn = next_lane_key(row);
    if (row >= 0 && (lane == 63 || rn != row)) {
      const size_t o = 3 * (size_t)row;
      double o0 = acc[0], o1 = acc[1], o2 = acc[2];
      if (MODE == SPMV_AX_C) {
        double s0, s1, s2;
        operand((size_t)row, s0, s1, s2);
        a.x_out[o] = s0; a.x_out[o + 1] = s1; a.x_out[o + 2] = s2;
        dotacc[0] += s0 * o0 + s1 * o1 + s2 * o2;
        dotacc[1] += s0 * a.dotC[o] + s1 * a.dotC[o + 1] + s2 * a.dotC[o + 2];
      } else if (MODE != SPMV_AX) {
        double b0 = a.b[o], b1 = a.b[o + 1], b2 = a.b[o + 2];
        if (MODE == SPMV_PRE_RESID_S) {
          b0 -= c1 * a.bsub[o]; b1 -= c1 * a.bsub[o + 1]; b2 -= c1 * a.bsub[o + 2];
          a.b_out[o] = b0; a.b_out[o + 1] = b1; a.b_out[o + 2] = b2;
        }
        const double r0 = b0 - acc[0], r1 = b1 - acc[1], r2 = b2 - acc[2];
        if (MODE == SPMV_JACOBI || MODE == SPMV_JACOBI_P) {
          const double* di = A.dinv + 6 * (size_t)row;
          double s0, s1, s2;
          operand((size_t)row, s0, s1, s2);
          o0 = s0 + a.omega * (di[0] * r0 + di[1] * r1 + di[2] * r2);
          o1 = s1 + a.omega * (di[1] * r0 + di[3] * r1 + di[4] * r2);
          o2 = s2 + a.omega * (di[2] * r0 + di[4] * r1 + di[5] * r2);
        } else {
          o0 = r0; o1 = r1; o2 = r2;
        }
        if (MODE == SPMV_PRE_RESID || MODE == SPMV_PRE_RESID_S || MODE == SPMV_PRE_RESID_ACC) {
          const double* di = A.dinv + 6 * (size_t)row;
          double v0 = a.omega * (di[0] * b0 + di[1] * b1 + di[2] * b2);
          double v1 = a.omega * (di[1] * b0 + di[3] * b1 + di[4] * b2);
          double v2 = a.omega * (di[2] * b0 + di[4] * b1 + di[5] * b2);
          if (MODE == SPMV_PRE_RESID_ACC) {
            v0 += a.y2[o]; v1 += a.y2[o + 1]; v2 += a.y2[o + 2];
          }
          a.y2[o] = v0; a.y2[o + 1] = v1; a.y2[o + 2] = v2;
        }
      }
      a.y[o] = o0; a.y[o + 1] = o1; a.y[o + 2] = o2;
      if (MODE != SPMV_AX_C) {
        if (a.dotA) dotacc[0] += a.dotA[o] * o0 + a.dotA[o + 1] * o1 + a.dotA[o + 2] * o2;
        if (a.dotA2) dotacc[1] += a.dotA2[o] * o0 + a.dotA2[o + 1] * o1 + a.dotA2[o + 2] * o2;
        else if (a.dotB) dotacc[1] += a.dotB[o] * a.dotC[o] + a.dotB[o + 1] * a.dotC[o + 1] + a.dotB[o + 2] * a.dotC[o + 2];
      }
    }
  }
  if (a.partials) block_sum_store<2>(dotacc, a.partials, kMaxPartials);
}

// ---------------------------------------------------------------------------- k_spmv0
// The level-0 (finest) Hessian product on the SYMMETRIC storage (Sym0Dev, sgo_internal.h): one lane
// per compact slot; an owned slot streams its block, a transposed slot fetches the block stored with
// the other endpoint's row (an L2 hit thanks to the Hilbert row order) and multiplies by its transpose;
// wavefront segmented scan per row; the row's last lane adds the diagonal block's product and applies
// the mode's epilogue:
//   S0_AX      y = H x
//   S0_RESID   y = b - H x                           (x = omega Dinv b was made by the producer of b:
//                                                     first smoothing sweep from zero + residual)
//   S0_JACOBI  y = x + omega Dinv (b - H x)          (damped block-Jacobi sweep)
// Optional dot partials: partials[0] += dotA[row].y[row], partials[1] += dotA2[row].y[row].
// (Graphs below kSmallGraphPairs take it: a few hundred to a thousand workgroups; the row data held across the scan costs
// registers, 4 workgroups per CU are plenty.)
template <int MODE>
__global__ __launch_bounds__(kBlock, 4) void k_spmv0(Sym0Dev A, Spmv0Args a) {
  spmv0_groups<MODE>(A, a, (int)gridDim.x);
}

// ---------------------------------------------------------------------------- k_spmv0t
// The level-0 product on the TILE view (Tile0Dev, sgo_internal.h): every stored block is streamed by
// exactly one lane, operands come from LDS (the tile's slice + its halo, fetched once per tile), transposed
// contributions of intra-tile pairs travel through LDS staging.  Modes and epilogues as k_spmv0.  One
// workgroup works on one tile at a time; workgroups of XCD x walk the x-th contiguous eighth of the tiles.
// one wave's loads for the first 64 slots of a phase-1 group, issued a whole group ahead of their use
// a stored block as it comes from memory: fp64 pair-SoA (four 16-byte loads + one 8-byte load) or the fp32 copy of the
// preconditioner's passes (two 16-byte loads + one 4-byte load)
template <bool F32> struct TileBlk;
template <> struct TileBlk<false> {
  double2 p0, p1, p2, p3;
  double b8;
  __device__ __forceinline__ void load(const Sym0Dev& A, size_t nu, int k) {
    const double2* __restrict__ bp = reinterpret_cast<const double2*>(A.ublk);
    p0 = bp[k]; p1 = bp[nu + k]; p2 = bp[2 * nu + k]; p3 = bp[3 * nu + k];
    b8 = A.ublk8[k];
  }
  __device__ __forceinline__ void get(double (&b)[9]) const {
    b[0] = p0.x; b[1] = p0.y; b[2] = p1.x; b[3] = p1.y; b[4] = p2.x; b[5] = p2.y; b[6] = p3.x; b[7] = p3.y; b[8] = b8;
  }
};
template <> struct TileBlk<true> {
  float4 q0, q1;
  float b8;
  __device__ __forceinline__ void load(const Sym0Dev& A, size_t nu, int k) {
    const float4* __restrict__ fp = reinterpret_cast<const float4*>(A.fblk);
    q0 = fp[k]; q1 = fp[nu + k];
    b8 = A.fblk8[k];
  }
  __device__ __forceinline__ void get(double (&b)[9]) const {
    b[0] = q0.x; b[1] = q0.y; b[2] = q0.z; b[3] = q0.w; b[4] = q1.x; b[5] = q1.y; b[6] = q1.z; b[7] = q1.w; b[8] = b8;
  }
};
template <bool F32>
struct TileGroupLoad {
  int gb, ge, r0;
  bool valid;
  unsigned cw;
  int off;
  TileBlk<F32> blk;
};
template <bool F32>
__device__ __forceinline__ void tile_group_load(const Sym0Dev& A, const Tile0Dev& TL, size_t nu, int g, int lane, TileGroupLoad<F32>& L) {
  L.gb = TL.grp1[g];
  L.ge = TL.grp1[g + 1];
  L.r0 = TL.grow1[g];
  const int k = L.gb + lane;
  L.valid = k < L.ge;
  if (L.valid) {
    L.cw = TL.cv[k];
    L.off = TL.off1[k];
    L.blk.load(A, nu, k);
  }
}
// u = B x_col into acc, v = B^T x_row to the twin's staging slot
__device__ __forceinline__ void tile_slot(const double* __restrict__ xs, double* __restrict__ vst, int rowl, unsigned cw,
                                          const double (&b)[9], double (&acc)[3]) {
  const double* xc = xs + 3 * (cw & 0xFFFFu);
  const double x0 = xc[0], x1 = xc[1], x2 = xc[2];
  acc[0] += b[0] * x0 + b[1] * x1 + b[2] * x2;
  acc[1] += b[3] * x0 + b[4] * x1 + b[5] * x2;
  acc[2] += b[6] * x0 + b[7] * x1 + b[8] * x2;
  const unsigned vp = cw >> 16;
  if (vp != 0xFFFFu) {   // the twin row is in this tile: hand it B^T x_row through LDS
    const double* xr = xs + 3 * rowl;
    const double s0 = xr[0], s1 = xr[1], s2 = xr[2];
    double* v = vst + 3 * vp;
    v[0] = b[0] * s0 + b[3] * s1 + b[6] * s2;
    v[1] = b[1] * s0 + b[4] * s1 + b[7] * s2;
    v[2] = b[2] * s0 + b[5] * s1 + b[8] * s2;
  }
}

template <int MODE, int kTileThreads, bool F32 = false>
__global__ __launch_bounds__(kTileThreads) void k_spmv0t(Sym0Dev A, Tile0Dev TL, Spmv0Args a) {
  // the stop flag, the tile descriptor and the tile's first halo column numbers are requested together (none of their
  // addresses depends on another's value): phase 0 is a chain of two memory round trips instead of four
  const int stop_flag = a.S ? a.S->stop : 0;
  extern __shared__ double lds[];
  __shared__ int next_group_cell;
  int* next_group = &next_group_cell;
  constexpr int NW = kTileThreads / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t nu = (size_t)A.nus;
  double dotacc[2] = {0.0, 0.0};
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int ulo = a.u1 > 0 ? a.u0 : 0, nun = (a.u1 > 0 ? a.u1 : TL.ntile) - ulo;
  const int tlo = ulo + (int)(((long long)nun * xcd) >> 3), thi = ulo + (int)(((long long)nun * (xcd + 1)) >> 3);
  for (int t = tlo + slot; t < thi; t += per_xcd) {
    const int hc_first = tid < TL.hstride ? TL.hfirst[(size_t)t * TL.hstride + tid] : -1;
    const TileDesc T = TL.tile[t];
    if (stop_flag) return;
    const int nr = T.row1 - T.row0, nh = T.h1 - T.h0;
    long long* stamp = a.dbg_stamps ? a.dbg_stamps + 8 * (size_t)t : nullptr;
    if (stamp && tid == 0) stamp[0] = __builtin_amdgcn_s_memtime();
    double* xs = lds;                    // [nr + nh][3] operand: the tile's rows, then its halo columns
    double* ys = xs + 3 * (nr + nh);     // [nr][3] owned part of the row sums
    double* vst = ys + 3 * nr;           // [nstaged][3]
    // ---- phase 0: operand slice and halo to LDS, owned sums cleared.  Order of issue matters (a wave's loads
    // return in order): the operand slice and the halo column numbers first, then -- while those are in flight --
    // this wave's first phase-1 group and the phase-2 row data, then the halo operands
    double xsl[3] = {0.0, 0.0, 0.0};   // up to three slice elements per thread (3 * rows <= 3 * kTileThreads)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int i = tid + q * kTileThreads;
      if (i < 3 * nr) xsl[q] = a.x[3 * (size_t)T.row0 + i];
    }
    const int hc0 = tid < TL.hstride ? hc_first : (tid < nh ? TL.hcol[T.h0 + tid] : -1);
    if (tid == 0) *next_group = T.g0 + NW;   // groups are handed out through an LDS counter (the first NW statically)
    int g = T.g0 + wave;
    bool has = g < T.g1;
    TileGroupLoad<F32> cur;
    cur.valid = false;
    if (has) tile_group_load<F32>(A, TL, nu, g, lane, cur);
    int e0 = 0, e1 = 0;
    double dd0 = 0, dd1 = 0, dd2 = 0, dd3 = 0, dd4 = 0, dd5 = 0;
    if (tid < nr) {
      const int r = T.row0 + tid;
      e0 = TL.trowptr[r] - T.e0;
      e1 = TL.trowptr[r + 1] - T.e0;
      const double* dd = A.dblk + 6 * (size_t)r;
      dd0 = dd[0]; dd1 = dd[1]; dd2 = dd[2]; dd3 = dd[3]; dd4 = dd[4]; dd5 = dd[5];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int i = tid + q * kTileThreads;
      if (i < 3 * nr) {
        xs[i] = xsl[q];
        ys[i] = 0.0;
      }
    }
    for (int i = tid + 3 * kTileThreads; i < 3 * nr; i += kTileThreads) {   // tiles of more rows than threads (rare)
      xs[i] = a.x[3 * (size_t)T.row0 + i];
      ys[i] = 0.0;
    }
    if (hc0 >= 0) {
      const size_t c3 = 3 * (size_t)hc0;
      const double h0 = a.x[c3], h1 = a.x[c3 + 1], h2 = a.x[c3 + 2];
      double* d = xs + 3 * (nr + tid);
      d[0] = h0; d[1] = h1; d[2] = h2;
    }
    for (int i = tid + kTileThreads; i < nh; i += kTileThreads) {
      const size_t c3 = 3 * (size_t)TL.hcol[T.h0 + i];
      const double h0 = a.x[c3], h1 = a.x[c3 + 1], h2 = a.x[c3 + 2];
      double* d = xs + 3 * (nr + i);
      d[0] = h0; d[1] = h1; d[2] = h2;
    }
    if (stamp && tid == 0) stamp[1] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (stamp && tid == 0) stamp[2] = __builtin_amdgcn_s_memtime();
    // ---- phase 1: one lane per stored block; groups hold whole rows.  A wave takes its next group from the LDS
    // counter and issues that group's loads before it works on the current one (which wave sums which group does
    // not change a single bit of the result: every group writes its own rows)
    while (has) {
      int gn = 0;
      if (lane == 0) gn = atomicAdd(next_group, 1);
      gn = __builtin_amdgcn_readfirstlane(gn);
      const bool hasn = gn < T.g1;
      TileGroupLoad<F32> nxt;
      nxt.valid = false;
      if (hasn) tile_group_load<F32>(A, TL, nu, gn, lane, nxt);
      double acc[3] = {0.0, 0.0, 0.0};
      int row = -1 - lane;
      if (cur.valid) {
        row = cur.r0 + cur.off;
        double b[9];
        cur.blk.get(b);
        tile_slot(xs, vst, row - T.row0, cur.cw, b, acc);
      }
      for (int k = cur.gb + 64 + lane; k < cur.ge; k += 64) {   // a row longer than one wave
        const unsigned cw = TL.cv[k];
        row = cur.r0 + TL.off1[k];
        TileBlk<F32> bl;
        bl.load(A, nu, k);
        double b[9];
        bl.get(b);
        tile_slot(xs, vst, row - T.row0, cw, b, acc);
      }
      seg_scan<3>(row, acc, lane);
      const int rn = next_lane_key(row);
      if (row >= 0 && (lane == 63 || rn != row)) {   // a row's owned slots sit in exactly one group: single writer
        double* d = ys + 3 * (row - T.row0);
        d[0] = acc[0]; d[1] = acc[1]; d[2] = acc[2];
      }
      cur = nxt;
      g = gn;
      has = hasn;
    }
    if (stamp && tid == 0) stamp[3] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (stamp && tid == 0) stamp[4] = __builtin_amdgcn_s_memtime();
    // ---- phase 2: one thread per row: staged entries in order, owned part, diagonal block, epilogue
    for (int i = tid; i < nr; i += kTileThreads) {
      const int r = T.row0 + i;
      if (i >= kTileThreads) {   // tiles of more rows than threads (rare): the later rows load here
        e0 = TL.trowptr[r] - T.e0;
        e1 = TL.trowptr[r + 1] - T.e0;
        const double* dd = A.dblk + 6 * (size_t)r;
        dd0 = dd[0]; dd1 = dd[1]; dd2 = dd[2]; dd3 = dd[3]; dd4 = dd[4]; dd5 = dd[5];
      }
      const double s0 = xs[3 * i], s1 = xs[3 * i + 1], s2 = xs[3 * i + 2];
      double o0 = ys[3 * i], o1 = ys[3 * i + 1], o2 = ys[3 * i + 2];
      {
        // the staged entries in their fixed order, four entries' LDS reads in flight at a time (the same additions in the same
        // order: a loop that reads and adds one entry per trip is a chain of LDS latencies)
        constexpr int kB = F32 ? 4 : 2;   // (the fp64 instantiations sit at 126 of 128 VGPRs: four entries in flight spilled)
        int e = e0;
        for (; e + kB <= e1; e += kB) {
          double v[3 * kB];
#pragma unroll
          for (int q = 0; q < 3 * kB; ++q) v[q] = vst[3 * e + q];
#pragma unroll
          for (int q = 0; q < kB; ++q) {
            o0 += v[3 * q]; o1 += v[3 * q + 1]; o2 += v[3 * q + 2];
          }
        }
        for (; e < e1; ++e) {
          o0 += vst[3 * e]; o1 += vst[3 * e + 1]; o2 += vst[3 * e + 2];
        }
      }
      o0 += dd0 * s0 + dd1 * s1 + dd2 * s2;
      o1 += dd1 * s0 + dd3 * s1 + dd4 * s2;
      o2 += dd2 * s0 + dd4 * s1 + dd5 * s2;
      const size_t o = 3 * (size_t)r;
      if (MODE != S0_AX) {
        const double t0 = a.b[o] - o0, t1 = a.b[o + 1] - o1, t2 = a.b[o + 2] - o2;
        if (MODE == S0_JACOBI) {
          const double* di = A.dinv + 6 * (size_t)r;
          o0 = s0 + a.omega * (di[0] * t0 + di[1] * t1 + di[2] * t2);
          o1 = s1 + a.omega * (di[1] * t0 + di[3] * t1 + di[4] * t2);
          o2 = s2 + a.omega * (di[2] * t0 + di[4] * t1 + di[5] * t2);
        } else {
          o0 = t0; o1 = t1; o2 = t2;
        }
      }
      a.y[o] = o0; a.y[o + 1] = o1; a.y[o + 2] = o2;
      if (MODE == S0_AX && a.dotA == a.x) {
        // p . H p: the operand row is in LDS already (s0..s2, the same values): no global round trip at the tail of the launch
        dotacc[0] += s0 * o0 + s1 * o1 + s2 * o2;
      } else if (a.dotA) {
        dotacc[0] += a.dotA[o] * o0 + a.dotA[o + 1] * o1 + a.dotA[o + 2] * o2;
      }
      if (a.dotA2) dotacc[1] += a.dotA2[o] * o0 + a.dotA2[o + 1] * o1 + a.dotA2[o + 2] * o2;
    }
    if (stamp && tid == 0) stamp[5] = __builtin_amdgcn_s_memtime();
    if (t + per_xcd < thi) __syncthreads();   // the next tile reuses the LDS (a workgroup's last -- usually only -- tile: nothing to wait for)
    if (stamp && tid == 0) stamp[6] = __builtin_amdgcn_s_memtime();
  }
  if (a.partials) {
    __shared__ double sm[2][NW];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double sum = wave_sum(dotacc[i]);
      if (lane == 0) sm[i][wave] = sum;
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        double sum = sm[i][0];
#pragma unroll
        for (int k = 1; k < NW; ++k) sum += sm[i][k];
        a.partials[(size_t)i * kMaxPartials + blockIdx.x] = sum;
      }
    }
  }
}

// alpha = r.z / p.q from the partials of p.q (every workgroup re-reduces them in the same fixed
// order, so all agree bit for bit and no single-workgroup scalar kernel sits between the product
// and the update; workgroup 0 records the scalars), then
// x += alpha p ; r -= alpha q ; z = Dinv r ; partials: r.z, r.r
// With the multigrid preconditioner (xs != nullptr) the update also leaves xs = omega Dinv r, the cycle's
// first level-0 smoothing sweep from zero, so that the level-0 residual pass gathers a plain vector.
__global__ __launch_bounds__(kBlock) void k_update_xr(int n, PcgScalars* S, const double* __restrict__ pq_parts, int n_pq,
                                                      const double* __restrict__ dinv, const double* __restrict__ p,
                                                      const double* __restrict__ q, double* __restrict__ x,
                                                      double* __restrict__ r, double* __restrict__ z,
                                                      double* __restrict__ xs, double omega,
                                                      double* __restrict__ partials) {
  // the scalars of the previous launches in one go, before the reduction's barriers -- and this thread's first row
  // of operands too (the grid gives every thread at most one row on all but the largest graphs): the partial sums'
  // round trip and the operands' overlap instead of following each other
  const int i0 = blockIdx.x * kBlock + threadIdx.x;
  double pr[3] = {0, 0, 0}, pq_[3] = {0, 0, 0}, pp[3] = {0, 0, 0}, px[3] = {0, 0, 0}, pd[6] = {0, 0, 0, 0, 0, 0};
  if (i0 < n) {
    const size_t o = 3 * (size_t)i0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      pr[c] = r[o + c];
      pq_[c] = q[o + c];
      pp[c] = p[o + c];
      px[c] = x[o + c];
    }
    if (dinv) {
#pragma unroll
      for (int c = 0; c < 6; ++c) pd[c] = dinv[6 * (size_t)i0 + c];
    }
  }
  const int stop0 = S->stop, iter0 = S->iter;
  const double rz = S->rz;
  if (stop0) return;
  const double pq = block_reduce_parts(pq_parts, n_pq);
  const bool bad = !(pq > 0.0) || !isfinite(pq);
  const double alpha = rz / pq;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    S->pq = pq;
    S->rz_prev = rz;
    S->iter_prev = iter0;
    if (bad) S->stop = 3;
    else S->alpha = alpha;
  }
  if (bad) return;
  double acc[2] = {0.0, 0.0};
  for (int i = i0; i < n; i += gridDim.x * kBlock) {
    const size_t o = 3 * (size_t)i;
    if (i != i0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        pr[c] = r[o + c];
        pq_[c] = q[o + c];
        pp[c] = p[o + c];
        px[c] = x[o + c];
      }
      if (dinv) {
#pragma unroll
        for (int c = 0; c < 6; ++c) pd[c] = dinv[6 * (size_t)i + c];
      }
    }
    const double r0 = pr[0] - alpha * pq_[0], r1 = pr[1] - alpha * pq_[1], r2 = pr[2] - alpha * pq_[2];
    x[o] = px[0] + alpha * pp[0]; x[o + 1] = px[1] + alpha * pp[1]; x[o + 2] = px[2] + alpha * pp[2];
    r[o] = r0; r[o + 1] = r1; r[o + 2] = r2;
    if (dinv) {
      const double z0 = pd[0] * r0 + pd[1] * r1 + pd[2] * r2;
      const double z1 = pd[1] * r0 + pd[3] * r1 + pd[4] * r2;
      const double z2 = pd[2] * r0 + pd[4] * r1 + pd[5] * r2;
      if (xs) {
        xs[o] = omega * z0; xs[o + 1] = omega * z1; xs[o + 2] = omega * z2;
      } else {
        z[o] = z0; z[o + 1] = z1; z[o + 2] = z2;
        acc[0] += r0 * z0 + r1 * z1 + r2 * z2;
      }
    }
    acc[1] += r0 * r0 + r1 * r1 + r2 * r2;
  }
  block_sum_store<2>(acc, partials, kMaxPartials);
}

// beta and the stopping test from the partials of r.z, r.r (and z.q), re-reduced by every
// workgroup as above, then p = z + beta p (flat over 3n).  Reads rz_prev / alpha, which the
// k_update_xr of this iteration recorded, so workgroup 0 can overwrite S->rz while the others
// are still running.
__global__ __launch_bounds__(kBlock) void k_update_p(int n3, PcgScalars* S, const double* __restrict__ rz_parts, int n_rz,
                                                     const double* __restrict__ rr_parts, int n_rr,
                                                     const double* __restrict__ zq_parts, const double* __restrict__ z,
                                                     double* __restrict__ p, RecDev rec) {
  // the scalars of the previous launches in one go, before the reduction's barriers, and this thread's first pair of
  // operands (see k_update_xr)
  const int i0 = blockIdx.x * kBlock + threadIdx.x;
  double z0 = 0.0, p0 = 0.0;
  if (i0 < n3) {
    z0 = z[i0];
    p0 = p[i0];
  }
  const int stop0 = S->stop, iter_prev = S->iter_prev, maxit = S->maxit;
  const double alpha = S->alpha, rz_prev = S->rz_prev, tol2 = S->tol2, bb = S->bb;
  if (stop0) {
    if (rec.mirror && blockIdx.x == 0 && threadIdx.x == 0) *rec.mirror = *S;   // (a solve stopped by another kernel, or iterations past convergence)
    return;
  }
  // flexible CG (variable preconditioner, e.g. the K-cycle): z_new.(r_new - r_old) = -alpha z_new.q
  const double* const parts[3] = {rz_parts, rr_parts, zq_parts};
  const int cnt[3] = {n_rz, n_rr, n_rz};
  double v[3];
  block_reduce_parts_n<3>(parts, cnt, v);
  const double rz = v[0], rr = v[1], zq = v[2];
  const double beta = zq_parts ? -alpha * zq / rz_prev : rz / rz_prev;
  int stop = 0;
  if (!isfinite(rz) || !isfinite(rr)) stop = 3;
  else if (rr <= tol2 * bb) stop = 1;
  else if (iter_prev + 1 >= maxit) stop = 2;
  const bool probe = iter_prev + 1 == S->probe_k;
  if (probe && !stop && S->probe_max > 0.0 && rr > S->probe_max * bb) stop = 4;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (probe) S->probe_rel = rr / bb;
    S->beta = beta;
    S->rz = rz;
    S->rr = rr;
    S->iter = iter_prev + 1;
    if (stop) S->stop = stop;
    if (rec.lanczos && iter_prev < kLanczosMax) {   // the recurrence's coefficients of this iteration (Lanczos matrix of M^-1 H)
      rec.lanczos[3 * iter_prev] = alpha;
      rec.lanczos[3 * iter_prev + 1] = beta;
      rec.lanczos[3 * iter_prev + 2] = rz_prev;
    }
    if (rec.mirror) {
      PcgScalars m = *S;   // (pq, alpha, rz_prev, iter_prev as this iteration's k_update_xr left them)
      m.beta = beta; m.rz = rz; m.rr = rr; m.iter = iter_prev + 1;
      if (probe) m.probe_rel = rr / bb;
      if (stop) m.stop = stop;
      *rec.mirror = m;
    }
  }
  if (stop) return;
  if (i0 < n3) {
    p[i0] = z0 + beta * p0;
  }
  for (int i = i0 + gridDim.x * kBlock; i < n3; i += gridDim.x * kBlock) {
    p[i] = z[i] + beta * p[i];
  }
}

// partials[blk] = sum a[i] * b[i]   (flat; used when the product had to be all-reduced first)
__global__ __launch_bounds__(kBlock) void k_dot(int n3, const double* __restrict__ a, const double* __restrict__ b,
                                                double* __restrict__ partials, const PcgScalars* S) {
  if (S && S->stop) return;
  double acc[1] = {0.0};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n3; i += gridDim.x * kBlock) acc[0] += a[i] * b[i];
  block_sum_store<1>(acc, partials, kMaxPartials);
}

// z = scale Dinv r
__global__ __launch_bounds__(kBlock) void k_precond_bj(int n, const double* __restrict__ dinv,
                                                       const double* __restrict__ r, double* __restrict__ z, double scale) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const size_t o = 3 * (size_t)i;
    const double* di = dinv + 6 * (size_t)i;
    const double r0 = r[o], r1 = r[o + 1], r2 = r[o + 2];
    z[o] = scale * (di[0] * r0 + di[1] * r1 + di[2] * r2);
    z[o + 1] = scale * (di[1] * r0 + di[3] * r1 + di[4] * r2);
    z[o + 2] = scale * (di[2] * r0 + di[4] * r1 + di[5] * r2);
  }
}

// SparseOptimizer::update -> VertexSE2::oplusImpl: t += d[0:2]; theta = normalize(theta + d[2])
__global__ __launch_bounds__(kBlock) void k_pose_update(int n, const int* __restrict__ free_id,
                                                        const double* __restrict__ x, double* __restrict__ poses) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const size_t v = 3 * (size_t)free_id[i], o = 3 * (size_t)i;
    poses[v] += x[o];
    poses[v + 1] += x[o + 1];
    poses[v + 2] = norm_theta(poses[v + 2] + x[o + 2]);
  }
}

// ---------------------------------------------------------------------------- row-owner mode: halo-row recurrences
__global__ __launch_bounds__(kBlock) void k_update_xr_rows(int nrows, const int* __restrict__ rows, const PcgScalars* S,
                                                           const double* __restrict__ dinv, const double* __restrict__ p,
                                                           const double* __restrict__ q, double* __restrict__ r,
                                                           double* __restrict__ xs, double omega) {
  if (S->stop) return;   // (stopped before, or this iteration's k_update_xr met a breakdown: nothing moved there either)
  const double alpha = S->alpha;
  for (int t = blockIdx.x * kBlock + threadIdx.x; t < nrows; t += gridDim.x * kBlock) {
    const size_t i = (size_t)rows[t], o = 3 * i;
    const double r0 = r[o] - alpha * q[o], r1 = r[o + 1] - alpha * q[o + 1], r2 = r[o + 2] - alpha * q[o + 2];
    r[o] = r0; r[o + 1] = r1; r[o + 2] = r2;
    if (xs) {
      const double* pd = dinv + 6 * i;
      const double z0 = pd[0] * r0 + pd[1] * r1 + pd[2] * r2;
      const double z1 = pd[1] * r0 + pd[3] * r1 + pd[4] * r2;
      const double z2 = pd[2] * r0 + pd[4] * r1 + pd[5] * r2;
      xs[o] = omega * z0; xs[o + 1] = omega * z1; xs[o + 2] = omega * z2;
    }
  }
}
__global__ __launch_bounds__(kBlock) void k_update_p_rows(int nrows, const int* __restrict__ rows, const PcgScalars* S,
                                                          const double* __restrict__ z, double* __restrict__ p) {
  if (S->stop) return;   // (this iteration's k_update_p did not move p either)
  const double beta = S->beta;
  for (int t = blockIdx.x * kBlock + threadIdx.x; t < nrows; t += gridDim.x * kBlock) {
    const size_t o = 3 * (size_t)rows[t];
    p[o] = z[o] + beta * p[o]; p[o + 1] = z[o + 1] + beta * p[o + 1]; p[o + 2] = z[o + 2] + beta * p[o + 2];
  }
}

// ---------------------------------------------------------------------------- row-owner exchanges
// pack: workgroup 0 reduces the scalar partial arrays (fixed order) into the packet's head; all workgroups copy the
// records idx[0 .. cnt) of `width` doubles each (padding entries, idx < 0, travel as zeros)
__global__ __launch_bounds__(kBlock) void k_halo_pack(const double* __restrict__ data, int width, const int* __restrict__ idx,
                                                      int cnt, double* __restrict__ send, HaloScalars sc) {
  if (blockIdx.x == 0) {
    const double* const parts[kHaloScalars] = {sc.parts[0], sc.parts[1], sc.parts[2], sc.parts[3]};
    const int n[kHaloScalars] = {sc.n[0], sc.n[1], sc.n[2], sc.n[3]};
    double v[kHaloScalars];
    block_reduce_parts_n<kHaloScalars>(parts, n, v);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < kHaloScalars; ++k) send[k] = v[k];
    }
  }
  const long long total = (long long)cnt * width;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long long)gridDim.x * kBlock) {
    const int i = (int)(e / width), c = (int)(e % width);
    const int g = idx[i];
    send[kHaloScalars + e] = g >= 0 ? data[(size_t)g * width + c] : 0.0;
  }
}
// unpack: the other ranks' records to their global places, every rank's scalars side by side (slot-major)
__global__ __launch_bounds__(kBlock) void k_halo_unpack(double* __restrict__ data, int width, const int* __restrict__ idx_all,
                                                        int cnt, int G, int me, const double* __restrict__ recv, size_t stride,
                                                        double* __restrict__ gparts) {
  if (blockIdx.x == 0)
    for (int t = threadIdx.x; t < kHaloScalars * G; t += kBlock) gparts[t] = recv[(size_t)(t % G) * stride + (size_t)(t / G)];
  const long long per = (long long)cnt * width, total = per * G;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long long)gridDim.x * kBlock) {
    const int s = (int)(e / per);
    if (s == me) continue;
    const long long rem = e - (long long)s * per;
    const int i = (int)(rem / width), c = (int)(rem % width);
    const int g = idx_all[(size_t)s * cnt + i];
    if (g >= 0) data[(size_t)g * width + c] = recv[(size_t)s * stride + kHaloScalars + rem];
  }
}
// the other ranks' owned slices of a flat vector to their places
__global__ __launch_bounds__(kBlock) void k_slices_unpack(double* __restrict__ vec, int width, const int* __restrict__ rank_row, int G, int me,
                                                          const double* __restrict__ recv, size_t stride) {
  for (int s = 0; s < G; ++s) {
    if (s == me) continue;
    const size_t base = (size_t)width * (size_t)rank_row[s], cnt = (size_t)width * (size_t)(rank_row[s + 1] - rank_row[s]);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < cnt; i += (size_t)gridDim.x * kBlock)
      vec[base + i] = recv[(size_t)s * stride + i];
  }
}

}  // namespace

// ---------------------------------------------------------------------------- launchers
void launch_update_xr_rows(hipStream_t s, int nrows, const int* rows, const PcgScalars* S, const double* dinv, const double* p,
                           const double* q, double* r, double* xs, double omega) {
  if (nrows > 0) SGO_LAUNCH(k_update_xr_rows, dim3(grid_for(nrows, kBlock)), dim3(kBlock), 0, s, nrows, rows, S, dinv, p, q, r, xs, omega);
}
void launch_update_p_rows(hipStream_t s, int nrows, const int* rows, const PcgScalars* S, const double* z, double* p) {
  if (nrows > 0) SGO_LAUNCH(k_update_p_rows, dim3(grid_for(nrows, kBlock)), dim3(kBlock), 0, s, nrows, rows, S, z, p);
}
bool halo_exchange(const HaloDev& H, hipStream_t s, double* data, int width, const int* idx, int idx_max, const HaloScalars& sc,
                   std::string* err) {
  if (!H.comm) return true;
  const int cnt = (data && idx) ? idx_max : 0;
  const size_t stride = (size_t)kHaloScalars + (size_t)cnt * width;
  if (stride > H.cap) {
    if (err) *err = "row-owner exchange: packet larger than the exchange buffers";
    if (H.failed) *H.failed = true;
    return false;
  }
  SGO_LAUNCH(k_halo_pack, dim3(grid_for((long long)cnt * width, kBlock)), dim3(kBlock), 0, s, (const double*)data, width,
             idx ? idx + (size_t)H.me * idx_max : (const int*)nullptr, cnt, H.send, sc);
  std::string e;
  if (!H.comm->allgather_f64(H.send, H.recv, stride, s, &e)) {
    if (err) *err = e;
    if (H.failed) *H.failed = true;
    return false;
  }
  SGO_LAUNCH(k_halo_unpack, dim3(grid_for((long long)cnt * width * H.G, kBlock)), dim3(kBlock), 0, s, data, width, idx, cnt, H.G, H.me,
             (const double*)H.recv, stride, H.gparts);
  return true;
}
bool halo_gather_slices(const HaloDev& H, hipStream_t s, double* vec, int width, std::string* err) {
  if (!H.comm || H.G <= 1) return true;
  const size_t stride = (size_t)width * (size_t)H.maxrows;
  if (stride > H.cap) {
    if (err) *err = "row-owner exchange: slice larger than the exchange buffers";
    if (H.failed) *H.failed = true;
    return false;
  }
  if (hipMemcpyAsync(H.send, vec + (size_t)width * (size_t)H.row0, sizeof(double) * (size_t)width * (size_t)(H.row1 - H.row0), hipMemcpyDeviceToDevice, s) != hipSuccess) {
    if (err) *err = "row-owner exchange: device copy failed";
    if (H.failed) *H.failed = true;
    return false;
  }
  std::string e;
  if (!H.comm->allgather_f64(H.send, H.recv, stride, s, &e)) {
    if (err) *err = e;
    if (H.failed) *H.failed = true;
    return false;
  }
  SGO_LAUNCH(k_slices_unpack, dim3(grid_for((long long)width * H.maxrows, kBlock)), dim3(kBlock), 0, s, vec, width, H.rank_row, H.G, H.me,
             (const double*)H.recv, stride);
  return true;
}

void launch_chi2(hipStream_t s, const EdgeListDev& el, int e0, int e1, const double* poses, double* e2_out,
                 double* partials, int* grid_out, const EdgeListDev* el2) {
  const int grid = grid_for(e1 - e0, kBlock);
  SGO_LAUNCH(k_chi2, dim3(grid), dim3(kBlock), 0, s, el, e0, e1, el2 ? *el2 : EdgeListDev(), poses, e2_out, partials);
  *grid_out = grid;
}
void launch_reduce2(hipStream_t s, const double* partials, int nparts, double* out2) {
  SGO_LAUNCH(k_reduce2, dim3(1), dim3(kBlock), 0, s, partials, nparts, out2);
}
void launch_linearize(hipStream_t s, const Sym0Dev& A, int g0, int g1, const EdgeSlotsDev& es, const double* poses,
                      double* dgb) {
  const int grid = grid_for(g1 - g0, kWavesPerBlock);
  SGO_LAUNCH(k_linearize, dim3(grid), dim3(kBlock), 0, s, A, g0, g1, es, poses, dgb);
}
void launch_finalize(hipStream_t s, const Sym0Dev& A, int row0, int row1, const double* dgb, double* b, double* x, double* r, double* z,
                     double* p, double* xs, double omega, double* partials, int* grid_out) {
  const int grid = grid_for(row1 - row0, kBlock);
  SGO_LAUNCH(k_finalize, dim3(grid), dim3(kBlock), 0, s, A, row0, row1, dgb, b, x, r, z, p, xs, omega, partials);
  *grid_out = grid;
}
int launch_diag_change(hipStream_t s, int row0, int row1, const double* dblk, double* dref, bool store_ref, double* partials) {
  const int grid = grid_for(row1 - row0, kBlock);
  SGO_LAUNCH(k_diag_change, dim3(grid), dim3(kBlock), 0, s, row0, row1, dblk, dref, store_ref ? 1 : 0, partials);
  return grid;
}
void launch_force_stop(hipStream_t s, PcgScalars* S, PcgScalars* mirror) { SGO_LAUNCH(k_force_stop, dim3(1), dim3(1), 0, s, S, mirror); }
void launch_set_probe(hipStream_t s, PcgScalars* S, int probe_k, double probe_max) {
  SGO_LAUNCH(k_set_probe, dim3(1), dim3(1), 0, s, S, probe_k, probe_max);
}
void launch_restart_scalars(hipStream_t s, PcgScalars* S, const double* rz_parts, int n_rz, int maxit, int keep_stop) {
  SGO_LAUNCH(k_restart_scalars, dim3(1), dim3(kBlock), 0, s, S, rz_parts, n_rz, maxit, keep_stop);
}
void launch_warm_start(hipStream_t s, int n3, const double* xp, const double* q, const double* b, double* x, double* r,
                       const double* xq_parts, int n_xq, const double* bx_parts, int n_bx) {
  SGO_LAUNCH(k_warm_start, dim3(grid_for(n3, kBlock)), dim3(kBlock), 0, s, n3, xp, q, b, x, r, xq_parts, n_xq, bx_parts, n_bx);
}
void launch_init_scalars(hipStream_t s, PcgScalars* S, const double* rz_parts, int n_rz, const double* bb_parts,
                         int n_bb, double tol, int maxit, double bb_ref, double tol_cap) {
  SGO_LAUNCH(k_init_scalars, dim3(1), dim3(kBlock), 0, s, S, rz_parts, n_rz, bb_parts, n_bb, tol, maxit, bb_ref, tol_cap);
}
void launch_edge_prepare(hipStream_t s, int E, const double* meas, const double* info, double* zinv, double* info_soa, size_t stride, size_t at) {
  SGO_LAUNCH(k_edge_prepare, dim3(grid_for(E, kBlock)), dim3(kBlock), 0, s, E, meas, info, zinv, info_soa, stride, at);
}
void launch_early_strength(hipStream_t s, const EdgeListDev& el, const double* poses, int n, const int* rowptr, const int* eidx,
                           const unsigned char* flags, const int* hrowptr, double* w) {
  if (n > 0) SGO_LAUNCH(k_row_strength, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, rowptr, eidx, flags, hrowptr, el, poses, w);
}
void launch_slot_expand(hipStream_t s, int k0, int k1, const int* eidx, const EdgeListDev& el, const EdgeSlotsDev& es) {
  SGO_LAUNCH(k_slot_expand, dim3(grid_for(k1 - k0, kBlock)), dim3(kBlock), 0, s, k0, k1, eidx, el, es);
}
int launch_spmv_ex(hipStream_t s, const BsrDev& A, int mode, const SpmvArgs& a) {
  const int grid = grid_for(A.ngrp, kWavesPerBlock);
  switch (mode) {
    case SPMV_AX: SGO_LAUNCH((k_spmv<SPMV_AX>), dim3(grid), dim3(kBlock), 0, s, A, a); break;
    case SPMV_RESID: SGO_LAUNCH((k_spmv<SPMV_RESID>), dim3(grid), dim3(kBlock), 0, s, A, a); break;
    case SPMV_JACOBI:
      SGO_LAUNCH((k_spmv<SPMV_JACOBI>), dim3(grid), dim3(kBlock), 0, s, A, a);
      break;
    case SPMV_JACOBI_P: SGO_LAUNCH((k_spmv<SPMV_JACOBI_P>), dim3(grid), dim3(kBlock), 0, s, A, a); break;
    case SPMV_PRE_RESID_S: SGO_LAUNCH((k_spmv<SPMV_PRE_RESID_S>), dim3(grid), dim3(kBlock), 0, s, A, a); break;
    case SPMV_PRE_RESID_ACC: SGO_LAUNCH((k_spmv<SPMV_PRE_RESID_ACC>), dim3(grid), dim3(kBlock), 0, s, A, a); break;
    case SPMV_AX_C: SGO_LAUNCH((k_spmv<SPMV_AX_C>), dim3(grid), dim3(kBlock), 0, s, A, a); break;
    default:
      SGO_LAUNCH((k_spmv<SPMV_PRE_RESID>), dim3(grid), dim3(kBlock), 0, s, A, a);
      break;
  }
  return grid;
}
int launch_spmv0(hipStream_t s, const Sym0Dev& A, int mode, const Spmv0Args& a) {
  const int grid = grid_for(A.ngrp, kWavesPerBlock);
  switch (mode) {
    case S0_AX: SGO_LAUNCH((k_spmv0<S0_AX>), dim3(grid), dim3(kBlock), 0, s, A, a); break;
    case S0_RESID: SGO_LAUNCH((k_spmv0<S0_RESID>), dim3(grid), dim3(kBlock), 0, s, A, a); break;
    default: SGO_LAUNCH((k_spmv0<S0_JACOBI>), dim3(grid), dim3(kBlock), 0, s, A, a); break;
  }
  return grid;
}
template <int MODE, int NT, bool F32>
void launch_spmv0t_inst(hipStream_t s, const Sym0Dev& A, const Tile0Dev& T, const Spmv0Args& a, int grid) {
  static int lds_allowed = 0;   // dynamic LDS beyond 64 KB needs the attribute once per kernel
  if (T.lds_bytes > 65536 && T.lds_bytes > lds_allowed) {
    constexpr int kDynMax = 160 * 1024 - 2048;   // the kernel's static LDS (scalars, dot partials) comes on top
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmv0t<MODE, NT, F32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kDynMax) == hipSuccess)
      lds_allowed = kDynMax;
  }
  SGO_LAUNCH((k_spmv0t<MODE, NT, F32>), dim3(grid), dim3(NT), (size_t)T.lds_bytes, s, A, T, a);
}
int launch_spmv0t(hipStream_t s, const Sym0Dev& A, const Tile0Dev& T, int mode, const Spmv0Args& a) {
  // as many tile workgroups as fit the CUs' LDS are resident; more tiles are walked in a loop
  const int per_cu = std::max(1, std::min(2, (160 * 1024) / std::max(T.lds_bytes + 512, 1)));
  long long g = std::min<long long>(T.ntile, 256LL * per_cu);
  if (g < 8) g = 8;
  const int grid = (int)((g + 7) / 8 * 8);
  // the preconditioner's two passes read the fp32 copy of the blocks when there is one; H p always the fp64 blocks
  const bool f32 = A.fblk != nullptr && !a.force_f64;
  switch (mode) {
    case S0_AX: launch_spmv0t_inst<S0_AX, 1024, false>(s, A, T, a, grid); break;
    case S0_RESID:
      if (f32) launch_spmv0t_inst<S0_RESID, 1024, true>(s, A, T, a, grid);
      else launch_spmv0t_inst<S0_RESID, 1024, false>(s, A, T, a, grid);
      break;
    default:
      if (f32) launch_spmv0t_inst<S0_JACOBI, 1024, true>(s, A, T, a, grid);
      else launch_spmv0t_inst<S0_JACOBI, 1024, false>(s, A, T, a, grid);
      break;
  }
  return grid;
}
void launch_update_xr(hipStream_t s, int n, PcgScalars* S, const double* pq_parts, int n_pq, const double* dinv,
                      const double* p, const double* q, double* x, double* r, double* z, double* xs, double omega,
                      double* partials, int* grid_out) {
  const int grid = grid_for(n, kBlock);
  SGO_LAUNCH(k_update_xr, dim3(grid), dim3(kBlock), 0, s, n, S, pq_parts, n_pq, dinv, p, q, x, r, z, xs, omega, partials);
  if (grid_out) *grid_out = grid;
}
void launch_update_p(hipStream_t s, int n, PcgScalars* S, const double* rz_parts, int n_rz, const double* rr_parts,
                     int n_rr, const double* zq_parts, const double* z, double* p, const RecDev& rec) {
  const int grid = grid_for(3LL * n, kBlock);
  SGO_LAUNCH(k_update_p, dim3(grid), dim3(kBlock), 0, s, 3 * n, S, rz_parts, n_rz, rr_parts, n_rr, zq_parts, z, p, rec);
}
// Score-weighted sample covariance of a scan-match window and its inverse: one wave per match,
// lanes stride over the window's samples (k fastest, as the reference's loops), ten fp64 sums
// combined by a fixed-shape wave reduction.
__global__ __launch_bounds__(kBlock) void k_closure_cov(int n, const sgo_match_window* __restrict__ win,
                                                        const float* __restrict__ scores, double* __restrict__ cov,
                                                        double* __restrict__ info) {
  const int lane = threadIdx.x & 63;
  for (int q = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); q < n; q += gridDim.x * kWavesPerBlock) {
    const sgo_match_window W = win[q];
    const int nw = 2 * W.w_size + 1, nk = 2 * W.scan_window + 1, total = nw * nw * nk;
    const float* sc = scores + W.score_offset;
    double a[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // Kxx Kxy Kxt Kyy Kyt Ktt ux uy ut s
    for (int e = lane; e < total; e += 64) {
      const int kk = e % nk, jj = (e / nk) % nw, ii = e / (nk * nw);
      const int i = W.x_index_offset - W.w_size + ii, j = W.y_index_offset - W.w_size + jj;
      const int k = W.scan_index - W.scan_window + kk;
      const double x = -j * W.resolution, y = -i * W.resolution;
      const double t = (k - W.num_angular_perturbations) * W.angular_step;
      const double w = (double)sc[e];
      a[0] += x * x * w; a[1] += x * y * w; a[2] += x * t * w;
      a[3] += y * y * w; a[4] += y * t * w; a[5] += t * t * w;
      a[6] += x * w; a[7] += y * w; a[8] += t * w;
      a[9] += w;
    }
#pragma unroll
    for (int c = 0; c < 10; ++c) a[c] = wave_sum(a[c]);
    if (lane == 0) {
      const double d = 1.0 / a[9], d2 = d * d;
      const double c00 = d * a[0] - d2 * a[6] * a[6], c01 = d * a[1] - d2 * a[6] * a[7], c02 = d * a[2] - d2 * a[6] * a[8];
      const double c11 = d * a[3] - d2 * a[7] * a[7], c12 = d * a[4] - d2 * a[7] * a[8], c22 = d * a[5] - d2 * a[8] * a[8];
      double* C = cov + 9 * (size_t)q;
      C[0] = c00; C[1] = c01; C[2] = c02;
      C[3] = c01; C[4] = c11; C[5] = c12;
      C[6] = c02; C[7] = c12; C[8] = c22;
      // inverse by cofactors / determinant
      const double k00 = c11 * c22 - c12 * c12, k01 = c02 * c12 - c01 * c22, k02 = c01 * c12 - c02 * c11;
      const double k11 = c00 * c22 - c02 * c02, k12 = c01 * c02 - c00 * c12, k22 = c00 * c11 - c01 * c01;
      const double id = 1.0 / (c00 * k00 + c01 * k01 + c02 * k02);
      double* I = info + 9 * (size_t)q;
      I[0] = k00 * id; I[1] = k01 * id; I[2] = k02 * id;
      I[3] = k01 * id; I[4] = k11 * id; I[5] = k12 * id;
      I[6] = k02 * id; I[7] = k12 * id; I[8] = k22 * id;
    }
  }
}

void launch_closure_cov(hipStream_t s, int n, const sgo_match_window* win, const float* scores, double* cov,
                        double* info) {
  const int grid = grid_for(n, kWavesPerBlock);
  SGO_LAUNCH(k_closure_cov, dim3(grid), dim3(kBlock), 0, s, n, win, scores, cov, info);
}
void launch_pose_update(hipStream_t s, int n, const int* free_id, const double* x, double* poses) {
  const int grid = grid_for(n, kBlock);
  SGO_LAUNCH(k_pose_update, dim3(grid), dim3(kBlock), 0, s, n, free_id, x, poses);
}
void launch_dot(hipStream_t s, int n3, const double* a, const double* b, double* partials, const PcgScalars* S,
                int* grid_out) {
  const int grid = grid_for(n3, kBlock);
  SGO_LAUNCH(k_dot, dim3(grid), dim3(kBlock), 0, s, n3, a, b, partials, S);
  if (grid_out) *grid_out = grid;
}
void launch_precond_bj(hipStream_t s, int n, const double* dinv, const double* r, double* z, double scale) {
  const int grid = grid_for(n, kBlock);
  SGO_LAUNCH(k_precond_bj, dim3(grid), dim3(kBlock), 0, s, n, dinv, r, z, scale);
}

}  // namespace sgo
