// sgo_device.h -- device-side helpers shared by the HIP translation units (wave64 reductions,
// the wavefront segmented scan, the XCD-aware group walk).
#pragma once
#include "sgo_internal.h"

namespace sgo {
namespace {

constexpr double kPi = 3.14159265358979323846;

// g2o::normalize_theta, branch structure kept literal (result in [-pi, pi)).
__device__ __forceinline__ double norm_theta(double t) {
  if (t >= -kPi && t < kPi) return t;
  double m = floor(t / (2 * kPi));
  t = t - m * 2 * kPi;
  if (t >= kPi) t -= 2 * kPi;
  if (t < -kPi) t += 2 * kPi;
  return t;
}

// EdgeSE2::computeError with the cached inverse measurement Zi:
//   e = toVector( Zi * (Xi^-1 * Xj) ),  SE2 algebra literal (g2o SE2::operator* / inverse).
__device__ __forceinline__ void edge_error(double xi, double yi, double ti, double xj, double yj, double tj,
                                           double zx, double zy, double zt, double sz, double cz,
                                           double (&e)[3]) {
  const double tin = norm_theta(-ti);
  double s1, c1;
  sincos(tin, &s1, &c1);
  const double ix = c1 * (-xi) - s1 * (-yi);
  const double iy = s1 * (-xi) + c1 * (-yi);
  const double dx = ix + c1 * xj - s1 * yj;
  const double dy = iy + s1 * xj + c1 * yj;
  const double dth = norm_theta(tin + tj);
  e[0] = zx + cz * dx - sz * dy;
  e[1] = zy + sz * dx + cz * dy;
  e[2] = norm_theta(zt + dth);
}

// RobustKernelDCS::robustify; phi < 0: no kernel.
__device__ __forceinline__ void dcs(double e2, double phi, double* rho0, double* rho1) {
  double r0 = e2, r1 = 1.0;
  if (phi >= 0.0) {
    const double scale = (2.0 * phi) / (phi + e2);
    if (!(scale >= 1.0)) {
      r0 = scale * e2 * scale;
      r1 = scale * scale;
    }
  }
  *rho0 = r0;
  *rho1 = r1;
}

__device__ __forceinline__ double wave_sum(double v);   // (below, with the DPP helpers)

// Block-wide sums of N values; thread 0 stores them to out[i * stride + blockIdx.x].
template <int N>
__device__ __forceinline__ void block_sum_store(double (&v)[N], double* out, int stride) {
  __shared__ double sm[N][kWavesPerBlock];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = wave_sum(v[i]);
    if (lane == 0) sm[i][w] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double s = sm[i][0];
#pragma unroll
      for (int k = 1; k < kWavesPerBlock; ++k) s += sm[i][k];
      out[(size_t)i * stride + blockIdx.x] = s;
    }
  }
}

// Deterministic sum of nparts partials by one block (fixed order), result in every thread.
__device__ __forceinline__ double block_reduce_parts(const double* parts, int nparts) {
  __shared__ double sm2[kWavesPerBlock];
  __shared__ double res;
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kBlock) s += parts[i];
  s = wave_sum(s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm2[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = sm2[0];
#pragma unroll
    for (int k = 1; k < kWavesPerBlock; ++k) t += sm2[k];
    res = t;
  }
  __syncthreads();
  return res;
}

// N such sums at once (same summation order as block_reduce_parts for each of them, so the values
// are bit-identical) with ONE LDS exchange: 2 barriers instead of 3 N.  parts[c] == nullptr gives 0.
template <int N>
__device__ __forceinline__ void block_reduce_parts_n(const double* const (&parts)[N], const int (&cnt)[N],
                                                     double (&out)[N]) {
  __shared__ double smn[N][kWavesPerBlock];
  double s[N];
#pragma unroll
  for (int c = 0; c < N; ++c) {
    s[c] = 0.0;
    if (parts[c])
      for (int i = threadIdx.x; i < cnt[c]; i += kBlock) s[c] += parts[c][i];
  }
#pragma unroll
  for (int c = 0; c < N; ++c) s[c] = wave_sum(s[c]);
  __syncthreads();  // readers of an earlier call are done with smn
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int c = 0; c < N; ++c) smn[c][threadIdx.x >> 6] = s[c];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < N; ++c) {
    double t = smn[c][0];
#pragma unroll
    for (int k = 1; k < kWavesPerBlock; ++k) t += smn[c][k];
    out[c] = t;
  }
}

// Inclusive segmented scan over the wave: lanes with equal `row` that are contiguous form a
// segment; after the scan the LAST lane of a segment holds the segment sum.
//
// Cross-lane traffic goes through DPP moves on the vector ALU, not through ds_bpermute: the LDS crossbar
// is what the tile kernel's staging and the 40-odd permutes of a shuffle-based scan would otherwise
// share (measured: SQ_WAIT_INST_LDS 18 % of the wave cycles of k_spmv0t with __shfl_up).  Steps 1, 2, 4, 8
// stay inside a row of 16 lanes (row_shr); the carries into rows 1 / 3 and then 2 / 3 come from lane 15 / 47
// and lane 31 (row_bcast15 / row_bcast31).  The two carry steps rely on what every caller guarantees:
// the keys of the ACTIVE lanes are non-decreasing along the wave (slots sorted by target), so a lane whose
// key equals that of the last lane of the previous row(s) belongs to a segment that spans everything in
// between.  Inactive lanes carry unique negative keys and never match.
// (Full row mask: bound_ctrl makes the lanes without a source lane read 0, so no lane keeps `old` and the compiler has no
// destination to initialise -- two moves per double and step less.  A key of 0 can then match row 0 at the start of a row
// of 16 lanes, where the value that comes with it is 0 as well: nothing is added.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_int(int old, int v) {
  if (ROW_MASK == 0xF) return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
// key of the next lane (lane 63: 0), by a wavefront shift on the vector ALU instead of a permute through the LDS crossbar
__device__ __forceinline__ int next_lane_key(int key) { return __builtin_amdgcn_update_dpp(0, key, 0x130, 0xF, 0xF, true); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_double(double v) {
  const int lo = dpp_int<CTRL, ROW_MASK>(0, __double2loint(v)), hi = dpp_int<CTRL, ROW_MASK>(0, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// Sum over the wave, the same in every lane: an inclusive scan by DPP moves on the vector ALU (the total ends up in lane
// 63) and two v_readlane.  (The shuffle tree it replaces went through ds_bpermute: twelve LDS-crossbar round trips in a
// dependent chain of six, in the prologue of every kernel that re-reduces partial sums and at the end of every kernel
// that leaves some.)
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_double<0x111, 0xF>(v);   // row_shr:1
  v += dpp_double<0x112, 0xF>(v);   // row_shr:2
  v += dpp_double<0x114, 0xF>(v);   // row_shr:4
  v += dpp_double<0x118, 0xF>(v);   // row_shr:8
  v += dpp_double<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
  v += dpp_double<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

template <int N, int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_scan_step(int row, double (&v)[N]) {
  constexpr int kNoKey = (int)0x80000000;   // lanes without a source lane see a key no lane has
  // (v += f u with f = 1.0 / 0.0 instead of a select and an add: one fused multiply-add, bit-identical for finite u)
  const double f = dpp_int<CTRL, ROW_MASK>(kNoKey, row) == row ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double u = dpp_double<CTRL, ROW_MASK>(v[i]);
    v[i] = fma(f, u, v[i]);
  }
}
template <int N>
__device__ __forceinline__ void seg_scan(int row, double (&v)[N], int /*lane*/) {
  seg_scan_step<N, 0x111, 0xF>(row, v);   // row_shr:1
  seg_scan_step<N, 0x112, 0xF>(row, v);   // row_shr:2
  seg_scan_step<N, 0x114, 0xF>(row, v);   // row_shr:4
  seg_scan_step<N, 0x118, 0xF>(row, v);   // row_shr:8
  seg_scan_step<N, 0x142, 0xA>(row, v);   // row_bcast15 into rows 1 and 3
  seg_scan_step<N, 0x143, 0xC>(row, v);   // row_bcast31 into rows 2 and 3
}

// Row-major 3x3 block of LOGICAL slot k of a level's operator: from the slot-indexed pair-SoA array on
// the coarse levels; on level 0 through ref[] from the symmetric storage (diagonal blocks in symmetric
// packing, off-diagonal blocks stored once and transposed for the other endpoint's row).
__device__ __forceinline__ void load_block(const BsrDev& A, size_t k, double (&b)[9]) {
  // All three sources fill nine named scalars and ONE sequence of stores hands them to the caller's array.  (With a store
  // sequence per branch the optimiser merges the branches' last stores into one store through a phi of POINTERS -- b[8] on
  // two paths, b[7] on the third -- and the caller's array can then no longer live in registers: the Galerkin products,
  // k_p_values and k_block_norms went through scratch memory for it.)
  double v0, v1, v2, v3, v4, v5, v6, v7, v8;
  if (A.ref == nullptr) {
    const size_t ns = (size_t)A.nslot;
    v0 = A.blk[blk_at(0, k, ns)]; v1 = A.blk[blk_at(1, k, ns)]; v2 = A.blk[blk_at(2, k, ns)];
    v3 = A.blk[blk_at(3, k, ns)]; v4 = A.blk[blk_at(4, k, ns)]; v5 = A.blk[blk_at(5, k, ns)];
    v6 = A.blk[blk_at(6, k, ns)]; v7 = A.blk[blk_at(7, k, ns)]; v8 = A.blk[blk_at(8, k, ns)];
  } else {
    const int r = A.ref[k];
    if (r < 0) {
      const double* d = A.dblk + 6 * (size_t)(~r);
      v0 = d[0]; v1 = d[1]; v2 = d[2];
      v3 = v1; v4 = d[3]; v5 = d[4];
      v6 = v2; v7 = v5; v8 = d[5];
    } else {
      const size_t u = (size_t)(r >> 1), nu = (size_t)A.nus;
      const double2* __restrict__ bp = reinterpret_cast<const double2*>(A.ublk);
      const double2 p0 = bp[u], p1 = bp[nu + u], p2 = bp[2 * nu + u], p3 = bp[3 * nu + u];
      const bool tr = r & 1;
      // stored row-major p0.x p0.y p1.x | p1.y p2.x p2.y | p3.x p3.y ublk8 ; transposed: swap (1,3) (2,6) (5,7)
      v0 = p0.x; v4 = p2.x; v8 = A.ublk8[u];
      v1 = tr ? p1.y : p0.y; v3 = tr ? p0.y : p1.y;
      v2 = tr ? p3.x : p1.x; v6 = tr ? p1.x : p3.x;
      v5 = tr ? p3.y : p2.y; v7 = tr ? p2.y : p3.y;
    }
  }
  b[0] = v0; b[1] = v1; b[2] = v2; b[3] = v3; b[4] = v4; b[5] = v5; b[6] = v6; b[7] = v7; b[8] = v8;
}

// Map (block, wave) -> first group and stride so that XCD x (blocks with blockIdx % 8 == x under
// the observed round-robin dispatch; speed only, never correctness) walks the contiguous band
// [x * ngrp / 8, (x + 1) * ngrp / 8) of groups.
// (nblocks: the workgroups that walk -- a multiple of 8; a launch may carry more behind them with another job)
__device__ __forceinline__ void group_walk_n(int ngrp, int nblocks, int* first, int* last, int* stride) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = nblocks >> 3;
  const int lo = (int)(((long long)ngrp * xcd) >> 3), hi = (int)(((long long)ngrp * (xcd + 1)) >> 3);
  *first = lo + slot * kWavesPerBlock + (threadIdx.x >> 6);
  *last = hi;
  *stride = per_xcd * kWavesPerBlock;
}
// (block: this workgroup's index among the nblocks walkers, when they do not start at blockIdx 0)
__device__ __forceinline__ void group_walk_b(int ngrp, int nblocks, int block, int* first, int* last, int* stride) {
  const int xcd = block & 7, slot = block >> 3, per_xcd = nblocks >> 3;
  const int lo = (int)(((long long)ngrp * xcd) >> 3), hi = (int)(((long long)ngrp * (xcd + 1)) >> 3);
  *first = lo + slot * kWavesPerBlock + (threadIdx.x >> 6);
  *last = hi;
  *stride = per_xcd * kWavesPerBlock;
}
__device__ __forceinline__ void group_walk(int ngrp, int* first, int* last, int* stride) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int lo = (int)(((long long)ngrp * xcd) >> 3), hi = (int)(((long long)ngrp * (xcd + 1)) >> 3);
  *first = lo + slot * kWavesPerBlock + (threadIdx.x >> 6);
  *last = hi;
  *stride = per_xcd * kWavesPerBlock;
}

// Body of the wave-group level-0 product k_spmv0 (sgo_kernels.hip; modes and storage described there): the first
// `nblocks` workgroups of a launch walk the groups -- the launch may carry workgroups with another job behind them
// (sgo_amg.hip: the folded cycle's level-0 pass and its restriction in one launch).
template <int MODE>
__device__ __forceinline__ void spmv0_groups(const Sym0Dev& A, const Spmv0Args& a, int nblocks) {
  // On the graphs that take this kernel a launch is a chain of dependent round trips (~0.3 us each on top of the 1.8-us
  // node of a replayed hipGraph: scripts/micro/graph_floor.hip), so the chain is kept short: the first group's descriptors
  // are requested before the stop flag is waited for, a slot's column with its meta byte, and the row's own data (diagonal
  // block, operand, right-hand side, block-diagonal inverse) before the segmented scan instead of after it.
  const int lane = threadIdx.x & 63;
  const size_t nu = (size_t)A.nus;
  const double2* __restrict__ bp = reinterpret_cast<const double2*>(A.ublk);
  double dotacc[2] = {0.0, 0.0};
  const int ulo = a.u1 > 0 ? a.u0 : 0, uhi = a.u1 > 0 ? a.u1 : A.ngrp;
  int g, gend, gstride;
  group_walk_n(uhi - ulo, nblocks, &g, &gend, &gstride);
  g += ulo;
  gend += ulo;
  int f_gb = 0, f_ge = 0, f_r0 = 0, f_ob = 0, f_tb = 0;
  if (g < gend) {
    f_gb = A.grp[g]; f_ge = A.grp[g + 1]; f_r0 = A.grow[g]; f_ob = A.gown[g]; f_tb = A.gtr[g];
  }
  if (a.S && a.S->stop) return;
  for (bool first = true; g < gend; g += gstride, first = false) {
    const int gb = first ? f_gb : A.grp[g], ge = first ? f_ge : A.grp[g + 1], r0 = first ? f_r0 : A.grow[g];
    int ob = first ? f_ob : A.gown[g], tb = first ? f_tb : A.gtr[g];
    double acc[3] = {0.0, 0.0, 0.0};
    int row = -1 - lane;
    for (int kb = gb; kb < ge; kb += 64) {
      const int k = kb + lane;
      const bool active = k < ge;
      const int m = active ? (int)A.meta[k] : (kSlotNoBlock << 6);
      const int cj = active ? A.col[k] : 0;
      const int type = m >> 6;
      const unsigned long long omask = __ballot(type == kSlotOwned), tmask = __ballot(type == kSlotTransposed);
      const int orank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(omask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)omask, 0u));
      const int trank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(tmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)tmask, 0u));
      int idx = ob + orank;
      const bool tr = type == kSlotTransposed;
      if (tr) idx = A.tref[tb + trank];
      ob += __popcll(omask);
      tb += __popcll(tmask);
      if (active) row = r0 + (m & 63);
      if (type != kSlotNoBlock) {
        const size_t c = 3 * (size_t)cj;
        const double x0 = a.x[c], x1 = a.x[c + 1], x2 = a.x[c + 2];
        const double2 p0 = bp[idx], p1 = bp[nu + idx], p2 = bp[2 * nu + idx], p3 = bp[3 * nu + idx];
        const double b8 = A.ublk8[idx];
        // row-major b0..b8 = p0.x p0.y p1.x | p1.y p2.x p2.y | p3.x p3.y b8 ; transposed: swap (1,3) (2,6) (5,7)
        const double m01 = tr ? p1.y : p0.y, m02 = tr ? p3.x : p1.x;
        const double m10 = tr ? p0.y : p1.y, m12 = tr ? p3.y : p2.y;
        const double m20 = tr ? p1.x : p3.x, m21 = tr ? p2.y : p3.y;
        acc[0] += p0.x * x0 + m01 * x1 + m02 * x2;
        acc[1] += m10 * x0 + p2.x * x1 + m12 * x2;
        acc[2] += m20 * x0 + m21 * x1 + b8 * x2;
      }
    }
    // the row's own data, requested by every lane of the row (one address per row) before the scan
    double dd0 = 0, dd1 = 0, dd2 = 0, dd3 = 0, dd4 = 0, dd5 = 0, s0 = 0, s1 = 0, s2 = 0, rb0 = 0, rb1 = 0, rb2 = 0;
    double di0 = 0, di1 = 0, di2 = 0, di3 = 0, di4 = 0, di5 = 0;
    if (row >= 0) {
      const size_t o = 3 * (size_t)row;
      const double* dd = A.dblk + 6 * (size_t)row;
      dd0 = dd[0]; dd1 = dd[1]; dd2 = dd[2]; dd3 = dd[3]; dd4 = dd[4]; dd5 = dd[5];
      s0 = a.x[o]; s1 = a.x[o + 1]; s2 = a.x[o + 2];
      if (MODE != S0_AX) {
        rb0 = a.b[o]; rb1 = a.b[o + 1]; rb2 = a.b[o + 2];
      }
      if (MODE == S0_JACOBI) {
        const double* di = A.dinv + 6 * (size_t)row;
        di0 = di[0]; di1 = di[1]; di2 = di[2]; di3 = di[3]; di4 = di[4]; di5 = di[5];
      }
    }
    seg_scan<3>(row, acc, lane);
    const int rn = next_lane_key(row);
    if (row >= 0 && (lane == 63 || rn != row)) {
      const size_t o = 3 * (size_t)row;
      double o0 = acc[0] + dd0 * s0 + dd1 * s1 + dd2 * s2;
      double o1 = acc[1] + dd1 * s0 + dd3 * s1 + dd4 * s2;
      double o2 = acc[2] + dd2 * s0 + dd4 * s1 + dd5 * s2;
      if (MODE != S0_AX) {
        const double t0 = rb0 - o0, t1 = rb1 - o1, t2 = rb2 - o2;
        if (MODE == S0_JACOBI) {
          o0 = s0 + a.omega * (di0 * t0 + di1 * t1 + di2 * t2);
          o1 = s1 + a.omega * (di1 * t0 + di3 * t1 + di4 * t2);
          o2 = s2 + a.omega * (di2 * t0 + di4 * t1 + di5 * t2);
        } else {
          o0 = t0; o1 = t1; o2 = t2;
        }
      }
      a.y[o] = o0; a.y[o + 1] = o1; a.y[o + 2] = o2;
      if (a.dotA) dotacc[0] += a.dotA[o] * o0 + a.dotA[o + 1] * o1 + a.dotA[o + 2] * o2;
      if (a.dotA2) dotacc[1] += a.dotA2[o] * o0 + a.dotA2[o + 1] * o1 + a.dotA2[o + 2] * o2;
    }
  }
  if (a.partials) block_sum_store<2>(dotacc, a.partials, kMaxPartials);
}

}  // namespace
}  // namespace sgo
