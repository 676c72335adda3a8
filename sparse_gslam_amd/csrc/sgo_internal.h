// sgo_internal.h -- shared declarations of libsgo (host side <-> HIP kernels).
// Not part of the public ABI (that is include/sgo.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "sgo.h"
#include "sgo_hostpool.h"

namespace sgo {

// ----------------------------------------------------------------------------------------
// Block-CSR matrix with 3x3 fp64 blocks in "slot list sorted by row" form (the LOGICAL view of a
// multigrid level's operator, and the storage of every level but the finest):
//   * slot k belongs to block row row[k] and multiplies column col[k];
//   * the FIRST slot of every row is that row's diagonal block;
//   * blk is SoA over slots in component pairs (blk_at below): component c (row-major 3x3) of
//     slot k is blk[blk_at(c, k, nslot)], so a wave reading 64 consecutive slots issues four
//     fully coalesced 1-KiB loads (16 B per lane) and one 512-byte load;
//   * grp[g] .. grp[g+1] is the slot range of wave-group g.  Groups are row aligned: a group
//     holds whole rows (<= 64 slots) or exactly one long row (> 64 slots), so the segmented
//     reduction of a row never leaves the wave and needs neither LDS hand-off nor atomics.
// The finest level keeps its VALUES in the symmetric storage of Sym0Dev (every off-diagonal block
// once); its logical view then has blk == nullptr and ref[k] says where slot k's block lives:
//   ref[k] < 0   diagonal block of row ~ref[k]  (dblk, symmetric packing)
//   ref[k] >= 0  off-diagonal block ublk[ref[k] >> 1], transposed when ref[k] & 1.
// Only the per-GN-iteration set-up kernels of the multigrid hierarchy read level 0 through this
// view (load_block in sgo_device.h); the products of the solve run on Sym0Dev directly.
// ----------------------------------------------------------------------------------------
struct BsrDev {
  int n = 0;          // block rows
  int nslot = 0;      // slots (diagonal slots included)
  int ngrp = 0;       // wave groups
  int* row = nullptr;
  int* col = nullptr;
  int* grp = nullptr;     // [ngrp + 1]
  int* rowptr = nullptr;  // [n + 1]
  double* blk = nullptr;  // 9 * nslot doubles, layout blk_at()   (nullptr on level 0)
  double* dinv = nullptr; // [n][6] inverse of the diagonal block, symmetric packing
  // level 0 only: references into the symmetric storage
  const int* ref = nullptr;     // [nslot]
  const double* ublk = nullptr; // pair-SoA over the stored off-diagonal blocks (components 0..7; see Sym0Dev)
  const double* ublk8 = nullptr;// component 8
  size_t nu = 0;                // blocks the symmetric storage holds in all
  size_t nus = 0;               // pair stride of ublk (= nu on one GPU; the blocks THIS rank holds in row-owner mode)
  const double* dblk = nullptr; // [n][6]
};

// Block storage ("pair-SoA"): components (0,1), (2,3), (4,5), (6,7) of slot k are adjacent pairs --
// pair p of slot k sits at double offset 2 * (p * nslot + k) -- and component 8 is at 8 * nslot + k.
// A wave reading 64 consecutive slots issues four 16-byte-per-lane loads of one contiguous KiB
// each plus one 512-B load, instead of nine 8-byte-per-lane loads.
__host__ __device__ inline size_t blk_at(int c, size_t k, size_t ns) {
  return c < 8 ? 2 * ((size_t)(c >> 1) * ns + k) + (size_t)(c & 1) : 8 * ns + k;
}

// ----------------------------------------------------------------------------------------
// Level-0 Hessian in SYMMETRIC storage: H is symmetric, so every off-diagonal block is stored once
// (72 B + indices per edge) and serves both y_i += B x_j and y_j += B^T x_i.
//   * rows are numbered in Hilbert-curve order of the poses (sgo_plan.cpp), which puts the two
//     endpoints of almost every edge -- odometry steps and closures between nearby poses alike --
//     a few hundred rows apart;
//   * the compact slot list has, per row, one slot per incident off-diagonal block: the slot is
//     OWNED (a block oriented for this row is stored with it; owned slots of consecutive rows are
//     consecutive in ublk, so their loads stream) or TRANSPOSED (the block is stored only with the other
//     endpoint's row, which lies in the same tile -- Tile0Dev below -- and has the lower index).  Pairs of
//     rows in different tiles own a copy each.  The tile kernel serves a transposed slot through LDS;
//     the wave-group kernel k_spmv0 (graphs without a tile view) fetches the owner's block through tref
//     and multiplies by its transpose;
//   * an edge to a FIXED vertex has a slot without a block (it contributes to the row's diagonal
//     block and right-hand side only), a row without any edge to a free vertex gets one such slot;
//   * diagonal blocks live apart in symmetric packing (dblk, 48 B per row) and are applied by the
//     row's last lane.
// Per slot: col (4 B) and meta (1 B: row - grow[g] in bits 0..5, type in bits 6..7); per transposed
// slot: tref (4 B).  The storage index of an owned slot is gown[g] + its rank among the group's owned
// slots (ballot + popcount), that of a transposed slot tref[gtr[g] + rank].
// ----------------------------------------------------------------------------------------
enum : int { kSlotOwned = 0, kSlotTransposed = 1, kSlotNoBlock = 2 };
struct Sym0Dev {
  int n = 0;        // rows
  int npairs = 0;   // connected pairs of free rows = the blocks a fully symmetric storage holds (SURVEY 8(d)'s E)
  int nu = 0;       // stored off-diagonal blocks (pairs inside a tile once, pairs across tiles twice)
  int ncs = 0;      // compact slots
  int ngrp = 0;
  int* col = nullptr;            // [ncs]
  unsigned char* meta = nullptr; // [ncs]
  int* tref = nullptr;           // [number of transposed slots]
  int* grp = nullptr;            // [ngrp + 1] slot range of wave group g (whole rows)
  int* grow = nullptr;           // [ngrp] first row of group g
  int* gown = nullptr;           // [ngrp] storage index of the group's first owned slot
  int* gtr = nullptr;            // [ngrp] position in tref of the group's first transposed slot
  // The block arrays are indexed by GLOBAL storage numbers.  In multi-GPU row-owner mode a rank holds the blocks (and the
  // per-slot / per-block index arrays) of its own rows only: the arrays are allocated for the rank's range and the base
  // pointers are shifted so that global numbers address them -- which is why component 8 has a base pointer of its own
  // (pair p of block u sits at ublk + 2 (p nus + u), component 8 at ublk8 + u) and the pair stride is a field (nus).
  int nus = 0;                   // pair stride of ublk: stored blocks held by this rank (= nu on one GPU)
  double* ublk = nullptr;        // pair-SoA, components 0..7
  double* ublk8 = nullptr;       // component 8
  // fp32 copy of the stored blocks for the PRECONDITIONER's two level-0 passes (residual pass and post-smoothing sweep of
  // the multigrid cycle; H p of the CG recurrence, residuals and results stay fp64: SURVEY.md section 7 names this
  // mitigation): components (0..3), (4..7) as two float4 per block -- 16-byte loads per lane as the fp64 pairs -- and
  // component 8; same numbering, stride and shifting as ublk.  nullptr: the passes read ublk (env SGO_PRECOND_F32=0).
  float* fblk = nullptr;         // quad-SoA: quad q of block u at fblk + 4 (q nus + u)
  float* fblk8 = nullptr;
  double* dblk = nullptr;        // [n][6]
  double* dinv = nullptr;        // [n][6]
};

// ----------------------------------------------------------------------------------------
// Tile view of the level-0 storage for the products of the solve (k_spmv0t).  Measured on MI355X
// (scripts/micro/ta_micro.hip): a wave64 global load costs >= 26 cycles of its CU's address path whatever
// it moves, and ~2.4 cycles per lane when the lanes hit different cache lines, even on L2 hits -- gathers,
// not bytes, bound this kernel class.  So the rows are cut into TILES of consecutive (Hilbert-ordered)
// rows, one workgroup per tile at a time, and every global access of the hot loop is a coalesced stream:
//   * a pair of rows INSIDE one tile stores its block once (with the lower row); the other row receives
//     B^T x through an LDS staging slot: symmetric storage, 72 B per edge;
//   * a pair that straddles two tiles (10-30 % in Hilbert order) stores the block with BOTH rows, each
//     copy oriented for its row, so that no block is ever gathered;
//   * phase 0  the tile's slice of the operand and its HALO (the distinct columns outside the tile, each
//              fetched once per tile instead of once per slot) go to LDS;
//     phase 1  one lane per stored block of the tile's rows: block streamed (5 coalesced loads), operand
//              from LDS, u = B x_c segment-summed per row, v = B^T x_r to the twin's staging slot (vpos);
//     phase 2  one thread per row sums the row's staged entries in a fixed order, adds the owned part and
//              the diagonal block's product, applies the mode's epilogue and stores the row (coalesced).
// No atomics, every sum in a fixed order (bitwise reproducible).
// ----------------------------------------------------------------------------------------
struct TileDesc {
  int row0, row1;   // rows [row0, row1)
  int g0, g1;       // phase-1 wave groups (whole rows of owned slots, <= 64 slots or one long row)
  int h0, h1;       // halo columns hcol[h0 .. h1)
  int e0;           // first staged entry (transposed slot number) of the tile
  int nstaged;      // staged entries
};
struct Tile0Dev {
  int ntile = 0;
  int lds_bytes = 0;          // dynamic LDS per workgroup (largest tile)
  TileDesc* tile = nullptr;
  unsigned int* cv = nullptr;       // [nu] owned slot u: LDS operand index (local row, or rows + halo number) in
                                    // bits 0..15, tile-relative staging slot of its twin in bits 16..31 (0xFFFF: none)
  unsigned char* off1 = nullptr;    // [nu] row - grow1[g]
  int* grp1 = nullptr;              // [ng1 + 1]
  int* grow1 = nullptr;             // [ng1]
  int* trowptr = nullptr;           // [n + 1] staged entries of row r: [trowptr[r], trowptr[r+1])  (global numbering)
  int* hcol = nullptr;              // halo columns (global rows) of all tiles
  int* hfirst = nullptr;            // [ntile][hstride] the first hstride halo columns of every tile at a fixed stride (-1 beyond the
  int hstride = 0;                  // tile's own): their address does not depend on the tile descriptor, one round trip less
};

// Edge operands aligned with the level-0 compact slots (SoA over slots).  For a slot of row r that
// came from edge e = (i, j):  dir = 0 when r is the i side (row Jacobian A), 1 when r is the j side
// (row Jacobian B).
enum : int { kSlotDir = 1, kSlotNoEdge = 2, kSlotFixedCol = 4 };   // FixedCol: the edge's other endpoint is fixed (no block)
struct EdgeSlotsDev {
  int stride = 0;         // component stride of zinv / info: the slots held (= ncs on one GPU; this rank's slots in row-owner mode,
                          // with base pointers shifted so that global slot numbers address them)
  int* vi = nullptr;      // vertex id of EdgeSE2::vertices()[0]
  int* vj = nullptr;      // vertex id of EdgeSE2::vertices()[1]
  unsigned char* flags = nullptr;
  double* zinv = nullptr; // [3][stride] cached inverse measurement (EdgeSE2::setMeasurement)
  double* info = nullptr; // [6][stride]
  double* phi = nullptr;  // [ncs]
};

// Original edge list (edge order of sgo_set_graph_se2), SoA, for chi2 / per-edge chi2.
struct EdgeListDev {
  int E = 0;              // edges = component stride of zinv / info (the overlay's list: its capacity, see cnt)
  int cnt = 0;            // valid edges when the list is kept at a fixed capacity (OverlayDev::el); otherwise unused
  int* vi = nullptr;
  int* vj = nullptr;
  double* zinv = nullptr; // [3][E]
  double* info = nullptr; // [6][E]
  double* phi = nullptr;  // [E]
};

// Device-resident scalars of the PCG recurrence (one cache line; read uniformly by kernels).
struct PcgScalars {
  double rz;      // r.z of the current iterate
  double pq;      // p.Hp
  double rr;      // r.r
  double bb;      // b.b
  double alpha;
  double beta;
  double tol2;    // pcg_tol^2
  double rz_prev; // r.z before this iteration's update (recorded by k_update_xr for k_update_p)
  int iter;
  int maxit;
  int stop;       // 0 run, 1 converged, 2 maxit, 3 breakdown (pq <= 0 or non-finite), 4 progress probe (below)
  int iter_prev;  // iter as k_update_xr saw it
  // Progress probe (lagged refresh of the multigrid hierarchy's coarse operators, sgo_solve.cpp): the k_update_p of iteration probe_k
  // records r.r / b.b, and with probe_max > 0 stops the solve with stop = 4 when that ratio is above probe_max -- a solve behind
  // kept coarse operators that converges visibly slower than the last one behind fresh ones
  int probe_k;
  int pad_;
  double probe_rel;
  double probe_max;
};

// k_spmv modes and arguments (see sgo_kernels.hip)
enum : int {
  SPMV_AX = 0, SPMV_RESID = 1, SPMV_JACOBI = 2, SPMV_PRE_RESID = 3,
  SPMV_JACOBI_P = 4, SPMV_PRE_RESID_S = 5, SPMV_AX_C = 6,
  SPMV_PRE_RESID_ACC = 7   // PRE_RESID whose correction is ADDED to y2 (second and later pre-smoothing sweeps)
};
// scalar = sum(num[0..n_num)) / sum(den[0..n_den)); num == den == nullptr means 1
struct SpmvRatio {
  const double* num = nullptr;
  int n_num = 0;
  const double* den = nullptr;
  int n_den = 0;
};
struct SpmvArgs {
  const double* x = nullptr;     // gathered operand (unused by SPMV_PRE_RESID)
  double* y = nullptr;           // output
  const double* b = nullptr;     // right-hand side (modes != AX)
  double* y2 = nullptr;          // SPMV_PRE_RESID: omega Dinv b
  double omega = 0.0;
  const double* dotA = nullptr;  // partials[0] += dotA . out
  const double* dotA2 = nullptr; // partials[1] += dotA2 . out   (takes precedence over dotB . dotC)
  const double* dotB = nullptr;  // partials[1] += dotB . dotC
  const double* dotC = nullptr;
  double* partials = nullptr;    // [2][kMaxPartials]
  const PcgScalars* S = nullptr; // optional early-out flag
  // fused coarse-level variants
  SpmvRatio c1, c2;
  const int* agg = nullptr;      // JACOBI_P: aggregate of each vertex, lever arms d, coarse vectors u1, u2
  const double* d = nullptr;
  const double* u1 = nullptr;
  const double* u2 = nullptr;
  const double* bsub = nullptr;  // PRE_RESID_S: b' = b - c1 bsub, stored to b_out
  double* b_out = nullptr;
  const double* x2 = nullptr;    // AX_C: x' = x - c1 x2, stored to x_out
  double* x_out = nullptr;
};

// k_spmv0 (level 0, symmetric storage) modes and arguments
enum : int { S0_AX = 0, S0_RESID = 1, S0_JACOBI = 2 };
struct Spmv0Args {
  const double* x = nullptr;     // gathered operand
  double* y = nullptr;           // output
  const double* b = nullptr;     // right-hand side (RESID, JACOBI)
  double omega = 0.0;
  const double* dotA = nullptr;  // partials[0] += dotA . y
  const double* dotA2 = nullptr; // partials[1] += dotA2 . y
  double* partials = nullptr;    // [2][kMaxPartials]
  const PcgScalars* S = nullptr; // optional early-out flag
  long long* dbg_stamps = nullptr;   // diagnostic builds: [ntile][8] s_memtime stamps of the tile kernel's phases
  bool force_f64 = false;        // RESID / JACOBI on the fp64 blocks although an fp32 copy exists (measurements)
  int u0 = 0, u1 = 0;            // multi-GPU: only the work units [u0, u1) -- tiles (k_spmv0t) or wave groups (k_spmv0) --
                                 // are evaluated, i.e. only their rows of y are written (u1 == 0: all)
};

// ----------------------------------------------------------------------------------------
// Multi-GPU, row-owner mode.  Rank r owns a contiguous range of tiles = the rows
// [row0, row1): it linearises, multiplies, smooths, restricts, prolongates and updates these rows only.  What another
// rank needs of a vector computed here are its BOUNDARY rows -- the rows with an edge into another rank's range; with
// Hilbert-ordered rows a few per cent of the range -- and the partial sums of the dot products.  One exchange =
// pack (this rank's scalars + boundary rows) -> all-gather of one fixed-size packet per rank -> unpack (every other
// rank's boundary rows into the full-length vector at their global row numbers, the ranks' scalars side by side for
// the consumers' fixed-order reduction: bit-identical on all ranks).  The same packets carry 72-byte records
// (boundary rows of the smoothed prolongator, once per GN iteration); whole owned slices travel the same way once
// per GN iteration (the step, for the replicated pose update).
// ----------------------------------------------------------------------------------------
struct Comm;
constexpr int kHaloScalars = 4;      // scalar slots at the head of every packet
struct HaloDev {
  Comm* comm = nullptr;
  int G = 1, me = 0;
  int row0 = 0, row1 = 0;            // owned rows
  int u0 = 0, u1 = 0;                // owned tiles
  int g0 = 0, g1 = 0;                // owned wave groups of the compact slot list (k_linearize)
  int bmax = 0;                      // boundary rows per rank (padded to the largest)
  const int* bnd = nullptr;          // [G][bmax] boundary rows of every rank (global row numbers), -1 beyond a rank's own
  int maxrows = 0;                   // owned rows per rank (largest)
  const int* rank_row = nullptr;     // device [G + 1] first row of every rank
  int nhalo = 0;                     // boundary rows of all OTHER ranks ...
  const int* halo_rows = nullptr;    // ... listed: the rows whose copies this rank keeps current (a superset of what its tiles gather)
  int pemax = 0;                     // entries of P in the boundary rows, per rank (padded): set by the multigrid set-up
  const int* pent = nullptr;         // [G][pemax] their entry numbers, -1 beyond a rank's own
  double* send = nullptr;            // one packet
  double* recv = nullptr;            // [G] packets
  size_t cap = 0;                    // doubles per packet the buffers hold
  double* gparts = nullptr;          // [kHaloScalars][G] the ranks' scalars of the last exchange (slot-major)
  bool* failed = nullptr;            // host flag: a collective failed
};
// scalars of an exchange: up to kHaloScalars arrays of per-workgroup partial sums, reduced in a fixed order by the pack kernel
struct HaloScalars {
  const double* parts[kHaloScalars] = {nullptr, nullptr, nullptr, nullptr};
  int n[kHaloScalars] = {0, 0, 0, 0};
};

// One profiling slot per __global__ symbol (template instantiations separately), named as
// rocprofv3 --kernel-trace prints them, so bench.py's event timings can be checked 1:1 against
// the committed rocprof summaries.
enum KernelId : int {
  K_CHI2 = 0,
  K_REDUCE2,
  K_LINEARIZE,
  K_FINALIZE,
  K_INIT_SCALARS,
  K_SPMV_AX,
  K_SPMV_RESID,
  K_SPMV_JACOBI,
  K_SPMV_PRE_RESID,
  K_SPMV_JACOBI_P,
  K_SPMV_PRE_RESID_S,
  K_SPMV_AX_C,
  K_ALPHA,
  K_UPDATE_XR,
  K_BETA,
  K_UPDATE_P,
  K_DOT,
  K_POSE_UPDATE,
  K_POSITIONS0,
  K_CENTRES,
  K_GALERKIN,
  K_LEVEL_DINV,
  K_RESTRICT,
  K_PROLONG,
  K_DENSE_INVERT,
  K_DENSE_APPLY,
  K_SA_P,
  K_SA_AP,
  K_SA_RAP,
  K_RESTRICT_P,
  K_PROLONG_P,
  K_SPMV_PRE_RESID_ACC,
  K_SPMV0_AX,           // the level-0 products, wave-group kernel (graphs without a tile view)
  K_SPMV0_RESID,
  K_SPMV0_JACOBI,
  K_SPMV0T_AX,          // the level-0 products, tile kernel
  K_SPMV0T_RESID,
  K_SPMV0T_JACOBI,
  K_DIRECT,
  // the transfer / set-up kernels' launches on the FINEST level in slots of their own ("<kernel> @level0"): these are the
  // ones a rank of a multi-GPU run evaluates for its own rows only; the same kernels' coarse-level launches stay in the
  // plain slots
  K_SPMV0T_RESID_F32,   // the preconditioner's two level-0 passes on the fp32 copy of the blocks
  K_SPMV0T_JACOBI_F32,
  K_RESTRICT_P0,
  K_PROLONG_P0,
  K_SA_P0,
  K_SA_AP0,
  K_SA_RAP0,
  K_GALERKIN0,
  K_RESTRICT0,
  K_PROLONG0,
  // folded cycle (sgo_amg.hip): values of the folded transfer operator, its prolongation launches
  K_PTILDE,
  K_PTILDE0,
  K_UP_FOLD,
  K_PROLONG_FOLD0,
  K_JACOBI0_RESTRICT,
  K_MFRONT,             // the multifrontal path: all launches of one optimize() (sgo_mfront.h)
  K_COUNT
};
extern const char* const kKernelNames[K_COUNT];

// Launch geometry shared by host and kernels.
constexpr int kBlock = 256;          // 4 waves
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kMaxGrid = 2048;       // 256 CUs x 8 blocks
constexpr int kMaxPartials = kMaxGrid;

inline int grid_for(long long work_items, int per_block) {
  long long g = (work_items + per_block - 1) / per_block;
  if (g < 8) g = 8;
  if (g > kMaxGrid) g = kMaxGrid;
  g = (g + 7) / 8 * 8;  // multiple of 8: one contiguous band of groups per XCD
  return (int)g;
}

// Host scratch that survives between set-ups: blocks are kept and handed out again after rewind(),
// so the 10^8-byte product lists of the multigrid set-up are not re-mapped, page-faulted and
// unmapped on every sgo_set_graph_se2 (uninitialised memory, 64-byte aligned, single-threaded use).
struct ChunkArena {
  struct Block {
    char* raw = nullptr;   // as allocated
    char* base = nullptr;  // 64-byte aligned start
    size_t cap = 0, used = 0;
  };
  std::vector<Block> blocks;
  ChunkArena() = default;
  ChunkArena(const ChunkArena&) = delete;
  ChunkArena& operator=(const ChunkArena&) = delete;
  ~ChunkArena() {
    for (Block& b : blocks) delete[] b.raw;
  }
  void rewind() {
    for (Block& b : blocks) b.used = 0;
  }
  void* take(size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    for (Block& b : blocks)
      if (b.cap - b.used >= bytes) {
        void* q = b.base + b.used;
        b.used += bytes;
        return q;
      }
    Block nb;
    nb.cap = std::max<size_t>(bytes, (size_t)64 << 20);
    nb.raw = new char[nb.cap + 64];
    nb.base = (char*)(((uintptr_t)nb.raw + 63) & ~(uintptr_t)63);
    nb.used = bytes;
    blocks.push_back(nb);
    return nb.base;
  }
};

// Hilbert-curve index of the cell (x, y) of a 2^order x 2^order grid.
inline uint32_t hilbert_index(uint32_t x, uint32_t y, int order) {
  uint32_t d = 0;
  for (uint32_t s = 1u << (order - 1); s > 0; s >>= 1) {
    const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
    d += s * s * ((3u * rx) ^ ry);
    if (ry == 0) {   // rotate the quadrant
      if (rx == 1) {
        x = s - 1 - (x & (s - 1));
        y = s - 1 - (y & (s - 1));
      } else {
        x &= s - 1;
        y &= s - 1;
      }
      const uint32_t t = x;
      x = y;
      y = t;
    } else {
      x &= s - 1;
      y &= s - 1;
    }
  }
  return d;
}


// Device memory that survives between set-ups: a graph's ~60 arrays (and the multigrid hierarchy's ~100) are
// carved out of a few large hipMalloc'ed chunks that are rewound, not freed, when the next graph arrives --
// the reference re-initialises its slowly growing graph before every optimize(20), and 60 hipFree + 60
// hipMalloc calls cost 15-20 ms per set-up on C4.  Single-threaded use (one context = one thread at a time).
struct DevArena {
  struct Chunk {
    char* base = nullptr;
    size_t cap = 0, used = 0;
  };
  std::vector<Chunk> chunks;
  size_t next_chunk = (size_t)8 << 20;
  DevArena() = default;
  DevArena(const DevArena&) = delete;
  DevArena& operator=(const DevArena&) = delete;
  void* take(size_t bytes) {   // nullptr when the device is out of memory
    bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    for (Chunk& c : chunks)
      if (c.cap - c.used >= bytes) {
        void* q = c.base + c.used;
        c.used += bytes;
        return q;
      }
    Chunk n;
    n.cap = std::max(bytes, next_chunk);
    void* q = nullptr;
    if (hipMalloc(&q, n.cap) != hipSuccess) {
      n.cap = bytes;
      if (hipMalloc(&q, n.cap) != hipSuccess) return nullptr;
    }
    next_chunk = std::min<size_t>(next_chunk * 2, (size_t)1 << 30);
    n.base = (char*)q;
    n.used = bytes;
    chunks.push_back(n);
    return q;
  }
  void rewind() {
    for (Chunk& c : chunks) c.used = 0;
  }
  // A position to come back to: everything taken after mark() is handed out again after rewind_to(mark) (the device set-up's
  // refused attempts and per-stage temporaries; stream-ordered reuse: whoever takes the memory next is queued behind its last user)
  struct Mark {
    std::vector<size_t> used;
  };
  Mark mark() const {
    Mark m;
    for (const Chunk& c : chunks) m.used.push_back(c.used);
    return m;
  }
  void rewind_to(const Mark& m) {
    for (size_t i = 0; i < chunks.size(); ++i) chunks[i].used = i < m.used.size() ? m.used[i] : 0;
  }
  void release() {
    for (Chunk& c : chunks) hipFree(c.base);
    chunks.clear();
  }
};

// ---- launch macro -------------------------------------------------------------------------
// In profile mode (sgo_opts.profile) the bracket of the NEXT single launch is the kernel's own
// dispatch: hipExtLaunchKernelGGL stamps the start/stop events from the dispatch packet, which is
// what rocprofv3 --kernel-trace reports, instead of two extra event packets around it.
struct LaunchEvents {
  hipEvent_t start = nullptr;
  hipEvent_t stop = nullptr;
};
extern thread_local LaunchEvents tl_launch_ev;
#define SGO_LAUNCH(kernel, grid, block, shmem, stream, ...)                                              \
  do {                                                                                                   \
    const ::sgo::LaunchEvents ev_ = ::sgo::tl_launch_ev;                                                 \
    ::sgo::tl_launch_ev = ::sgo::LaunchEvents();                                                         \
    if (ev_.start) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ev_.start, ev_.stop, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                            \
  } while (0)

// ---- kernel launchers (sgo_kernels.hip) --------------------------------------------------
// All take the stream; none allocates or synchronises (hipGraph-capturable).
// el2 (optional): a second list whose first el2->cnt edges are summed by the same launch (the incremental set-up's
// appended edges, sgo_overlay.h); their per-edge values follow the first list's in e2_out
void launch_chi2(hipStream_t s, const EdgeListDev& el, int e0, int e1, const double* poses, double* e2_out,
                 double* partials /*[2][kMaxPartials]*/, int* grid_out, const EdgeListDev* el2 = nullptr);
void launch_reduce2(hipStream_t s, const double* partials, int nparts, double* out2);
// meas / info: E raw rows; outputs at SoA positions [at, at + E) of arrays with component stride `stride`
void launch_edge_prepare(hipStream_t s, int E, const double* meas, const double* info, double* zinv, double* info_soa, size_t stride, size_t at);
void launch_slot_expand(hipStream_t s, int k0, int k1, const int* eidx, const EdgeListDev& el, const EdgeSlotsDev& es);   // slots [k0, k1)
// strength weights of the logical slots (hrowptr: logical row pointers) straight from the edge list (sgo_kernels.hip)
void launch_early_strength(hipStream_t s, const EdgeListDev& el, const double* poses, int n, const int* rowptr, const int* eidx,
                           const unsigned char* flags, const int* hrowptr, double* w);
void launch_linearize(hipStream_t s, const Sym0Dev& A, int g0, int g1, const EdgeSlotsDev& es, const double* poses,
                      double* dgb /*[n][9]*/);
// (partials: [3][kMaxPartials], host-mapped -- the host adds the workgroups' sums itself, in order; returns the number of workgroups)
int launch_diag_change(hipStream_t s, int row0, int row1, const double* dblk, double* dref, bool store_ref, double* partials);
void launch_finalize(hipStream_t s, const Sym0Dev& A, int row0, int row1, const double* dgb, double* b, double* x, double* r, double* z,
                     double* p, double* xs, double omega, double* partials, int* grid_out);
void launch_init_scalars(hipStream_t s, PcgScalars* S, const double* rz_parts, int n_rz, const double* bb_parts,
                         int n_bb, double tol, int maxit, double bb_ref, double tol_cap);
void launch_set_probe(hipStream_t s, PcgScalars* S, int probe_k, double probe_max);
void launch_force_stop(hipStream_t s, PcgScalars* S, PcgScalars* mirror);
void launch_restart_scalars(hipStream_t s, PcgScalars* S, const double* rz_parts, int n_rz, int maxit, int keep_stop);
void launch_warm_start(hipStream_t s, int n3, const double* xp, const double* q, const double* b, double* x, double* r,
                       const double* xq_parts, int n_xq, const double* bx_parts, int n_bx);
int launch_spmv0(hipStream_t s, const Sym0Dev& A, int mode, const Spmv0Args& a);   // returns grid
int launch_spmv0t(hipStream_t s, const Sym0Dev& A, const Tile0Dev& T, int mode, const Spmv0Args& a);   // returns grid
// tile kernel when the graph has a tile view, the wave-group kernel otherwise
inline int launch_spmv0_any(hipStream_t s, const Sym0Dev& A, const Tile0Dev& T, int mode, const Spmv0Args& a) {
  return T.ntile > 0 ? launch_spmv0t(s, A, T, mode, a) : launch_spmv0(s, A, mode, a);
}
constexpr int kTileLdsMax = 157 * 1024;   // one 1024-thread tile workgroup per CU (160 KiB of LDS)
int launch_spmv_ex(hipStream_t s, const BsrDev& A, int mode, const SpmvArgs& a);  // returns grid
void launch_update_xr(hipStream_t s, int n, PcgScalars* S, const double* pq_parts, int n_pq, const double* dinv,
                      const double* p, const double* q, double* x, double* r, double* z, double* xs, double omega,
                      double* partials, int* grid_out);
constexpr int kLanczosMax = 2048;     // PCG iterations whose (alpha, beta, r.z) the diagnostic record keeps
struct RecDev {                       // what k_update_p leaves for the host besides the recurrence
  PcgScalars* mirror = nullptr;       // pinned host copy of the scalars, rewritten by every iteration's k_update_p (also by the
                                      // early-exit ones): the host reads the stop flag there behind an event instead of
                                      // queueing a device-to-host copy kernel after every graph replay
  double* lanczos = nullptr;          // [kLanczosMax][3] alpha_j, beta_j, r_j . z_j
};
void launch_update_p(hipStream_t s, int n, PcgScalars* S, const double* rz_parts, int n_rz, const double* rr_parts,
                     int n_rr, const double* zq_parts, const double* z, double* p, const RecDev& rec = RecDev());
void launch_pose_update(hipStream_t s, int n, const int* free_id, const double* x, double* poses);
void launch_closure_cov(hipStream_t s, int n, const sgo_match_window* win, const float* scores, double* cov,
                        double* info);
void launch_dot(hipStream_t s, int n3, const double* a, const double* b, double* partials, const PcgScalars* S,
                int* grid_out);
void launch_precond_bj(hipStream_t s, int n, const double* dinv, const double* r, double* z, double scale);
// row-owner mode: the recurrences of k_update_xr / k_update_p repeated on a LIST of rows (the copies of the neighbours'
// boundary rows this rank keeps: same inputs, same arithmetic, hence bit-identical to the owner's values) with the
// alpha / beta the preceding k_update_xr / k_update_p left in S
void launch_update_xr_rows(hipStream_t s, int nrows, const int* rows, const PcgScalars* S, const double* dinv, const double* p,
                           const double* q, double* r, double* xs, double omega);
void launch_update_p_rows(hipStream_t s, int nrows, const int* rows, const PcgScalars* S, const double* z, double* p);
// row-owner mode: one exchange of `width`-double records (3: rows of a vector through H.bnd; 9: entries of P through H.pent;
// idx == nullptr, width 0: scalars only) + the scalars; returns false when the collective failed
bool halo_exchange(const HaloDev& H, hipStream_t s, double* data, int width, const int* idx, int idx_max, const HaloScalars& sc,
                   std::string* err);
// every rank's owned slice of a [n][width] array to every rank (the full-length array is valid everywhere afterwards)
bool halo_gather_slices(const HaloDev& H, hipStream_t s, double* vec, int width, std::string* err);

}  // namespace sgo
