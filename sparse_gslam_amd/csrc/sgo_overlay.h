// sgo_overlay.h -- incremental re-initialisation (sgo_update_graph_se2): the resident graph plus APPENDED vertices and
// edges without rebuilding the level-0 structure or the multigrid hierarchy.
//
// The reference calls initializeOptimization(); optimize(20) after every accepted loop closure on the previous graph +
// a chain of new poses with their odometry edges + one closure (src/sparse_gslam/src/submap_loop_closer.cpp:205-226,
// :272-287).  The appended part is kept beside the resident ("base") structures as an overlay:
//   * new rows N  -- the appended free poses; the edges among consecutive ones form a chain in pose order (block-tridiagonal
//     H_NN), which is what slc.cpp:205-226 appends;
//   * hub rows X  -- appended poses with an edge to a NON-neighbouring appended pose (a closure that ends in a pose of an
//     earlier update, slc.cpp:279): one endpoint of every such edge is taken out of the chain, which leaves N a set of chain
//     segments; the hubs are eliminated after them from the small dense Schur complement;
//   * touched rows T -- the base rows an appended edge ends in (the chain's anchor, the closure's old endpoint).
// With x = [x_O; x_N] the Gauss-Newton system is solved EXACTLY as
//     S x_O = g,   S = H_OO - H_ON H_NN^-1 H_NO,   g = b_O - H_ON H_NN^-1 b_N,   x_N = H_NN^-1 (b_N - H_NO x_O)
// where S = H_base + U M U^T: H_base is the resident structure with its values at the current poses (the level-0 product
// kernels, unchanged), U selects the touched rows and M (3|T| x 3|T|, symmetric positive semi-definite) collects the
// appended edges' contributions to the touched rows minus the Schur complement of the chain.  PCG runs on the base rows
// with the operator H_base p + U M U^T p (one single-wave kernel after the product, k_ov_ax) and the resident multigrid
// hierarchy -- a preconditioner for H_base, i.e. for S up to a perturbation of rank <= 3|T| -- as preconditioner.  Per
// Gauss-Newton iteration: k_ov_lin (the appended edges' linearisation, one thread per overlay row, recompute instead of
// scatter: no atomics), k_ov_solve (block-tridiagonal LDL^T of H_NN with the 3|T| + 1 right-hand sides [H_NT | b_N] on the
// threads of one workgroup, M and g), k_ov_finish (x_N and the new poses' update).
// The overlay accumulates over successive updates; sgo_update_graph_se2 falls back to the full set-up when the appended
// part does not have this shape, outgrows the capacities below, or the resident structures cannot take it (multi-GPU,
// direct path).
#pragma once
#include <string>
#include <vector>

#include "sgo_internal.h"

namespace sgo {

constexpr int kOvMaxRows = 512;      // new rows (poses of the appended chain)
constexpr int kOvMaxTouched = 64;    // touched base rows + hub rows (round 6: 16 before -- the right-hand sides of the chain's elimination
                                     // sat on the lanes of ONE wave; they now sit on the threads of the workgroup, 3 x 64 + 1 columns)
constexpr int kOvMaxHubs = 8;        // of which hub rows (their Gauss-Jordan elimination lives in the elimination's LDS tile)
constexpr int kOvMaxEdges = 4096;    // appended edges
constexpr int kOvMaxVerts = 4096;    // appended vertices (active or not)
constexpr int kOvOtherFixed = -(1 << 30);   // entry code: the edge's other endpoint is fixed (no block)

struct OverlayDev {
  int k = 0, nt = 0, ncol = 1;       // chain rows, touched base rows, right-hand-side columns 3 (nt + nx) + 1
  int nnz = 0;                       // chain rows with a block into a touched or hub row
  int nx = 0;                        // hub rows (numbered nt .. nt + nx - 1 among the "kept" rows K = T u X)
  EdgeListDev el;                    // appended edges (capacity kOvMaxEdges, el.cnt valid)
  // structure (host-made per update)
  const int* hdr = nullptr;          // [4] = {k, nt, ncol, nnz} on the device (k_ov_ax reads its sizes here: a captured hipGraph stays valid)
  const int* rp = nullptr;           // [k + nt + 1] entries of overlay row r (new rows first, then touched rows)
  const int* ent_edge = nullptr;     // entry: index into el
  const int* ent_other = nullptr;    // entry: other endpoint: >= 0 new row, -1 - t touched row t, kOvOtherFixed
  const unsigned char* ent_side = nullptr;   // entry: 0 = this row is vertices()[0] (Jacobian A), 1 = vertices()[1] (B)
  const int* vtx = nullptr;          // [k + nx] vertex id of the chain rows, then of the hub rows
  const int* trow = nullptr;         // [nt] base row of touched row t
  const int* nz = nullptr;           // [nnz] the new rows with a block into a touched row
  // values (per Gauss-Newton iteration)
  double* Dn = nullptr;              // [k][6]  diagonal blocks of H_NN (symmetric packing)
  double* Un = nullptr;              // [k][9]  H_{i, i+1}
  double* H0 = nullptr;              // [3 k][ncol] row-major: [H_NT | b_N]
  double* Y = nullptr;               // [3 k][ncol] H_NN^-1 [H_NT | b_N]
  double* Sinv = nullptr;            // [k][6]  inverses of the pivot blocks
  double* M0 = nullptr;              // [3 nk][3 nk] appended edges' direct contributions to the kept rows (nk = nt + nx)
  double* bt = nullptr;              // [3 nk]       ... to their right-hand side
  double* S = nullptr;               // [3 nk][3 nk] M0 - H_KN H_NN^-1 H_NK, symmetrised; gk [3 nk] the matching right-hand side
  double* gk = nullptr;
  double* Wx = nullptr;              // [3 nx][3 nt + 1] S_XX^-1 [S_XT | g_X]: the hubs' back-substitution
  double* M = nullptr;               // [3 nt][3 nt] S_TT - S_TX S_XX^-1 S_XT, symmetrised: the operator's term on the touched rows
};

// Host state of the overlay (lives in the context)
struct Overlay {
  bool active = false;
  int base_V = 0, base_E = 0, base_n = 0;     // the resident graph the structures were built for
  std::vector<int> hpos;                      // base: vertex id -> internal row (-1: fixed or inactive)
  std::vector<unsigned char> fixed;           // fixed flags of all vertices (base + appended)
  std::vector<int32_t> ei, ej;                // appended edges (vertex ids), all updates since the base
  std::vector<int> new_vertex;                // vertex ids of the new rows, chain order
  OverlayDev dev;
  // device buffers (hipMalloc once per context, capacity sized)
  void* buf = nullptr;
  int* d_int = nullptr;                       // structure arrays
  unsigned char* d_side = nullptr;
  int* h_int = nullptr;                       // their pinned host images (a pageable source of this size makes the runtime pin and
  unsigned char* h_side = nullptr;            // unpin it per copy: 10-20 ms per update, measured)
  double* h_edge = nullptr;                   // pinned staging of an update's appended edges: [10][kOvMaxEdges] doubles + 2 x ints
  int updates = 0;                            // updates absorbed since the base was built
};

// Builds the overlay for the appended edges ov.ei / ov.ej on top of the base described by ov.hpos / ov.fixed.
// Returns true when the appended part has the supported shape (then ov.dev is ready and uploaded on `s`); false with
// `why` set otherwise (the caller falls back to the full set-up).  Device buffers are allocated on first use.
bool overlay_build(Overlay& ov, int V, hipStream_t s, std::string* why, std::string* err);
void overlay_release(Overlay& ov);
// appends `cnt` raw edges (host arrays) to the overlay's device edge list at position `at`
bool overlay_upload_edges(Overlay& ov, hipStream_t s, int at, int cnt, const int32_t* ei, const int32_t* ej, const double* meas,
                          const double* info, const double* phi, std::string* err);

void launch_ov_lin(hipStream_t s, const OverlayDev& O, const double* poses);
// factorisation + Schur complement; adds g to the right-hand sides of the touched rows in dgb ([n][9]: entries 6..8)
void launch_ov_solve(hipStream_t s, const OverlayDev& O, double* dgb);
// q_T += M p_T; partials0[0] += p_T . (M p_T) when partials0 != nullptr
void launch_ov_ax(hipStream_t s, const OverlayDev& O, const double* p, double* q, double* partials0, const PcgScalars* S);
// x_N = Y_b - Y_T x_T, then the new poses' update (VertexSE2::oplusImpl)
void launch_ov_finish(hipStream_t s, const OverlayDev& O, const double* x, double* poses);

}  // namespace sgo
