// sgo_amg_host.h -- host side of the multigrid set-up (sgo_amg_host.cpp): aggregation, patterns of the smoothed transfer
// and of the Galerkin products, structure of the next level.  No device code; amg_create (sgo_amg.hip) uploads what this
// produces and makes the product lists on the device from the patterns.
// Round 6: on one GPU the default set-up takes only the AGGREGATION from here (host_aggregate) and makes everything else on the
// device (amg_create_dev, sgo_amg_dev.inc), bit-identical to what host_coarsen + amg_create produce; this path stays as the
// multi-GPU modes' set-up, as SGO_AMG_SETUP=host, as the fallback of a device set-up that cannot be made, and as the reference the
// device set-up is tested against (tests/test_gpu_device_setup.py).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "sgo_amg.h"
#include "sgo_internal.h"

namespace sgo {

// wave groups over segments [ptr[i], ptr[i+1]): whole segments packed up to 64 items; a longer
// segment is its own group (same rule as the level-0 row groups in sgo_structure.cpp)
constexpr int kGroupChunk = 256;   // segments per chunk of the grouping rule (host: make_groups; device: k_group_chunks)
std::vector<int> make_groups(const std::vector<int>& ptr);

// Host side of the smoothed-aggregation set-up: patterns of P, AP = A P and A_c = P^T AP and, for
// every entry of each, the block products that make it up, listed in target order (all sorting is
// per row -- a counting sort over the row's few distinct columns -- so the rows run in parallel).
// int array without value-initialisation (the product lists are 10^7 entries and fully overwritten)
struct UVec {
  std::unique_ptr<int[]> own;
  int* p = nullptr;
  size_t n = 0;
  ChunkArena* arena = nullptr;   // when set, the memory is the arena's (kept for the next set-up)
  void resize(size_t count) {
    if (arena) {
      p = (int*)arena->take(std::max<size_t>(count, 1) * sizeof(int));
    } else {
      own.reset(new int[std::max<size_t>(count, 1)]);
      p = own.get();
    }
    n = count;
  }
  int* data() { return p; }
  int* begin() { return p; }
  int& operator[](size_t i) { return p[i]; }
  const int& operator[](size_t i) const { return p[i]; }
  size_t size() const { return n; }
};
struct SaHost {
  // FILTERED smoothing (the second attempt of host_coarsen, when the transfer smoothed with the whole operator is refused as too
  // dense): P = (I - w D_F^-1 A_F) T with A_F the operator of the STRONG connections only (the aggregation's criterion) -- the
  // weak blocks dropped together with their share of the diagonal, so that A_F still annihilates the rigid motions (k_filtered_diag,
  // sgo_amg.hip).  P then has the pattern of the strong neighbours' aggregates; A P and P^T A P are made from the WHOLE operator.
  bool filtered = false;
  std::vector<unsigned char> strong;   // [nslot] 1 = kept by the filter (diagonal and alias slots included); empty when !filtered
  std::vector<int> val_rowptr;         // [n + 1] range of row i's value products in val_src / val_tgt (the row's KEPT slots)
  std::vector<int> p_rowptr, p_row, p_col, val_grp;
  UVec val_src, val_tgt;   // (the large lists live in the set-up's scratch arena: storage kept between set-ups, no fresh pages)
  std::vector<int> r_grp, t_pos, t_row, t_col, t_grp;
  int nap = 0;
  UVec ap_a, ap_b, ap_tgt;
  std::vector<int> ap_grp;
  HostLevel Hc;
  UVec rap_a, rap_b, rap_tgt;
  std::vector<int> rap_grp, rap_mirror;
  // lists_on_device: the product lists (ap_*, rap_*) are NOT made here; the patterns they follow from are kept instead
  bool lists_on_device = false;
  std::vector<int> ap_rowptr, t_ptr, t_idx;
  UVec ap_col, ap_row;
  long long n_ap_prod = 0, n_rap_prod = 0;
};

// Everything the host decides about one coarsening step: aggregates, patterns and product lists of the smoothed
// transfer (or the tentative one's Galerkin map), structure of the next level.  Depends on the level's structure and
// on the strength weights only -- for level 0 it can therefore run on a helper thread while the caller still builds
// the level-0 storage (build_structure, sgo_structure.cpp).
struct HostCoarse {
  std::vector<int> agg, visit_c, mem_ptr, mem;
  int nc = 0;
  bool reuse_agg = false;   // agg / visit_c / nc are GIVEN (AmgConfig::keep_agg): host_coarsen skips the aggregation
  bool stop = false;     // the level cannot be coarsened further
  bool smooth = false;   // (sa.filtered says which smoothing)
  SaHost sa;
  HostLevel Hc;
  std::vector<int> order, tgt, cptr, grp_g, grp_c;
  double t_agg = 0, t_sort = 0, t_all = 0;
  std::string err;
};

// nc (0: the level cannot be coarsened further); agg renumbered in order of first appearance when H.visit is set, visit_c as host_coarsen's
int host_aggregate(const HostLevel& H, const std::vector<double>& w, const AmgConfig& cfg, int l, ChunkArena* scratch, std::vector<int>& agg,
                   std::vector<int>& visit_c, double* theta_used);
void host_coarsen(const HostLevel& H, const std::vector<double>& w, const AmgConfig& cfg, int l, ChunkArena* scratch, HostCoarse& o);

}  // namespace sgo
