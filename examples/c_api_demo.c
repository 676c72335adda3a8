/* c_api_demo.c -- the C-ABI of include/sgo.h used from plain C (C99): a square loop of 4 poses
 * with one loop closure, optimised with sgo_optimize_gn.  Shows the call sequence a foreign-
 * function binding makes; mirrors what g2o::SparseOptimizer::optimize() does in the compat
 * header (initializeOptimization -> sgo_set_graph_se2, optimize -> sgo_optimize_gn,
 * estimates -> sgo_get_poses).
 *
 *   gcc -std=c99 -Iinclude examples/c_api_demo.c -Lsparse_gslam_amd/csrc -lsgo \
 *       -Wl,-rpath,$PWD/sparse_gslam_amd/csrc -o c_api_demo && ./c_api_demo
 */
#include <stdio.h>
#include <stdlib.h>

#include "sgo.h"

int main(void) {
  /* ground truth: a 1 m square walked counter-clockwise, heading along the direction of travel */
  const double PI = 3.14159265358979323846;
  double poses[4 * 3] = {0, 0, 0, 1.05, 0.02, PI / 2 + 0.03, 0.97, 1.04, PI - 0.02, -0.04, 0.98, -PI / 2 + 0.05};
  const uint8_t fixed[4] = {1, 0, 0, 0};
  const int32_t ei[4] = {0, 1, 2, 3}, ej[4] = {1, 2, 3, 0};
  double meas[4 * 3], info[4 * 6], phi[4];
  for (int e = 0; e < 4; ++e) {
    meas[3 * e] = 1.0;            /* one step forward ...            */
    meas[3 * e + 1] = 0.0;
    meas[3 * e + 2] = PI / 2;     /* ... then a left turn            */
    const double o[6] = {400, 0, 0, 400, 0, 2500};
    for (int q = 0; q < 6; ++q) info[6 * e + q] = o[q];
    phi[e] = e == 3 ? 1.0 : -1.0; /* the closing edge carries a DCS kernel (delta = 1) */
  }
  sgo_ctx* ctx = sgo_create(-1, NULL);
  if (!ctx) {
    fprintf(stderr, "sgo_create: %s\n", sgo_last_error(NULL));
    return 1;
  }
  if (sgo_set_graph_se2(ctx, 4, poses, fixed, 4, ei, ej, meas, info, phi) != SGO_OK) {
    fprintf(stderr, "sgo_set_graph_se2: %s\n", sgo_last_error(ctx));
    return 1;
  }
  sgo_stats* st = (sgo_stats*)calloc(1, sizeof(sgo_stats));
  const int done = sgo_optimize_gn(ctx, 10, st);
  if (done < 0) {
    fprintf(stderr, "sgo_optimize_gn: %s\n", sgo_last_error(ctx));
    return 1;
  }
  sgo_get_poses(ctx, poses);
  printf("iterations %d  chi2 %.6g -> %.3g\n", done, st->chi2[0], st->chi2[done]);
  for (int v = 0; v < 4; ++v) printf("pose %d: %.6f %.6f %.6f\n", v, poses[3 * v], poses[3 * v + 1], poses[3 * v + 2]);
  int ok = done == 10 && st->chi2[done] < 1e-16;
  /* the graph grows -- one more pose a step further along, with its odometry edge --: the new graph's first four edges are the
   * resident graph, which sgo_update_graph_se2 is told; where the resident structures can take the appended part (graphs of the
   * multigrid path, include/sgo.h) they are kept, otherwise -- as for this toy -- the call is sgo_set_graph_se2 */
  {
    double poses5[5 * 3], meas5[5 * 3], info5[5 * 6], phi5[5];
    const uint8_t fixed5[5] = {1, 0, 0, 0, 0};
    const int32_t ei5[5] = {0, 1, 2, 3, 3}, ej5[5] = {1, 2, 3, 0, 4};
    for (int q = 0; q < 12; ++q) poses5[q] = poses[q], meas5[q] = meas[q];
    for (int q = 0; q < 24; ++q) info5[q] = info[q];
    for (int q = 0; q < 4; ++q) phi5[q] = phi[q];
    poses5[12] = 0.1; poses5[13] = 0.1; poses5[14] = -PI / 2 + 0.05;     /* a rough guess for the new pose (truth: the origin, heading -y) */
    meas5[12] = 1.0; meas5[13] = 0.0; meas5[14] = 0.0;                  /* one step forward from pose 3 (heading -y) */
    for (int q = 0; q < 6; ++q) info5[24 + q] = info[q];
    phi5[4] = -1.0;
    if (sgo_update_graph_se2(ctx, 5, poses5, fixed5, 5, ei5, ej5, meas5, info5, phi5, 4) != SGO_OK) {
      fprintf(stderr, "sgo_update_graph_se2: %s\n", sgo_last_error(ctx));
      return 1;
    }
    const int done2 = sgo_optimize_gn(ctx, 10, st);
    sgo_get_poses(ctx, poses5);
    printf("after the update: iterations %d  chi2 %.3g; pose 4: %.6f %.6f %.6f  [%s]\n", done2, st->chi2[done2 > 0 ? done2 : 0], poses5[12],
           poses5[13], poses5[14], sgo_solver_description(ctx));
    ok = ok && done2 == 10 && st->chi2[done2] < 1e-16 && poses5[12] * poses5[12] + poses5[13] * poses5[13] < 1e-12;
  }
  free(st);
  sgo_destroy(ctx);
  return ok ? 0 : 2;
}
