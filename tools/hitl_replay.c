/*
 * hitl_replay.c -- the reference binary's batch path as a plain C program over the C ABI (include/hitl_gpu.h only):
 *     HitL-SLAM -P map.stfs.covars -L session.log        (HitLSLAM_main.cpp:953-1049)
 * load the pose graph (loadPoseGraph, :192-300), load the recording (LoadLogFile, :676-764), replay every entry through
 * one correction cycle (HitLSLAM::replayLog, HitLSLAM.cpp:278-360: verify, fit, order, explicit correction,
 * back-propagation, angle normalisation, batch re-solve), write the poses (saveHitLResults, :572-581).
 * Also the shortest complete example of driving libhitlgpu.so from compiled code.
 *
 *   hitl_replay map.stfs.covars session.log out_poses.txt [mode=0|2] [stf_window=3] [fix_normals=1] [fix_parallel=0]
 *
 * Build: gcc -std=c11 -O2 -I include tools/hitl_replay.c -o hitl_replay -L hitl_slam_amd/csrc -lhitlgpu -lm
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hitl_gpu.h"

#define CHECK(call)                                                                               \
  do {                                                                                            \
    const int st_ = (call);                                                                       \
    if (st_ != HITL_OK) {                                                                         \
      fprintf(stderr, "hitl_replay: %s -> %s (%s)\n", #call, hitl_status_string(st_), hitl_last_error()); \
      return st_ == HITL_ERR_NO_DEVICE ? 3 : 70;                                                  \
    }                                                                                             \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: hitl_replay map.stfs.covars session.log out_poses.txt [mode] [stf_window] [fix_normals] [fix_parallel]\n");
    return 64;
  }
  const int mode = argc > 4 ? atoi(argv[4]) : HITL_MODE_ODOM_HC;
  const int window = argc > 5 ? atoi(argv[5]) : 3;
  const int fix_normals = argc > 6 ? atoi(argv[6]) : 1;
  const int fix_parallel = argc > 7 ? atoi(argv[7]) : 0;

  hitl_posegraph* g = NULL;
  hitl_input_log* lg = NULL;
  if (hitl_io_load_stfs_covars(argv[1], fix_normals, &g) != HITL_OK) { fprintf(stderr, "hitl_replay: cannot read %s\n", argv[1]); return 65; }
  if (hitl_io_load_input_log(argv[2], fix_parallel, &lg) != HITL_OK) { fprintf(stderr, "hitl_replay: cannot read %s\n", argv[2]); return 66; }
  const int64_t n = hitl_posegraph_num_poses(g);

  hitl_options opt;
  hitl_options_default(&opt);
  hitl_solver* s = NULL;
  CHECK(hitl_solver_create(&opt, &s));
  CHECK(hitl_solver_set_map(s, n, hitl_posegraph_obs_offset(g), hitl_posegraph_px(g), hitl_posegraph_py(g), hitl_posegraph_nx(g),
                            hitl_posegraph_ny(g), hitl_posegraph_cov9(g)));

  float* poses = (float*)malloc(sizeof(float) * 3 * (size_t)n);
  float* cov = (float*)malloc(sizeof(float) * 9 * (size_t)n);
  memcpy(poses, hitl_posegraph_poses(g), sizeof(float) * 3 * (size_t)n);
  memcpy(cov, hitl_posegraph_cov9(g), sizeof(float) * 9 * (size_t)n);
  int32_t* counts = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n);
  int32_t* corrected = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int32_t* anchor = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  hitl_human_constraint_t* all_rows = NULL;       /* human_constraints_, flattened */
  size_t n_rows = 0;

  const int entries = hitl_input_log_num_entries(lg);
  for (int e = 0; e < entries; ++e) {
    uint32_t type; int32_t undone, npts; const float* pts;
    CHECK(hitl_input_log_entry(lg, e, &type, &undone, &npts, &pts));
    if ((type != HITL_HC_LINE_SEGMENT && type != HITL_HC_COLINEAR && type != HITL_HC_PERPENDICULAR && type != HITL_HC_PARALLEL) || npts != 4) {
      printf("entry %d: type %u with %d points is not supported (nor in the reference)\n", e, type, npts);
      continue;
    }
    float sp[8];
    memcpy(sp, pts, sizeof(sp));
    int32_t seen[4];
    CHECK(hitl_solver_verify_points(s, poses, 4, sp, 0.05f, seen));                       /* verifyUserInput */
    if (seen[0] + seen[1] + seen[2] + seen[3] != 4) { printf("entry %d: user input was not verified as close to observations\n", e); continue; }
    CHECK(hitl_solver_fit_segments(s, poses, 2, sp, NULL, NULL, NULL));                   /* EMInput::Run, first half */
    CHECK(hitl_solver_observation_sets(s, poses, sp, 0.03, counts, NULL));
    int32_t nc = 0, na = 0, bounds[2];
    const int ord = hitl_order_user_input(n, counts, sp, corrected, &nc, anchor, &na, bounds);
    if (ord != HITL_OK || !(bounds[0] >= 0 && bounds[1] >= 1)) { printf("entry %d: selection overlap\n", e); continue; }
    float correction[3];
    all_rows = (hitl_human_constraint_t*)realloc(all_rows, sizeof(hitl_human_constraint_t) * (n_rows + (size_t)na * nc));
    CHECK(hitl_solver_apply_explicit_correction(s, type, sp, n, poses, nc, corrected, na, anchor, correction, all_rows + n_rows));
    n_rows += (size_t)na * nc;
    CHECK(hitl_solver_backprop(s, n, poses, cov, bounds[0], bounds[1], correction));
    CHECK(hitl_normalize_angles(n, poses));
    CHECK(hitl_solver_set_human_constraints(s, (int64_t)n_rows, all_rows));
    if (mode != HITL_MODE_ODOM_HC) CHECK(hitl_solver_find_correspondences(s, poses, window));
    hitl_summary sum;
    const int st = hitl_solver_solve(s, mode, poses, poses, &sum);
    if (st == HITL_ERR_SOLVER_FAILURE) { printf("\nCeres failure\n\n"); return 1; }        /* JointOptimization.cpp:1339-1342 */
    if (st != HITL_OK) { fprintf(stderr, "hitl_replay: solve -> %s (%s)\n", hitl_status_string(st), hitl_last_error()); return 70; }
    printf("entry %d: type %u anchors %d corrected %d bounds (%d, %d) rows %zu iterations %d final_cost %.9g\n", e, type, na, nc,
           bounds[0], bounds[1], n_rows, sum.iterations, sum.final_cost);
  }
  CHECK(hitl_io_save_results(argv[3], n, poses));
  hitl_solver_destroy(s);
  hitl_input_log_free(lg);
  hitl_posegraph_free(g);
  free(poses); free(cov); free(counts); free(corrected); free(anchor); free(all_rows);
  return 0;
}
