/*
 * hitl_gpu.h -- C ABI of libhitlgpu.so, the MI355X (gfx950) backend for hitl-slam's
 * batch re-optimisation hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI for this
 * path; the seam is the JointOpt object (public members in, Run(), poses_ out --
 * HitL-SLAM/src/human_in_the_loop_slam/JointOptimization.h:49-89) whose body reaches
 * Ceres at ceres::Solve (JointOptimization.cpp:1093 and :1208).  Each entry point below
 * cites the reference interface it replaces.  Plain pointers and sizes only; no C++ or
 * torch types cross this boundary; every function returns an int status (0 = OK) and
 * never throws.  The library installs no signal handlers and is meant to be created
 * once and called repeatedly from one thread (the reference's single ROS spinner,
 * HitLSLAM_main.cpp:1049).
 *
 * Numerics contract: inputs are fp32 (poses, scans, human-constraint rows), all solver
 * state and arithmetic is fp64, outputs are rounded to fp32 with theta wrapped by
 * theta - 2*pi*rint(theta/2pi) (shared/math/util.h:433-439 via JointOptimization.cpp:388).
 * Pose 0 is held constant (JointOptimization.cpp:824, :1197).
 */
#ifndef HITL_GPU_H_
#define HITL_GPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HITL_ABI_VERSION 1

/* ---- status codes ------------------------------------------------------------------ */
enum {
  HITL_OK = 0,
  HITL_ERR_INVALID_ARGUMENT = 1,
  HITL_ERR_NO_DEVICE = 2,       /* no HIP device / HIP runtime error at create        */
  HITL_ERR_HIP = 3,             /* a HIP call failed (message via hitl_last_error)    */
  HITL_ERR_OUT_OF_MEMORY = 4,
  HITL_ERR_NOT_READY = 5,       /* e.g. solve before set_map, mode B without pairs    */
  HITL_ERR_UNSUPPORTED = 6,     /* e.g. a scan with more than 65535 points            */
  HITL_ERR_SOLVER_FAILURE = 7,  /* == ceres::FAILURE; the JointOpt shim exit(1)s here */
  HITL_ERR_COMM = 8             /* RCCL error                                          */
};

/* ---- solve modes --------------------------------------------------------------------
 * A: odometry + human rows   == JointOpt::SolveHumanConstraints (JointOptimization.cpp:1064-1138)
 * B: scan-to-scan (STF) only == JointOpt::PostHumanOptimization  (JointOptimization.cpp:1156-1256)
 * C: A and B jointly         == the "full batch re-solve" of BASELINE.json                  */
enum { HITL_MODE_ODOM_HC = 0, HITL_MODE_STF_ONLY = 1, HITL_MODE_JOINT = 2 };

/* ---- termination (ceres::TerminationType as used at JointOptimization.cpp:1339-1349) -- */
enum { HITL_CONVERGENCE = 0, HITL_NO_CONVERGENCE = 1, HITL_FAILURE = 2 };

/* == CorrectionType (human_constraints.h:8-16).  Types 0,1,3 add no residual. */
enum {
  HITL_HC_UNKNOWN = 0, HITL_HC_POINT = 1, HITL_HC_LINE_SEGMENT = 2, HITL_HC_CORNER = 3,
  HITL_HC_COLINEAR = 4, HITL_HC_PERPENDICULAR = 5, HITL_HC_PARALLEL = 6
};

/* == struct HumanConstraint (human_constraints.h:29-41), 28 bytes, same field order. */
typedef struct hitl_human_constraint_t {
  uint32_t type;                 /* CorrectionType                                   */
  int32_t constrained_pose_id;
  int32_t anchor_pose_id;
  float delta_parallel;
  float delta_perpendicular;
  float delta_angle;
  float relative_penalty_dir;    /* colinear rows only                               */
} hitl_human_constraint_t;

/* Options.  STF fields mirror VectorMappingOptions (vector_mapping.h:121-186), which the
 * reference binary leaves uninitialised; defaults are the cobot block of
 * config/non_markov_localization.cfg (:11,:16,:47-50).  Odometry sigmas are the
 * constants hard-coded at JointOptimization.cpp:770,776,783.  Trust-region fields are the
 * ceres::Solver::Options defaults in effect at JointOptimization.cpp:1066-1070 / :145-162. */
typedef struct hitl_options {
  int32_t abi_version;                    /* HITL_ABI_VERSION                         */
  int32_t device;                         /* HIP device ordinal (default 0)           */
  /* STF / correspondence */
  float laser_std_dev;                    /* kLaserStdDev                 0.05        */
  float point_correlation_factor;         /* kPointPointCorrelationFactor 1/40        */
  float point_match_threshold;            /* kPointMatchThreshold         0.15        */
  float max_stf_angle_error;              /* kMaxStfAngleError  rad(25 deg)           */
  int32_t max_correspondences_per_point;  /* kMaxCorrespondencesPerPoint  6           */
  int32_t num_skip_readings;              /* num_skip_readings            1           */
  int32_t min_inter_pose_correspondence;  /* pair kept iff count > this   10          */
  /* odometry sigmas */
  float odom_radial_std_dev;              /* 0.03 */
  float odom_tangential_std_dev;          /* 0.03 */
  float odom_angular_std_dev;             /* 0.01 */
  /* Levenberg-Marquardt / trust region */
  int32_t max_num_iterations;             /* 100 */
  int32_t jacobi_scaling;                 /* 1   */
  double function_tolerance;              /* 1e-6  */
  double gradient_tolerance;              /* 1e-10 */
  double parameter_tolerance;             /* 1e-8  */
  double initial_trust_region_radius;     /* 1e4   */
  double max_trust_region_radius;         /* 1e16  */
  double min_trust_region_radius;         /* 1e-32 */
  double min_relative_decrease;           /* 1e-3  */
  double min_lm_diagonal;                 /* 1e-6  */
  double max_lm_diagonal;                 /* 1e32  */
  /* linear solver: PCG preconditioned by the exact block-cyclic-reduction factorisation
   * of the in-band part of the Hessian.                                                 */
  double linear_solver_tolerance;         /* relative residual, 1e-12                  */
  int32_t max_linear_iterations;          /* 500 (Ceres max_linear_solver_iterations); PCG only */
  int32_t max_band_poses;                 /* widest pose distance kept in the band, 1..4 (default 4) */
  /* multi-GPU (pose-range submaps).  world_size 1 = single device, no RCCL.             */
  int32_t rank;
  int32_t world_size;
  const void* rccl_unique_id;             /* 128-byte ncclUniqueId from hitl_comm_unique_id, same on all ranks */
  /* 0 (default): every linearisation streams all point matches, as the reference's functor does
   * (residual_functors.h:786-830).  1: the scan-to-scan factors are evaluated from 2 x 15 moments per pose pair that
   * are summed once per correspondence set -- the same algebra (sum e^2 = k^2 z^T M z), a different summation order
   * (agreement ~1e-10 relative), and a linearisation cost that no longer depends on the number of matches. */
  int64_t stf_moments;
  /* device timings in hitl_summary.  An event record drains the queue (~6 us each on this stack), so: 0 = none;
   * 1 (default) = every 4th k_linearize_stf launch is bracketed (linearize_ms over linearize_timed_calls launches),
   * no linear-solver timing; 2 = every launch bracketed and linear_solver_ms filled. */
  int64_t timing_level;
  /* hitl_solver_find_correspondences: 1 (default) = the reference's own per-scan KD-trees and pruned walk (BuildKDTrees,
   * JointOptimization.cpp:514-537; perception_tools/kdtree.cpp:37-197) reproduced bit for bit -- std::sort's tie order
   * taken as the stable one.  The walk prunes with the plane distance and therefore returns a tree-dependent winner
   * wherever several target points score almost equally (flat walls); these are the index sets JointOpt::Run computes.
   * 0 = the brute-force definition of FindNearestPointNormal (among the target points inside the Euclidean gate, the
   * smallest point-to-plane distance, first index on ties): tree-independent, a different set on ~40 % of the matches of
   * a corridor map, converged poses up to 3e-3 m away from the reference's. */
  int64_t kd_tree_search;
  /* Where the Levenberg-Marquardt loop tests parameter_tolerance and function_tolerance.  0 (default): on the candidate,
   * BEFORE the accept test; reaching one ends the solve at the PRE-step state (the order of Ceres'
   * TrustRegionMinimizer: ...ParameterToleranceReached / FunctionToleranceReached, then IsStepSuccessful).  1: after an
   * accepted step, on the updated state.  The reference's Ceres version is not pinned (CMakeLists.txt:83); the two orders
   * differ in the last step of a solve only.  tests/golden/ holds fixtures for both. */
  int64_t tolerances_after_accept;
} hitl_options;

/* Per-solve report (what Ceres' Solver::Summary gives the caller at
 * JointOptimization.cpp:1093-1095, plus per-phase device timings). */
typedef struct hitl_summary {
  int32_t termination;          /* HITL_CONVERGENCE / NO_CONVERGENCE / FAILURE          */
  int32_t iterations;           /* LM iterations executed (successful + unsuccessful)   */
  int32_t successful_steps;
  int32_t unsuccessful_steps;
  int32_t linear_iterations;    /* total PCG iterations                                 */
  int32_t num_residuals;        /* rows of r                                            */
  int32_t num_hc_residuals;     /* == JointOpt::num_hc_residuals_ (JointOptimization.cpp:973-1050) */
  int32_t band_poses;           /* pose half-width of the banded part                   */
  int64_t num_pairs;            /* directed STF pose pairs in the problem               */
  int64_t num_matches;          /* point matches in the problem                         */
  int64_t num_offband_pairs;
  double initial_cost;
  double final_cost;
  double final_gradient_max_norm;
  double final_trust_region_radius;
  double total_ms;              /* wall clock of hitl_solver_solve                      */
  double linearize_ms;          /* device time of the linearize_timed_calls bracketed k_linearize_stf launches */
  double linear_solver_ms;
  int32_t linearize_calls;
  int32_t linearize_timed_calls; /* how many of them were bracketed by events (hitl_options.timing_level) */
  /* how a pose-range submap rank solved its linear systems: 0 = single GPU; 1 = interface columns (1 + 6 Wi right-hand sides
   * through the rank's own factorisation, dense interface system); 2 = separators (each rank eliminates between its first and
   * last super-block, one chain of 2 * world_size separator blocks: needs every rank but the last to own a multiple of the
   * pose band -- hitl_partition_range_aligned); 3 = distributed PCG (pose pairs outside the band). */
  double submap_linear_solver;
  /* bytes this solve moved (the counterpart of the reference's FunctionTimer / summary prints, JointOptimization.cpp:1158-1202;
   * SURVEY.md 5.1 "per-phase ms, iterations, CG iterations, bytes moved").  hbm_bytes: what the solve's kernels stream through
   * device memory, counted from the array sizes each launch reads and writes once (compulsory bytes of this layout; the PMC
   * counters of profiles/ sit within a few per cent of it); pcie_bytes: host <-> device copies and pinned-memory writes inside
   * hitl_solver_solve (poses in and out, factor tables, the per-iteration partial sums). */
  double hbm_bytes;
  double pcie_bytes;
  /* the persistent PCG launch of systems with pose pairs outside the band (one launch per linear solve): device time of the
   * pcg_timed_calls launches that were bracketed by events (hitl_options.timing_level: 1 = every fourth solve, 2 = all) and the CG
   * iterations those launches ran -- pcg_ms / pcg_timed_iterations is the time of one CG iteration.  All 0 when no such launch ran. */
  double pcg_ms;
  double pcg_timed_calls;
  double pcg_timed_iterations;
} hitl_summary;

typedef struct hitl_solver hitl_solver;   /* opaque; owns device memory */

/* Fill *opt with the defaults listed above. */
void hitl_options_default(hitl_options* opt);

/* Replaces: JointOpt::JointOpt() (JointOptimization.cpp ctor).  Picks the device, creates
 * streams and, when world_size > 1, the RCCL communicator. */
int hitl_solver_create(const hitl_options* opt, hitl_solver** out);
void hitl_solver_destroy(hitl_solver*);

/* Replaces: the hand-off of robot_frame_point_clouds_, robot_frame_normal_clouds_ and
 * covariances_ (HitLSLAM.cpp:450-454; JointOptimization.h:62-71) and BuildKDTrees
 * (JointOptimization.cpp:514-537: with kd_tree_search = 1, the default, every scan's median-split tree is built on the
 * device by the first hitl_solver_find_correspondences after a map load and kept until the next set_map).
 * SoA robot-frame scans: obs_offset has n_poses+1 entries; cov9 (n_poses*9) may be NULL.
 * Arrays are copied to the device; the caller keeps ownership.  In a multi-GPU solver
 * every rank passes the WHOLE map's obs_offset but may pass scan data only for the poses
 * it needs (see hitl_solver_owned_range); simplest is to pass everything. */
int hitl_solver_set_map(hitl_solver*, int64_t n_poses, const int64_t* obs_offset,
                        const float* px, const float* py, const float* nx, const float* ny,
                        const float* cov9);

/* Replaces: human_constraints_ hand-off (HitLSLAM.cpp:454) + AddHumanConstraints
 * (JointOptimization.cpp:969-1054).  Rows are the flattened vector<vector<HumanConstraint>>. */
int hitl_solver_set_human_constraints(hitl_solver*, int64_t n, const hitl_human_constraint_t* rows);

/* Replaces: FindSTFCorrespondences (JointOptimization.cpp:561-642).  Every source pose i
 * is matched against target poses j in [i-window, i+window], j != i, ascending; window <= 0
 * or >= n_poses reproduces the reference's all-pairs search. */
int hitl_solver_find_correspondences(hitl_solver*, const float* poses_xyt, int64_t window);

/* Replaces: BuildKDTrees (JointOptimization.cpp:514-537) + KDTree::BuildKDTree (perception_tools/kdtree.cpp:37-139), read
 * back.  Builds the per-scan trees of the resident map if they do not exist yet (they are built once per map, like
 * `kdtrees_.size() == 0` at :1307) and copies them out in the library's pointer-free layout: the tree of scan i occupies
 * the scan's own range [obs_offset[i], obs_offset[i+1]); the node of a sub-range [lo, lo + n) sits at lo + n / 2, its
 * children are the nodes of [lo, lo + n / 2) and [lo + n / 2 + 1, lo + n).  pts4[4*p ..] = (x, y, nx, ny) of the node's
 * KDNodeValue, meta[p] = KDNodeValue::index | splitting_dimension_ << 16.  Either output may be NULL. */
int hitl_solver_get_kd_trees(hitl_solver*, float* pts4 /* n_obs * 4 */, uint32_t* meta /* n_obs */);

/* Install a correspondence set directly (parity tests; == point_point_glob_correspondences_,
 * vector_mapping.h:102-119 as index pairs).  Directed pairs p: source pose pair_src[p],
 * target pose pair_tgt[p]; matches [match_offset[p], match_offset[p+1]) index into the two
 * scans (src_idx into scan pair_src[p], tgt_idx into scan pair_tgt[p]). */
int hitl_solver_set_correspondences(hitl_solver*, int64_t n_pairs, const int32_t* pair_src,
                                    const int32_t* pair_tgt, const int64_t* match_offset,
                                    const int32_t* src_idx, const int32_t* tgt_idx);

/* Read back the current correspondence set (sizes first, then arrays sized by them). */
int hitl_solver_get_correspondence_counts(hitl_solver*, int64_t* n_pairs, int64_t* n_matches);
int hitl_solver_get_correspondences(hitl_solver*, int32_t* pair_src, int32_t* pair_tgt,
                                    int64_t* match_offset, int32_t* src_idx, int32_t* tgt_idx);

/* Replaces: the `info_mat_` side effect of JointOpt::Run -- an N x N float image (JointOptimization.cpp:1313-1324, written
 * to info_mat.png at :1381-1382; 10 GB at 50 000 poses) in which every pose pair that a factor or a matched point connects
 * is set to 255 -- as the sparse list it is: the symmetric adjacency in coordinate form, upper triangle (row <= col),
 * sorted by (row, col), no duplicates.  Entries (mode selects the factor sets, as for hitl_solver_solve):
 *   odometry edges (i - 1, i)                                  (:744-745; modes A and C)
 *   human rows (anchor_pose_id, constrained_pose_id), any type (:978-979; modes A and C)
 *   pose pairs (i, j) with at least one accepted point match in the last correspondence search -- whether or not the
 *   pair passed the more-than-kMinInterPoseCorrespondence filter (:624-625 marks per match; modes B and C); after
 *   hitl_solver_set_correspondences: the installed pairs.
 * Call with rows == cols == NULL for the count, then with arrays of that size.  Single-GPU entry point. */
int hitl_solver_get_info_mat(hitl_solver*, int32_t mode, int64_t* n_entries, int32_t* rows, int32_t* cols);

/* Replaces: SetParams + SolveHumanConstraints / PostHumanOptimization + CopyParams, i.e. the
 * ceres::Solve call sites (JointOptimization.cpp:392-402, :1064-1138, :1156-1256, :380-390).
 * poses_in_xyt / poses_out_xyt are n_poses*3 fp32 (x,y,theta); they may alias.  Synchronous. */
int hitl_solver_solve(hitl_solver*, int32_t mode, const float* poses_in_xyt,
                      float* poses_out_xyt, hitl_summary* summary);

/* Replaces: problem.Evaluate(..., &residuals, &gradients_, ...) (JointOptimization.cpp:1247-1252).
 * cost = 1/2 |r|^2; gradient (3*n_poses, pose 0 entries are 0) may be NULL.  The odometry
 * measurements and human targets are re-derived from poses_xyt exactly as a solve would.
 * Submap ranks (collective: every rank calls it; mode A or C): poses_xyt is the GLOBAL array, cost the global cost (every
 * factor is counted by the rank owning it), gradient (3 * n_global) receives the rank's OWNED rows and zeros elsewhere. */
int hitl_solver_evaluate(hitl_solver*, int32_t mode, const float* poses_xyt,
                         double* cost, double* gradient);

/* Replaces: problem.Evaluate(evalOptions, NULL, &residuals, &gradients_, &ceres_jacobian_) at the end of
 * PostHumanOptimization (JointOptimization.cpp:1247-1252): the scan-to-scan problem's residual vector and its Jacobian
 * as a ceres::CRSMatrix (members JointOptimization.h:79-86).  Rows: 2 per directed pose pair, in the order of the
 * correspondence set as the caller installed / would read it back (hitl_solver_get_correspondences).  Columns: 3 per
 * pose (x, y, theta), all n_poses poses -- the columns of the constant pose 0 are present and hold the block's
 * partials (Ceres drops constant blocks from its column space; shift by 3 to compare).  Every row has 6 entries, column
 * indices ascending.  Sizes: residuals 2P, crs_rows 2P + 1, crs_cols / crs_values 12P, P = number of pairs
 * (hitl_solver_get_correspondence_counts); any output may be NULL.  Single-GPU entry point. */
int hitl_solver_evaluate_jacobian(hitl_solver*, const float* poses_xyt, double* cost, double* residuals,
                                  int32_t* crs_rows, int32_t* crs_cols, double* crs_values);

/* ---- the O(N*M) observation scans either side of the solve ----------
 * They run over the resident scans with the robot->world transform fused in, so the
 * WORLD_FRAME_point_clouds_ copies of the reference never have to exist.  fp32 like the
 * reference; results are bit-identical to the CPU restatement.  (Submap ranks: see the end of this comment.)
 * hitl_solver_transform_scans  == HitLSLAM::transformPointCloudsToWorldFrame (HitLSLAM.cpp:245-254):
 *     world_x/world_y receive n_obs values each (pose_transform * point).
 * hitl_solver_verify_points    == the scan of HitLSLAM::verifyUserInput (HitLSLAM.cpp:218-243):
 *     seen[q] = 1 iff some world-frame observation lies within `threshold` of points_xy[q]
 *     (the caller sums them and applies the two equality checks of :237-241).  n_points <= 16.
 * hitl_solver_observation_sets == EMInput::EstablishObservationSets (EMinput.cpp:281-323) with
 *     distToLineSeg (:268-279): seg_xy8 = selected_points_[0..3]; counts[2*i + s] = number of
 *     observations of pose i within `threshold` (0.03 in the reference) of selection s; mask (may be
 *     NULL) gets bit s set per observation.  The caller keeps poses with counts > 5 (:309-316).
 * Submap ranks (world_size > 1): poses_xyt is the GLOBAL pose array; the three calls run over the rank's RESIDENT scans --
 * global poses [begin - hl, end + hh), halos included, in that order -- so world_x / world_y / mask hold one value per
 * resident observation and counts two per resident pose; `seen` flags are per rank (the caller ORs them). */
int hitl_solver_transform_scans(hitl_solver*, const float* poses_xyt, float* world_x, float* world_y);
/* The same re-projection, left where a renderer or a later kernel wants it: *d_world_x / *d_world_y receive DEVICE pointers
 * to n_obs floats each, owned by the solver and valid until the next transform_scans* call, set_map or destroy.  The 160 MB
 * of world-frame clouds of a 20 M-observation map never cross PCIe (0.1 ms instead of ~11 ms into pageable host memory).
 * hitl_solver_transform_scans itself copies at pinned-memory speed when world_x / world_y come from hitl_host_alloc. */
int hitl_solver_transform_scans_device(hitl_solver*, const float* poses_xyt, const float** d_world_x, const float** d_world_y);
/* Page-locked host memory for the large outputs of the library (transform_scans, get_correspondences ...): copies into it
 * run at the full PCIe rate and need no staging.  NULL on failure. */
void* hitl_host_alloc(size_t bytes);
void hitl_host_free(void* p);
int hitl_solver_verify_points(hitl_solver*, const float* poses_xyt, int32_t n_points, const float* points_xy,
                              float threshold, int32_t* seen);
int hitl_solver_observation_sets(hitl_solver*, const float* poses_xyt, const float* seg_xy8, double threshold,
                                 int32_t* counts, uint8_t* mask);

/* ---- the correction cycle around the solve (HitLSLAM::Run, HitLSLAM.cpp:380-480) -----
 * One user correction is: verify_points -> fit_segments -> observation_sets -> order_user_input ->
 * apply_explicit_correction -> backprop -> normalize_angles -> set_human_constraints + solve -> transform_scans.
 * fp32 like the reference; float transcendental scalars are the C library's.  On submap ranks: fit_segments takes the
 * GLOBAL pose array and the SAME segments on every rank, scans the owned poses, all-gathers the inlier points in pose order
 * and runs the same M step everywhere (identical results on all ranks, equal to the single-GPU ones); the pose-array steps
 * (order_user_input, apply_explicit_correction, backprop, normalize_angles) need no map and run on any rank.
 *
 * hitl_solver_fit_segments == EMInput::AutomaticEndpointAdjustment + SegFitEM (EMinput.cpp:153-250): each of the
 *     n_segments rows of seg_xy4 (x0,y0,x1,y1; in/out) is re-fitted to the world-frame observations within 3 cm of it
 *     (Eigen::DistanceToLineSegment as written, shared/math/eigen_helper.h:66-81) by a 1-parameter trust-region solve
 *     (25 iterations, Ceres defaults), repeated until both endpoints move less than 5 cm.  Outputs (may be NULL):
 *     inliers of the last pass, number of passes, total LM iterations per segment.
 * hitl_order_user_input == EMInput::OrderAndFilterUserInput (EMinput.cpp:325-455), pure index logic on the counts
 *     of hitl_solver_observation_sets: corrected / anchor pose ids (capacity n_poses each), backprop bounds
 *     ((-1,-1) on overlap); seg_xy8 is swapped in place when the selections came in the wrong order.
 * hitl_solver_apply_explicit_correction == AppExpCorrect::Run (ApplyExplicitCorrection.cpp:150-180, 229-494) for
 *     types 2, 4, 5, 6: moves the first contiguous group of corrected poses and everything after it (poses_xyt in/out),
 *     returns correction_ and the n_anchor x n_corrected new_human_constraints_ rows (anchor-major).
 * hitl_solver_backprop == Backprop::Run (Backprop.cpp:98-213): poses in [lo, hi] and covariances in [lo, hi) are
 *     updated in place; the two O(span^2) loops are reproduced update by update (bit-identical fp32 poses).
 * hitl_normalize_angles == HitLSLAM.cpp:442-447. */
int hitl_solver_fit_segments(hitl_solver*, const float* poses_xyt, int32_t n_segments, float* seg_xy4, int32_t* n_inliers,
                             int32_t* passes, int32_t* lm_iterations);
int hitl_order_user_input(int64_t n_poses, const int32_t* counts, float* seg_xy8, int32_t* corrected, int32_t* n_corrected,
                          int32_t* anchor, int32_t* n_anchor, int32_t bounds[2]);
int hitl_solver_apply_explicit_correction(hitl_solver*, uint32_t type, const float* seg_xy8, int64_t n_poses, float* poses_xyt,
                                          int32_t n_corrected, const int32_t* corrected, int32_t n_anchor, const int32_t* anchor,
                                          float correction[3], hitl_human_constraint_t* rows);
int hitl_solver_backprop(hitl_solver*, int64_t n_poses, float* poses_xyt, float* cov9, int32_t lo, int32_t hi,
                         const float correction[3]);
int hitl_normalize_angles(int64_t n_poses, float* poses_xyt);

/* Pose covariances == the ceres::Covariance block at
 * episodic_non_markov_localization/vector_mapping.cpp:2772-2812: for every pose the 3x3 diagonal block of (J^T J)^-1 at
 * poses_xyt (constant poses removed, no trust-region diagonal), rotated into the robot frame with
 * transform = blockdiag(Rotation2Df(-angle), 1).  cov9 = n_poses*9 doubles, row-major; zeros for constant poses
 * (the reference never writes pose min_poses).  Every pose pair inside the band: from the cyclic-reduction factors by selected
 * inversion -- no dense inverse, no SVD.  Pose pairs outside the band (the all-pairs search on a map that revisits places; one GPU,
 * at most 10 922 poses): the dense form of the scaled J^T J, inverted by rocSOLVER's Cholesky (bound at run time with dlopen: no
 * link-time dependency; the first such call loads it, ~0.9 s; HITL_ERR_UNSUPPORTED where librocsolver.so is not to be found).
 * Submap ranks (collective; mode A or C): poses_xyt is the GLOBAL array, cov9 (9 * n_global) receives the blocks of the rank's
 * OWNED poses (zeros elsewhere) -- blocks of the GLOBAL inverse: the rank's own selected inversion minus a low-rank correction
 * through its interface columns, with one all-gather of the interface record. */
int hitl_solver_pose_covariances(hitl_solver*, int32_t mode, const float* poses_xyt, double* cov9);

/* Debug / parity: linearise at double-precision poses x (3*n_poses) with the odometry
 * measurements and human targets derived from the fp32 poses_ref_xyt; returns the cost, the
 * gradient (3n), the per-pair STF residuals (2 per pair) and Jacobians (12 per pair,
 * row-major 2x6) -- any output may be NULL. */
int hitl_solver_linearize(hitl_solver*, int32_t mode, const float* poses_ref_xyt, const double* x,
                          double* cost, double* gradient, double* pair_residuals,
                          double* pair_jacobians);

/* ---- pose-range submaps across GPUs (world_size > 1; nothing in the reference) ------------
 * One process per GPU; rank r owns the contiguous GLOBAL pose range [begin, end) and also holds
 * the scans and state of `halo_poses` poses on each side (fewer at the trajectory's ends), so it
 * can evaluate EVERY factor that touches one of its own poses -- the odometry edges across its
 * cuts and the scan-to-scan pairs within the matching window of a cut included.  Its rows of the
 * normal equations are therefore complete without exchanging any Hessian data; the shared
 * blocks are simply computed on both sides.  halo_poses must be >= the pose band of the
 * correspondence set (<= 4) and should be 2 x the matching window, so that the per-point cap of
 * the correspondence search (counted across targets) sees the same targets as a global search.
 * Protocol per rank:
 *   hitl_solver_create(opt with rank/world_size [+ rccl_unique_id])
 *   hitl_solver_set_partition(begin, n_global, halo_poses)        -- before set_map
 *   hitl_solver_set_map(n_local, ...)   -- scans of GLOBAL poses [begin - hl, end + hh), hl = halo_poses
 *                                          on rank > 0 else 0, hh = halo_poses on rank < world-1 else 0
 *   hitl_solver_set_human_constraints(rows)   -- GLOBAL pose ids; rows constraining another rank's
 *                                                pose are ignored here
 *   hitl_solver_find_correspondences(poses GLOBAL n_global*3, window)
 *   hitl_solver_solve(mode, poses_in GLOBAL, poses_out GLOBAL, summary)
 *        every rank takes identical LM decisions; poses_out receives the owned range (other
 *        entries are copied from poses_in); summary costs/iterations are global.
 * Per LM iteration the ranks exchange TWO small all-gathers.  The first belongs to the linear solve and has two forms
 * (hitl_summary.submap_linear_solver): separators (round 4; when every rank but the last owns a multiple of the pose
 * band, see hitl_partition_range_aligned) -- each rank eliminates everything between its first and last owned
 * super-block and contributes their Schur complement (4 B^2 + 4 B doubles, B = 3 * band); the chain of 2 * world
 * separators is solved on the device, identically on every rank, and with the built-in RCCL transport the host is
 * not involved at all -- or interface columns (any partition): the interface rows of a multi-rhs solve + Jacobi
 * scales + cross-cut blocks (<= 5 KB per rank) and a 6*band*world interface system solved on the host.  Both
 * eliminate the coupling exactly.  The second all-gather carries 7 doubles per rank: the step scalars together with
 * (cost, |g|max) of the candidate, whose linearisation is queued speculatively behind the step.  (cost, |g|max) of
 * the initial linearisation travel once, before the loop.
 * Pose pairs farther apart than the band (window > max_band_poses, the reference's all-pairs search
 * JointOptimization.cpp:575-635, revisits) on ANY rank switch all ranks to a distributed PCG: block-Jacobi
 * preconditioner = each rank's own band factorisation, two all-gathers per CG iteration -- (r.z, r.r, b.b) + the
 * owned part of one vector (halo rows are filled from it), then p.q.  A pair that straddles a cut must reach no
 * deeper than halo_poses on either side (both ranks it touches must see it), else HITL_ERR_UNSUPPORTED on every
 * rank; halos may span several ranks (two ranks whose halos cover the whole map = all-pairs search over the map).
 * The all-gather is either the built-in RCCL one (opt.rccl_unique_id != NULL) or a caller
 * supplied callback on HOST buffers (tests use torch.distributed/gloo). */
typedef int (*hitl_allgather_fn)(void* ctx, const void* send, void* recv, int64_t bytes_per_rank);
typedef struct hitl_comm_ops {
  hitl_allgather_fn allgather;   /* recv holds world_size * bytes_per_rank, rank order; 0 = OK */
  void* ctx;
} hitl_comm_ops;
int hitl_solver_set_comm(hitl_solver*, const hitl_comm_ops* ops);
int hitl_solver_set_partition(hitl_solver*, int64_t global_begin, int64_t n_global_poses, int32_t halo_poses);
/* The same with the owned range [global_begin, global_end) spelled out, so that the halos can be clipped at the ends of the
 * map: the rank then holds the scans of [begin - min(halo, begin), end + min(halo, n_global - end)).  halo_poses < 0 =
 * REPLICATED SCANS: every rank holds the whole map (C4 is 320 MB of 288 GB) and owns the rows of its pose range -- a pose pair
 * may then reach any distance across any number of cuts (the reference's all-pairs loop, JointOptimization.cpp:575-635, and
 * revisits; the off-band blocks go through the distributed PCG), where a finite halo refuses pairs that reach deeper than it. */
int hitl_solver_set_partition_range(hitl_solver*, int64_t global_begin, int64_t global_end, int64_t n_global_poses, int32_t halo_poses);
/* one small all-gather through the active communicator, payload verified (bring-up check) */
int hitl_solver_comm_selftest(hitl_solver*);
/* ranks the active transport really holds: ncclCommCount of the built-in RCCL communicator, world_size for a caller
 * callback, 1 when neither is installed.  kind (may be NULL): 0 = none, 1 = built-in RCCL, 2 = caller callback. */
int hitl_solver_comm_ranks(hitl_solver*, int32_t* ranks, int32_t* kind);

/* Multi-GPU helpers.  hitl_comm_unique_id: rank 0 calls it and broadcasts the 128 bytes.
 * hitl_partition_range: the contiguous pose range rank r of w owns for n_poses poses,
 * balanced by observation count (host-only; runs without a GPU). */
int hitl_comm_unique_id(void* out128);
int hitl_partition_range(int64_t n_poses, const int64_t* obs_offset, int32_t rank,
                         int32_t world_size, int64_t* begin, int64_t* end);
/* The same with every cut moved to the nearest multiple of `align` poses (counted from pose 0), so that every rank but the
 * last owns a multiple of it.  With align = a common multiple of the pose bands the solves may use (the search window; 12
 * covers bands of 1, 2, 3, 4 and 6 poses) the ranks' linear solves take the separator form (hitl_summary.submap_linear_solver
 * = 2), whose cost does not depend on how many poses a factor reaches across a cut. */
int hitl_partition_range_aligned(int64_t n_poses, const int64_t* obs_offset, int32_t rank, int32_t world_size, int32_t align,
                                 int64_t* begin, int64_t* end);

/* ---- file formats either side of the path (host-only, no GPU needed) ----
 * hitl_io_load_stfs_covars   == loadPoseGraph      (HitLSLAM_main.cpp:192-300): rows of 16 numbers
 *     "pose_x,pose_y,pose_theta,obs_x,obs_y, n_x,n_y, c00..c22" with obs/normal in the WORLD frame;
 *     rows are grouped into poses on any bit difference of (x,y,theta); scans come back in the
 *     robot frame, SoA, ready for hitl_solver_set_map.  fix_normals = 0 reproduces the reference's
 *     normal transform R(-theta)(n - t) (:269, :293); != 0 rotates normals only.
 * hitl_io_save_stfs_covars   == SaveStfsandCovars  (vector_mapping_main.cpp:1855-1901), "%.4f" x7, "%f" x9.
 * hitl_io_save_results       == saveHitLResults    (HitLSLAM_main.cpp:572-581), "%f %f %f\n" per pose. */
typedef struct hitl_posegraph hitl_posegraph;
typedef struct hitl_input_log hitl_input_log;
int hitl_io_load_stfs_covars(const char* path, int fix_normals, hitl_posegraph** out);
void hitl_posegraph_free(hitl_posegraph*);
int64_t hitl_posegraph_num_poses(const hitl_posegraph*);
int64_t hitl_posegraph_num_obs(const hitl_posegraph*);
double hitl_posegraph_timestamp(const hitl_posegraph*);
const char* hitl_posegraph_map_name(const hitl_posegraph*);
const float* hitl_posegraph_poses(const hitl_posegraph*);        /* n*3 (x, y, theta)      */
const float* hitl_posegraph_cov9(const hitl_posegraph*);         /* n*9 row-major          */
const int64_t* hitl_posegraph_obs_offset(const hitl_posegraph*); /* n+1                    */
const float* hitl_posegraph_px(const hitl_posegraph*);
const float* hitl_posegraph_py(const hitl_posegraph*);
const float* hitl_posegraph_nx(const hitl_posegraph*);
const float* hitl_posegraph_ny(const hitl_posegraph*);
int hitl_io_save_stfs_covars(const char* path, const char* map_name, double timestamp, int64_t n_poses,
                             const float* poses_xyt, const int64_t* obs_offset, const float* px, const float* py,
                             const float* nx, const float* ny, const float* cov9 /* may be NULL */);
int hitl_io_save_results(const char* path, int64_t n_poses, const float* poses_xyt);
/* The recording of a session's corrections: hitl_io_load_input_log == LoadLogFile (HitLSLAM_main.cpp:676-764),
 * hitl_io_save_input_log == LogActivity (:776-822; "%d \n", "%d, %d\n", "%.4f, %.4f\n").  An entry is
 * (type, undone, its clicked points); 2 points for type 1, 8 for type 3, 4 for types 2, 4, 5, 6.  The reference reader
 * maps the number 7 (never written) to the parallel type and leaves a logged 6 uninitialised: reproduced as
 * HITL_HC_UNKNOWN unless fix_parallel != 0.  Each entry is replayed by one correction cycle (HitLSLAM::replayLog,
 * HitLSLAM.cpp:278-360), whether or not it is marked undone (the skip at HitLSLAM_main.cpp:889-896 is commented out). */
int hitl_io_load_input_log(const char* path, int fix_parallel, hitl_input_log** out);
void hitl_input_log_free(hitl_input_log*);
int32_t hitl_input_log_num_entries(const hitl_input_log*);
int hitl_input_log_entry(const hitl_input_log*, int32_t i, uint32_t* type, int32_t* undone, int32_t* n_points,
                         const float** points_xy);
int hitl_io_save_input_log(const char* path, int32_t n_entries, const uint32_t* types, const int32_t* undone,
                           const int32_t* n_points, const float* points_xy);

/* Changes hitl_options.timing_level of a live solver (0 none, 1 every 4th scan-to-scan launch bracketed by HIP
 * events, 2 every launch + the linear solve of every LM iteration: adds ~20 us of queue drain per iteration). */
int hitl_solver_set_timing_level(hitl_solver*, int32_t level);

/* Free / total memory of a HIP device in bytes (hipMemGetInfo); for leak checks and capacity planning. */
int hitl_device_memory(int32_t device, int64_t* free_bytes, int64_t* total_bytes);
/* Measured device-to-device copy rate of a HIP device: a float4-per-thread copy kernel over `bytes` (>= 1 GiB recommended; rounded
 * down to 16), `repeats` launches between two HIP events after 2 warm-up launches; *gb_per_s counts bytes read + bytes written.
 * The measured ceiling SURVEY.md 8(d) asks for beside the 8 TB/s vendor peak (bench.py `roofline.measured_copy_ceiling`). */
int hitl_device_copy_bandwidth(int32_t device, int64_t bytes, int32_t repeats, double* gb_per_s);
const char* hitl_status_string(int status);
const char* hitl_last_error(void);       /* thread-local detail of the last failure      */
int hitl_abi_version(void);
/* Host check behind the correspondence search's trigonometry: 1 if sincosf(x) == (sinf(x), cosf(x)) and sinf(-x) == -sinf(x),
 * cosf(-x) == cosf(x) hold bit for bit on a 200 000-value sample of this C library (glibc: exhaustively true); 0 makes the
 * search fall back to one cosf and one sinf per candidate.  No GPU involved. */
int hitl_libm_sincos_exact(void);

#ifdef __cplusplus
}
#endif
#endif  /* HITL_GPU_H_ */
