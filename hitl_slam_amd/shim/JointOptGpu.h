// JointOptGpu.h -- drop-in replacement for the reference's JointOpt class
// (HitL-SLAM/src/human_in_the_loop_slam/JointOptimization.h:49-89) whose Run() body is
// marshaling + libhitlgpu.so calls.  Same public data members in,
// Run(), poses_ out, so HitLSLAM::Run / replayLog (HitLSLAM.cpp:345-353, :450-458) compile
// unchanged.  Header-only; needs only include/hitl_gpu.h and -lhitlgpu.
//
// In the reference tree:
//     #include "JointOptGpu.h"
//     typedef hitl::JointOptGpuT<perception_2d::Pose2Df, Eigen::Vector2f, Eigen::Matrix3f, HumanConstraint> JointOpt;
// The template only asks of its parameters what the reference types offer:
//     Pose2Df : .translation.x(), .translation.y(), .angle       (perception_2d.h:31-55)
//     Vec2f   : .x(), .y()                                       (Eigen::Vector2f)
//     Mat3f   : operator()(row, col)                             (Eigen::Matrix3f)
//     HumanConstraint : the 28-byte POD of human_constraints.h:29-41
#ifndef HITL_JOINT_OPT_GPU_H_
#define HITL_JOINT_OPT_GPU_H_

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hitl_gpu.h"

namespace hitl {

template <class Pose2Df, class Vec2f, class Mat3f, class HumanConstraint>
class JointOptGpuT {
 public:
  JointOptGpuT() : num_hc_residuals_(0), mode_(HITL_MODE_ODOM_HC), stf_window_(80), solver_(NULL), map_poses_(0), stf_moments_(false) {
    std::memset(&summary_, 0, sizeof(summary_));
  }
  virtual ~JointOptGpuT() { if (solver_) hitl_solver_destroy(solver_); }

  // ---- the reference's public interface (JointOptimization.h:55-89) ----------------------
  std::vector<Pose2Df> poses_;
  std::vector<std::vector<Vec2f> > robot_frame_point_clouds_;
  std::vector<std::vector<Vec2f> > robot_frame_normal_clouds_;
  std::vector<Mat3f> covariances_;
  std::vector<std::vector<HumanConstraint> > human_constraints_;
  std::vector<double> gradients_;
  int num_hc_residuals_;

  // Reference semantics: exit(1) on solver FAILURE (JointOptimization.cpp:1339-1342),
  // NO_CONVERGENCE accepted (:1343-1349).
  void Run() {
    const int st = RunStatus();
    if (st == HITL_ERR_SOLVER_FAILURE) { std::printf("\nCeres failure\n\n"); std::exit(1); }
    if (st != HITL_OK) { std::fprintf(stderr, "hitl-gpu: %s (%s)\n", hitl_status_string(st), hitl_last_error()); std::exit(1); }
  }
  void ClearPoses() { poses_.clear(); robot_frame_point_clouds_.clear(); robot_frame_normal_clouds_.clear(); covariances_.clear(); map_poses_ = 0; }
  std::vector<Mat3f> GetPoseCovariances() { return covariances_; }

  // ---- extensions ------------------------------------------------------------------------
  // mode: HITL_MODE_ODOM_HC is what the shipped binary runs (SolveHumanConstraints, :1335);
  // HITL_MODE_JOINT adds the scan-to-scan factors of the commented-out PostHumanOptimization
  // (:1353-1373), matched within +-stf_window poses (kMaxHistory-style window, default 80).
  void SetMode(int mode, int64_t stf_window = 80) { mode_ = mode; stf_window_ = stf_window; }
  const hitl_summary& Summary() const { return summary_; }
  bool want_gradients = false;     // fill gradients_ like problem.Evaluate at :1247-1252

  // The session's solver (created on first use): the correction-cycle entry points (hitl_solver_fit_segments,
  // hitl_solver_apply_explicit_correction, hitl_solver_backprop ...) run on the same resident map.
  hitl_solver* Solver() {
    if (!solver_) {
      hitl_options opt;
      hitl_options_default(&opt);
      opt.stf_moments = stf_moments_ ? 1 : 0;
      create_status_ = hitl_solver_create(&opt, &solver_);
      if (create_status_ != HITL_OK) solver_ = NULL;
    }
    return solver_;
  }
  // opt-in moments form of the scan-to-scan factors (hitl_options.stf_moments); call before the first Run()
  void SetStfMoments(bool on) { stf_moments_ = on; }

  int RunStatus() {
    if (!Solver()) return create_status_;
    const size_t n = poses_.size();
    if (n == 0 || robot_frame_point_clouds_.size() != n || robot_frame_normal_clouds_.size() != n) return HITL_ERR_INVALID_ARGUMENT;
    if (map_poses_ != n) {   // == BuildKDTrees on the first call (:1307): the scans never change afterwards
      std::vector<int64_t> off(n + 1, 0);
      for (size_t i = 0; i < n; ++i) off[i + 1] = off[i] + (int64_t)robot_frame_point_clouds_[i].size();
      std::vector<float> px(off[n]), py(off[n]), nx(off[n]), ny(off[n]), cov(9 * n, 0.f);
      for (size_t i = 0; i < n; ++i) {
        for (size_t k = 0; k < robot_frame_point_clouds_[i].size(); ++k) {
          px[off[i] + k] = robot_frame_point_clouds_[i][k].x(); py[off[i] + k] = robot_frame_point_clouds_[i][k].y();
          nx[off[i] + k] = robot_frame_normal_clouds_[i][k].x(); ny[off[i] + k] = robot_frame_normal_clouds_[i][k].y();
        }
        if (covariances_.size() == n)
          for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) cov[9 * i + 3 * r + c] = covariances_[i](r, c);
      }
      const int st = hitl_solver_set_map(solver_, (int64_t)n, off.data(), px.data(), py.data(), nx.data(), ny.data(), cov.data());
      if (st != HITL_OK) return st;
      map_poses_ = n;
    }
    static_assert(sizeof(HumanConstraint) == sizeof(hitl_human_constraint_t), "HumanConstraint layout (human_constraints.h:29-41)");
    std::vector<hitl_human_constraint_t> rows;
    for (size_t g = 0; g < human_constraints_.size(); ++g)
      for (size_t j = 0; j < human_constraints_[g].size(); ++j) {
        hitl_human_constraint_t r;
        std::memcpy(&r, &human_constraints_[g][j], sizeof(r));
        rows.push_back(r);
      }
    int st = hitl_solver_set_human_constraints(solver_, (int64_t)rows.size(), rows.data());
    if (st != HITL_OK) return st;
    std::vector<float> xyt(3 * n);     // == SetParams (:392-402)
    for (size_t i = 0; i < n; ++i) { xyt[3 * i] = poses_[i].translation.x(); xyt[3 * i + 1] = poses_[i].translation.y(); xyt[3 * i + 2] = poses_[i].angle; }
    if (mode_ != HITL_MODE_ODOM_HC) {
      st = hitl_solver_find_correspondences(solver_, xyt.data(), stf_window_);
      if (st != HITL_OK) return st;
    }
    st = hitl_solver_solve(solver_, mode_, xyt.data(), xyt.data(), &summary_);
    if (st != HITL_OK && st != HITL_ERR_SOLVER_FAILURE) return st;
    if (st == HITL_ERR_SOLVER_FAILURE) return st;
    num_hc_residuals_ = summary_.num_hc_residuals;
    for (size_t i = 0; i < n; ++i) {   // == CopyParams (:380-390)
      poses_[i].translation.x() = xyt[3 * i]; poses_[i].translation.y() = xyt[3 * i + 1]; poses_[i].angle = xyt[3 * i + 2];
    }
    if (want_gradients) {
      gradients_.assign(3 * n, 0.0);
      double cost = 0.0;
      st = hitl_solver_evaluate(solver_, mode_, xyt.data(), &cost, gradients_.data());
      if (st != HITL_OK) return st;
    }
    return HITL_OK;
  }

 private:
  int mode_;
  int64_t stf_window_;
  hitl_solver* solver_;
  size_t map_poses_;
  bool stf_moments_;
  int create_status_ = HITL_OK;
  hitl_summary summary_;
};

}  // namespace hitl
#endif  // HITL_JOINT_OPT_GPU_H_
