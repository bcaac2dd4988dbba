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

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <utility>
#include <vector>

#include "hitl_gpu.h"

namespace hitl {

// same members as ceres::CRSMatrix (what JointOpt::ceres_jacobian_ is, JointOptimization.h:86)
struct CRSMatrix {
  CRSMatrix() : num_rows(0), num_cols(0) {}
  int num_rows, num_cols;
  std::vector<int> cols, rows;
  std::vector<double> values;
};

template <class Pose2Df, class Vec2f, class Mat3f, class HumanConstraint>
class JointOptGpuT {
 public:
  JointOptGpuT() : num_hc_residuals_(0), mode_(HITL_MODE_ODOM_HC), stf_window_(3), solver_(NULL), map_poses_(0), map_print_(0), stf_moments_(false), brute_force_search_(false) {
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
  CRSMatrix ceres_jacobian_;          // filled with gradients_ when want_gradients and the mode has scan-to-scan factors
  // info_mat_ (JointOptimization.h:256; an N x N float image saved as info_mat.png, :1313-1324, :1381-1382) as what it
  // encodes: the pose pairs connected by a factor or a matched point, upper triangle, sorted, filled when want_info_mat
  std::vector<std::pair<int, int> > info_mat_;
  int num_hc_residuals_;
  // JointOptimization.cpp:342-344 returns ceres_cost_, which nothing ever fills (its code at :836-838 is commented out):
  // an empty vector, as in the reference
  std::vector<float> GetCeresCost() { return ceres_cost_; }

  // Reference semantics: exit(1) on solver FAILURE (JointOptimization.cpp:1339-1342),
  // NO_CONVERGENCE accepted (:1343-1349).
  void Run() {
    const int st = RunStatus();
    if (st == HITL_ERR_SOLVER_FAILURE) { std::printf("\nCeres failure\n\n"); std::exit(1); }
    if (st != HITL_OK) { std::fprintf(stderr, "hitl-gpu: %s (%s)\n", hitl_status_string(st), hitl_last_error()); std::exit(1); }
  }
  void ClearPoses() { poses_.clear(); robot_frame_point_clouds_.clear(); robot_frame_normal_clouds_.clear(); covariances_.clear(); map_poses_ = 0; map_print_ = 0; }
  std::vector<Mat3f> GetPoseCovariances() { return covariances_; }

  // ---- extensions ------------------------------------------------------------------------
  // mode: HITL_MODE_ODOM_HC is what the shipped binary runs (SolveHumanConstraints, :1335);
  // HITL_MODE_JOINT adds the scan-to-scan factors of the commented-out PostHumanOptimization
  // (:1353-1373), matched within +-stf_window poses.  Default 3: within hitl_options.max_band_poses (4) every pair stays
  // in the block band and the linear solve is the exact cyclic reduction.  A wider window (the legacy kMaxHistory = 80, or
  // 0 = all pairs like FindSTFCorrespondences) takes the off-band PCG path (since round 6 one persistent launch per linear
  // solve and one host wait per LM iteration; pose_covariances through a dense inverse, up to 10 922 poses) and a candidate
  // table of n * 2*window * max_scan * 2 bytes.
  void SetMode(int mode, int64_t stf_window = 3) { mode_ = mode; stf_window_ = stf_window; }
  // The correspondence search is the reference's own (per-scan KD-trees + pruned FindNearestPointNormal walk) unless this is
  // called with true before the first Run(): then the brute-force definition of the same query (hitl_options.kd_tree_search = 0)
  void SetBruteForceSearch(bool on) { brute_force_search_ = on; }
  // forget the resident map: the next Run() uploads the scans again whatever their fingerprint says
  void InvalidateMap() { map_poses_ = 0; map_print_ = 0; }
  const hitl_summary& Summary() const { return summary_; }
  bool want_gradients = false;     // fill gradients_ like problem.Evaluate at :1247-1252
  bool want_info_mat = false;      // fill info_mat_

  // The session's solver (created on first use): the correction-cycle entry points (hitl_solver_fit_segments,
  // hitl_solver_apply_explicit_correction, hitl_solver_backprop ...) run on the same resident map.
  hitl_solver* Solver() {
    if (!solver_) {
      hitl_options opt;
      hitl_options_default(&opt);
      opt.stf_moments = stf_moments_ ? 1 : 0;
      opt.kd_tree_search = brute_force_search_ ? 0 : 1;
      create_status_ = hitl_solver_create(&opt, &solver_);
      if (create_status_ != HITL_OK) solver_ = NULL;
    }
    return solver_;
  }
  // opt-in moments form of the scan-to-scan factors (hitl_options.stf_moments); call before the first Run()
  void SetStfMoments(bool on) { stf_moments_ = on; }
  // host time of the last Run(): the map fingerprint alone, and the whole call up to CopyParams (fingerprint, marshaling,
  // search, solve); threads the fingerprint may use (default: min(8, hardware threads); 1 = no threads at all)
  double LastFingerprintMs() const { return fingerprint_ms_; }
  double LastRunMs() const { return run_ms_; }
  void SetHashThreads(int t) { hash_threads_ = t < 1 ? 1 : t; }

  // 64-bit fingerprint of all scans: every bit of every point and normal and every scan size enters it
  unsigned long long Fingerprint() const {
    const size_t n = robot_frame_point_clouds_.size();
    std::vector<unsigned long long> per(n, 0ull);
    int nt = hash_threads_ > 0 ? hash_threads_ : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 8 && hash_threads_ <= 0) nt = 8;
    if ((size_t)nt > n / 64 + 1) nt = (int)(n / 64 + 1);
    const JointOptGpuT* self = this;
    unsigned long long* out = per.data();
    bool threaded = false;
    if (nt > 1) {
      std::vector<std::thread> th;
      try {
        for (int t = 0; t < nt; ++t) th.push_back(std::thread(&JointOptGpuT::HashRange, self, n * t / nt, n * (t + 1) / nt, out));
        threaded = true;
      } catch (...) { threaded = false; }            // no thread to be had: the started ones finish, then the serial loop redoes all
      for (size_t t = 0; t < th.size(); ++t) th[t].join();
    }
    if (!threaded) HashRange(this, 0, n, out);
    unsigned long long h = 0x9e3779b97f4a7c15ull ^ (unsigned long long)n;
    for (size_t i = 0; i < n; ++i) { h = (h ^ per[i]) * 0x100000001b3ull; h ^= h >> 29; }
    return h;
  }

  int RunStatus() {
    if (!Solver()) return create_status_;
    const size_t n = poses_.size();
    if (n == 0 || robot_frame_point_clouds_.size() != n || robot_frame_normal_clouds_.size() != n) return HITL_ERR_INVALID_ARGUMENT;
    // == BuildKDTrees on the first call (:1307).  The scans normally never change afterwards, but HitLSLAM::Run re-copies
    // every cloud into this object before every solve (HitLSLAM.cpp:450-454), so neither pointers nor sizes can say "same
    // map": a different map -- any bit of any point or normal, any scan size -- is uploaded again, the same map is not.
    // The fingerprint therefore reads every coordinate on every Run(); it is the one O(n_obs) host cost of a steady-state
    // call, so it runs over the raw 64-bit words of each scan with four independent multiply-mix chains, scans spread over
    // a few threads (20 M observations = 320 MB: ~4 ms on 8 threads, measured by tests/shim_driver.cpp, against 25-50 ms
    // for the byte-serial form of round 3).  LastFingerprintMs() / LastRunMs() report what a call spent.
    const std::chrono::steady_clock::time_point t_run0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n; ++i)
      if (robot_frame_normal_clouds_[i].size() != robot_frame_point_clouds_[i].size()) return HITL_ERR_INVALID_ARGUMENT;
    const unsigned long long print = Fingerprint();
    fingerprint_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run0).count();
    if (map_poses_ != n || map_print_ != print) {
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
      map_print_ = print;
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
    run_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run0).count();   // what JointOpt::Run costs the caller
    if (want_info_mat) {
      int64_t ne = 0;
      st = hitl_solver_get_info_mat(solver_, mode_, &ne, NULL, NULL);
      if (st != HITL_OK) return st;
      std::vector<int32_t> r(ne > 0 ? ne : 1), c(ne > 0 ? ne : 1);
      st = hitl_solver_get_info_mat(solver_, mode_, &ne, r.data(), c.data());
      if (st != HITL_OK) return st;
      info_mat_.resize((size_t)ne);
      for (int64_t k = 0; k < ne; ++k) info_mat_[(size_t)k] = std::make_pair((int)r[k], (int)c[k]);
    }
    if (want_gradients) {
      gradients_.assign(3 * n, 0.0);
      double cost = 0.0;
      st = hitl_solver_evaluate(solver_, mode_, xyt.data(), &cost, gradients_.data());
      if (st != HITL_OK) return st;
      if (mode_ != HITL_MODE_ODOM_HC) {   // ceres_jacobian_ of PostHumanOptimization's Problem::Evaluate (:1247-1252)
        int64_t P = 0, Cn = 0;
        st = hitl_solver_get_correspondence_counts(solver_, &P, &Cn);
        if (st != HITL_OK) return st;
        ceres_jacobian_.num_rows = (int)(2 * P); ceres_jacobian_.num_cols = (int)(3 * n);
        ceres_jacobian_.rows.assign(2 * P + 1, 0); ceres_jacobian_.cols.assign(12 * P, 0); ceres_jacobian_.values.assign(12 * P, 0.0);
        std::vector<int32_t> rows(2 * P + 1), cols(12 * P);
        st = hitl_solver_evaluate_jacobian(solver_, xyt.data(), &cost, NULL, rows.data(), cols.data(), ceres_jacobian_.values.data());
        if (st != HITL_OK) return st;
        for (size_t k = 0; k < rows.size(); ++k) ceres_jacobian_.rows[k] = rows[k];
        for (size_t k = 0; k < cols.size(); ++k) ceres_jacobian_.cols[k] = cols[k];
      }
    }
    return HITL_OK;
  }

 private:
  static unsigned long long HashBytes(const void* data, size_t bytes, unsigned long long seed) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    unsigned long long a = seed ^ 0xcbf29ce484222325ull, b = seed + 0x9e3779b97f4a7c15ull, c = ~seed, d = seed * 0xc2b2ae3d27d4eb4full + 1;
    const unsigned long long K = 0x100000001b3ull * 0x9e3779b97f4a7c15ull | 1ull;
    size_t k = 0;
    for (; k + 32 <= bytes; k += 32) {
      unsigned long long w[4];
      std::memcpy(w, p + k, 32);
      a = (a ^ w[0]) * K; b = (b ^ w[1]) * K; c = (c ^ w[2]) * K; d = (d ^ w[3]) * K;
      a ^= a >> 31; b ^= b >> 31; c ^= c >> 31; d ^= d >> 31;
    }
    for (; k < bytes; ++k) { a = (a ^ p[k]) * K; a ^= a >> 31; }
    return ((a * 3) ^ (b * 5) ^ (c * 7) ^ (d * 11)) + (unsigned long long)bytes;
  }
  // scans [i0, i1): Vec2f that is two packed floats (Eigen::Vector2f) is hashed as raw memory, anything else through x() / y()
  static void HashRange(const JointOptGpuT* self, size_t i0, size_t i1, unsigned long long* out) {
    for (size_t i = i0; i < i1; ++i) {
      const std::vector<Vec2f>& P = self->robot_frame_point_clouds_[i];
      const std::vector<Vec2f>& N = self->robot_frame_normal_clouds_[i];
      const size_t m = P.size();
      unsigned long long h = 0x1000193ull * (unsigned long long)(m + 1);
      if (m > 0) {
        if (sizeof(Vec2f) == 2 * sizeof(float)) {
          h ^= HashBytes(&P[0], m * sizeof(Vec2f), 1);
          h = h * 0x100000001b3ull ^ HashBytes(&N[0], m * sizeof(Vec2f), 2);
        } else {
          std::vector<float> tmp(4 * m);
          for (size_t k = 0; k < m; ++k) { tmp[4 * k] = P[k].x(); tmp[4 * k + 1] = P[k].y(); tmp[4 * k + 2] = N[k].x(); tmp[4 * k + 3] = N[k].y(); }
          h ^= HashBytes(&tmp[0], tmp.size() * sizeof(float), 3);
        }
      }
      out[i] = h;
    }
  }
  int hash_threads_ = 0;
  double fingerprint_ms_ = 0.0, run_ms_ = 0.0;
  int mode_;
  int64_t stf_window_;
  hitl_solver* solver_;
  size_t map_poses_;
  unsigned long long map_print_;
  bool stf_moments_, brute_force_search_;
  std::vector<float> ceres_cost_;
  int create_status_ = HITL_OK;
  hitl_summary summary_;
};

}  // namespace hitl
#endif  // HITL_JOINT_OPT_GPU_H_
