// CPU-only driver of JointOptGpuT::Fingerprint (hitl_slam_amd/shim/JointOptGpu.h): the O(n_obs) host cost of every Run().
// argv: poses obs_per_pose.  Prints: serial_ms threaded_ms equal(0/1) edit_seen(0/1) size_edit_seen(0/1) plain_sum_ms checksum
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "JointOptGpu.h"

struct V2 { float v[2]; float& x() { return v[0]; } float& y() { return v[1]; } float x() const { return v[0]; } float y() const { return v[1]; } };
struct P2 { V2 translation; float angle; };
struct M3 { float m[9]; float& operator()(int r, int c) { return m[3 * r + c]; } float operator()(int r, int c) const { return m[3 * r + c]; } };
struct HC { unsigned type; int constrained_pose_id, anchor_pose_id; float delta_parallel, delta_perpendicular, delta_angle, relative_penalty_dir; };
typedef hitl::JointOptGpuT<P2, V2, M3, HC> JointOpt;

static double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)std::atol(argv[1]) : 2000, m = argc > 2 ? (size_t)std::atol(argv[2]) : 400;
  JointOpt jo;
  jo.robot_frame_point_clouds_.resize(n); jo.robot_frame_normal_clouds_.resize(n);
  unsigned s = 12345u;
  for (size_t i = 0; i < n; ++i) {
    jo.robot_frame_point_clouds_[i].resize(m); jo.robot_frame_normal_clouds_[i].resize(m);
    for (size_t k = 0; k < m; ++k) {
      s = s * 1664525u + 1013904223u; jo.robot_frame_point_clouds_[i][k].v[0] = (float)(s >> 8) * 1e-6f;
      s = s * 1664525u + 1013904223u; jo.robot_frame_point_clouds_[i][k].v[1] = (float)(s >> 8) * 1e-6f;
      s = s * 1664525u + 1013904223u; jo.robot_frame_normal_clouds_[i][k].v[0] = (float)(s >> 8) * 1e-7f;
      s = s * 1664525u + 1013904223u; jo.robot_frame_normal_clouds_[i][k].v[1] = (float)(s >> 8) * 1e-7f;
    }
  }
  jo.SetHashThreads(1);
  unsigned long long h1 = jo.Fingerprint();
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  h1 = jo.Fingerprint();
  const double serial = ms_since(t0);
  jo.SetHashThreads(8);
  unsigned long long h8 = jo.Fingerprint();
  t0 = std::chrono::steady_clock::now();
  h8 = jo.Fingerprint();
  const double threaded = ms_since(t0);
  // one bit of one normal in the middle of the map
  unsigned bits;
  std::memcpy(&bits, &jo.robot_frame_normal_clouds_[n / 2][m / 3].v[1], 4);
  bits ^= 1u;
  std::memcpy(&jo.robot_frame_normal_clouds_[n / 2][m / 3].v[1], &bits, 4);
  const unsigned long long h_edit = jo.Fingerprint();
  bits ^= 1u;
  std::memcpy(&jo.robot_frame_normal_clouds_[n / 2][m / 3].v[1], &bits, 4);
  const unsigned long long h_back = jo.Fingerprint();
  // a point moved from one scan to its neighbour: same totals, other sizes
  V2 pt = jo.robot_frame_point_clouds_[3].back(), nn = jo.robot_frame_normal_clouds_[3].back();
  jo.robot_frame_point_clouds_[3].pop_back(); jo.robot_frame_normal_clouds_[3].pop_back();
  jo.robot_frame_point_clouds_[4].insert(jo.robot_frame_point_clouds_[4].begin(), pt);
  jo.robot_frame_normal_clouds_[4].insert(jo.robot_frame_normal_clouds_[4].begin(), nn);
  const unsigned long long h_size = jo.Fingerprint();
  // the floor on this host: one pass of plain 64-bit adds over the same bytes
  t0 = std::chrono::steady_clock::now();
  unsigned long long sum = 0;
  for (size_t i = 0; i < n; ++i) {
    const unsigned long long* a = reinterpret_cast<const unsigned long long*>(&jo.robot_frame_point_clouds_[i][0]);
    const unsigned long long* b = reinterpret_cast<const unsigned long long*>(&jo.robot_frame_normal_clouds_[i][0]);
    for (size_t k = 0; k < jo.robot_frame_point_clouds_[i].size(); ++k) sum += a[k] + b[k];
  }
  const double floor_ms = ms_since(t0);
  // argv[3] == "run" (a GPU is present): what JointOpt::Run costs its caller at this size in mode A (the solve the shipped binary
  // runs), first call (map upload) and steady state (same map: fingerprint + marshaling + solve)
  if (argc > 3 && std::string(argv[3]) == "run") {
    jo.poses_.resize(n); jo.covariances_.resize(n);
    for (size_t i = 0; i < n; ++i) {
      jo.poses_[i].translation.v[0] = 0.3f * (float)i; jo.poses_[i].translation.v[1] = 0.01f * (float)(i % 7); jo.poses_[i].angle = 0.001f * (float)(i % 11);
      for (int q = 0; q < 9; ++q) jo.covariances_[i].m[q] = (q % 4 == 0) ? 1e-4f : 0.f;
    }
    std::vector<HC> group;
    for (int k = 0; k < 20; ++k) {
      HC h; h.type = 4; h.constrained_pose_id = (int)(n / 2 + 40 * k); h.anchor_pose_id = 10 + k; h.delta_parallel = 0.3f * (float)(n / 2 + 40 * k - 10 - k);
      h.delta_perpendicular = 0.f; h.delta_angle = 0.f; h.relative_penalty_dir = 1.5707963f;
      group.push_back(h);
    }
    jo.human_constraints_.push_back(group);
    jo.SetMode(HITL_MODE_ODOM_HC, 3);
    if (jo.RunStatus() != HITL_OK) { std::fprintf(stderr, "shim run failed: %s\n", hitl_last_error()); return 2; }
    const double first_ms = jo.LastRunMs();
    if (jo.RunStatus() != HITL_OK) return 3;
    std::printf("run_first_ms %.3f run_steady_ms %.3f fingerprint_ms %.3f solve_ms %.3f iterations %d\n", first_ms, jo.LastRunMs(), jo.LastFingerprintMs(),
                jo.Summary().total_ms, jo.Summary().iterations);
  }
  std::printf("%.3f %.3f %d %d %d %.3f %llu\n", serial, threaded, (int)(h1 == h8 && h_back == h1), (int)(h_edit != h1), (int)(h_size != h1), floor_ms, sum);
  return 0;
}
