// hitl_obs.hip -- the O(N*M) observation scans either side of the solve,
// fused with the robot->world transform so the world-frame clouds never have to exist:
//   k_transform   == HitLSLAM::transformPointCloudsToWorldFrame     (HitLSLAM.cpp:245-254)
//   k_verify      == the scan inside HitLSLAM::verifyUserInput       (HitLSLAM.cpp:218-243)
//   k_obs_sets    == EMInput::EstablishObservationSets + distToLineSeg (EMinput.cpp:268-323)
// All arithmetic is fp32 like the reference (Vector2f / Affine2f); compiled with
// -ffp-contract=off and correctly rounded sqrt/divide, and the per-pose cos/sin come from the host,
// so flags, counts and masks are bit-identical to the CPU restatement.
// One workgroup per pose (consecutive poses on one XCD); streaming 16 B per observation.

#include "hitl_internal.h"

namespace hitl {

constexpr int kObsBlock = 256;

struct PoseRT { float c, s, tx, ty; };   // Rotation2Df(angle), Translation2f

__device__ __forceinline__ float2 to_world(const PoseRT& T, float px, float py) {
  // Affine2f (translation * rotation) * p  ==  R p + t, row-by-row like Eigen's 2x2 product
  return make_float2((T.c * px - T.s * py) + T.tx, (T.s * px + T.c * py) + T.ty);
}

__device__ __forceinline__ int pose_of_block(int n_poses, int chunk) {
  const int b = blockIdx.x, idx = b >> 3;
  const int i = (b & 7) * chunk + idx;
  return (idx < chunk && i < n_poses) ? i : -1;
}

__global__ __launch_bounds__(kObsBlock) void k_transform(MapView map, int chunk, const PoseRT* __restrict__ T,
                                                         float2* __restrict__ world) {
  const int i = pose_of_block(map.n_poses, chunk);
  if (i < 0) return;
  const PoseRT t = T[i];
  const int64_t o = map.obs_off[i];
  const int m = (int)(map.obs_off[i + 1] - o);
  for (int k = threadIdx.x; k < m; k += kObsBlock) {
    const float4 a = map.obs[o + k];
    world[o + k] = to_world(t, a.x, a.y);
  }
}

struct VerifyArgs { int n; float x[16], y[16]; float thr; };

__global__ __launch_bounds__(kObsBlock) void k_verify(MapView map, int chunk, const PoseRT* __restrict__ T, VerifyArgs va,
                                                      int* __restrict__ seen) {
  const int i = pose_of_block(map.n_poses, chunk);
  if (i < 0) return;
  const PoseRT t = T[i];
  const int64_t o = map.obs_off[i];
  const int m = (int)(map.obs_off[i + 1] - o);
  unsigned hit = 0;
  for (int k = threadIdx.x; k < m; k += kObsBlock) {
    const float4 a = map.obs[o + k];
    const float2 w = to_world(t, a.x, a.y);
    for (int q = 0; q < va.n; ++q) {
      const float dx = w.x - va.x[q], dy = w.y - va.y[q];
      if (__fsqrt_rn(dx * dx + dy * dy) < va.thr) hit |= 1u << q;     // (cloud - selected).norm() < thresh
    }
  }
  for (int q = 0; q < va.n; ++q)
    if (__any((hit >> q) & 1u)) { if ((threadIdx.x & 63) == 0) atomicOr(&seen[q], 1); }
}

__device__ __forceinline__ float dist_to_line_seg(float p1x, float p1y, float p2x, float p2y, float px, float py) {
  // EMInput::distToLineSeg (EMinput.cpp:268-279), fp32
  const float ex = p2x - p1x, ey = p2y - p1y, dx = px - p1x, dy = py - p1y;
  const float t = (dx * ex + dy * ey) / (ex * ex + ey * ey);
  if (t < 0.0f) return __fsqrt_rn(dx * dx + dy * dy);
  if (t > 1.0f) { const float fx = px - p2x, fy = py - p2y; return __fsqrt_rn(fx * fx + fy * fy); }
  const float qx = px - (p1x + t * ex), qy = py - (p1y + t * ey);
  return __fsqrt_rn(qx * qx + qy * qy);
}

struct SegArgs { float s[8]; double thr; };   // selected_points_[0..3]

__global__ __launch_bounds__(kObsBlock) void k_obs_sets(MapView map, int chunk, const PoseRT* __restrict__ T, SegArgs sa,
                                                        int* __restrict__ counts, unsigned char* __restrict__ mask) {
  __shared__ int s_cnt[2];
  const int i = pose_of_block(map.n_poses, chunk);
  if (i < 0) return;
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const PoseRT t = T[i];
  const int64_t o = map.obs_off[i];
  const int m = (int)(map.obs_off[i + 1] - o);
  int c0 = 0, c1 = 0;
  for (int k = threadIdx.x; k < m; k += kObsBlock) {
    const float4 a = map.obs[o + k];
    const float2 w = to_world(t, a.x, a.y);
    const bool in0 = (double)dist_to_line_seg(sa.s[0], sa.s[1], sa.s[2], sa.s[3], w.x, w.y) < sa.thr;
    const bool in1 = (double)dist_to_line_seg(sa.s[4], sa.s[5], sa.s[6], sa.s[7], w.x, w.y) < sa.thr;
    c0 += in0; c1 += in1;
    if (mask) mask[o + k] = (unsigned char)((in0 ? 1 : 0) | (in1 ? 2 : 0));
  }
  for (int sft = 32; sft >= 1; sft >>= 1) { c0 += __shfl_xor(c0, sft, 64); c1 += __shfl_xor(c1, sft, 64); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cnt[0], c0); atomicAdd(&s_cnt[1], c1); }
  __syncthreads();
  if (threadIdx.x < 2) counts[2 * i + threadIdx.x] = s_cnt[threadIdx.x];
}

void launch_transform(hipStream_t st, const MapView& map, const float* T4, float2* world) {
  const int chunk = (map.n_poses + 7) / 8;
  hipLaunchKernelGGL(k_transform, dim3(8 * chunk), dim3(kObsBlock), 0, st, map, chunk, (const PoseRT*)T4, world);
}

void launch_verify(hipStream_t st, const MapView& map, const float* T4, int n, const float* pts_xy, float thr, int* seen) {
  VerifyArgs va;
  va.n = n; va.thr = thr;
  for (int q = 0; q < 16; ++q) { va.x[q] = q < n ? pts_xy[2 * q] : 0.f; va.y[q] = q < n ? pts_xy[2 * q + 1] : 0.f; }
  const int chunk = (map.n_poses + 7) / 8;
  hipLaunchKernelGGL(k_verify, dim3(8 * chunk), dim3(kObsBlock), 0, st, map, chunk, (const PoseRT*)T4, va, seen);
}

void launch_obs_sets(hipStream_t st, const MapView& map, const float* T4, const float* seg8, double thr, int* counts,
                     unsigned char* mask) {
  SegArgs sa;
  for (int q = 0; q < 8; ++q) sa.s[q] = seg8[q];
  sa.thr = thr;
  const int chunk = (map.n_poses + 7) / 8;
  hipLaunchKernelGGL(k_obs_sets, dim3(8 * chunk), dim3(kObsBlock), 0, st, map, chunk, (const PoseRT*)T4, sa, counts, mask);
}

}  // namespace hitl
