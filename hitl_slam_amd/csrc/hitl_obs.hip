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
                                                         float* __restrict__ world_x, float* __restrict__ world_y) {
  const int i = pose_of_block(map.n_poses, chunk);
  if (i < 0) return;
  const PoseRT t = T[i];
  const int64_t o = map.obs_off[i];
  const int m = (int)(map.obs_off[i + 1] - o);
  for (int k = threadIdx.x; k < m; k += kObsBlock) {
    const float4 a = map.obs[o + k];
    const float2 w = to_world(t, a.x, a.y);
    world_x[o + k] = w.x; world_y[o + k] = w.y;      // SoA, the layout the caller's two arrays have
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

// ---- EMInput::AutomaticEndpointAdjustment (EMinput.cpp:195-250): inlier scan + ordered gather ----------------
__device__ __forceinline__ float distance_to_line_segment(float p0x, float p0y, float p1x, float p1y, float px, float py) {
  // Eigen::DistanceToLineSegment (shared/math/eigen_helper.h:66-81), fp32.  t is a length along the UNIT direction
  // and is compared with 1.0 there, so beyond 1 m along the segment the distance is taken to p1 -- kept as is.
  const float dx = p1x - p0x, dy = p1y - p0y;
  const float nrm = __fsqrt_rn(dx * dx + dy * dy);
  const float ux = dx / nrm, uy = dy / nrm;
  const float qx = px - p0x, qy = py - p0y;
  const float t = qx * ux + qy * uy;
  if (t < 0.0f) return __fsqrt_rn(qx * qx + qy * qy);
  if (t > 1.0f) { const float ex = px - p1x, ey = py - p1y; return __fsqrt_rn(ex * ex + ey * ey); }
  return fabsf(-uy * qx + ux * qy);
}

struct FitSeg { float s[4]; };

// pass 1: inliers per pose.  pass 2 (offsets != nullptr): world points of the inliers, in (pose, observation) order.
__global__ __launch_bounds__(kObsBlock) void k_fit_inliers(MapView map, int chunk, const PoseRT* __restrict__ T, FitSeg fs,
                                                           int* __restrict__ counts, const int64_t* __restrict__ offsets,
                                                           float2* __restrict__ points) {
  __shared__ int s_wave[kObsBlock / 64];
  __shared__ int s_base;
  const int i = pose_of_block(map.n_poses, chunk);
  if (i < 0) return;
  const PoseRT t = T[i];
  const int64_t o = map.obs_off[i];
  const int m = (int)(map.obs_off[i + 1] - o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int k0 = 0; k0 < m; k0 += kObsBlock) {
    const int k = k0 + threadIdx.x;
    bool in = false;
    float2 w = make_float2(0.f, 0.f);
    if (k < m) {
      const float4 a = map.obs[o + k];
      w = to_world(t, a.x, a.y);
      in = (double)distance_to_line_segment(fs.s[0], fs.s[1], fs.s[2], fs.s[3], w.x, w.y) < 0.03;
    }
    const unsigned long long bal = __ballot(in);
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int before = s_base;
    for (int v = 0; v < wave; ++v) before += s_wave[v];
    if (in && offsets) points[offsets[i] + before + __popcll(bal & ((1ull << lane) - 1ull))] = w;
    __syncthreads();
    if (threadIdx.x == 0) { int tot = 0; for (int v = 0; v < kObsBlock / 64; ++v) tot += s_wave[v]; s_base += tot; }
    __syncthreads();
  }
  if (threadIdx.x == 0 && counts) counts[i] = s_base;
}

// segDistResidualEM (EMinput.cpp:107-150).  The VALUE is computed with the functor's own expressions in its own order
// (p1 = cm + len a, p2 = cm - len a, t = (p - p1).(p2 - p1) / |p2 - p1|^2, three branches, sqrt of a sum of squares), so
// r -- and with it the branch taken and the r == 0 case -- is the number the Jet evaluation produces.  The derivative is
// the closed form of that branch, with n = (-sin, cos):
//     t < 0 : dr = -len (q.n) / r, q = p - p1      t > 1 : dr = len (f.n) / r, f = p - p2
//     else  : dr = -sign(q.n) (len + q.a)
// and NaN when r == 0, which is what Jet sqrt yields there (1 / (2 sqrt(0)) * 0) and what makes Ceres give up.
// partial[g] = (sum r^2, sum dr r, sum dr^2) of a fixed contiguous slice, summed in a fixed order.
constexpr int kFitBlock = 256;
struct FitLine { double cmx, cmy, len, c, s; };

__global__ __launch_bounds__(kFitBlock) void k_fit_eval(const float2* __restrict__ pts, int64_t m, int64_t per_block, FitLine fl,
                                                        double* __restrict__ partial) {
  __shared__ double s_red[3][kFitBlock / 64];
  const int64_t b0 = (int64_t)blockIdx.x * per_block, b1 = (b0 + per_block < m) ? b0 + per_block : m;
  double a0 = 0, a1 = 0, a2 = 0;
  for (int64_t q = b0 + threadIdx.x; q < b1; q += kFitBlock) {
    const float2 p = pts[q];
    const double px = (double)p.x, py = (double)p.y;
    const double p1x = fl.cmx + fl.len * fl.c, p1y = fl.cmy + fl.len * fl.s;
    const double p2x = fl.cmx - fl.len * fl.c, p2y = fl.cmy - fl.len * fl.s;
    const double ex = p2x - p1x, ey = p2y - p1y;
    const double qx = px - p1x, qy = py - p1y;
    const double t = (qx * ex + qy * ey) * (1.0 / (ex * ex + ey * ey));   // Jet division: f * (1 / g)
    double r, dr;
    if (t < 0.0) {
      r = sqrt(qx * qx + qy * qy);
      dr = -fl.len * (-qx * fl.s + qy * fl.c) / r;
    } else if (t > 1.0) {
      const double fx = px - p2x, fy = py - p2y;
      r = sqrt(fx * fx + fy * fy);
      dr = fl.len * (-fx * fl.s + fy * fl.c) / r;
    } else {
      const double gx = px - (p1x + t * ex), gy = py - (p1y + t * ey);
      r = sqrt(gx * gx + gy * gy);
      const double qn = -qx * fl.s + qy * fl.c;
      dr = (qn < 0.0) ? (fl.len + (qx * fl.c + qy * fl.s)) : -(fl.len + (qx * fl.c + qy * fl.s));
    }
    if (r == 0.0) dr = __longlong_as_double(0x7ff8000000000000ll);
    a0 += r * r; a1 += dr * r; a2 += dr * dr;
  }
  for (int sft = 32; sft >= 1; sft >>= 1) { a0 += __shfl_xor(a0, sft, 64); a1 += __shfl_xor(a1, sft, 64); a2 += __shfl_xor(a2, sft, 64); }
  if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = a0; s_red[1][threadIdx.x >> 6] = a1; s_red[2][threadIdx.x >> 6] = a2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0;
    for (int v = 0; v < kFitBlock / 64; ++v) t += s_red[threadIdx.x][v];
    partial[3 * blockIdx.x + threadIdx.x] = t;
  }
}

void launch_fit_inliers(hipStream_t st, const MapView& map, const float* T4, const float* seg4, int* counts, const int64_t* offsets,
                        float2* points) {
  FitSeg fs;
  for (int q = 0; q < 4; ++q) fs.s[q] = seg4[q];
  const int chunk = (map.n_poses + 7) / 8;
  hipLaunchKernelGGL(k_fit_inliers, dim3(8 * chunk), dim3(kObsBlock), 0, st, map, chunk, (const PoseRT*)T4, fs, counts, offsets, points);
}

int fit_eval_blocks(int64_t m) {
  const int64_t want = (m + 4 * kFitBlock - 1) / (4 * kFitBlock);
  return (int)(want < 1 ? 1 : (want > 256 ? 256 : want));
}

void launch_fit_eval(hipStream_t st, const float2* pts, int64_t m, double cmx, double cmy, double len, double c, double s,
                     double* partial /* 3 * fit_eval_blocks(m) */) {
  const int g = fit_eval_blocks(m);
  const int64_t per = (m + g - 1) / g;
  FitLine fl{cmx, cmy, len, c, s};
  hipLaunchKernelGGL(k_fit_eval, dim3(g), dim3(kFitBlock), 0, st, pts, m, per, fl, partial);
}

void launch_transform(hipStream_t st, const MapView& map, const float* T4, float* world_x, float* world_y) {
  const int chunk = (map.n_poses + 7) / 8;
  hipLaunchKernelGGL(k_transform, dim3(8 * chunk), dim3(kObsBlock), 0, st, map, chunk, (const PoseRT*)T4, world_x, world_y);
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
