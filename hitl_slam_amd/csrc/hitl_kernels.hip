// hitl_kernels.hip -- CDNA4 (gfx950) kernels for hitl-slam's batch re-optimisation path.
//
// Kernel inventory (SURVEY.md section 2.3 "compute sites" -> here):
//   k_linearize_stf  : K4  PointToPointGlobConstraint residual + analytic 2x6 Jacobian per
//                      directed pose pair (residual_functors.h:786-830).  The HBM-bound kernel.
//   k_assemble       : K5/K6/K7a odometry (residual_functors.h:1081-1125), human rows
//                      (:1299-1415), J^T J 3x3 blocks + gradient, gathered per pose (no atomics,
//                      bit-reproducible).
//   k_build_system   : Jacobi scaling + LM diagonal -> block-tridiagonal super-block system.
//   k_cr_*           : K7b block cyclic reduction = recursive red/black Schur complement of the
//                      banded normal equations (what Ceres SPARSE_SCHUR does on a chain,
//                      JointOptimization.cpp:148), exact preconditioner of the PCG.
//   k_spmv, k_dots...: PCG pieces.
// All arithmetic is fp64 (T = double in the reference functors); observations stay fp32.
// Wavefront = 64 everywhere.

#include "hitl_internal.h"

namespace hitl {

#define WAVE 64

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
  return v;
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// ===========================================================================================
// linearize_stf: one workgroup per SOURCE pose (its scan staged once in LDS as float4), one
// wave per directed pair (i -> j), lanes stride the pair's matches.
//
// All per-match arithmetic is done in the SOURCE pose's frame: with R01 = R0^T R1 and
// t01 = R0^T (t1 - t0),   u = R01 p1,  w = u + t01,  q = w - p0  (= R0^T delta),  n1' = R01 n1,
//   e0/k = n0 . q                      e1/k = n1' . q
//   d(e0/k)/dtheta0 = cross(n0, w)     d(e0/k)/dtheta1 = -cross(n0, u)
//   d(e1/k)/dtheta0 = cross(n1', p0)   d(e1/k)/dtheta1 = cross(n1', t01 - p0)
// (identical to SURVEY.md A.2 after rotating by R0^T; cross(a,b) = ax*by - ay*bx), and the
// translation partials -+k R0 n0, -+k R0 n1' are rotated once per pair after the reduction.
// 38 fp64 operations per match instead of ~110 for the world-frame form.
//
// The 10 sums are reduced with a reduce-scatter butterfly (17 64-bit shuffles per pair
// instead of 60): at each step a lane keeps half of its values and ships the other half.
//
// blockIdx -> pose mapping keeps consecutive poses on one XCD (blocks b and b+8 share an
// XCD) so the target scans i+-1..W, which the neighbouring workgroups stage as THEIR source,
// are served from that XCD's L2.
// ===========================================================================================
constexpr int kStfBlock = 192;

// v[0..2h) -> v[0..h): lanes with bit `mask` set keep the upper half
#define HITL_XSTEP(mask, h)                                        \
  {                                                                \
    const bool up = (lane & (mask)) != 0;                          \
    _Pragma("unroll") for (int q_ = 0; q_ < (h); ++q_) {           \
      const double send_ = up ? v[q_] : v[q_ + (h)];               \
      const double keep_ = up ? v[q_ + (h)] : v[q_];               \
      v[q_] = keep_ + __shfl_xor(send_, (mask), WAVE);             \
    }                                                              \
  }

constexpr int kStfSub = 32;     // lanes per directed pair: two pairs per wave

// per-pose (cos theta, sin theta), computed once per linearisation instead of per pair
__global__ void k_pose_trig(int n, const double* __restrict__ x, double2* __restrict__ cs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  sincos(x[3 * i + 2], &s, &c);
  cs[i] = make_double2(c, s);
}

// Tuning notes (MI355X, C4 workload, rocprofv3 PMC; see DESIGN.md "k_linearize_stf"):
//  * 2 matches per lane in flight at <= 80 VGPRs (6 waves/SIMD) beats deeper unrolling at lower
//    occupancy; staging the target scans in LDS too (50 KB/workgroup) or software-pipelining the
//    gathers by hand were both slower (LDS-limited occupancy / register spills).
//  * requesting the NEXT trip's index words before this trip's arithmetic (2 VGPRs) took 376 -> 344 us; going one
//    step further (next trip's gathers in flight too, 8 more VGPRs) spills at the 80-VGPR budget: 539 us.
//  * VALU and TA are each ~55 % busy: the kernel is balanced between fp64 issue and L1/L2
//    gather handling, at 2x the HBM floor of its compulsory bytes.
constexpr int kStfUnroll = 2;

template <bool USE_LDS>
__global__ __launch_bounds__(kStfBlock, 6) void k_linearize_stf(MapView map, PairView pr, int chunk, int lds_cap,
                                                                 const double* __restrict__ x,
                                                                 const double2* __restrict__ cs, double corr, double sd,
                                                                 double* __restrict__ pairRJ) {
  extern __shared__ float4 s_scan[];  // [lds_cap] : (px, py, nx, ny) of the source scan
  const int b = blockIdx.x;
  const int idx = b >> 3;
  const int i = (b & 7) * chunk + idx;
  if (idx >= chunk || i >= map.n_poses) return;
  const int p_begin = pr.pose_pair_off[i], p_end = pr.pose_pair_off[i + 1];
  if (p_begin == p_end) return;
  const int64_t oi = map.obs_off[i];
  const int mi = (int)(map.obs_off[i + 1] - oi);
  if (USE_LDS) {
    for (int k = threadIdx.x; k < mi; k += kStfBlock) s_scan[k] = map.obs[oi + k];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int sub = threadIdx.x / kStfSub;        // pair slot of this half-wave within the workgroup
  const int hl = threadIdx.x & (kStfSub - 1);   // lane within the half-wave
  const double xi = x[3 * i], yi = x[3 * i + 1];
  const double2 cs0 = cs[i];
  const double c0 = cs0.x, s0 = cs0.y;
  const double kf = corr / sd;
  const float4* __restrict__ src_g = map.obs + oi;
  const uint32_t* __restrict__ mt = pr.matches;
  constexpr int kSlots = kStfBlock / kStfSub;
  for (int pb = p_begin; pb < p_end; pb += kSlots) {   // workgroup-uniform trip count
    const int p = pb + sub;
    const bool live = p < p_end;
    const int pc = live ? p : p_begin;
    const int j = pr.pair_tgt[pc];
    const float4* __restrict__ tgt_g = map.obs + map.obs_off[j];
    const double ddx = x[3 * j] - xi, ddy = x[3 * j + 1] - yi;
    const double2 cs1 = cs[j];
    const double cR = c0 * cs1.x + s0 * cs1.y, sR = c0 * cs1.y - s0 * cs1.x;   // R01 = R0^T R1
    const double t01x = c0 * ddx + s0 * ddy, t01y = c0 * ddy - s0 * ddx;       // R0^T (t1 - t0)
    const int64_t m_begin = pr.match_off[pc];
    const int64_t m_end = live ? pr.match_off[pc + 1] : m_begin;
    // 8 running sums; the two remaining theta partials follow from them after the reduction:
    //   sum e0 cross(n0, w)        = -T01 + cross(A0, t01)     (w = u + t01)
    //   sum e1 cross(n1', t01-p0)  = cross(A1, t01) - T10
    double S0 = 0, A0x = 0, A0y = 0, T01 = 0;
    double S1 = 0, A1x = 0, A1y = 0, T10 = 0;
    // the index words of the NEXT trip are requested before this trip's arithmetic, so a trip waits for one memory
    // round (the gathers) instead of two dependent ones (index, then gather)
    uint32_t un[kStfUnroll];
    {
      const int64_t m0 = m_begin + hl;
#pragma unroll
      for (int q = 0; q < kStfUnroll; ++q) {
        const int64_t mq = m0 + (int64_t)q * kStfSub;
        un[q] = (mq < m_end) ? mt[mq] : 0u;
      }
    }
    for (int64_t m = m_begin + hl; m < m_end; m += kStfSub * kStfUnroll) {
      uint32_t uu[kStfUnroll];
      float4 av[kStfUnroll], tv[kStfUnroll];
#pragma unroll
      for (int q = 0; q < kStfUnroll; ++q) uu[q] = un[q];   // tail lanes carry index 0 (valid), weighted 0 below
      {
        const int64_t mn = m + kStfSub * kStfUnroll;
#pragma unroll
        for (int q = 0; q < kStfUnroll; ++q) {
          const int64_t mq = mn + (int64_t)q * kStfSub;
          un[q] = (mq < m_end) ? mt[mq] : 0u;
        }
      }
#pragma unroll
      for (int q = 0; q < kStfUnroll; ++q) {
        const int k0 = (int)(uu[q] & 0xffffu), k1 = (int)(uu[q] >> 16);
        av[q] = USE_LDS ? s_scan[k0] : src_g[k0];
        tv[q] = tgt_g[k1];
      }
#pragma unroll
      for (int q = 0; q < kStfUnroll; ++q) {
        const double wgt = (m + (int64_t)q * kStfSub < m_end) ? 1.0 : 0.0;
        const double ax = av[q].x, ay = av[q].y, anx = av[q].z, any = av[q].w;
        const double bx = tv[q].x, by = tv[q].y, bnx = tv[q].z, bny = tv[q].w;
        const double ux = cR * bx - sR * by, uy = sR * bx + cR * by;        // u = R01 p1
        const double mx = cR * bnx - sR * bny, my = sR * bnx + cR * bny;    // n1' = R01 n1
        const double qx = (ux + t01x) - ax, qy = (uy + t01y) - ay;          // q = R0^T delta
        const double e0 = wgt * (anx * qx + any * qy);                      // e0 / k
        const double e1 = wgt * (mx * qx + my * qy);                        // e1 / k
        const double d01 = any * ux - anx * uy;                             // -cross(n0, u)
        const double d10 = mx * ay - my * ax;                               //  cross(n1', p0)
        S0 += e0 * e0; A0x += e0 * anx; A0y += e0 * any; T01 += e0 * d01;
        S1 += e1 * e1; A1x += e1 * mx; A1y += e1 * my; T10 += e1 * d10;
      }
    }
    // reduce-scatter over the 32 lanes of this pair: 8 64-bit shuffles for both pairs of the wave
    double v[8] = {S0, A0x, A0y, T01, S1, A1x, A1y, T10};
    HITL_XSTEP(16, 4)
    HITL_XSTEP(8, 2)
    HITL_XSTEP(4, 1)
    double tot = v[0];
    tot += __shfl_xor(tot, 2, WAVE);
    tot += __shfl_xor(tot, 1, WAVE);
    // lane base + 4*id now holds the pair total of value id (base = 0 or 32)
    const bool hi = (lane & 32) != 0;
#define HITL_PICK(id) (hi ? readlane_f64(tot, 32 + 4 * (id)) : readlane_f64(tot, 4 * (id)))
    S0 = HITL_PICK(0); A0x = HITL_PICK(1); A0y = HITL_PICK(2); T01 = HITL_PICK(3);
    S1 = HITL_PICK(4); A1x = HITL_PICK(5); A1y = HITL_PICK(6); T10 = HITL_PICK(7);
#undef HITL_PICK
    const double T00 = (A0x * t01y - A0y * t01x) - T01;
    const double T11 = (A1x * t01y - A1y * t01x) - T10;
    if (hl == 0 && live) {
      const double dn = (double)(m_end - m_begin);
      const double inv_n = 1.0 / dn;
      double* o = pairRJ + (size_t)kPairRJ * p;
      // sums are of e/k: r = k sqrt(S/n) (residual_functors.h:822-827); dr/dq = k^2 (sum e' de')/(n r)
      //                                                                       = k (sum e' de') / (n sqrt(S/n))
      const double a0 = S0 * inv_n, a1 = S1 * inv_n;
      const double i0 = (S0 != 0.0) ? rsqrt(a0) : 0.0, i1 = (S1 != 0.0) ? rsqrt(a1) : 0.0;
      const double r0 = kf * a0 * i0, r1 = kf * a1 * i1;
      const double f0 = kf * inv_n * i0, f1 = kf * inv_n * i1;
      const double g0x = c0 * A0x - s0 * A0y, g0y = s0 * A0x + c0 * A0y;   // R0 * sum e0' n0
      const double g1x = c0 * A1x - s0 * A1y, g1y = s0 * A1x + c0 * A1y;   // R0 * sum e1' n1'
      o[0] = r0; o[1] = r1;
      o[2] = -f0 * g0x; o[3] = -f0 * g0y; o[4] = f0 * T00; o[5] = f0 * g0x; o[6] = f0 * g0y; o[7] = f0 * T01;
      o[8] = -f1 * g1x; o[9] = -f1 * g1y; o[10] = f1 * T10; o[11] = f1 * g1x; o[12] = f1 * g1y; o[13] = f1 * T11;
    }
  }
}

// ===========================================================================================
// Opt-in (hitl_options.stf_moments): the scan-to-scan factor through its sufficient statistics.
// In the source frame  e0_i / k = m_i . z  with  m_i = (n0.p1, n0 x p1, n0x, n0y, -n0.p0)  and
// z = (cos D, sin D, tau_x, tau_y, 1),  D = theta1 - theta0,  tau = R0^T (t1 - t0);  in the target frame
// e1_i / k = m'_i . z'  with  m'_i = (-n1.p0, -(n1 x p0), n1x, n1y, n1.p1),  z' = (cos D, sin D, tau'_x, tau'_y, 1),
// tau' = R1^T (t1 - t0).  So  sum e0^2 = k^2 z^T M0 z  with  M0 = sum m m^T  (5x5, 15 numbers) -- the point data enter
// only through M0, M1, which do not depend on the poses.  k_stf_moments forms them once per correspondence set;
// k_linearize_stf_moments then evaluates r and J of every pair from 30 doubles, whatever the number of matches.
// Same algebra as k_linearize_stf, different summation order: results agree to ~1e-10 relative.
// ===========================================================================================
constexpr int kMomStride = 32;     // doubles per pair: M0 (15, upper triangle by rows), M1 (15), 2 spare

__global__ __launch_bounds__(256) void k_stf_moments(MapView map, PairView pr, double* __restrict__ pairM) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= pr.n_pairs) return;
  const float4* __restrict__ src = map.obs + map.obs_off[pr.pair_src[p]];
  const float4* __restrict__ tgt = map.obs + map.obs_off[pr.pair_tgt[p]];
  const int64_t m_begin = pr.match_off[p], m_end = pr.match_off[p + 1];
  double acc[30];
#pragma unroll
  for (int q = 0; q < 30; ++q) acc[q] = 0.0;
  for (int64_t m = m_begin + lane; m < m_end; m += 64) {
    const uint32_t u = pr.matches[m];
    const float4 a = src[u & 0xffffu], b = tgt[u >> 16];
    const double ax = a.x, ay = a.y, anx = a.z, any = a.w, bx = b.x, by = b.y, bnx = b.z, bny = b.w;
    const double v0[5] = {anx * bx + any * by, any * bx - anx * by, anx, any, -(anx * ax + any * ay)};
    const double v1[5] = {-(bnx * ax + bny * ay), -(bnx * ay - bny * ax), bnx, bny, bnx * bx + bny * by};
    int q = 0;
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
      for (int c = r; c < 5; ++c) { acc[q] += v0[r] * v0[c]; acc[15 + q] += v1[r] * v1[c]; ++q; }
  }
#pragma unroll
  for (int q = 0; q < 30; ++q) {
    double v = acc[q];
    for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft, WAVE);
    if (lane == 0) pairM[(size_t)p * kMomStride + q] = v;
  }
}

__device__ __forceinline__ void sym5_mul(const double* __restrict__ M, const double (&z)[5], double (&g)[5]) {
  // M = upper triangle by rows: 00 01 02 03 04 | 11 12 13 14 | 22 23 24 | 33 34 | 44
  g[0] = M[0] * z[0] + M[1] * z[1] + M[2] * z[2] + M[3] * z[3] + M[4] * z[4];
  g[1] = M[1] * z[0] + M[5] * z[1] + M[6] * z[2] + M[7] * z[3] + M[8] * z[4];
  g[2] = M[2] * z[0] + M[6] * z[1] + M[9] * z[2] + M[10] * z[3] + M[11] * z[4];
  g[3] = M[3] * z[0] + M[7] * z[1] + M[10] * z[2] + M[12] * z[3] + M[13] * z[4];
  g[4] = M[4] * z[0] + M[8] * z[1] + M[11] * z[2] + M[13] * z[3] + M[14] * z[4];
}

__global__ __launch_bounds__(128) void k_linearize_stf_moments(PairView pr, const double* __restrict__ pairM, const double* __restrict__ x,
                                                               const double2* __restrict__ cs, double corr, double sd,
                                                               double* __restrict__ pairRJ) {
  const int p = blockIdx.x * 128 + threadIdx.x;
  if (p >= pr.n_pairs) return;
  const int i = pr.pair_src[p], j = pr.pair_tgt[p];
  const double2 cs0 = cs[i], cs1 = cs[j];
  const double c0 = cs0.x, s0 = cs0.y, c1 = cs1.x, s1 = cs1.y;
  const double c = c0 * c1 + s0 * s1, s = c0 * s1 - s0 * c1;             // cos, sin of theta1 - theta0
  const double dx = x[3 * j] - x[3 * i], dy = x[3 * j + 1] - x[3 * i + 1];
  const double tx = c0 * dx + s0 * dy, ty = c0 * dy - s0 * dx;            // R0^T (t1 - t0)
  const double ux = c1 * dx + s1 * dy, uy = c1 * dy - s1 * dx;            // R1^T (t1 - t0)
  const double z0[5] = {c, s, tx, ty, 1.0}, z1[5] = {c, s, ux, uy, 1.0};
  double M[30];
  const double* __restrict__ src = pairM + (size_t)p * kMomStride;
#pragma unroll
  for (int q = 0; q < 30; ++q) M[q] = src[q];
  double g0[5], g1[5];
  sym5_mul(M, z0, g0);
  sym5_mul(M + 15, z1, g1);
  double S0 = 0.0, S1 = 0.0;
#pragma unroll
  for (int q = 0; q < 5; ++q) { S0 += z0[q] * g0[q]; S1 += z1[q] * g1[q]; }
  S0 = fmax(S0, 0.0); S1 = fmax(S1, 0.0);                                 // a sum of squares; cancellation can leave -eps
  const double dn = (double)(pr.match_off[p + 1] - pr.match_off[p]), inv_n = 1.0 / dn, kf = corr / sd;
  const double a0 = S0 * inv_n, a1 = S1 * inv_n;
  const double i0 = (S0 != 0.0) ? rsqrt(a0) : 0.0, i1 = (S1 != 0.0) ? rsqrt(a1) : 0.0;
  const double f0 = kf * inv_n * i0, f1 = kf * inv_n * i1;
  double* o = pairRJ + (size_t)kPairRJ * p;
  o[0] = kf * a0 * i0; o[1] = kf * a1 * i1;
  const double gx0 = c0 * g0[2] - s0 * g0[3], gy0 = s0 * g0[2] + c0 * g0[3];   // world-frame sum e0 n0
  o[2] = -f0 * gx0; o[3] = -f0 * gy0; o[4] = f0 * (s * g0[0] - c * g0[1] + ty * g0[2] - tx * g0[3]);
  o[5] = f0 * gx0; o[6] = f0 * gy0; o[7] = f0 * (-s * g0[0] + c * g0[1]);
  const double gx1 = c1 * g1[2] - s1 * g1[3], gy1 = s1 * g1[2] + c1 * g1[3];
  o[8] = -f1 * gx1; o[9] = -f1 * gy1; o[10] = f1 * (s * g1[0] - c * g1[1]);
  o[11] = f1 * gx1; o[12] = f1 * gy1; o[13] = f1 * (-s * g1[0] + c * g1[1] + uy * g1[2] - ux * g1[3]);
}

void launch_stf_moments(hipStream_t st, const MapView& map, const PairView& pr, double* pairM) {
  if (pr.n_pairs == 0) return;
  hipLaunchKernelGGL(k_stf_moments, dim3((pr.n_pairs + 3) / 4), dim3(256), 0, st, map, pr, pairM);
}

void launch_linearize_stf_moments(hipStream_t st, const PairView& pr, const double* pairM, const double* x, const double2* cs, double corr,
                                  double sd, double* pairRJ) {
  if (pr.n_pairs == 0) return;
  hipLaunchKernelGGL(k_linearize_stf_moments, dim3((pr.n_pairs + 127) / 128), dim3(128), 0, st, pr, pairM, x, cs, corr, sd, pairRJ);
}

void launch_pose_trig(hipStream_t st, int n_poses, const double* x, double2* cs) {
  hipLaunchKernelGGL(k_pose_trig, dim3((n_poses + 255) / 256), dim3(256), 0, st, n_poses, x, cs);
}

void launch_linearize_stf(hipStream_t st, const MapView& map, const PairView& pr, int max_scan, const double* x,
                          const double2* cs, float corr, float std_dev, double* pairRJ) {
  if (pr.n_pairs == 0) return;
  const int chunk = (map.n_poses + 7) / 8;
  int lds_cap = (max_scan + 63) & ~63;
  if ((size_t)lds_cap * 16 > 64 * 1024) lds_cap = 0;  // oversized scans read the source from global memory
  if (lds_cap > 0)
    hipLaunchKernelGGL(k_linearize_stf<true>, dim3(8 * chunk), dim3(kStfBlock), (size_t)lds_cap * 16, st, map, pr, chunk,
                       lds_cap, x, cs, (double)corr, (double)std_dev, pairRJ);
  else
    hipLaunchKernelGGL(k_linearize_stf<false>, dim3(8 * chunk), dim3(kStfBlock), 0, st, map, pr, chunk, 0, x, cs,
                       (double)corr, (double)std_dev, pairRJ);
}

// SoA (the caller's layout, JointOptimization.h:62-68) -> device-resident float4 per observation
__global__ void k_pack_obs(int64_t n, const float* __restrict__ px, const float* __restrict__ py,
                           const float* __restrict__ nx, const float* __restrict__ ny, float4* __restrict__ obs) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) obs[t] = make_float4(px[t], py[t], nx[t], ny[t]);
}

void launch_pack_obs(hipStream_t st, int64_t n, const float* px, const float* py, const float* nx, const float* ny,
                     float4* obs) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_pack_obs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, px, py, nx, ny, obs);
}

// ===========================================================================================
// assemble: one thread per pose gathers every factor touching it.
// ===========================================================================================
struct OdomEval { double r[3]; double J[18]; };

// residual_functors.h:1081-1125 with the analytic Jacobian of SURVEY.md A.1
__device__ __forceinline__ void eval_odom(const double* x1, const double* x2, float4 A, float2 rr, float sr_, float st_,
                                          float sa_, OdomEval& o) {
  const double dx = x2[0] - x1[0], dy = x2[1] - x1[1];
  double s, c;
  sincos(-x1[2], &s, &c);  // Rotation2D(-theta1)
  const double ux = c * dx - s * dy, uy = s * dx + c * dy;
  const double a00 = A.x, a01 = A.y, a10 = A.z, a11 = A.w;
  const double sr = sr_, st = st_, sa = sa_;
  o.r[0] = ((a00 * ux + a01 * uy) - (double)rr.x) / sr;
  o.r[1] = (a10 * ux + a11 * uy) / st;
  const double phi = x2[2] - x1[2] - (double)rr.y;
  o.r[2] = atan2(sin(phi), cos(phi)) / sa;
  const double dux[6] = {-c, s, uy, c, -s, 0.0};
  const double duy[6] = {-s, -c, -ux, s, c, 0.0};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    o.J[k] = (a00 * dux[k] + a01 * duy[k]) / sr;
    o.J[6 + k] = (a10 * dux[k] + a11 * duy[k]) / st;
    o.J[12 + k] = 0.0;
  }
  o.J[12 + 2] = -1.0 / sa;
  o.J[12 + 5] = 1.0 / sa;
}

__device__ __forceinline__ double quad_sum(double v) {     // fixed order: (l0 + l1) + (l2 + l3) in every lane of the quad
  v += __shfl_xor(v, 1, WAVE);
  v += __shfl_xor(v, 2, WAVE);
  return v;
}

// Four lanes per pose (there are too few poses to fill the machine with one thread each, and a single thread's chain of
// dependent loads is what the kernel waits for): lane r of the quad takes the incidence entries e = r (mod 4), lane 0
// the odometry edge (i-1, i), lane 1 the edge (i, i+1), lane 2 the human rows, lane d-1 the band block (i, i+d); the
// diagonal block, gradient and cost are then summed across the quad in a fixed order.  No atomics: bit-reproducible.
constexpr int kAsmBlock = 256;     // 64 poses per workgroup
__global__ __launch_bounds__(kAsmBlock) void k_assemble(MapView map, PairView pr, FactorView fv, const double* __restrict__ x,
                                                         LinView lin) {
  const int i = (blockIdx.x * kAsmBlock + (int)threadIdx.x) >> 2;
  const int r = threadIdx.x & 3;
  const int n = map.n_poses;
  const bool act = i < n;
  const int Wb = fv.band_poses;
  double Hd[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double gi[3] = {0, 0, 0};
  double H1[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // odometry part of block (i, i+1), produced by lane 1
  double cost = 0.0;
  if (act && fv.with_stf) {
    for (int e = fv.inc_off[i] + r; e < fv.inc_off[i + 1]; e += 4) {
      const int ent = fv.inc_ent[e];
      const double* RJ = lin.pairRJ + (size_t)kPairRJ * (ent >> 1);
      const int co = 2 + 3 * (ent & 1);
      const double r0 = RJ[0], r1 = RJ[1];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const double ja0 = RJ[co + a], ja1 = RJ[6 + co + a];
        gi[a] += ja0 * r0 + ja1 * r1;
#pragma unroll
        for (int c = 0; c < 3; ++c) Hd[a * 3 + c] += ja0 * RJ[co + c] + ja1 * RJ[6 + co + c];
      }
      if ((ent & 1) == 0) cost += 0.5 * (r0 * r0 + r1 * r1);      // a pair's cost is counted at its source pose
    }
  }
  if (act && fv.with_odom) {
    OdomEval o;
    if (r == 0 && i >= 1) {  // edge (i-1, i): this pose is pose2 (columns 3..5)
      eval_odom(x + 3 * (i - 1), x + 3 * i, fv.odom_A[i], fv.odom_rr[i], fv.sig_r, fv.sig_t, fv.sig_a, o);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          gi[a] += o.J[m * 6 + 3 + a] * o.r[m];
#pragma unroll
          for (int c = 0; c < 3; ++c) Hd[a * 3 + c] += o.J[m * 6 + 3 + a] * o.J[m * 6 + 3 + c];
        }
      }
      cost += 0.5 * (o.r[0] * o.r[0] + o.r[1] * o.r[1] + o.r[2] * o.r[2]);
    }
    if (r == 1 && i + 1 < n) {  // edge (i, i+1): this pose is pose1 (columns 0..2)
      eval_odom(x + 3 * i, x + 3 * (i + 1), fv.odom_A[i + 1], fv.odom_rr[i + 1], fv.sig_r, fv.sig_t, fv.sig_a, o);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          gi[a] += o.J[m * 6 + a] * o.r[m];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            Hd[a * 3 + c] += o.J[m * 6 + a] * o.J[m * 6 + c];
            H1[a * 3 + c] += o.J[m * 6 + a] * o.J[m * 6 + 3 + c];
          }
        }
      }
    }
  }
  if (act && fv.with_hc && r == 2) {
    for (int h = fv.hc_off[i]; h < fv.hc_off[i + 1]; ++h) {  // residual_functors.h:1305-1411
      const uint32_t type = fv.hc_type[h];
      const float4 tg = fv.hc_tgt[h];
      const double ex = (double)tg.x - x[3 * i], ey = (double)tg.y - x[3 * i + 1], et = (double)tg.z - x[3 * i + 2];
      if (type == HITL_HC_LINE_SEGMENT) {  // r = (ex, ey, et), J = -I
        gi[0] -= ex; gi[1] -= ey; gi[2] -= et;
        Hd[0] += 1.0; Hd[4] += 1.0; Hd[8] += 1.0;
        cost += 0.5 * (ex * ex + ey * ey + et * et);
      } else if (type == HITL_HC_COLINEAR) {
        double sp, cp;
        sincos((double)tg.w, &sp, &cp);
        const double r0 = cp * ex + sp * ey;
        gi[0] -= cp * r0; gi[1] -= sp * r0; gi[2] -= et;
        Hd[0] += cp * cp; Hd[1] += cp * sp; Hd[3] += sp * cp; Hd[4] += sp * sp; Hd[8] += 1.0;
        cost += 0.5 * (r0 * r0 + et * et);
      } else if (type == HITL_HC_PERPENDICULAR || type == HITL_HC_PARALLEL) {
        gi[2] -= et;
        Hd[8] += 1.0;
        cost += 0.5 * et * et;
      }
    }
  }
  // quad totals (all four lanes take part, active or not)
#pragma unroll
  for (int q = 0; q < 9; ++q) Hd[q] = quad_sum(Hd[q]);
#pragma unroll
  for (int q = 0; q < 3; ++q) gi[q] = quad_sum(gi[q]);
  cost = quad_sum(cost);
#pragma unroll
  for (int q = 0; q < 9; ++q) H1[q] = __shfl(H1[q], (threadIdx.x & 60) + 1, WAVE);   // lane 1's odometry block, for lane 0
  double pose_cost = 0.0, pose_gmax = 0.0;
  if (act) {
    // Constant poses (global pose 0: JointOptimization.cpp:824, :1197; submap halos) keep their
    // assembled blocks here; k_build_system / k_spmv / k_jacobi_scale treat them as fixed.  Their
    // rows are only meaningful as COLUMNS of free neighbours (cross-cut coupling blocks).
    const bool is_fixed = fv.fixed[i] != 0;
    const bool is_owned = (i >= fv.own_lo) && (i < fv.own_hi);
    if (is_fixed) { gi[0] = gi[1] = gi[2] = 0.0; }
    if (!is_owned) cost = 0.0;        // every factor's cost is counted once: by the rank owning it
    double* Hrow = lin.Hband + (size_t)i * (Wb + 1) * 9;
    if (r == 0) {
#pragma unroll
      for (int q = 0; q < 9; ++q) Hrow[q] = Hd[q];
      lin.g[3 * i] = gi[0]; lin.g[3 * i + 1] = gi[1]; lin.g[3 * i + 2] = gi[2];
      pose_cost = cost;
      pose_gmax = fmax(fabs(gi[0]), fmax(fabs(gi[1]), fabs(gi[2])));
      if (gi[0] != gi[0] || gi[1] != gi[1] || gi[2] != gi[2]) pose_gmax = gi[0] + gi[1] + gi[2];  // NaN propagates
      if (!is_owned) pose_gmax = 0.0;
    }
    const int d = r + 1;             // this lane's band block (i, i+d)
    if (d <= Wb) {
      double blk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (i + d < n) {
        if (fv.with_stf) {
          const int fwd = fv.band_pair[((size_t)i * Wb + (d - 1)) * 2];
          const int rev = fv.band_pair[((size_t)i * Wb + (d - 1)) * 2 + 1];
          if (fwd >= 0) {  // pair (i -> i+d): block = J0^T J1
            const double* J = lin.pairRJ + (size_t)kPairRJ * fwd + 2;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
              for (int c = 0; c < 3; ++c) blk[a * 3 + c] += J[a] * J[3 + c] + J[6 + a] * J[9 + c];
          }
          if (rev >= 0) {  // pair (i+d -> i): block (i, i+d) = J1^T J0
            const double* J = lin.pairRJ + (size_t)kPairRJ * rev + 2;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
              for (int c = 0; c < 3; ++c) blk[a * 3 + c] += J[3 + a] * J[c] + J[9 + a] * J[6 + c];
          }
        }
        if (d == 1 && fv.with_odom) {
#pragma unroll
          for (int q = 0; q < 9; ++q) blk[q] += H1[q];
        }
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) Hrow[d * 9 + q] = blk[q];
    }
  }
  // per-workgroup partials (fixed order => bit-reproducible); the host adds the few hundred partials
  __shared__ double sm_c[kAsmBlock / WAVE], sm_g[kAsmBlock / WAVE];
  double c = wave_sum(pose_cost), gm = pose_gmax;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const double o = __shfl_xor(gm, m, WAVE); gm = (o > gm || o != o) ? o : gm; }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm_c[wv] = c; sm_g[wv] = gm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double cs = sm_c[0], gs = sm_g[0];
    for (int k = 1; k < kAsmBlock / WAVE; ++k) { cs += sm_c[k]; const double g1 = sm_g[k]; gs = (g1 > gs || g1 != g1) ? g1 : gs; }
    lin.partials[2 * blockIdx.x] = cs;
    lin.partials[2 * blockIdx.x + 1] = gs;
  }
}

__global__ __launch_bounds__(128) void k_assemble_offband(FactorView fv, LinView lin) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= fv.n_offband) return;
  double blk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  {
    const int fwd = fv.ob_fwd[e], rev = fv.ob_rev[e];
    if (fwd >= 0) {
      const double* J = lin.pairRJ + (size_t)kPairRJ * fwd + 2;
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) blk[a * 3 + c] += J[a] * J[3 + c] + J[6 + a] * J[9 + c];
    }
    if (rev >= 0) {
      const double* J = lin.pairRJ + (size_t)kPairRJ * rev + 2;
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) blk[a * 3 + c] += J[3 + a] * J[c] + J[9 + a] * J[6 + c];
    }
  }
  for (int q = 0; q < 9; ++q) lin.obblk[(size_t)e * 9 + q] = blk[q];
}

void launch_assemble(hipStream_t st, const MapView& map, const PairView& pr, const FactorView& fv, const double* x,
                     const LinView& lin) {
  const int n = map.n_poses;
  hipLaunchKernelGGL(k_assemble, dim3(assemble_blocks(n)), dim3(kAsmBlock), 0, st, map, pr, fv, x, lin);
  if (fv.with_stf && fv.n_offband > 0)
    hipLaunchKernelGGL(k_assemble_offband, dim3((fv.n_offband + 127) / 128), dim3(128), 0, st, fv, lin);
}

// ===========================================================================================
// single-workgroup deterministic reductions (vectors are <= a few 100k doubles)
// ===========================================================================================
constexpr int kRedBlock = 1024;

__device__ __forceinline__ double block_sum(double v, double* sm) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  double t = (threadIdx.x < kRedBlock / WAVE) ? sm[threadIdx.x] : 0.0;
  if (wave == 0) t = wave_sum(t);
  if (threadIdx.x == 0) sm[0] = t;
  __syncthreads();
  return sm[0];
}

__device__ __forceinline__ double block_max(double v, double* sm) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, WAVE));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  double t = (threadIdx.x < kRedBlock / WAVE) ? sm[threadIdx.x] : 0.0;
  if (wave == 0) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) t = fmax(t, __shfl_xor(t, m, WAVE));
  }
  if (threadIdx.x == 0) sm[0] = t;
  __syncthreads();
  return sm[0];
}

__global__ void k_jacobi_scale(int n_poses, int npad, int Wb, const double* __restrict__ Hband, int enable,
                               const unsigned char* __restrict__ fixed, double* __restrict__ scale) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= npad) return;
  double s = 1.0;
  if (enable && a < 3 * n_poses && !fixed[a / 3]) {
    const int i = a / 3, c = a % 3;
    s = 1.0 / (1.0 + sqrt(Hband[(size_t)i * (Wb + 1) * 9 + c * 3 + c]));
  }
  scale[a] = s;
}

void launch_jacobi_scale(hipStream_t st, int n_poses, int npad, int Wb, const double* Hband, int enable,
                         const unsigned char* fixed, double* scale) {
  hipLaunchKernelGGL(k_jacobi_scale, dim3((npad + 255) / 256), dim3(256), 0, st, n_poses, npad, Wb, Hband, enable, fixed, scale);
}

// ===========================================================================================
// build_system: banded 3x3-block Hessian -> block-tridiagonal system of B x B super-blocks
// (Wb consecutive poses each) in the Jacobi-scaled space with the LM diagonal
//   D^2 = clamp(diag(S H S), min, max) / mu     
// ===========================================================================================
// one thread per (super-block k, row a, column b): coalesced stores of D and L
__global__ void k_build_system(int n_poses, int Wb, const double* __restrict__ Hband, const double* __restrict__ g,
                               const double* __restrict__ scale, const unsigned char* __restrict__ fixed, double mu,
                               double min_diag, double max_diag, CrView cr, double* __restrict__ rhs, double* __restrict__ rhs_work) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int B = cr.B, BB = B * B;
  if (t >= cr.nsb * BB) return;
  const int k = t / BB, e = t % BB, a = e / B, b = e % B;
  const int i = k * Wb + a / 3, ca = a % 3;
  if (i >= n_poses || fixed[i]) {     // padding rows and constant poses: identity
    cr.D[t] = (b == a) ? 1.0 : 0.0;
    cr.L[t] = 0.0;
    if (b == 0) { rhs[k * B + a] = 0.0; if (rhs_work) rhs_work[k * B + a] = 0.0; }
    return;
  }
  const int W1 = Wb + 1;
  const double sa = scale[3 * i + ca];
  const int i2 = k * Wb + b / 3, cb = b % 3;
  double v = 0.0;
  if (i2 < n_poses && !fixed[i2]) {
    v = (i2 >= i) ? Hband[((size_t)i * W1 + (i2 - i)) * 9 + ca * 3 + cb]
                  : Hband[((size_t)i2 * W1 + (i - i2)) * 9 + cb * 3 + ca];
    v *= sa * scale[3 * i2 + cb];
  }
  if (b == a) v += fmin(fmax(v, min_diag), max_diag) / mu;
  cr.D[t] = v;
  double l = 0.0;
  if (k > 0) {
    const int i3 = (k - 1) * Wb + b / 3;
    const int d = i - i3;
    if (d <= Wb && !fixed[i3]) l = Hband[((size_t)i3 * W1 + d) * 9 + cb * 3 + ca] * sa * scale[3 * i3 + cb];
  }
  cr.L[t] = l;
  if (b == 0) {
    const double v = -sa * g[3 * i + ca];
    rhs[k * B + a] = v;
    if (rhs_work) rhs_work[k * B + a] = v;      // the copy the factor-solve consumes in place
  }
}

void launch_build_system(hipStream_t st, int n_poses, int Wb, const LinView& lin, const double* scale,
                         const unsigned char* fixed, double mu, double min_diag, double max_diag, const CrView& cr,
                         double* rhs, double* rhs_work) {
  const int tot = cr.nsb * cr.B * cr.B;
  hipLaunchKernelGGL(k_build_system, dim3((tot + 255) / 256), dim3(256), 0, st, n_poses, Wb, lin.Hband, lin.g, scale, fixed,
                     mu, min_diag, max_diag, cr, rhs, rhs_work);
}

// ===========================================================================================
// block cyclic reduction.  One 64-lane workgroup per super-block; B x B tiles live in LDS.
// Level stride s: nodes s, 3s, 5s.. are eliminated, nodes 0, 2s, 4s.. survive and absorb the
// Schur complement.  k_cr_invert inverts the level-0 odd nodes; k_cr_level then does one whole
// level per launch (survivor update + rhs sweep + inverse of the next level's odd nodes) and
// k_cr_top finishes the last kCrTopNodes nodes, the root and their back-substitution.
// ===========================================================================================
// Register-resident Gauss-Jordan sweep of one SPD super-block: lane (grp, c) owns column c of
// the augmented matrix [D | I | L_k | L_{k+s}^T]; for pivot p the pivot column is broadcast from
// lane p with v_readlane (no LDS, no barriers, no pivoting needed for SPD tiles).  Result:
// [I | Dinv | Gl = Dinv L_k | Gr = Dinv L_{k+s}^T].
template <int B>
__device__ __forceinline__ void gauss_jordan_cols(double (&col)[B]) {
#pragma unroll
  for (int p = 0; p < B; ++p) {
    double m[B];
#pragma unroll
    for (int r = 0; r < B; ++r) m[r] = readlane_f64(col[r], p);   // current column p (lane p, group 0)
    const double inv = 1.0 / m[p];
    const double vk = col[p] * inv;
#pragma unroll
    for (int r = 0; r < B; ++r) col[r] = (r == p) ? vk : col[r] - m[r] * vk;
  }
}

template <int B>
__device__ __forceinline__ void cr_eliminate_node(const CrView& cr, int s, int k, int lane) {
  static_assert(4 * B <= WAVE, "four column groups must fit one wave");
  const size_t off = (size_t)k * B * B;
  const bool has_right = (k + s) < cr.nsb;
  const int grp = lane / B, c = lane % B;
  const double* Dk = cr.D + off;
  const double* Lk = cr.L + off;
  const double* Lr = cr.L + (size_t)(has_right ? (k + s) : k) * B * B;
  double col[B];
#pragma unroll
  for (int r = 0; r < B; ++r) {
    double v = 0.0;
    if (grp == 0) v = Dk[r * B + c];
    else if (grp == 1) v = (r == c) ? 1.0 : 0.0;
    else if (grp == 2) v = Lk[r * B + c];
    else if (grp == 3 && has_right) v = Lr[c * B + r];      // column c of L_{k+s}^T
    col[r] = v;
  }
  gauss_jordan_cols<B>(col);
  double* dst = (grp == 1) ? cr.Dinv + off : (grp == 2) ? cr.Gl + off : cr.Gr + off;
  if (grp >= 1 && grp <= 3) {
#pragma unroll
    for (int r = 0; r < B; ++r) dst[r * B + c] = col[r];
  }
}

// Schur-complement update of survivor j at level stride s (LDS-free):
//   D_j  -= L_j Gr[k1] + L_{k2}^T Gl[k2]        L'_j = -L_j Gl[k1]
// Lane (grp, c) owns column c of the right operands (grp 0: Gr[k1], Gl[k2] -> D_j; grp 1: Gl[k1]
// -> L'_j); the left matrices L_j and L_{k2}^T are the same for every lane (uniform loads).
// With w != nullptr the right-hand side rides along (w_j -= Gr[k1]^T w_k1 + Gl[k2]^T w_k2), which
// saves the separate forward sweep when a factorisation is followed by exactly one solve.
template <int B>
__device__ __forceinline__ void cr_update_node(const CrView& cr, int s, int j, int lane, double* w, int nrhs, size_t stride) {
  const size_t off = (size_t)j * B * B;
  const int k1 = j - s, k2 = j + s;
  const bool has1 = k1 >= 0, has2 = k2 < cr.nsb, has_ll = (k1 - s) >= 0;
  const int grp = lane / B, c = lane % B;
  if (grp > 1) return;
  const double* Lj = cr.L + off;                                   // overwritten below: no __restrict__
  const double* L2 = cr.L + (size_t)(has2 ? k2 : j) * B * B;
  double b1[B], b2[B], acc[B];
  const double* B1 = (grp == 0 ? cr.Gr : cr.Gl) + (size_t)(has1 ? k1 : j) * B * B;
  const double* B2 = cr.Gl + (size_t)(has2 ? k2 : j) * B * B;
#pragma unroll
  for (int r = 0; r < B; ++r) {
    b1[r] = has1 ? B1[r * B + c] : 0.0;
    b2[r] = (has2 && grp == 0) ? B2[r * B + c] : 0.0;
    acc[r] = 0.0;
  }
  if (has1) {
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int q = 0; q < B; ++q) acc[r] += Lj[r * B + q] * b1[q];          // (L_j * B1)[r][c]
  }
  if (has2) {
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int q = 0; q < B; ++q) acc[r] += L2[q * B + r] * b2[q];          // (L_{k2}^T * Gl[k2])[r][c]
  }
  if (grp == 0) {
    if (w != nullptr) {   // rhs forward sweep: w_j[c] -= Gr[k1][:,c] . w_k1 + Gl[k2][:,c] . w_k2
      for (int v = 0; v < nrhs; ++v) {
        double* wv = w + stride * v;
        double wacc = wv[(size_t)j * B + c];
        if (has1) {
          const double* wk = wv + (size_t)k1 * B;
#pragma unroll
          for (int r = 0; r < B; ++r) wacc -= b1[r] * wk[r];
        }
        if (has2) {
          const double* wk = wv + (size_t)k2 * B;
#pragma unroll
          for (int r = 0; r < B; ++r) wacc -= b2[r] * wk[r];
        }
        wv[(size_t)j * B + c] = wacc;
      }
    }
#pragma unroll
    for (int r = 0; r < B; ++r) cr.D[off + r * B + c] -= acc[r];
  } else if (has1) {
    // every read of the old L_j happened above (same wave, in order); now it is replaced
#pragma unroll
    for (int r = 0; r < B; ++r) cr.L[off + r * B + c] = has_ll ? -acc[r] : 0.0;
  }
}

// One launch per big level (instead of eliminate + update): the workgroup of survivor j
//  1. forms, from the inverses Dinv[k1], Dinv[k2] of its two eliminated neighbours (computed by
//     the PREVIOUS level's launch, step 3 below), the products
//        Gr[k1] = Dinv_k1 L_j^T     Gl[k1] = Dinv_k1 L_k1     Gl[k2] = Dinv_k2 L_k2
//     (Gr/Gl of node k are stored by its right-hand survivor; the back substitution reads them),
//  2. applies the Schur complement  D_j -= L_j Gr[k1] + L_k2^T Gl[k2],  L'_j = -L_j Gl[k1]  and the
//     right-hand-side sweep  w_j -= Gr[k1]^T w_k1 + Gl[k2]^T w_k2,
//  3. if j itself is eliminated at the next level, inverts its new D_j (register-resident
//     Gauss-Jordan) so that level needs no separate inversion launch either.
template <int B>
__device__ __forceinline__ void lds_load(double* dst, const double* src) {
  for (int e = threadIdx.x; e < B * B; e += WAVE) dst[e] = src[e];
}

// C = A * Bm (TA: use A^T, TB: use Bm^T), all B x B tiles in LDS, one wave
template <int B, bool TA, bool TB>
__device__ __forceinline__ void lds_matmul(double* C, const double* A, const double* Bm) {
  for (int e = threadIdx.x; e < B * B; e += WAVE) {
    const int r = e / B, c = e % B;
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < B; ++q) v += (TA ? A[q * B + r] : A[r * B + q]) * (TB ? Bm[c * B + q] : Bm[q * B + c]);
    C[e] = v;
  }
}

template <int B>
__global__ __launch_bounds__(WAVE) void k_cr_level(CrView cr, int s, int invert_next, double* w, int nrhs, size_t stride) {
  __shared__ double sLj[B * B], sL1[B * B], sL2[B * B], sI1[B * B], sI2[B * B], sGr1[B * B], sGl1[B * B], sGl2[B * B], sD[B * B];
  const int j = 2 * s * blockIdx.x;
  if (j >= cr.nsb) return;
  const size_t BB = (size_t)B * B, off = (size_t)j * BB;
  const int k1 = j - s, k2 = j + s;
  const bool has1 = k1 >= 0, has2 = k2 < cr.nsb, has_ll = (k1 - s) >= 0;
  lds_load<B>(sD, cr.D + off);
  if (has1) { lds_load<B>(sLj, cr.L + off); lds_load<B>(sL1, cr.L + (size_t)k1 * BB); lds_load<B>(sI1, cr.Dinv + (size_t)k1 * BB); }
  if (has2) { lds_load<B>(sL2, cr.L + (size_t)k2 * BB); lds_load<B>(sI2, cr.Dinv + (size_t)k2 * BB); }
  __syncthreads();
  if (has1) {
    lds_matmul<B, false, true>(sGr1, sI1, sLj);       // Dinv_k1 * L_j^T
    if (has_ll) lds_matmul<B, false, false>(sGl1, sI1, sL1);   // Dinv_k1 * L_k1
    else for (int e = threadIdx.x; e < B * B; e += WAVE) sGl1[e] = 0.0;
  }
  if (has2) lds_matmul<B, false, false>(sGl2, sI2, sL2);       // Dinv_k2 * L_k2
  __syncthreads();
  // factors of the eliminated neighbours, for the back substitution
  if (has1) for (int e = threadIdx.x; e < B * B; e += WAVE) { cr.Gr[(size_t)k1 * BB + e] = sGr1[e]; cr.Gl[(size_t)k1 * BB + e] = sGl1[e]; }
  if (has2 && (k2 + s) >= cr.nsb)      // k2 has no right-hand survivor: its Gl comes from here, its Gr is zero
    for (int e = threadIdx.x; e < B * B; e += WAVE) { cr.Gl[(size_t)k2 * BB + e] = sGl2[e]; cr.Gr[(size_t)k2 * BB + e] = 0.0; }
  // right-hand-side sweep
  if (w != nullptr && (int)threadIdx.x < B) {
    for (int v = 0; v < nrhs; ++v) {
      double* wv = w + stride * v;
      double acc_w = wv[(size_t)j * B + threadIdx.x];
      if (has1) { const double* wk = wv + (size_t)k1 * B;
#pragma unroll
        for (int r = 0; r < B; ++r) acc_w -= sGr1[r * B + threadIdx.x] * wk[r]; }
      if (has2) { const double* wk = wv + (size_t)k2 * B;
#pragma unroll
        for (int r = 0; r < B; ++r) acc_w -= sGl2[r * B + threadIdx.x] * wk[r]; }
      wv[(size_t)j * B + threadIdx.x] = acc_w;
    }
  }
  // Schur complement
  for (int e = threadIdx.x; e < B * B; e += WAVE) {
    const int r = e / B, c = e % B;
    double d = sD[e], l = 0.0;
    if (has1) {
#pragma unroll
      for (int q = 0; q < B; ++q) { d -= sLj[r * B + q] * sGr1[q * B + c]; l -= sLj[r * B + q] * sGl1[q * B + c]; }
    }
    if (has2) {
#pragma unroll
      for (int q = 0; q < B; ++q) d -= sL2[q * B + r] * sGl2[q * B + c];
    }
    cr.D[off + e] = d;
    if (has1) cr.L[off + e] = l;
    sI1[e] = d;                       // reuse: new D_j for the inversion below (sI1 is dead after the matmuls... see barrier)
  }
  // inversion of the new D_j when j is eliminated at the next level
  if (invert_next && (blockIdx.x & 1)) {
    __syncthreads();
    const int grp = threadIdx.x / B, c = threadIdx.x % B;
    double col[B];
#pragma unroll
    for (int r = 0; r < B; ++r) col[r] = (grp == 0) ? sI1[r * B + c] : ((grp == 1 && r == c) ? 1.0 : 0.0);
    gauss_jordan_cols<B>(col);
    if (grp == 1) {
#pragma unroll
      for (int r = 0; r < B; ++r) cr.Dinv[off + r * B + c] = col[r];
    }
  }
}

// Dinv of the nodes eliminated at the first level (one wave per node)
template <int B>
__global__ __launch_bounds__(WAVE) void k_cr_invert(CrView cr, int s) {
  const int k = s * (2 * blockIdx.x + 1);
  if (k >= cr.nsb) return;
  const size_t off = (size_t)k * B * B;
  const int grp = threadIdx.x / B, c = threadIdx.x % B;
  double col[B];
#pragma unroll
  for (int r = 0; r < B; ++r) col[r] = (grp == 0) ? cr.D[off + r * B + c] : ((grp == 1 && r == c) ? 1.0 : 0.0);
  gauss_jordan_cols<B>(col);
  if (grp == 1) {
#pragma unroll
    for (int r = 0; r < B; ++r) cr.Dinv[off + r * B + c] = col[r];
  }
}

template <int B>
__device__ __forceinline__ void cr_root_node(const CrView& cr, int lane) {
  const int grp = lane / B, c = lane % B;
  double col[B];
#pragma unroll
  for (int r = 0; r < B; ++r) col[r] = (grp == 0) ? cr.D[r * B + c] : ((grp == 1 && r == c) ? 1.0 : 0.0);
  gauss_jordan_cols<B>(col);
  if (grp == 1) {
#pragma unroll
    for (int r = 0; r < B; ++r) cr.Dinv[r * B + c] = col[r];
  }
}

template <int B>
__global__ __launch_bounds__(WAVE) void k_cr_root(CrView cr) { cr_root_node<B>(cr, threadIdx.x); }

// back substitution of one row: x_k[r] = (Dinv_k w_k - Gl_k x_{k-s} - Gr_k x_{k+s})[r]
template <int B>
__device__ __forceinline__ void cr_backward_row(const CrView& cr, int s, int k, int r, const double* w, double* x) {
  const size_t off = (size_t)k * B * B;
  const double* wk = w + (size_t)k * B;
  const double* xl = x + (size_t)(k - s) * B;
  const bool has_right = (k + s) < cr.nsb;
  const double* xr = x + (size_t)(has_right ? (k + s) : k) * B;
  double v = 0.0;
#pragma unroll
  for (int c = 0; c < B; ++c) {
    v += cr.Dinv[off + r * B + c] * wk[c];
    v -= cr.Gl[off + r * B + c] * xl[c];
    if (has_right) v -= cr.Gr[off + r * B + c] * xr[c];
  }
  x[(size_t)k * B + r] = v;
}

// All levels from stride s_start up to the root, the root itself and (x != nullptr) the
// matching back-substitution levels in ONE single-workgroup launch: these levels hold at most a
// few hundred super-blocks, so they are pure launch latency when issued as 3 kernels per level.
// 16 waves, one super-block per wave per step, __syncthreads between phases (all traffic is
// workgroup-local global memory).
constexpr int kCrTopThreads = 1024;
template <int B>
__global__ __launch_bounds__(kCrTopThreads) void k_cr_top(CrView cr, int s_start, double* w, double* x, int nrhs, size_t stride) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NW = kCrTopThreads / WAVE;
  int s_last = 0;
  for (int s = s_start; s < cr.nsb; s *= 2) {
    for (int q = wave;; q += NW) {
      const int k = s * (2 * q + 1);
      if (k >= cr.nsb) break;
      cr_eliminate_node<B>(cr, s, k, lane);
    }
    __syncthreads();
    for (int q = wave;; q += NW) {
      const int j = 2 * s * q;
      if (j >= cr.nsb) break;
      cr_update_node<B>(cr, s, j, lane, w, nrhs, stride);
    }
    __syncthreads();
    s_last = s;
  }
  if (wave == 0) cr_root_node<B>(cr, lane);
  __syncthreads();
  if (x == nullptr) return;
  if ((int)threadIdx.x < B * nrhs) {
    const int vv = threadIdx.x / B, r = threadIdx.x % B;
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < B; ++c) v += cr.Dinv[r * B + c] * w[stride * vv + c];
    x[stride * vv + r] = v;
  }
  __syncthreads();
  for (int s = s_last; s >= s_start && s >= 1; s /= 2) {
    for (int vv = 0; vv < nrhs; ++vv)
      for (int t = threadIdx.x;; t += kCrTopThreads) {
        const int k = s * (2 * (t / B) + 1);
        if (k >= cr.nsb) break;
        cr_backward_row<B>(cr, s, k, t % B, w + stride * vv, x + stride * vv);
      }
    __syncthreads();
  }
}

// The same top of the tree with every tile resident in LDS: the M = ceil(nsb / s_start) nodes alive at stride s_start
// are copied in once (D, L, right-hand sides), all levels, the root and the back-substitution run on a LOCAL system of
// M nodes with local strides 1, 2, 4 .. through the very same node routines (they only see a CrView whose arrays
// happen to live in LDS), and Dinv / Gl / Gr / w / x go back out once.  Global memory is touched twice instead of at
// every one of the ~20 dependent phases.
template <int B>
__global__ __launch_bounds__(kCrTopThreads) void k_cr_top_lds(CrView cr, int s_start, int M, double* w, double* x, int nrhs, size_t stride) {
  extern __shared__ double lds[];
  constexpr int BB = B * B;
  double* sD = lds;
  double* sL = sD + (size_t)M * BB;
  double* sDinv = sL + (size_t)M * BB;
  double* sGl = sDinv + (size_t)M * BB;
  double* sGr = sGl + (size_t)M * BB;
  double* sw = sGr + (size_t)M * BB;                 // nrhs vectors of M * B, lstride apart
  double* sx = sw + (size_t)M * B * nrhs;
  const size_t lstride = (size_t)M * B;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NW = kCrTopThreads / WAVE;
  for (int e = threadIdx.x; e < M * BB; e += kCrTopThreads) {
    const int m = e / BB, q = e % BB;
    const size_t g = (size_t)m * s_start * BB + q;
    sD[e] = cr.D[g]; sL[e] = cr.L[g]; sGl[e] = 0.0; sGr[e] = 0.0; sDinv[e] = 0.0;
  }
  if (w != nullptr)
    for (int e = threadIdx.x; e < M * B * nrhs; e += kCrTopThreads) {
      const int v = e / (M * B), r = e % (M * B);
      sw[(size_t)v * lstride + r] = w[stride * v + (size_t)(r / B) * s_start * B + (r % B)];
    }
  __syncthreads();
  CrView lc;
  lc.nsb = M; lc.B = B; lc.D = sD; lc.L = sL; lc.Dinv = sDinv; lc.Gl = sGl; lc.Gr = sGr;
  double* lw = (w != nullptr) ? sw : nullptr;
  int s_last = 0;
  for (int s = 1; s < M; s *= 2) {
    for (int q = wave;; q += NW) {
      const int k = s * (2 * q + 1);
      if (k >= M) break;
      cr_eliminate_node<B>(lc, s, k, lane);
    }
    __syncthreads();
    for (int q = wave;; q += NW) {
      const int j = 2 * s * q;
      if (j >= M) break;
      cr_update_node<B>(lc, s, j, lane, lw, nrhs, lstride);
    }
    __syncthreads();
    s_last = s;
  }
  if (wave == 0) cr_root_node<B>(lc, lane);
  __syncthreads();
  if (x != nullptr) {
    if ((int)threadIdx.x < B * nrhs) {
      const int vv = threadIdx.x / B, r = threadIdx.x % B;
      double v = 0.0;
#pragma unroll
      for (int c = 0; c < B; ++c) v += sDinv[r * B + c] * sw[lstride * vv + c];
      sx[lstride * vv + r] = v;
    }
    __syncthreads();
    for (int s = s_last; s >= 1; s /= 2) {
      for (int vv = 0; vv < nrhs; ++vv)
        for (int t = threadIdx.x;; t += kCrTopThreads) {
          const int k = s * (2 * (t / B) + 1);
          if (k >= M) break;
          cr_backward_row<B>(lc, s, k, t % B, sw + lstride * vv, sx + lstride * vv);
        }
      __syncthreads();
    }
  }
  for (int e = threadIdx.x; e < M * BB; e += kCrTopThreads) {
    const int m = e / BB, q = e % BB;
    const size_t g = (size_t)m * s_start * BB + q;
    cr.Dinv[g] = sDinv[e]; cr.Gl[g] = sGl[e]; cr.Gr[g] = sGr[e];
  }
  if (w != nullptr)
    for (int e = threadIdx.x; e < M * B * nrhs; e += kCrTopThreads) {
      const int v = e / (M * B), r = e % (M * B);
      const size_t g = stride * v + (size_t)(r / B) * s_start * B + (r % B);
      w[g] = sw[(size_t)v * lstride + r];
      if (x != nullptr) x[g] = sx[(size_t)v * lstride + r];
    }
}

// forward sweep on right-hand sides (blockIdx.y selects the rhs, `stride` doubles apart):
// w_j -= Gr[k1]^T w_k1 + Gl[k2]^T w_k2
template <int B>
__global__ void k_cr_forward(CrView cr, int s, double* __restrict__ w, size_t stride) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int q = t / B, c = t % B;
  const int j = 2 * s * q;
  if (j >= cr.nsb) return;
  w += stride * blockIdx.y;
  const int k1 = j - s, k2 = j + s;
  double v = w[(size_t)j * B + c];
  if (k1 >= 0) {
    const double* G = cr.Gr + (size_t)k1 * B * B;
    const double* wk = w + (size_t)k1 * B;
#pragma unroll
    for (int r = 0; r < B; ++r) v -= G[r * B + c] * wk[r];
  }
  if (k2 < cr.nsb) {
    const double* G = cr.Gl + (size_t)k2 * B * B;
    const double* wk = w + (size_t)k2 * B;
#pragma unroll
    for (int r = 0; r < B; ++r) v -= G[r * B + c] * wk[r];
  }
  w[(size_t)j * B + c] = v;
}

// back substitution: x_k = Dinv_k w_k - Gl_k x_{k-s} - Gr_k x_{k+s}
template <int B>
__global__ void k_cr_backward(CrView cr, int s, const double* __restrict__ w, double* __restrict__ x, size_t stride) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int q = t / B, r = t % B;
  const int k = s * (2 * q + 1);
  if (k >= cr.nsb) return;
  cr_backward_row<B>(cr, s, k, r, w + stride * blockIdx.y, x + stride * blockIdx.y);
}

// The back-substitution is top-down, so once x is known at every multiple of S the S - 1 nodes between two such
// neighbours form an independent subtree: one workgroup walks it through strides S/2 .. 1 with a barrier per level.
// Replaces log2(S) latency-bound launches by one.
constexpr int kCrTreeThreads = 256;
template <int B>
__global__ __launch_bounds__(kCrTreeThreads) void k_cr_backward_tree(CrView cr, int S, const double* w, double* x, int nrhs, size_t stride) {
  // x of the interval's S + 1 nodes lives in LDS between the levels (one right-hand side at a time), so a level only
  // waits for its own tile loads, not for the previous level's stores to come back from L2
  extern __shared__ double sx[];                       // (S + 1) * B
  const int base = blockIdx.x * S;
  {
    const int vv = blockIdx.y;                          // one right-hand side per workgroup row
    const double* wv = w + stride * vv;
    double* xv = x + stride * vv;
    for (int t = threadIdx.x; t < 2 * B; t += kCrTreeThreads) {            // the two boundary nodes (known)
      const int node = (t < B) ? base : base + S, r = t % B;
      sx[(size_t)(node - base) * B + r] = (node < cr.nsb) ? xv[(size_t)node * B + r] : 0.0;
    }
    __syncthreads();
    for (int s = S / 2; s >= 1; s /= 2) {
      const int per = S / (2 * s);
      for (int t = threadIdx.x; t < per * B; t += kCrTreeThreads) {
        const int k = base + s * (2 * (t / B) + 1), r = t % B;
        if (k < cr.nsb) {
          const size_t off = (size_t)k * B * B;
          const double* wk = wv + (size_t)k * B;
          const double* xl = sx + (size_t)(k - s - base) * B;
          const double* xr = sx + (size_t)(k + s - base) * B;             // zero-filled beyond the last node
          double v = 0.0;
#pragma unroll
          for (int c = 0; c < B; ++c) {
            v += cr.Dinv[off + r * B + c] * wk[c];
            v -= cr.Gl[off + r * B + c] * xl[c];
            if (k + s < cr.nsb) v -= cr.Gr[off + r * B + c] * xr[c];
          }
          sx[(size_t)(k - base) * B + r] = v;
        }
      }
      __syncthreads();
    }
    for (int t = threadIdx.x; t < (S - 1) * B; t += kCrTreeThreads) {       // interior nodes go out once
      const int node = base + 1 + t / B;
      if (node < cr.nsb) xv[(size_t)node * B + t % B] = sx[(size_t)(node - base) * B + t % B];
    }
  }
}

template <int B>
static void cr_backward_levels(hipStream_t st, const CrView& cr, int s_from, const double* w, double* x, int nrhs, size_t stride) {
  // strides s_from, s_from / 2 .. 1: wide levels one launch each, the subtrees below stride kTree in one launch
  constexpr int kTree = 64;
  int s = s_from;
  for (; s >= kTree; s /= 2) {
    const int n_elim = (cr.nsb - s + 2 * s - 1) / (2 * s);
    hipLaunchKernelGGL(k_cr_backward<B>, dim3((n_elim * B + 127) / 128, nrhs), dim3(128), 0, st, cr, s, w, x, stride);
  }
  if (s >= 1) {
    const int S = 2 * s;
    hipLaunchKernelGGL(k_cr_backward_tree<B>, dim3((cr.nsb + S - 1) / S, nrhs), dim3(kCrTreeThreads), sizeof(double) * (S + 1) * B, st, cr, S, w, x,
                       nrhs, stride);
  }
}

template <int B>
__global__ void k_cr_root_solve(CrView cr, const double* __restrict__ w, double* __restrict__ x, size_t stride) {
  const int r = threadIdx.x;
  if (r >= B) return;
  w += stride * blockIdx.x;
  x += stride * blockIdx.x;
  double v = 0.0;
#pragma unroll
  for (int c = 0; c < B; ++c) v += cr.Dinv[r * B + c] * w[c];
  x[r] = v;
}

// first stride handled by the fused single-workgroup kernel: at most kCrTopNodes eliminations at that stride (one
// per wave) and every tile of the nodes alive there resident in LDS
constexpr int kCrTopNodes = 16;
constexpr size_t kCrTopLdsBytes = 144 * 1024;
static size_t cr_top_lds_bytes(int M, int B, int nrhs) { return sizeof(double) * ((size_t)5 * M * B * B + (size_t)2 * M * B * nrhs); }
static int cr_top_start(int nsb, int B, int nrhs) {
  int s = 1;
  while (s < nsb && ((nsb - s + 2 * s - 1) / (2 * s) > kCrTopNodes || cr_top_lds_bytes((nsb + s - 1) / s, B, nrhs) > kCrTopLdsBytes)) s *= 2;
  return s;
}

// x == nullptr: factorise only (w, if given, still receives the forward sweep)
template <int B>
static void cr_factor_t(hipStream_t st, const CrView& cr, double* w, double* x, int nrhs, size_t stride) {
  const int nr = (w != nullptr) ? nrhs : 0;
  const int s_top = cr_top_start(cr.nsb, B, nr);
  if (s_top > 1) {
    const int n_elim = (cr.nsb - 1 + 1) / 2;
    hipLaunchKernelGGL(k_cr_invert<B>, dim3(n_elim), dim3(WAVE), 0, st, cr, 1);
  }
  for (int s = 1; s < s_top; s *= 2) {
    const int n_surv = (cr.nsb + 2 * s - 1) / (2 * s);
    const int invert_next = (2 * s < s_top) ? 1 : 0;    // the fused top kernel inverts its own nodes
    hipLaunchKernelGGL(k_cr_level<B>, dim3(n_surv), dim3(WAVE), 0, st, cr, s, invert_next, w, nrhs, stride);
  }
  {
    const int M = (cr.nsb + s_top - 1) / s_top;
    const size_t lds_bytes = cr_top_lds_bytes(M, B, nr);
    static bool raised = false;      // more than the default 64 KB of dynamic LDS needs to be asked for once
    if (!raised) {
      raised = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cr_top_lds<B>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kCrTopLdsBytes) == hipSuccess;
      if (!raised) (void)hipGetLastError();
    }
    if (lds_bytes <= (raised ? kCrTopLdsBytes : (size_t)64 * 1024))
      hipLaunchKernelGGL(k_cr_top_lds<B>, dim3(1), dim3(kCrTopThreads), lds_bytes, st, cr, s_top, M, w, x, nr, stride);
    else
      hipLaunchKernelGGL(k_cr_top<B>, dim3(1), dim3(kCrTopThreads), 0, st, cr, s_top, w, x, nrhs, stride);
  }
  if (x != nullptr && s_top >= 2) cr_backward_levels<B>(st, cr, s_top / 2, w, x, nrhs, stride);
}

// back substitution only, any number of right-hand sides (after cr_fwd_t)
template <int B>
static void cr_back_t(hipStream_t st, const CrView& cr, const double* w, double* x, int nrhs, size_t stride) {
  int s_last = 0;
  for (int s = 1; s < cr.nsb; s *= 2) s_last = s;
  hipLaunchKernelGGL(k_cr_root_solve<B>, dim3(nrhs), dim3(WAVE), 0, st, cr, w, x, stride);
  if (s_last >= 1) cr_backward_levels<B>(st, cr, s_last, w, x, nrhs, stride);
}

template <int B>
static void cr_fwd_t(hipStream_t st, const CrView& cr, double* w, int nrhs, size_t stride) {
  for (int s = 1; s < cr.nsb; s *= 2) {
    const int n_surv = (cr.nsb + 2 * s - 1) / (2 * s);
    hipLaunchKernelGGL(k_cr_forward<B>, dim3((n_surv * B + 127) / 128, nrhs), dim3(128), 0, st, cr, s, w, stride);
  }
}

template <int B>
static void cr_solve_t(hipStream_t st, const CrView& cr, double* w, double* x, int nrhs, size_t stride) {
  cr_fwd_t<B>(st, cr, w, nrhs, stride);
  cr_back_t<B>(st, cr, w, x, nrhs, stride);
}

void launch_cr_factor(hipStream_t st, const CrView& cr) {
  switch (cr.B) {
    case 3: cr_factor_t<3>(st, cr, nullptr, nullptr, 0, 0); break;
    case 6: cr_factor_t<6>(st, cr, nullptr, nullptr, 0, 0); break;
    case 9: cr_factor_t<9>(st, cr, nullptr, nullptr, 0, 0); break;
    default: cr_factor_t<12>(st, cr, nullptr, nullptr, 0, 0); break;
  }
}

void launch_cr_factor_solve(hipStream_t st, const CrView& cr, double* w, double* x, int nrhs, size_t stride) {
  // nrhs right-hand sides `stride` doubles apart ride through the factorisation
  switch (cr.B) {
    case 3: cr_factor_t<3>(st, cr, w, x, nrhs, stride); break;
    case 6: cr_factor_t<6>(st, cr, w, x, nrhs, stride); break;
    case 9: cr_factor_t<9>(st, cr, w, x, nrhs, stride); break;
    default: cr_factor_t<12>(st, cr, w, x, nrhs, stride); break;
  }
}

void launch_cr_solve_multi(hipStream_t st, const CrView& cr, double* w, double* x, int nrhs, size_t stride) {
  switch (cr.B) {
    case 3: cr_solve_t<3>(st, cr, w, x, nrhs, stride); break;
    case 6: cr_solve_t<6>(st, cr, w, x, nrhs, stride); break;
    case 9: cr_solve_t<9>(st, cr, w, x, nrhs, stride); break;
    default: cr_solve_t<12>(st, cr, w, x, nrhs, stride); break;
  }
}

void launch_cr_solve(hipStream_t st, const CrView& cr, double* w, double* x) {
  launch_cr_solve_multi(st, cr, w, x, 1, 0);
}

// ===========================================================================================
// Selected inversion through the cyclic-reduction factors: the diagonal super-blocks of A^-1.
// A node k eliminated at stride s between survivors a = k - s and b = k + s satisfies
//     x_k = Dinv_k w_k - Gl_k x_a - Gr_k x_b ,
// so with Sigma = A^-1 and the blocks of the coarser level (Sigma_aa, Sigma_bb, Sigma_ab) known,
//     M1 = Gl Sigma_aa + Gr Sigma_ab^T        M2 = Gl Sigma_ab + Gr Sigma_bb
//     Sigma_kk = Dinv_k + M1 Gl^T + M2 Gr^T    Sigma_ak = -M1^T      Sigma_kb = -M2
// which are exactly the blocks the next finer level needs (a,k and k,b are adjacent there).
// Sd[node] = Sigma_node,node; So[node] = Sigma between node and its right neighbour at the level
// processed last.  One launch per level, one wave per eliminated node; ~log2(nsb) launches.
// ===========================================================================================
template <int B>
__global__ __launch_bounds__(WAVE) void k_cr_selinv_level(CrView cr, int s, double* Sd, double* So) {
  __shared__ double sGl[B * B], sGr[B * B], sAA[B * B], sBB[B * B], sAB[B * B], sM1[B * B], sM2[B * B];
  const int k = s * (2 * blockIdx.x + 1);
  if (k >= cr.nsb) return;
  const size_t BB = (size_t)B * B;
  const int a = k - s, b = k + s;
  const bool has_b = b < cr.nsb;
  lds_load<B>(sGl, cr.Gl + (size_t)k * BB);
  lds_load<B>(sAA, Sd + (size_t)a * BB);
  if (has_b) { lds_load<B>(sGr, cr.Gr + (size_t)k * BB); lds_load<B>(sBB, Sd + (size_t)b * BB); lds_load<B>(sAB, So + (size_t)a * BB); }
  else for (int e = threadIdx.x; e < B * B; e += WAVE) { sGr[e] = 0.0; sBB[e] = 0.0; sAB[e] = 0.0; }
  __syncthreads();
  for (int e = threadIdx.x; e < B * B; e += WAVE) {
    const int r = e / B, c = e % B;
    double m1 = 0.0, m2 = 0.0;
#pragma unroll
    for (int q = 0; q < B; ++q) {
      m1 += sGl[r * B + q] * sAA[q * B + c] + sGr[r * B + q] * sAB[c * B + q];
      m2 += sGl[r * B + q] * sAB[q * B + c] + sGr[r * B + q] * sBB[q * B + c];
    }
    sM1[e] = m1; sM2[e] = m2;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < B * B; e += WAVE) {
    const int r = e / B, c = e % B;
    double v = cr.Dinv[(size_t)k * BB + e];
#pragma unroll
    for (int q = 0; q < B; ++q) v += sM1[r * B + q] * sGl[c * B + q] + sM2[r * B + q] * sGr[c * B + q];
    Sd[(size_t)k * BB + e] = v;
    So[(size_t)a * BB + e] = -sM1[c * B + r];
    if (has_b) So[(size_t)k * BB + e] = -sM2[e];
  }
}

template <int B>
static void cr_selinv_t(hipStream_t st, const CrView& cr, double* Sd, double* So) {
  (void)hipMemcpyAsync(Sd, cr.Dinv, sizeof(double) * B * B, hipMemcpyDeviceToDevice, st);   // root
  int s_last = 0;
  for (int s = 1; s < cr.nsb; s *= 2) s_last = s;
  for (int s = s_last; s >= 1; s /= 2) {
    const int n_elim = (cr.nsb - s + 2 * s - 1) / (2 * s);
    hipLaunchKernelGGL(k_cr_selinv_level<B>, dim3(n_elim), dim3(WAVE), 0, st, cr, s, Sd, So);
  }
}

void launch_cr_selinv(hipStream_t st, const CrView& cr, double* Sd, double* So) {
  switch (cr.B) {
    case 3: cr_selinv_t<3>(st, cr, Sd, So); break;
    case 6: cr_selinv_t<6>(st, cr, Sd, So); break;
    case 9: cr_selinv_t<9>(st, cr, Sd, So); break;
    default: cr_selinv_t<12>(st, cr, Sd, So); break;
  }
}

// pose i: its 3x3 block of S Sigma S, rotated into the robot frame as vector_mapping.cpp:2792-2801 does
// (transform = blockdiag(Rotation2Df(-angle).cast<double>(), 1); one_cov = transform * one_cov * transform^T)
__global__ void k_cov_extract(int n_poses, int Wb, int B, const double* __restrict__ Sd, const double* __restrict__ scale,
                              const unsigned char* __restrict__ fixed, const float2* __restrict__ rot, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_poses) return;
  double* o = out + 9 * (size_t)i;
  if (fixed[i]) { for (int e = 0; e < 9; ++e) o[e] = 0.0; return; }
  const int k = i / Wb, a0 = 3 * (i % Wb);
  const double* blk = Sd + (size_t)k * B * B;
  double S[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) S[r][c] = blk[(a0 + r) * B + a0 + c] * scale[3 * i + r] * scale[3 * i + c];
  const double c = (double)rot[i].x, sn = (double)rot[i].y;     // cos, sin of -angle (fp32 values)
  const double T[3][3] = {{c, -sn, 0.0}, {sn, c, 0.0}, {0.0, 0.0, 1.0}};
  double M[3][3];
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) M[r][q] = T[r][0] * S[0][q] + T[r][1] * S[1][q] + T[r][2] * S[2][q];
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) o[3 * r + q] = M[r][0] * T[q][0] + M[r][1] * T[q][1] + M[r][2] * T[q][2];
}

void launch_cov_extract(hipStream_t st, int n_poses, int Wb, int B, const double* Sd, const double* scale, const unsigned char* fixed,
                        const float2* rot, double* out) {
  hipLaunchKernelGGL(k_cov_extract, dim3((n_poses + 127) / 128), dim3(128), 0, st, n_poses, Wb, B, Sd, scale, fixed, rot, out);
}

// ===========================================================================================
// y = (S H S [+ D^2]) v : band blocks (both triangles) + off-band blocks, one thread per pose.
// MODE 0: plain product (PCG).
// MODE 1: residual check after the cyclic-reduction solve: y -> r = rhs - A v, per-workgroup
//         partials (r.r, rhs.rhs) in slots 0,1 of the 8-wide record.
// MODE 2: trust-region step bookkeeping, fused: q = (S H S) v (no damping), partials
//         (v.rhs, v.q, sum (s v)^2 over free poses, sum x^2 over free poses, #non-finite) in
//         slots 2..6, and the candidate state xc = x + s*v.
// The host adds the few hundred per-workgroup partials in index order (bit-reproducible).
// ===========================================================================================
struct SpmvAux {
  const double* rhs;   // MODE 1, 2
  double* r_out;       // MODE 1
  const double* x;     // MODE 2
  double* xc;          // MODE 2
  double* partials;    // MODE 1, 2 : [gridDim.x][8]
};

template <int MODE>
__global__ __launch_bounds__(128) void k_spmv(int n_poses, int npad, FactorView fv, LinView lin,
                                               const double* __restrict__ scale, double mu, double min_diag,
                                               double max_diag, const double* __restrict__ v, double* __restrict__ y,
                                               SpmvAux aux) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // MODE 3 = MODE 1 and MODE 2 in one pass (banded single-GPU solve: v is exactly 0 on constant poses, so including
  // their columns changes nothing): residual partials in slots 0,1 and the step bookkeeping in slots 2..6
  constexpr bool kRes = (MODE == 1 || MODE == 3), kStep = (MODE == 2 || MODE == 3);
  double part[7] = {0, 0, 0, 0, 0, 0, 0};
  if (3 * i < npad) {
    if (i >= n_poses || fv.fixed[i]) {  // padding rows and constant poses: identity rows
      for (int c = 0; c < 3; ++c) {
        const int t = 3 * i + c;
        if (t < npad) {
          if (MODE == 0) y[t] = v[t];
          if (kRes) { const double r = aux.rhs[t] - v[t]; if (MODE == 1) aux.r_out[t] = r; part[0] += r * r; part[1] += aux.rhs[t] * aux.rhs[t]; }
          if (kStep) { y[t] = v[t]; if (i < n_poses) aux.xc[t] = aux.x[t]; }
        }
      }
    } else {
      // Columns of constant poses are not part of the linear system (MODE 0/1).  In MODE 2 they
      // ARE included: v then carries the neighbouring submap's step on the halo poses (0 for pose 0),
      // so owned rows see the cross-cut coupling in the model cost change.
      const int Wb = fv.band_poses, W1 = Wb + 1;
      double acc[3] = {0, 0, 0};
      const double* Hi = lin.Hband + (size_t)i * W1 * 9;
      for (int d = 0; d <= Wb && i + d < n_poses; ++d) {
        const int j = i + d;
        if (!kStep && fv.fixed[j]) continue;
        const double* blk = Hi + d * 9;
        const double v0 = scale[3 * j] * v[3 * j], v1 = scale[3 * j + 1] * v[3 * j + 1], v2 = scale[3 * j + 2] * v[3 * j + 2];
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[a] += blk[a * 3] * v0 + blk[a * 3 + 1] * v1 + blk[a * 3 + 2] * v2;
      }
      for (int d = 1; d <= Wb && i - d >= 0; ++d) {
        const int j = i - d;
        if (!kStep && fv.fixed[j]) continue;
        const double* blk = lin.Hband + ((size_t)j * W1 + d) * 9;  // block (j, i), used transposed
        const double v0 = scale[3 * j] * v[3 * j], v1 = scale[3 * j + 1] * v[3 * j + 1], v2 = scale[3 * j + 2] * v[3 * j + 2];
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[a] += blk[a] * v0 + blk[3 + a] * v1 + blk[6 + a] * v2;
      }
      if (fv.with_stf && fv.n_offband > 0) {
        for (int e = fv.obadj_off[i]; e < fv.obadj_off[i + 1]; ++e) {
          const int ent = fv.obadj_ent[e];
          const int en = ent >> 1, tr = ent & 1;
          const double* blk = lin.obblk + (size_t)en * 9;
          const int j = tr ? fv.ob_lo[en] : fv.ob_hi[en];
          if (!kStep && fv.fixed[j]) continue;
          const double v0 = scale[3 * j] * v[3 * j], v1 = scale[3 * j + 1] * v[3 * j + 1], v2 = scale[3 * j + 2] * v[3 * j + 2];
          if (!tr) {
            for (int a = 0; a < 3; ++a) acc[a] += blk[a * 3] * v0 + blk[a * 3 + 1] * v1 + blk[a * 3 + 2] * v2;
          } else {
            for (int a = 0; a < 3; ++a) acc[a] += blk[a] * v0 + blk[3 + a] * v1 + blk[6 + a] * v2;
          }
        }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int t = 3 * i + a;
        const double sa = scale[t];
        const double rq = sa * acc[a];              // (S H S v)[t]
        double r = rq;
        if (MODE != 2 && mu > 0.0) {
          const double dg = sa * sa * Hi[a * 3 + a];
          r += fmin(fmax(dg, min_diag), max_diag) / mu * v[t];
        }
        if (MODE == 0) y[t] = r;
        if (kRes) {
          const double b = aux.rhs[t], res = b - r;
          if (MODE == 1) aux.r_out[t] = res;
          part[0] += res * res; part[1] += b * b;
        }
        if (kStep) {
          constexpr int o = (MODE == 3) ? 2 : 0;
          const double vt = v[t];
          const double dl = sa * vt, xt = aux.x[t];
          y[t] = rq;
          part[o + 0] += vt * aux.rhs[t];
          part[o + 1] += vt * rq;
          part[o + 2] += dl * dl; part[o + 3] += xt * xt;
          aux.xc[t] = xt + dl;
          if (!isfinite(vt)) part[o + 4] += 1.0;
        }
      }
    }
  }
  if (MODE != 0) {
    __shared__ double sm[2][7];
    const int wv = threadIdx.x >> 6;
    constexpr int NP = (MODE == 1) ? 2 : (MODE == 2 ? 5 : 7);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const double t = wave_sum(part[k]);
      if ((threadIdx.x & 63) == 0) sm[wv][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < NP) aux.partials[(size_t)blockIdx.x * 8 + (MODE == 2 ? 2 : 0) + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x];
  }
}

int spmv_blocks(int npad) { return ((npad + 2) / 3 + 127) / 128; }

void launch_spmv(hipStream_t st, int n_poses, int npad, const FactorView& fv, const LinView& lin, const double* scale,
                 double mu, double min_diag, double max_diag, const double* v, double* y) {
  SpmvAux aux = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(k_spmv<0>, dim3(spmv_blocks(npad)), dim3(128), 0, st, n_poses, npad, fv, lin, scale, mu, min_diag,
                     max_diag, v, y, aux);
}

void launch_spmv_residual(hipStream_t st, int n_poses, int npad, const FactorView& fv, const LinView& lin,
                          const double* scale, double mu, double min_diag, double max_diag, const double* v,
                          const double* rhs, double* r_out, double* partials) {
  SpmvAux aux = {rhs, r_out, nullptr, nullptr, partials};
  hipLaunchKernelGGL(k_spmv<1>, dim3(spmv_blocks(npad)), dim3(128), 0, st, n_poses, npad, fv, lin, scale, mu, min_diag,
                     max_diag, v, (double*)nullptr, aux);
}

void launch_spmv_residual_step(hipStream_t st, int n_poses, int npad, const FactorView& fv, const LinView& lin, const double* scale, double mu,
                               double min_diag, double max_diag, const double* v, const double* rhs, const double* x, double* q, double* xc,
                               double* partials) {
  SpmvAux aux = {rhs, nullptr, x, xc, partials};
  hipLaunchKernelGGL(k_spmv<3>, dim3(spmv_blocks(npad)), dim3(128), 0, st, n_poses, npad, fv, lin, scale, mu, min_diag, max_diag, v, q, aux);
}

void launch_spmv_step(hipStream_t st, int n_poses, int npad, const FactorView& fv, const LinView& lin,
                      const double* scale, const double* dt, const double* rhs, const double* x, double* q, double* xc,
                      double* partials) {
  SpmvAux aux = {rhs, nullptr, x, xc, partials};
  hipLaunchKernelGGL(k_spmv<2>, dim3(spmv_blocks(npad)), dim3(128), 0, st, n_poses, npad, fv, lin, scale, 0.0, 0.0, 0.0, dt,
                     q, aux);
}

// ===========================================================================================
// small vector kernels
// ===========================================================================================
// submap exchange helpers --------------------------------------------------------------------
// unit right-hand sides for the interface solve: w[r][*] = 0 except w[r][cols[r]] = 1  (r < ncols)
__global__ void k_unit_rhs(int npad, double* __restrict__ w, size_t stride, const int* __restrict__ cols) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npad) return;
  const int r = blockIdx.y;
  w[stride * r + t] = (t == cols[r]) ? 1.0 : 0.0;
}

void launch_unit_rhs(hipStream_t st, int npad, double* w, size_t stride, const int* cols_dev, int ncols) {
  hipLaunchKernelGGL(k_unit_rhs, dim3((npad + 255) / 256, ncols), dim3(256), 0, st, npad, w, stride, cols_dev);
}

// interface rows of z0, of the ncols unit-rhs solutions and of the Jacobi scale:
// out[0..nc) = z0[cols], out[nc + nc*c + r] = Z[c][cols[r]], out[nc + nc*nc + r] = scale[cols[r]]
__global__ void k_gather_iface(const double* __restrict__ z0, const double* __restrict__ Z, size_t stride,
                               const double* __restrict__ scale, const int* __restrict__ cols, int nc,
                               double* __restrict__ out) {
  for (int t = threadIdx.x; t < nc * (nc + 2); t += blockDim.x) {
    if (t < nc) out[t] = z0[cols[t]];
    else if (t < nc + nc * nc) { const int c = (t - nc) / nc, r = (t - nc) % nc; out[t] = Z[stride * c + cols[r]]; }
    else out[t] = scale[cols[t - nc - nc * nc]];
  }
}

void launch_gather_iface(hipStream_t st, const double* z0, const double* Z, size_t stride, const double* scale,
                         const int* cols_dev, int ncols, double* out) {
  hipLaunchKernelGGL(k_gather_iface, dim3(1), dim3(256), 0, st, z0, Z, stride, scale, cols_dev, ncols, out);
}

// dt = z0 - sum_k coef[k] * Z[k]   (Z: ncols columns, `stride` apart; coef in device memory)
__global__ void k_combine(int npad, const double* __restrict__ z0, const double* __restrict__ Z, size_t stride,
                          const double* __restrict__ coef, int ncols, double* __restrict__ dt) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npad) return;
  double v = z0[t];
  for (int k = 0; k < ncols; ++k) v -= coef[k] * Z[stride * k + t];
  dt[t] = v;
}

void launch_combine(hipStream_t st, int npad, const double* z0, const double* Z, size_t stride, const double* coef_dev,
                    int ncols, double* dt) {
  hipLaunchKernelGGL(k_combine, dim3((npad + 255) / 256), dim3(256), 0, st, npad, z0, Z, stride, coef_dev, ncols, dt);
}

// halo poses (constant in the local system) take the neighbouring submap's step:
// dt[idx[k]] = val[k], scale[idx[k]] = sc[k]  (k < cnt)
__global__ void k_scatter_halo(int cnt, const int* __restrict__ idx, const double* __restrict__ val,
                               const double* __restrict__ sc, double* __restrict__ dt, double* __restrict__ scale) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  dt[idx[k]] = val[k];
  scale[idx[k]] = sc[k];
}

void launch_scatter_halo(hipStream_t st, int cnt, const int* idx, const double* val, const double* sc, double* dt,
                         double* scale) {
  if (cnt <= 0) return;
  hipLaunchKernelGGL(k_scatter_halo, dim3((cnt + 63) / 64), dim3(64), 0, st, cnt, idx, val, sc, dt, scale);
}

// candidate state of the halo poses: xc = x + scale * dt  (k_spmv<2> leaves xc = x on constant poses)
__global__ void k_halo_candidate(int cnt, const int* __restrict__ idx, const double* __restrict__ x,
                                 const double* __restrict__ scale, const double* __restrict__ dt, double* __restrict__ xc) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  const int t = idx[k];
  xc[t] = x[t] + scale[t] * dt[t];
}

void launch_halo_candidate(hipStream_t st, int cnt, const int* idx, const double* x, const double* scale, const double* dt,
                           double* xc) {
  if (cnt <= 0) return;
  hipLaunchKernelGGL(k_halo_candidate, dim3((cnt + 63) / 64), dim3(64), 0, st, cnt, idx, x, scale, dt, xc);
}

struct DotArgs { const double* a[4]; const double* b[4]; };

__global__ __launch_bounds__(kRedBlock) void k_dots(int n, int ndots, DotArgs args, double* out) {
  __shared__ double sm[kRedBlock / WAVE];
  for (int k = 0; k < ndots; ++k) {
    double s = 0.0;
    const double* a = args.a[k];
    const double* b = args.b[k];
    for (int t = threadIdx.x; t < n; t += kRedBlock) s += a[t] * b[t];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) out[k] = s;
  }
}

void launch_dots(hipStream_t st, int n, int ndots, const double* const* a, const double* const* b, double* out) {
  DotArgs args;
  for (int k = 0; k < 4; ++k) { args.a[k] = a[k < ndots ? k : 0]; args.b[k] = b[k < ndots ? k : 0]; }
  hipLaunchKernelGGL(k_dots, dim3(1), dim3(kRedBlock), 0, st, n, ndots, args, out);
}

__global__ void k_axpby(int n, double alpha, const double* __restrict__ x, double beta, double* __restrict__ y) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  y[t] = alpha * x[t] + ((beta != 0.0) ? beta * y[t] : 0.0);
}

void launch_axpby(hipStream_t st, int n, double alpha, const double* x, double beta, double* y) {
  hipLaunchKernelGGL(k_axpby, dim3((n + 255) / 256), dim3(256), 0, st, n, alpha, x, beta, y);
}

void launch_copy(hipStream_t st, int n, const double* src, double* dst) {
  (void)hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
}

const char* kernels_build_info() { return "gfx950 " __DATE__ " " __TIME__; }

}  // namespace hitl
