// hitl_kernels.hip -- CDNA4 (gfx950) kernels that linearise hitl-slam's batch re-optimisation problem.
//
// Kernel inventory (SURVEY.md section 2.3 "compute sites" -> here):
//   k_linearize_stf  : K4  PointToPointGlobConstraint residual + analytic 2x6 Jacobian per
//                      directed pose pair (residual_functors.h:786-830).  The HBM-bound kernel.
//   k_stf_moments, k_linearize_stf_moments : the same factors through their sufficient statistics (opt-in).
//   k_assemble       : K5/K6/K7a odometry (residual_functors.h:1081-1125), human rows
//                      (:1299-1415), J^T J 3x3 blocks + gradient, gathered per pose (no atomics,
//                      bit-reproducible).
// The linear algebra lives in hitl_cr.hip (block cyclic reduction, selected inversion) and hitl_vec.hip (band
// products, PCG pieces, submap interface helpers).
// All arithmetic is fp64 (T = double in the reference functors); observations stay fp32.

#include <algorithm>
#include <cstdlib>

#include "hitl_device.h"

namespace hitl {

// ===========================================================================================
// linearize_stf: one workgroup per SOURCE pose (its scan staged once in LDS, already widened to fp64), one 32-lane
// half-wave per directed pair (i -> j), lanes stride the pair's matches two at a time.
//
// All per-match arithmetic is done in the SOURCE pose's frame: with R01 = R0^T R1 and
// t01 = R0^T (t1 - t0),   u = R01 p1,  w = u + t01,  q = w - p0  (= R0^T delta),  n1' = R01 n1,
//   e0/k = n0 . q                      e1/k = n1' . q
//   d(e0/k)/dtheta0 = cross(n0, w)     d(e0/k)/dtheta1 = -cross(n0, u)
//   d(e1/k)/dtheta0 = cross(n1', p0)   d(e1/k)/dtheta1 = cross(n1', t01 - p0)
// (identical to SURVEY.md A.2 after rotating by R0^T; cross(a,b) = ax*by - ay*bx).  30 fp64 operations per match.
//
// Per match the kernel only SUMS: 8 numbers per pair (S0, A0 = sum e0 n0, T01 = sum e0 d01, S1, A1 = sum e1 n1', T10), reduced
// across the half-wave with a reduce-scatter butterfly (7 64-bit swizzles for both pairs of a wave); the 8 lanes that end up
// holding the totals pass them through LDS to lane 0 of the half-wave, which forms the residual, the square roots and the
// 2x6 Jacobian.  (A separate one-thread-per-pair finishing kernel was measured at 21 us: strided 112-byte rows.)
//
// What bounds it (round 2, C4, rocprofv3 kernel trace + PMC, profiles/r02_stf_*): 280 us per launch = 80 us with every
// trip removed (launch, scan staging, the pair prologue/epilogue) + ~35 us per trip.  Removing any ONE ingredient of a trip
// moves it by 5-14 % only (no target gathers 250 us, no index stream 253, no arithmetic 268, no LDS source reads 276);
// 27 % fewer VALU instructions, 5 instead of 6 waves per SIMD, index words 1-3 trips ahead, gathers one trip ahead, several
// poses per workgroup: all within +-3 %.  VALU issue slots ~58 % used, TA ~55 % busy, L1 72 % / L2 75 % hit rates, HBM
// 2.7 TB/s.  The kernel sits on several ceilings at once; the moments form (stf_moments) is the way past them.
// Round-2 instruction diet (rocprofv3 PMC, C4: 845 VALU instructions per wave before):
//  * full trips carry no tail weights; the last partial trip is a separate copy of the body with weights,
//  * the source scan sits in LDS as fp64 (4 conversions per match instead of 8),
//  * next-trip index words are fetched unguarded (the match array carries 64 words of padding),
//  * the 16 cross-lane picks of the per-pair epilogue became one LDS round trip.
//
// blockIdx -> pose mapping keeps consecutive poses on one XCD (blocks b and b+8 share an
// XCD) so the target scans i+-1..W, which the neighbouring workgroups stage as THEIR source,
// are served from that XCD's L2.
// ===========================================================================================
constexpr int kStfBlock = 192;

// value of lane (l ^ M) within each group of 32 lanes: ds_swizzle in bit mode (and 0x1f, or 0, xor M).  Unlike
// ds_bpermute (__shfl_xor) it needs no lane-address register, and this kernel has none to spare at 80 VGPRs.
template <int M>
__device__ __forceinline__ double swz_xor(double v) {
  static_assert(M > 0 && M < 32, "bit-mode swizzle stays inside 32 lanes");
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), (M << 10) | 0x1f);
  const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), (M << 10) | 0x1f);
  return __hiloint2double(hi, lo);
}

// v[0..2h) -> v[0..h): lanes with bit `mask` set keep the upper half
#define HITL_XSTEP(mask, h)                                        \
  {                                                                \
    const bool up = (lane & (mask)) != 0;                          \
    _Pragma("unroll") for (int q_ = 0; q_ < (h); ++q_) {           \
      const double send_ = up ? v[q_] : v[q_ + (h)];               \
      const double keep_ = up ? v[q_ + (h)] : v[q_];               \
      v[q_] = keep_ + swz_xor<(mask)>(send_);                      \
    }                                                              \
  }

constexpr int kStfSub = 32;     // lanes per directed pair: two pairs per wave
constexpr int kStfSums = 8;     // S0, A0x, A0y, T01, S1, A1x, A1y, T10 (of e / k)

// per-pose (cos theta, sin theta), computed once per linearisation instead of per pair
__global__ void k_pose_trig(int n, const double* __restrict__ x, double2* __restrict__ cs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  sincos(x[3 * i + 2], &s, &c);
  cs[i] = make_double2(c, s);
}

struct StfAcc { double S0, A0x, A0y, T01, S1, A1x, A1y, T10; };
struct StfLm { const LmDev* lm; const double* x_alt; double* pairRJ_alt; };

// one match: source point (ax, ay) + normal (anx, any) already fp64, target observation tv fp32
template <bool WEIGHTED>
__device__ __forceinline__ void stf_match(StfAcc& a, double ax, double ay, double anx, double any, const float4 tv, double cR, double sR,
                                          double t01x, double t01y, double wgt) {
  const double bx = tv.x, by = tv.y, bnx = tv.z, bny = tv.w;
  const double ux = cR * bx - sR * by, uy = sR * bx + cR * by;        // u = R01 p1
  const double mx = cR * bnx - sR * bny, my = sR * bnx + cR * bny;    // n1' = R01 n1
  const double qx = (ux + t01x) - ax, qy = (uy + t01y) - ay;          // q = R0^T delta
  double e0 = anx * qx + any * qy;                                    // e0 / k
  double e1 = mx * qx + my * qy;                                      // e1 / k
  if (WEIGHTED) { e0 *= wgt; e1 *= wgt; }
  const double d01 = any * ux - anx * uy;                             // -cross(n0, u)
  const double d10 = mx * ay - my * ax;                               //  cross(n1', p0)
  a.S0 += e0 * e0; a.A0x += e0 * anx; a.A0y += e0 * any; a.T01 += e0 * d01;
  a.S1 += e1 * e1; a.A1x += e1 * mx; a.A1y += e1 * my; a.T10 += e1 * d10;
}

// The same match with w = R01 p1 + t01 formed directly (two fused multiply-adds per component instead of a product and an add):
// two instructions fewer.  u = R01 p1 is then never formed, so the sum that needed it changes: instead of T01 = sum e0 (-cross(n0, u))
// the lane accumulates T00 = sum e0 cross(n0, w), and the pair's tail recovers T01 = cross(A0, t01) - T00 (w = u + t01).
template <bool WEIGHTED>
__device__ __forceinline__ void stf_match_w(StfAcc& a, double ax, double ay, double anx, double any, const float4 tv, double cR, double sR,
                                            double t01x, double t01y, double wgt) {
  const double bx = tv.x, by = tv.y, bnx = tv.z, bny = tv.w;
  const double wx = fma(cR, bx, fma(-sR, by, t01x)), wy = fma(sR, bx, fma(cR, by, t01y));   // w = R01 p1 + t01
  const double mx = cR * bnx - sR * bny, my = sR * bnx + cR * bny;                            // n1' = R01 n1
  const double qx = wx - ax, qy = wy - ay;                                                    // q = R0^T delta
  double e0 = anx * qx + any * qy;                                                            // e0 / k
  double e1 = mx * qx + my * qy;                                                              // e1 / k
  if (WEIGHTED) { e0 *= wgt; e1 *= wgt; }
  const double d00 = anx * wy - any * wx;                                                     //  cross(n0, w)
  const double d10 = mx * ay - my * ax;                                                       //  cross(n1', p0)
  a.S0 += e0 * e0; a.A0x += e0 * anx; a.A0y += e0 * any; a.T01 += e0 * d00;                   // (the T01 slot holds T00 here)
  a.S1 += e1 * e1; a.A1x += e1 * mx; a.A1y += e1 * my; a.T10 += e1 * d10;
}

// SRC: 0 = source scan read from global memory (scans beyond the LDS budget), 1 = staged in LDS as float4 (<= 4096
// points), 2 = staged in LDS widened to fp64 (<= 2048 points)
// OFF32: every observation and every index word lies within 4 GB of its array base, so a gather / index fetch is
// `scalar base + 32-bit lane offset` (global_load with an SGPR base: no 64-bit address arithmetic, no address to keep alive)
// AHEAD: how many trips ahead the index words are requested (1..3).  The index stream comes from HBM (about 1.2 us under
// load, rocprofv3 TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ), a trip lasts about as long: one trip ahead does not hide it.
template <int SRC>
__device__ __forceinline__ void stf_source(const double2* s_scan8, const double2* s_scanN, const float4* s_scan4, const float4* __restrict__ src_g, uint32_t u,
                                           double& ax, double& ay, double& anx, double& any) {
  const int k0 = (int)__builtin_amdgcn_ubfe(u, 0, 16);
  if (SRC == 2) {
    const double2 a = s_scan8[k0], b = s_scanN[k0];   // two planes: a 16-byte lane stride keeps the b128 reads off each other's banks
    ax = a.x; ay = a.y; anx = b.x; any = b.y;
  } else {
    const float4 a = (SRC == 1) ? s_scan4[k0] : src_g[k0];
    ax = a.x; ay = a.y; anx = a.z; any = a.w;
  }
}

template <int SRC, bool OFF32, int AHEAD, int MINB, bool WFOLD = false>
__global__ __launch_bounds__(kStfBlock, MINB) void k_linearize_stf(MapView map, PairView pr, int chunk,
                                                                    const double* __restrict__ x, const double2* __restrict__ cs,
                                                                    double kf, double* __restrict__ pairRJ, StfLm lmsel) {
  if (lmsel.lm != nullptr) {                    // device-driven LM loop: the candidate's buffers (uniform branch, scalar loads)
    if (lmsel.lm->done) return;
    if ((lmsel.lm->cur ^ 1) != 0) { x = lmsel.x_alt; pairRJ = lmsel.pairRJ_alt; }
  }
  __shared__ double s_tot[kStfBlock / kStfSub][kStfSums];           // the 8 totals of each half-wave's pair
  extern __shared__ double s_scan_raw[];
  float4* s_scan4 = reinterpret_cast<float4*>(s_scan_raw);            // SRC == 1
  double2* s_scan8 = reinterpret_cast<double2*>(s_scan_raw);          // SRC == 2: (px, py), (nx, ny) per point
  const int b = blockIdx.x;
  const int idx = b >> 3;
  const int i = (b & 7) * chunk + idx;
  if (idx >= chunk || i >= map.n_poses) return;
  const int p_begin = pr.pose_pair_off[i], p_end = pr.pose_pair_off[i + 1];
  if (p_begin == p_end) return;
  const int64_t oi = map.obs_off[i];
  const int mi = (int)(map.obs_off[i + 1] - oi);
  double2* s_scanN = s_scan8 + mi;                                    // SRC == 2: the (nx, ny) plane behind the (px, py) plane
  const float4* __restrict__ src_g = map.obs + oi;
  const int lane = threadIdx.x & 63;
  const int sub = threadIdx.x / kStfSub;        // pair slot of this half-wave within the workgroup
  const int hl = threadIdx.x & (kStfSub - 1);   // lane within the half-wave
  const uint32_t* __restrict__ mt = pr.matches;
  const char* __restrict__ mtb = reinterpret_cast<const char*>(mt);
  const char* __restrict__ obs_b = reinterpret_cast<const char*>(map.obs);
  constexpr int kSlots = kStfBlock / kStfSub;
  // A workgroup's life is a chain of dependent memory rounds (pose -> pair -> target pose / index words -> gathers), and
  // the machine holds only 8 workgroups per CU: every round taken off the chain is throughput.  Round 1 of the first
  // pair round (target pose, match range) is requested here, together with the scan; round 2 (target state, first index
  // words) right behind the scan's LDS stores, BEFORE the barrier -- not after it.
  int p = p_begin + sub;
  bool live = p < p_end;
  int pc = live ? p : p_begin;
  int j = pr.pair_tgt[pc];
  int64_t m_begin = pr.match_off[pc];
  int64_t m_end = pr.match_off[pc + 1];
  if (SRC == 1) {
    for (int k = threadIdx.x; k < mi; k += kStfBlock) s_scan4[k] = src_g[k];
  } else if (SRC == 2) {
    for (int k = threadIdx.x; k < mi; k += kStfBlock) {
      const float4 o = src_g[k];
      s_scan8[k] = make_double2((double)o.x, (double)o.y);
      s_scanN[k] = make_double2((double)o.z, (double)o.w);
    }
  }
  // workgroup-uniform values pinned to SGPRs
  const double xi = uniform_f64(x[3 * i]), yi = uniform_f64(x[3 * i + 1]);
  const double2 cs0 = cs[i];
  const double c0 = uniform_f64(cs0.x), s0 = uniform_f64(cs0.y);
  bool staged = (SRC == 0);
  for (int pb = p_begin;;) {                            // workgroup-uniform trip count
    const int64_t oj = map.obs_off[j];
    const float4* __restrict__ tgt_g = map.obs + oj;
    const uint32_t tgt_b = (uint32_t)oj << 4;                                   // OFF32: byte offset of the target scan
#define HITL_TGT(u) (OFF32 ? *reinterpret_cast<const float4*>(obs_b + (tgt_b + (__builtin_amdgcn_ubfe((u), 16, 16) << 4))) : tgt_g[(u) >> 16])
    const double xj = x[3 * j], yj = x[3 * j + 1];
    const double2 cs1 = cs[j];
    const int len = live ? (int)(m_end - m_begin) : 0;                          // a pair holds < 65536 matches
    // index words, AHEAD trips in flight.  Unguarded: the array ends with kMatchPad words of padding.
    const uint32_t* __restrict__ mp = mt + m_begin + hl;
    uint32_t mo = ((uint32_t)m_begin + (uint32_t)hl) << 2;
#define HITL_IDX(k) (OFF32 ? *reinterpret_cast<const uint32_t*>(mtb + (mo + 4u * (uint32_t)(k))) : mp[(k)])
    uint32_t ua[AHEAD], ub[AHEAD];
#pragma unroll
    for (int d = 0; d < AHEAD; ++d) { ua[d] = HITL_IDX(2 * kStfSub * d); ub[d] = HITL_IDX(2 * kStfSub * d + kStfSub); }
    if (!staged) { __syncthreads(); staged = true; }    // the source scan is in LDS
    const double ddx = xj - xi, ddy = yj - yi;
    const double cR = c0 * cs1.x + s0 * cs1.y, sR = c0 * cs1.y - s0 * cs1.x;   // R01 = R0^T R1
    const double t01x = c0 * ddx + s0 * ddy, t01y = c0 * ddy - s0 * ddx;       // R0^T (t1 - t0)
    StfAcc acc = {0, 0, 0, 0, 0, 0, 0, 0};
    const int nfull = len >> 6, rem = len & 63;
    double ax0, ay0, anx0, any0, ax1, ay1, anx1, any1;
    for (int t = 0; t < nfull; t += AHEAD) {
#pragma unroll
      for (int d = 0; d < AHEAD; ++d) {
        if (t + d < nfull) {
          const uint32_t u0 = ua[d], u1 = ub[d];
          const float4 tv0 = HITL_TGT(u0), tv1 = HITL_TGT(u1);
          stf_source<SRC>(s_scan8, s_scanN, s_scan4, src_g, u0, ax0, ay0, anx0, any0);
          stf_source<SRC>(s_scan8, s_scanN, s_scan4, src_g, u1, ax1, ay1, anx1, any1);
          ua[d] = HITL_IDX(2 * kStfSub * (d + AHEAD)); ub[d] = HITL_IDX(2 * kStfSub * (d + AHEAD) + kStfSub);   // trip t + d + AHEAD
          __builtin_amdgcn_sched_barrier(0);      // keep the index prefetch ahead of the arithmetic (the scheduler sinks it otherwise)
          if (WFOLD) {
            stf_match_w<false>(acc, ax0, ay0, anx0, any0, tv0, cR, sR, t01x, t01y, 1.0);
            stf_match_w<false>(acc, ax1, ay1, anx1, any1, tv1, cR, sR, t01x, t01y, 1.0);
          } else {
            stf_match<false>(acc, ax0, ay0, anx0, any0, tv0, cR, sR, t01x, t01y, 1.0);
            stf_match<false>(acc, ax1, ay1, anx1, any1, tv1, cR, sR, t01x, t01y, 1.0);
          }
        }
      }
      if (OFF32) mo += 8u * kStfSub * AHEAD; else mp += 2 * kStfSub * AHEAD;
    }
    if (rem > 0) {      // the last, partial trip: lanes past the end take index 0 (valid) with weight 0
      uint32_t w0 = ua[0], w1 = ub[0];
#pragma unroll
      for (int d = 1; d < AHEAD; ++d) { const bool here = (nfull % AHEAD) == d; w0 = here ? ua[d] : w0; w1 = here ? ub[d] : w1; }
      const bool v0 = hl < rem, v1 = hl + kStfSub < rem;
      const uint32_t u0 = v0 ? w0 : 0u, u1 = v1 ? w1 : 0u;
      const float4 tv0 = HITL_TGT(u0), tv1 = HITL_TGT(u1);
      stf_source<SRC>(s_scan8, s_scanN, s_scan4, src_g, u0, ax0, ay0, anx0, any0);
      stf_source<SRC>(s_scan8, s_scanN, s_scan4, src_g, u1, ax1, ay1, anx1, any1);
      if (WFOLD) {
        stf_match_w<true>(acc, ax0, ay0, anx0, any0, tv0, cR, sR, t01x, t01y, v0 ? 1.0 : 0.0);
        stf_match_w<true>(acc, ax1, ay1, anx1, any1, tv1, cR, sR, t01x, t01y, v1 ? 1.0 : 0.0);
      } else {
        stf_match<true>(acc, ax0, ay0, anx0, any0, tv0, cR, sR, t01x, t01y, v0 ? 1.0 : 0.0);
        stf_match<true>(acc, ax1, ay1, anx1, any1, tv1, cR, sR, t01x, t01y, v1 ? 1.0 : 0.0);
      }
    }
    // reduce-scatter over the 32 lanes of this pair: after the three exchange steps lane hl holds the 4-lane partial of
    // value (hl >> 2); two more exchanges complete it
    double v[kStfSums] = {acc.S0, acc.A0x, acc.A0y, acc.T01, acc.S1, acc.A1x, acc.A1y, acc.T10};
    HITL_XSTEP(16, 4)
    HITL_XSTEP(8, 2)
    HITL_XSTEP(4, 1)
    double tot = v[0];
    tot += swz_xor<2>(tot);
    tot += swz_xor<1>(tot);
    // lane hl = 4 id now holds total id: through LDS to lane 0 of the half-wave (the kernel is not VALU-bound, and no
    // lane-address register is needed this way), which finishes the pair: r = k sqrt(S/n) (residual_functors.h:822-827),
    // dr/dq = k (sum e' de') / (n sqrt(S/n)); the two remaining theta partials follow from the sums:
    //   sum e0 cross(n0, w) = cross(A0, t01) - T01 (w = u + t01),   sum e1 cross(n1', t01 - p0) = cross(A1, t01) - T10
    if ((hl & 3) == 0) s_tot[sub][hl >> 2] = tot;
    if (hl == 0 && live) {          // same wave: the LDS writes above are complete before these reads (in-order LDS queue)
      const double S0 = s_tot[sub][0], A0x = s_tot[sub][1], A0y = s_tot[sub][2], Tq = s_tot[sub][3];
      const double S1 = s_tot[sub][4], A1x = s_tot[sub][5], A1y = s_tot[sub][6], T10 = s_tot[sub][7];
      const double Tr = (A0x * t01y - A0y * t01x) - Tq;            // WFOLD: the lanes summed T00 (stf_match_w), T01 follows; else the other way round
      const double T00 = WFOLD ? Tq : Tr, T01 = WFOLD ? Tr : Tq;
      const double T11 = (A1x * t01y - A1y * t01x) - T10;
      const double inv_n = 1.0 / (double)len;
      const double a0 = S0 * inv_n, a1 = S1 * inv_n;
      const double i0 = (S0 != 0.0) ? rsqrt(a0) : 0.0, i1 = (S1 != 0.0) ? rsqrt(a1) : 0.0;
      const double f0 = kf * inv_n * i0, f1 = kf * inv_n * i1;
      const double g0x = c0 * A0x - s0 * A0y, g0y = s0 * A0x + c0 * A0y;   // R0 * sum e0' n0
      const double g1x = c0 * A1x - s0 * A1y, g1y = s0 * A1x + c0 * A1y;   // R0 * sum e1' n1'
      double* o = pairRJ + (size_t)kPairRJ * p;
      o[0] = kf * a0 * i0; o[1] = kf * a1 * i1;
      o[2] = -f0 * g0x; o[3] = -f0 * g0y; o[4] = f0 * T00; o[5] = f0 * g0x; o[6] = f0 * g0y; o[7] = f0 * T01;
      o[8] = -f1 * g1x; o[9] = -f1 * g1y; o[10] = f1 * T10; o[11] = f1 * g1x; o[12] = f1 * g1y; o[13] = f1 * T11;
    }
#undef HITL_TGT
#undef HITL_IDX
    pb += kSlots;
    if (pb >= p_end) break;
    p = pb + sub; live = p < p_end; pc = live ? p : p_begin;
    j = pr.pair_tgt[pc]; m_begin = pr.match_off[pc]; m_end = pr.match_off[pc + 1];
  }
}

// ===========================================================================================
// k_linearize_stf_r (round 4, VERDICT r3 lever (a)): the half-wave kernel with a PER-POSE DESCRIPTOR RECORD.
// A workgroup's life starts with a chain of dependent loads: pose -> (pair range, scan range) -> per pair (target pose, match
// range) -> (target scan offset, target state, first index words) -> gathers: four rounds of ~1 us under load before the first
// trip of 50 000 workgroups.  The record (128 B = one cache line per pose, written once per correspondence set by
// k_stf_pose_recs) holds what the first two rounds fetch -- pair range, scan range and, for the pose's first six pairs, target
// pose, target scan byte offset, first match and match count -- so the chain is record -> (states, index words, scan) ->
// gathers.  Pairs beyond the sixth of a pose (wider bands, all-pairs sets) take the old loads.
// ===========================================================================================
struct alignas(16) StfPoseSlot { uint32_t tgt_b; uint32_t m_begin; int len; int j; };     // OFF32: both offsets fit 32 bits
struct alignas(128) StfPoseRec { int p_begin, n_pairs, mi, pad0; int64_t src_off; int64_t pad1; StfPoseSlot s[kStfBlock / kStfSub]; };
static_assert(sizeof(StfPoseRec) == 128, "one cache line per pose");

__global__ void k_stf_pose_recs(MapView map, PairView pr, StfPoseRec* __restrict__ recs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= map.n_poses) return;
  StfPoseRec r;
  const int p0 = pr.pose_pair_off[i], p1 = pr.pose_pair_off[i + 1];
  const int64_t oi = map.obs_off[i];
  r.p_begin = p0; r.n_pairs = p1 - p0; r.mi = (int)(map.obs_off[i + 1] - oi); r.pad0 = 0; r.src_off = oi; r.pad1 = 0;
  for (int w = 0; w < kStfBlock / kStfSub; ++w) {
    const int p = (p0 + w < p1) ? p0 + w : p0;                      // slots without a pair repeat the first one with length 0
    StfPoseSlot q;
    if (p0 < p1) {
      q.j = pr.pair_tgt[p]; q.tgt_b = (uint32_t)(map.obs_off[q.j] << 4); q.m_begin = (uint32_t)pr.match_off[p];
      q.len = (p0 + w < p1) ? (int)(pr.match_off[p + 1] - pr.match_off[p]) : 0;
    } else { q.j = i; q.tgt_b = (uint32_t)(oi << 4); q.m_begin = 0; q.len = 0; }
    r.s[w] = q;
  }
  recs[i] = r;
}
void launch_build_stf_pose_recs(hipStream_t st, const MapView& map, const PairView& pr, void* recs) {
  hipLaunchKernelGGL(k_stf_pose_recs, dim3((map.n_poses + 127) / 128), dim3(128), 0, st, map, pr, reinterpret_cast<StfPoseRec*>(recs));
}
size_t stf_pose_rec_bytes() { return sizeof(StfPoseRec); }

// ABL = 1 (test hook stf_lds_targets; WRONG NUMBERS ON PURPOSE, a timing experiment: profiles/r06_stf_counters.md): every target point is
// read from an fp32 copy of the SOURCE scan in LDS (one ds_read_b128 at the target's index) instead of the 16-byte L1 gather -- what a
// kernel with its target scans resident in LDS would issue per match, without any of the staging such a kernel has to pay for.
template <int MINB, int ABL = 0>
__global__ __launch_bounds__(kStfBlock, MINB) void k_linearize_stf_r(MapView map, PairView pr, const StfPoseRec* __restrict__ recs, int chunk,
                                                                     const double* __restrict__ x, const double2* __restrict__ cs, double kf,
                                                                     double* __restrict__ pairRJ, StfLm lmsel) {
  if (lmsel.lm != nullptr) {                    // device-driven LM loop: the candidate's buffers (uniform branch, scalar loads)
    if (lmsel.lm->done) return;
    if ((lmsel.lm->cur ^ 1) != 0) { x = lmsel.x_alt; pairRJ = lmsel.pairRJ_alt; }
  }
  __shared__ double s_tot[kStfBlock / kStfSub][kStfSums];           // the 8 totals of each half-wave's pair
  extern __shared__ double s_scan_raw_r[];
  double2* s_scan8 = reinterpret_cast<double2*>(s_scan_raw_r);        // (px, py) plane, then the (nx, ny) plane
  const int b = blockIdx.x;
  const int idx = b >> 3;
  const int i = (b & 7) * chunk + idx;
  if (idx >= chunk || i >= map.n_poses) return;
  const StfPoseRec* __restrict__ rec = recs + i;
  const int p_begin = rec->p_begin, n_pairs = rec->n_pairs;           // scalar loads: the whole header is one 32-byte read
  if (n_pairs == 0) return;
  const int p_end = p_begin + n_pairs;
  const int mi = rec->mi;
  const int64_t oi = rec->src_off;
  double2* s_scanN = s_scan8 + mi;
  float4* s_scanF = reinterpret_cast<float4*>(s_scanN + mi);           // ABL only
  const float4* __restrict__ src_g = map.obs + oi;
  const int lane = threadIdx.x & 63;
  const int sub = threadIdx.x / kStfSub;        // pair slot of this half-wave within the workgroup
  const int hl = threadIdx.x & (kStfSub - 1);   // lane within the half-wave
  const uint32_t* __restrict__ mt = pr.matches;
  const char* __restrict__ mtb = reinterpret_cast<const char*>(mt);
  const char* __restrict__ obs_b = reinterpret_cast<const char*>(map.obs);
  constexpr int kSlots = kStfBlock / kStfSub;
  // round 1 of this half-wave's pair: one 16-byte read of the record
  const StfPoseSlot sl = rec->s[sub];
  int p = p_begin + sub;
  bool live = p < p_end;
  int j = sl.j;
  uint32_t tgt_b = sl.tgt_b;
  uint32_t m_begin32 = sl.m_begin;
  int len = sl.len;
  for (int k = threadIdx.x; k < mi; k += kStfBlock) {
    const float4 o = src_g[k];
    s_scan8[k] = make_double2((double)o.x, (double)o.y);
    s_scanN[k] = make_double2((double)o.z, (double)o.w);
    if (ABL) s_scanF[k] = o;
  }
  const double xi = uniform_f64(x[3 * i]), yi = uniform_f64(x[3 * i + 1]);
  const double2 cs0 = cs[i];
  const double c0 = uniform_f64(cs0.x), s0 = uniform_f64(cs0.y);
  bool staged = false;
  for (int pb = p_begin;;) {                            // workgroup-uniform trip count
#define HITL_RTGT(u) (ABL ? s_scanF[min((int)__builtin_amdgcn_ubfe((u), 16, 16), mi - 1)] \
                          : *reinterpret_cast<const float4*>(obs_b + (tgt_b + (__builtin_amdgcn_ubfe((u), 16, 16) << 4))))
    const double xj = x[3 * j], yj = x[3 * j + 1];
    const double2 cs1 = cs[j];
    uint32_t mo = (m_begin32 + (uint32_t)hl) << 2;
#define HITL_RIDX(k) (*reinterpret_cast<const uint32_t*>(mtb + (mo + 4u * (uint32_t)(k))))
    uint32_t ua = HITL_RIDX(0), ub = HITL_RIDX(kStfSub);
    if (!staged) { __syncthreads(); staged = true; }    // the source scan is in LDS
    const double ddx = xj - xi, ddy = yj - yi;
    const double cR = c0 * cs1.x + s0 * cs1.y, sR = c0 * cs1.y - s0 * cs1.x;   // R01 = R0^T R1
    const double t01x = c0 * ddx + s0 * ddy, t01y = c0 * ddy - s0 * ddx;       // R0^T (t1 - t0)
    StfAcc acc = {0, 0, 0, 0, 0, 0, 0, 0};
    const int nfull = len >> 6, rem = len & 63;
    double ax0, ay0, anx0, any0, ax1, ay1, anx1, any1;
    for (int t = 0; t < nfull; ++t) {
      const uint32_t u0 = ua, u1 = ub;
      const float4 tv0 = HITL_RTGT(u0), tv1 = HITL_RTGT(u1);
      stf_source<2>(s_scan8, s_scanN, nullptr, src_g, u0, ax0, ay0, anx0, any0);
      stf_source<2>(s_scan8, s_scanN, nullptr, src_g, u1, ax1, ay1, anx1, any1);
      ua = HITL_RIDX(2 * kStfSub); ub = HITL_RIDX(3 * kStfSub);                 // trip t + 1
      __builtin_amdgcn_sched_barrier(0);      // keep the index prefetch ahead of the arithmetic (the scheduler sinks it otherwise)
      stf_match_w<false>(acc, ax0, ay0, anx0, any0, tv0, cR, sR, t01x, t01y, 1.0);
      stf_match_w<false>(acc, ax1, ay1, anx1, any1, tv1, cR, sR, t01x, t01y, 1.0);
      mo += 8u * kStfSub;
    }
    if (rem > 0) {      // the last, partial trip: lanes past the end take index 0 (valid) with weight 0
      const bool v0 = hl < rem, v1 = hl + kStfSub < rem;
      const uint32_t u0 = v0 ? ua : 0u, u1 = v1 ? ub : 0u;
      const float4 tv0 = HITL_RTGT(u0), tv1 = HITL_RTGT(u1);
      stf_source<2>(s_scan8, s_scanN, nullptr, src_g, u0, ax0, ay0, anx0, any0);
      stf_source<2>(s_scan8, s_scanN, nullptr, src_g, u1, ax1, ay1, anx1, any1);
      stf_match_w<true>(acc, ax0, ay0, anx0, any0, tv0, cR, sR, t01x, t01y, v0 ? 1.0 : 0.0);
      stf_match_w<true>(acc, ax1, ay1, anx1, any1, tv1, cR, sR, t01x, t01y, v1 ? 1.0 : 0.0);
    }
    double v[kStfSums] = {acc.S0, acc.A0x, acc.A0y, acc.T01, acc.S1, acc.A1x, acc.A1y, acc.T10};
    HITL_XSTEP(16, 4)
    HITL_XSTEP(8, 2)
    HITL_XSTEP(4, 1)
    double tot = v[0];
    tot += swz_xor<2>(tot);
    tot += swz_xor<1>(tot);
    if ((hl & 3) == 0) s_tot[sub][hl >> 2] = tot;
    if (hl == 0 && live) {          // same wave: the LDS writes above are complete before these reads (in-order LDS queue)
      const double S0 = s_tot[sub][0], A0x = s_tot[sub][1], A0y = s_tot[sub][2], T00 = s_tot[sub][3];
      const double S1 = s_tot[sub][4], A1x = s_tot[sub][5], A1y = s_tot[sub][6], T10 = s_tot[sub][7];
      const double T01 = (A0x * t01y - A0y * t01x) - T00;          // the lanes summed T00 (stf_match_w)
      const double T11 = (A1x * t01y - A1y * t01x) - T10;
      const double inv_n = 1.0 / (double)len;
      const double a0 = S0 * inv_n, a1 = S1 * inv_n;
      const double i0 = (S0 != 0.0) ? rsqrt(a0) : 0.0, i1 = (S1 != 0.0) ? rsqrt(a1) : 0.0;
      const double f0 = kf * inv_n * i0, f1 = kf * inv_n * i1;
      const double g0x = c0 * A0x - s0 * A0y, g0y = s0 * A0x + c0 * A0y;   // R0 * sum e0' n0
      const double g1x = c0 * A1x - s0 * A1y, g1y = s0 * A1x + c0 * A1y;   // R0 * sum e1' n1'
      double* o = pairRJ + (size_t)kPairRJ * p;
      o[0] = kf * a0 * i0; o[1] = kf * a1 * i1;
      o[2] = -f0 * g0x; o[3] = -f0 * g0y; o[4] = f0 * T00; o[5] = f0 * g0x; o[6] = f0 * g0y; o[7] = f0 * T01;
      o[8] = -f1 * g1x; o[9] = -f1 * g1y; o[10] = f1 * T10; o[11] = f1 * g1x; o[12] = f1 * g1y; o[13] = f1 * T11;
    }
#undef HITL_RTGT
#undef HITL_RIDX
    pb += kSlots;
    if (pb >= p_end) break;
    // pairs beyond the record's six: the old dependent loads (wider bands, all-pairs sets)
    p = pb + sub; live = p < p_end;
    const int pc = live ? p : p_begin;
    j = pr.pair_tgt[pc];
    tgt_b = (uint32_t)map.obs_off[j] << 4;
    const int64_t mb64 = pr.match_off[pc];
    m_begin32 = (uint32_t)mb64;
    len = live ? (int)(pr.match_off[pc + 1] - mb64) : 0;
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
                                                               double* __restrict__ pairRJ, StfLm lmsel) {
  if (lmsel.lm != nullptr) {                    // device-driven LM loop: the candidate's buffers
    if (lmsel.lm->done) return;
    if ((lmsel.lm->cur ^ 1) != 0) { x = lmsel.x_alt; pairRJ = lmsel.pairRJ_alt; }
  }
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
                                  double sd, double* pairRJ, const LmDev* lm, const double* x_alt, double* pairRJ_alt) {
  if (pr.n_pairs == 0) return;
  const StfLm lmsel = {lm, x_alt, pairRJ_alt};
  hipLaunchKernelGGL(k_linearize_stf_moments, dim3((pr.n_pairs + 127) / 128), dim3(128), 0, st, pr, pairM, x, cs, corr, sd, pairRJ, lmsel);
}

void launch_pose_trig(hipStream_t st, int n_poses, const double* x, double2* cs) {
  hipLaunchKernelGGL(k_pose_trig, dim3((n_poses + 255) / 256), dim3(256), 0, st, n_poses, x, cs);
}

void launch_linearize_stf(hipStream_t st, const MapView& map, const PairView& pr, int max_scan, int64_t map_obs_count, int64_t n_matches, const double* x,
                          const double2* cs, float corr, float std_dev, double* pairRJ, const LmDev* lm, const double* x_alt, double* pairRJ_alt,
                          const void* pose_recs) {
  const StfLm lmsel = {lm, x_alt, pairRJ_alt};
  if (pr.n_pairs == 0) return;
  const int chunk = (map.n_poses + 7) / 8;
  const int cap = (max_scan + 63) & ~63;
  // 32-bit byte offsets into the scan and match arrays while both stay below 4 GB (test switch stf_addr64: the 64-bit
  // addressing instantiations on a small map)
  const bool off32 = (uint64_t)map_obs_count * 16 < ((uint64_t)1 << 32) && ((uint64_t)n_matches + kMatchPad) * 4 < ((uint64_t)1 << 32) &&
                     !dev_switches().stf_addr64;
  const double kf = (double)corr / (double)std_dev;                       // k = corr / sigma, divided here once
  const dim3 grid(8 * chunk), block(kStfBlock);
  const int src = ((size_t)cap * 32 <= 64 * 1024) ? 2 : ((size_t)cap * 16 <= 64 * 1024) ? 1 : 0;
  if (src == 2 && off32 && pose_recs != nullptr && dev_switches().stf_lds_targets && (size_t)cap * 48 <= 64 * 1024) {      // timing experiment (see the kernel)
    hipLaunchKernelGGL((k_linearize_stf_r<7, 1>), grid, block, (size_t)cap * 48, st, map, pr, reinterpret_cast<const StfPoseRec*>(pose_recs), chunk, x, cs, kf,
                       pairRJ, lmsel);
    return;
  }
  if (src == 2 && off32 && pose_recs != nullptr) {      // the shipped kernel: half-waves behind a per-pose descriptor record
    hipLaunchKernelGGL((k_linearize_stf_r<7>), grid, block, (size_t)cap * 32, st, map, pr, reinterpret_cast<const StfPoseRec*>(pose_recs), chunk, x, cs, kf,
                       pairRJ, lmsel);
    return;
  }
  // the general layouts: no record, scans beyond the fp64 / fp32 LDS budget, arrays beyond 4 GB (> 80 VGPRs: 5 waves per SIMD instead of spilling)
  const size_t lds = (src == 2) ? (size_t)cap * 32 : (src == 1) ? (size_t)cap * 16 : 0;
#define HITL_STF_GO(SRC, O32, MINB, WF) hipLaunchKernelGGL((k_linearize_stf<SRC, O32, 1, MINB, WF>), grid, block, lds, st, map, pr, chunk, x, cs, kf, pairRJ, lmsel)
  if (src == 2) { if (off32) HITL_STF_GO(2, true, 7, true); else HITL_STF_GO(2, false, 5, false); }
  else if (src == 1) { if (off32) HITL_STF_GO(1, true, 5, false); else HITL_STF_GO(1, false, 5, false); }
  else { if (off32) HITL_STF_GO(0, true, 5, false); else HITL_STF_GO(0, false, 5, false); }
#undef HITL_STF_GO
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
struct AsmLm { const LmDev* lm; const double* x_alt; LinView lin_alt; };
__global__ __launch_bounds__(kAsmBlock) void k_assemble(MapView map, PairView pr, FactorView fv, const double* __restrict__ x,
                                                         LinView lin, AsmLm lmsel, LmDecide dec) {
  if (lmsel.lm != nullptr) {                    // device-driven LM loop: the candidate's buffers
    if (lmsel.lm->done) return;
    if ((lmsel.lm->cur ^ 1) != 0) { x = lmsel.x_alt; lin = lmsel.lin_alt; }
  }
  const int i = (blockIdx.x * kAsmBlock + (int)threadIdx.x) >> 2;
  const int r = threadIdx.x & 3;
  const int n = map.n_poses;
  const bool act = i < n;
  const int Wb = fv.band_poses;
  double Hd[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double gi[3] = {0, 0, 0};
  double H1[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // odometry part of block (i, i+1), produced by lane 1
  double cost = 0.0;
  // the band block (i, i + d) of this lane: its two pair indices are requested first, their memory round trip runs under the
  // incidence loop instead of after it
  const int d = r + 1;
  int fwd = -1, rev = -1;
  if (act && fv.with_stf && d <= Wb && i + d < n) {
    fwd = fv.band_pair[((size_t)i * Wb + (d - 1)) * 2];
    rev = fv.band_pair[((size_t)i * Wb + (d - 1)) * 2 + 1];
  }
  if (act && fv.with_stf) {
    // incidence entries e = r (mod 4), four at a time: the entry words of a batch are loaded together, then the pair records
    // of the batch, then the sums in entry order (the order of the one-entry-at-a-time loop: same bits)
    const int e1 = fv.inc_off[i + 1];
    for (int eb = fv.inc_off[i] + r; eb < e1; eb += 16) {
      int ent[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) ent[k] = (eb + 4 * k < e1) ? fv.inc_ent[eb + 4 * k] : -1;
      double rj[4][8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (ent[k] >= 0) {
          const double* RJ = lin.pairRJ + (size_t)kPairRJ * (ent[k] >> 1);
          const int co = 2 + 3 * (ent[k] & 1);
          rj[k][0] = RJ[0]; rj[k][1] = RJ[1];
#pragma unroll
          for (int a = 0; a < 3; ++a) { rj[k][2 + a] = RJ[co + a]; rj[k][5 + a] = RJ[6 + co + a]; }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (ent[k] >= 0) {
          const double r0 = rj[k][0], r1 = rj[k][1];
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const double ja0 = rj[k][2 + a], ja1 = rj[k][5 + a];
            gi[a] += ja0 * r0 + ja1 * r1;
#pragma unroll
            for (int c = 0; c < 3; ++c) Hd[a * 3 + c] += ja0 * rj[k][2 + c] + ja1 * rj[k][5 + c];
          }
          if ((ent[k] & 1) == 0) cost += 0.5 * (r0 * r0 + r1 * r1);      // a pair's cost is counted at its source pose
        }
      }
    }
  }
  if (act && fv.with_odom && r < 2) {
    // lane 0: edge (i-1, i), this pose is pose2 (columns 3..5); lane 1: edge (i, i+1), this pose is pose1 (columns 0..2).
    // ONE call site: both lanes evaluate their edge together (two call sites run one after the other in a diverged wave,
    // and the functor's sincos / atan2 / divisions are most of this kernel's instructions)
    const int ea = i - 1 + r;
    if (ea >= 0 && ea + 1 < n) {
      OdomEval o;
      eval_odom(x + 3 * ea, x + 3 * (ea + 1), fv.odom_A[ea + 1], fv.odom_rr[ea + 1], fv.sig_r, fv.sig_t, fv.sig_a, o);
      const bool second = (r == 0);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const double ja = second ? o.J[m * 6 + 3 + a] : o.J[m * 6 + a];
          gi[a] += ja * o.r[m];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            Hd[a * 3 + c] += ja * (second ? o.J[m * 6 + 3 + c] : o.J[m * 6 + c]);
            if (!second) H1[a * 3 + c] += ja * o.J[m * 6 + 3 + c];
          }
        }
      }
      if (second) cost += 0.5 * (o.r[0] * o.r[0] + o.r[1] * o.r[1] + o.r[2] * o.r[2]);
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
    if (d <= Wb) {                   // this lane's band block (i, i+d)
      double blk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (i + d < n) {
        if (fv.with_stf) {
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
    if (dec.lm == nullptr && dec.host_flag == nullptr) {
      lin.partials[2 * blockIdx.x] = cs;
      lin.partials[2 * blockIdx.x + 1] = gs;
    } else if (dec.lm == nullptr) {
      // host-driven loop: the partials go to pinned host memory as SYSTEM-scope stores (written through every cache level on
      // their way out: nothing of them may linger in this XCD's L2 when another XCD's workgroup raises the flag); once they have
      // been accepted this workgroup counts itself in, and the last one to do so raises the flag the host polls (same order on
      // the link: partials, then flag)
      __hip_atomic_store(&lin.partials[2 * blockIdx.x], cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&lin.partials[2 * blockIdx.x + 1], gs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __builtin_amdgcn_s_waitcnt(0);
      if (__hip_atomic_fetch_add(dec.arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        *dec.arrivals = 0u;
        __threadfence_system();
        __hip_atomic_store(dec.host_flag, dec.flag_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    } else {
      // device-driven loop: the two numbers go out as device-scope stores (written through to where every XCD sees them) and have
      // arrived there before this workgroup counts itself in -- NOT a release fence: at device scope that writes back the whole
      // L2 slice, 14 MB of band blocks 782 times over (measured: +115 us per LM iteration)
      __hip_atomic_store(&lin.partials[2 * blockIdx.x], cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&lin.partials[2 * blockIdx.x + 1], gs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_waitcnt(0);
    }
  }
  if (dec.lm == nullptr) return;
  // ---- device-driven LM loop: the last workgroup to get here takes the iteration's decision (round 5; rounds 3-4: a launch of
  //      its own, k_lm_decide, that cost what the host's turn-around did).  Nine sums over the per-workgroup partials (7 of the band
  //      pass, cost and max |g| of this linearisation), each in the fixed order of ordered_sum32 (hitl_host.h): 32 lanes take the
  //      terms b = lane, lane + 32, ... in ascending order, then the 32 lane sums are added from lane 0 up. ----
  __shared__ int s_last;
  __shared__ double s_part[9][32];
  __shared__ double s_tot[9];
  if (threadIdx.x == 0) s_last = (__hip_atomic_fetch_add(dec.arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  {
    const int tid = threadIdx.x, j = tid & 31;
    const double* lp = (lmsel.lm->cur ^ 1) ? dec.lin_partials1 : dec.lin_partials0;      // the candidate's linearisation = this launch's
    for (int grp = tid >> 5; grp < 9; grp += kAsmBlock / 32) {
      double acc = 0.0;
      if (grp < 7) {
#pragma unroll 8
        for (int b = j; b < dec.nblk_step; b += 32) acc += dec.step_partials[8 * b + grp];
      } else if (grp == 7) {
#pragma unroll 8
        for (int b = j; b < dec.nblk_lin; b += 32) acc += __hip_atomic_load(&lp[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
#pragma unroll 8
        for (int b = j; b < dec.nblk_lin; b += 32) {
          const double g = __hip_atomic_load(&lp[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (g > acc || g != g) acc = g;
        }
      }
      s_part[grp][j] = acc;
    }
    __syncthreads();
    if (tid < 9) {
      double t = 0.0;
      if (tid != 8) { for (int q = 0; q < 32; ++q) t += s_part[tid][q]; }
      else { for (int q = 0; q < 32; ++q) { const double g = s_part[8][q]; if (g > t || g != g) t = g; } }
      s_tot[tid] = t;
    }
    __syncthreads();
    if (tid != 0) return;
    LmDev s = *dec.lm;
    lm_decide_step(s, dec.o, s_tot[0], s_tot[1], s_tot[2], s_tot[3], s_tot[4], s_tot[5], s_tot[6], s_tot[7], s_tot[8]);
    *dec.lm = s;
    *dec.arrivals = 0u;                                          // every workgroup of this launch has arrived: ready for the next one
    volatile double* progress = dec.progress;
    progress[1] = (double)s.done; progress[2] = (double)s.term; progress[3] = s.cost; progress[4] = s.radius;
    __threadfence_system();
    progress[0] = (double)s.iter;                                // last: the host polls this word
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
                     const LinView& lin, const LmDev* lm, const double* x_alt, const LinView* lin_alt, const LmDecide* decide) {
  const int n = map.n_poses;
  AsmLm lmsel = {lm, x_alt, lin_alt ? *lin_alt : lin};
  LmDecide dec = {};
  if (decide != nullptr && (lm != nullptr || (decide->lm == nullptr && decide->host_flag != nullptr))) dec = *decide;
  if (fv.with_stf && fv.n_offband > 0) dec.host_flag = nullptr;   // (k_assemble_offband follows: the caller keeps its own flag)
  hipLaunchKernelGGL(k_assemble, dim3(assemble_blocks(n)), dim3(kAsmBlock), 0, st, map, pr, fv, x, lin, lmsel, dec);
  if (fv.with_stf && fv.n_offband > 0)
    hipLaunchKernelGGL(k_assemble_offband, dim3((fv.n_offband + 127) / 128), dim3(128), 0, st, fv, lin);
}

const char* kernels_build_info() { return "gfx950 " __DATE__ " " __TIME__; }

}  // namespace hitl
