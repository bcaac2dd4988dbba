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
constexpr int kStfAheadDefault = 1;   // index words requested this many trips ahead: one trip costs 72 VGPRs = 7 waves per SIMD, and the
                                      // ninth workgroup per CU is worth more than the second trip (bench.py A/B: 254 vs 270 us; a 64-VGPR build with
                                      // the two source points read one after the other, 8 waves per SIMD, measured the same as 7)
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
// k_linearize_stf_w (round 4): the same sums with ONE WAVE PER DIRECTED PAIR.  A half-wave per pair (above) makes every
// per-pair quantity a per-lane one: rotation, translation, target base and trip counter live in VGPRs, the trip loop is an
// exec-masked loop over the longer of the wave's two pairs, and 14 of a trip's 78 vector instructions are address and
// loop arithmetic.  With a whole wave on one pair all of that is wave-uniform: R01 / t01 sit in SGPRs (one scalar operand
// per fp64 instruction), the target scan is a scalar base with a 32-bit lane offset, the index stream advances by scalar
// adds, the trip counter is a scalar loop.  The 8 sums of a pair are reduced through LDS (8 ds_write_b64, 4 ds_read_b128,
// 7 adds, 3 DPP steps) instead of a swizzle butterfly, and the ~100-instruction tail of a pair (1/n, two rsqrt, the 2x6
// Jacobian) runs once per workgroup in lanes 0..5 of wave 0 instead of once per wave.
// Workgroup = 6 waves = one source pose (the band holds <= 6 directed pairs per pose; more pairs take further rounds).
// ===========================================================================================
constexpr int kStfWWaves = 6;
constexpr int kStfWBlock = 64 * kStfWWaves;
constexpr int kStfWTot = 12;        // per pair in LDS: the 8 totals, t01x, t01y, match count, spare

typedef const char __attribute__((address_space(1)))* gbytes_t;       // global address space, spelled out: survives the integer round trip
__device__ __forceinline__ gbytes_t uniform_gptr(const void* p) {     // a wave-uniform global pointer, pinned to an SGPR pair
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (gbytes_t)(((uint64_t)hi << 32) | lo);
}

typedef float f4raw_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gload_f4(gbytes_t p) {
  const f4raw_t v = *(const f4raw_t __attribute__((address_space(1)))*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}

// (u & 0xffff) << 4 in one instruction (SDWA word select feeding the shift): the byte offset of source point u.lo16 in an LDS plane
__device__ __forceinline__ uint32_t lo16_shl4(uint32_t u) {
  uint32_t r;
  asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(u));
  return r;
}
typedef double d2raw_t __attribute__((ext_vector_type(2)));
typedef const d2raw_t __attribute__((address_space(3)))* lds_d2_t;

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
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

template <int MINB>
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
  }
  const double xi = uniform_f64(x[3 * i]), yi = uniform_f64(x[3 * i + 1]);
  const double2 cs0 = cs[i];
  const double c0 = uniform_f64(cs0.x), s0 = uniform_f64(cs0.y);
  bool staged = false;
  for (int pb = p_begin;;) {                            // workgroup-uniform trip count
#define HITL_RTGT(u) (*reinterpret_cast<const float4*>(obs_b + (tgt_b + (__builtin_amdgcn_ubfe((u), 16, 16) << 4))))
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
// k_linearize_stf_h (round 4): the half-wave kernel of rounds 1-3 (k_linearize_stf<2, true, 1, 7>, still the fastest structure
// measured: profiles/r04_stf_levers.md) with ten of a trip's 78 vector instructions removed:
//   * the dynamic LDS segment starts at LDS address 0 and the plane stride is a template constant, so a source point's two
//     reads are `ds_read_b128 v, off` and `ds_read_b128 v, off offset:16*CAP` with off = (u & 0xffff) << 4 from ONE
//     SDWA shift -- no base additions, no mask (-6 per trip);
//   * w = R01 p1 + t01 by fused multiply-adds (stf_match_w: -4 per trip); the lane sums T00 instead of T01.
// Same decomposition otherwise: one workgroup (192 threads) per source pose, one 32-lane half-wave per directed pair, index
// words one trip ahead, blockIdx -> pose keeps consecutive poses on one XCD.
// ===========================================================================================
template <int CAP, int MINB, int FLAGS>
__global__ __launch_bounds__(kStfBlock, MINB) void k_linearize_stf_h(MapView map, PairView pr, int chunk, const double* __restrict__ x,
                                                                     const double2* __restrict__ cs, double kf, double* __restrict__ pairRJ, StfLm lmsel) {
  if (lmsel.lm != nullptr) {                    // device-driven LM loop: the candidate's buffers (uniform branch, scalar loads)
    if (lmsel.lm->done) return;
    if ((lmsel.lm->cur ^ 1) != 0) { x = lmsel.x_alt; pairRJ = lmsel.pairRJ_alt; }
  }
  extern __shared__ double s_raw_h[];
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)s_raw_h != 0u) __builtin_trap();   // the trips address the planes from LDS address 0
  double2* s_pts = reinterpret_cast<double2*>(s_raw_h);                       // [CAP] (px, py), then [CAP] (nx, ny)
  double (*s_tot)[kStfSums] = reinterpret_cast<double (*)[kStfSums]>(s_raw_h + 4 * CAP);   // the 8 totals of each half-wave's pair
  const int b = blockIdx.x;
  const int idx = b >> 3;
  const int i = (b & 7) * chunk + idx;
  if (idx >= chunk || i >= map.n_poses) return;
  const int p_begin = pr.pose_pair_off[i], p_end = pr.pose_pair_off[i + 1];
  if (p_begin == p_end) return;
  const int64_t oi = map.obs_off[i];
  const int mi = (int)(map.obs_off[i + 1] - oi);
  const float4* __restrict__ src_g = map.obs + oi;
  const int lane = threadIdx.x & 63;
  const int sub = threadIdx.x / kStfSub;        // pair slot of this half-wave within the workgroup
  const int hl = threadIdx.x & (kStfSub - 1);   // lane within the half-wave
  const char* __restrict__ mtb = reinterpret_cast<const char*>(pr.matches);
  const char* __restrict__ obs_b = reinterpret_cast<const char*>(map.obs);
  constexpr int kSlots = kStfBlock / kStfSub;
  int p = p_begin + sub;
  bool live = p < p_end;
  int pc = live ? p : p_begin;
  int j = pr.pair_tgt[pc];
  int64_t m_begin = pr.match_off[pc];
  int64_t m_end = pr.match_off[pc + 1];
  for (int k = threadIdx.x; k < mi; k += kStfBlock) {
    const float4 o = src_g[k];
    s_pts[k] = make_double2((double)o.x, (double)o.y);
    s_pts[k + CAP] = make_double2((double)o.z, (double)o.w);
  }
  const double xi = uniform_f64(x[3 * i]), yi = uniform_f64(x[3 * i + 1]);
  const double2 cs0 = cs[i];
  const double c0 = uniform_f64(cs0.x), s0 = uniform_f64(cs0.y);
  bool staged = false;
  for (int pb = p_begin;;) {                            // workgroup-uniform trip count
    const int64_t oj = map.obs_off[j];
    const uint32_t tgt_b = (uint32_t)oj << 4;                                   // byte offset of the target scan
#define HITL_HTGT(u) (*reinterpret_cast<const float4*>(obs_b + (tgt_b + (__builtin_amdgcn_ubfe((u), 16, 16) << 4))))
    const double xj = x[3 * j], yj = x[3 * j + 1];
    const double2 cs1 = cs[j];
    const int len = live ? (int)(m_end - m_begin) : 0;                          // a pair holds < 65536 matches
    uint32_t mo = ((uint32_t)m_begin + (uint32_t)hl) << 2;
#define HITL_HIDX(k) (*reinterpret_cast<const uint32_t*>(mtb + (mo + 4u * (uint32_t)(k))))
    uint32_t ua = HITL_HIDX(0), ub = HITL_HIDX(kStfSub);
    if (!staged) { __syncthreads(); staged = true; }    // the source scan is in LDS
    const double ddx = xj - xi, ddy = yj - yi;
    const double cR = c0 * cs1.x + s0 * cs1.y, sR = c0 * cs1.y - s0 * cs1.x;   // R01 = R0^T R1
    const double t01x = c0 * ddx + s0 * ddy, t01y = c0 * ddy - s0 * ddx;       // R0^T (t1 - t0)
    StfAcc acc = {0, 0, 0, 0, 0, 0, 0, 0};
    const int nfull = len >> 6, rem = len & 63;
    for (int t = 0; t < nfull; ++t) {
      const uint32_t u0 = ua, u1 = ub;
      const float4 tv0 = HITL_HTGT(u0), tv1 = HITL_HTGT(u1);
      const lds_d2_t l0 = (lds_d2_t)(uintptr_t)((FLAGS & 2) ? ((u0 & 0xffffu) << 4) : lo16_shl4(u0)), l1 = (lds_d2_t)(uintptr_t)((FLAGS & 2) ? ((u1 & 0xffffu) << 4) : lo16_shl4(u1));
      const d2raw_t a0 = l0[0], n0 = l0[CAP], a1 = l1[0], n1 = l1[CAP];
      ua = HITL_HIDX(2 * kStfSub); ub = HITL_HIDX(3 * kStfSub);                 // trip t + 1
      __builtin_amdgcn_sched_barrier(0);      // keep the index prefetch ahead of the arithmetic (the scheduler sinks it otherwise)
      if (FLAGS & 1) {
        stf_match<false>(acc, a0.x, a0.y, n0.x, n0.y, tv0, cR, sR, t01x, t01y, 1.0);
        stf_match<false>(acc, a1.x, a1.y, n1.x, n1.y, tv1, cR, sR, t01x, t01y, 1.0);
      } else {
        stf_match_w<false>(acc, a0.x, a0.y, n0.x, n0.y, tv0, cR, sR, t01x, t01y, 1.0);
        stf_match_w<false>(acc, a1.x, a1.y, n1.x, n1.y, tv1, cR, sR, t01x, t01y, 1.0);
      }
      mo += 8u * kStfSub;
    }
    if (rem > 0) {      // the last, partial trip: lanes past the end take index 0 (valid) with weight 0
      const bool v0 = hl < rem, v1 = hl + kStfSub < rem;
      const uint32_t u0 = v0 ? ua : 0u, u1 = v1 ? ub : 0u;
      const float4 tv0 = HITL_HTGT(u0), tv1 = HITL_HTGT(u1);
      const lds_d2_t l0 = (lds_d2_t)(uintptr_t)((FLAGS & 2) ? ((u0 & 0xffffu) << 4) : lo16_shl4(u0)), l1 = (lds_d2_t)(uintptr_t)((FLAGS & 2) ? ((u1 & 0xffffu) << 4) : lo16_shl4(u1));
      const d2raw_t a0 = l0[0], n0 = l0[CAP], a1 = l1[0], n1 = l1[CAP];
      if (FLAGS & 1) {
        stf_match<true>(acc, a0.x, a0.y, n0.x, n0.y, tv0, cR, sR, t01x, t01y, v0 ? 1.0 : 0.0);
        stf_match<true>(acc, a1.x, a1.y, n1.x, n1.y, tv1, cR, sR, t01x, t01y, v1 ? 1.0 : 0.0);
      } else {
        stf_match_w<true>(acc, a0.x, a0.y, n0.x, n0.y, tv0, cR, sR, t01x, t01y, v0 ? 1.0 : 0.0);
        stf_match_w<true>(acc, a1.x, a1.y, n1.x, n1.y, tv1, cR, sR, t01x, t01y, v1 ? 1.0 : 0.0);
      }
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
      const double S0 = s_tot[sub][0], A0x = s_tot[sub][1], A0y = s_tot[sub][2], Tq = s_tot[sub][3];
      const double S1 = s_tot[sub][4], A1x = s_tot[sub][5], A1y = s_tot[sub][6], T10 = s_tot[sub][7];
      const double Tr = (A0x * t01y - A0y * t01x) - Tq;            // sum e0 (-cross(n0, u)),  u = w - t01 (or the other way round)
      const double T00 = (FLAGS & 1) ? Tr : Tq, T01 = (FLAGS & 1) ? Tq : Tr;
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
#undef HITL_HTGT
#undef HITL_HIDX
    pb += kSlots;
    if (pb >= p_end) break;
    p = pb + sub; live = p < p_end; pc = live ? p : p_begin;
    j = pr.pair_tgt[pc]; m_begin = pr.match_off[pc]; m_end = pr.match_off[pc + 1];
  }
}

// CAP: points per LDS plane (compile-time: the normals are read at a constant offset from the points' address)
template <int CAP, int MINW>
__global__ __launch_bounds__(kStfWBlock, MINW) void k_linearize_stf_w(MapView map, PairView pr, int chunk, const double* __restrict__ x,
                                                                      const double2* __restrict__ cs, double kf, double* __restrict__ pairRJ,
                                                                      StfLm lmsel) {
  if (lmsel.lm != nullptr) {
    if (lmsel.lm->done) return;
    if ((lmsel.lm->cur ^ 1) != 0) { x = lmsel.x_alt; pairRJ = lmsel.pairRJ_alt; }
  }
  extern __shared__ double s_raw_w[];
  double2* s_pts = reinterpret_cast<double2*>(s_raw_w);              // [CAP] (px, py), then [CAP] (nx, ny)
  double* s_red = s_raw_w + 4 * CAP;                                 // [waves][8][64]
  double* s_tot = s_red + kStfWWaves * kStfSums * 64;                // [2][waves][kStfWTot]
  const int b = blockIdx.x;
  const int idx = b >> 3;
  const int i = (b & 7) * chunk + idx;
  if (idx >= chunk || i >= map.n_poses) return;
  const int p_begin = pr.pose_pair_off[i], p_end = pr.pose_pair_off[i + 1];
  if (p_begin == p_end) return;
  const int64_t oi = map.obs_off[i];
  const int mi = (int)(map.obs_off[i + 1] - oi);
  const float4* __restrict__ src_g = map.obs + oi;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // round 1 of the dependent loads (target pose, match range of this wave's pair) is requested before the scan is staged
  int p = p_begin + wave;
  bool live = p < p_end;                                             // wave-uniform
  int pc = live ? p : p_begin;
  int j = pr.pair_tgt[pc];
  int64_t m_begin = pr.match_off[pc];
  int len = live ? (int)(pr.match_off[pc + 1] - m_begin) : 0;
  for (int k = threadIdx.x; k < mi; k += kStfWBlock) {
    const float4 o = src_g[k];
    s_pts[k] = make_double2((double)o.x, (double)o.y);
    s_pts[k + CAP] = make_double2((double)o.z, (double)o.w);
  }
  const double xi = x[3 * i], yi = x[3 * i + 1];
  const double2 cs0 = cs[i];
  const double c0 = cs0.x, s0 = cs0.y;
  double* red = s_red + wave * (kStfSums * 64);
  const int lane4 = lane << 2;
  // the kernel declares no static LDS: its dynamic segment starts at LDS address 0, and the trip loop addresses the planes from there
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)s_raw_w != 0u) __builtin_trap();
  bool staged = false;
  int par = 0;
  for (int pb = p_begin;;) {
    const gbytes_t tgt = uniform_gptr(map.obs + map.obs_off[j]);
    const gbytes_t mb = uniform_gptr(pr.matches + m_begin);
    int mo = lane4;                                                  // signed: the +256 of the second word folds into the instruction's offset
#define HITL_WIDX(k) (*(const uint32_t __attribute__((address_space(1)))*)(mb + (mo + 4 * (k))))
#define HITL_WTGT(u) gload_f4(tgt + (((u) >> 16) << 4))
    uint32_t ua = HITL_WIDX(0), ub = HITL_WIDX(64);                  // trip 0, unguarded (kMatchPad)
    const double xj = x[3 * j], yj = x[3 * j + 1];
    const double2 cs1 = cs[j];
    const double ddx = xj - xi, ddy = yj - yi;
    const double cR = uniform_f64(c0 * cs1.x + s0 * cs1.y), sR = uniform_f64(c0 * cs1.y - s0 * cs1.x);   // R01 = R0^T R1
    const double t01x = uniform_f64(c0 * ddx + s0 * ddy), t01y = uniform_f64(c0 * ddy - s0 * ddx);       // R0^T (t1 - t0)
    if (!staged) { __syncthreads(); staged = true; }                 // the source scan is in LDS
    StfAcc acc = {0, 0, 0, 0, 0, 0, 0, 0};
    const int nfull = len >> 7, rem = len & 127;
    for (int t = 0; t < nfull; ++t) {
      const uint32_t u0 = ua, u1 = ub;
      const float4 tv0 = HITL_WTGT(u0), tv1 = HITL_WTGT(u1);
      const lds_d2_t l0 = (lds_d2_t)(uintptr_t)lo16_shl4(u0), l1 = (lds_d2_t)(uintptr_t)lo16_shl4(u1);
      const d2raw_t a0 = l0[0], n0 = l0[CAP], a1 = l1[0], n1 = l1[CAP];
      mo += 512;
      ua = HITL_WIDX(0); ub = HITL_WIDX(64);
      __builtin_amdgcn_sched_barrier(0);
      stf_match<false>(acc, a0.x, a0.y, n0.x, n0.y, tv0, cR, sR, t01x, t01y, 1.0);
      stf_match<false>(acc, a1.x, a1.y, n1.x, n1.y, tv1, cR, sR, t01x, t01y, 1.0);
    }
    if (rem > 0) {                                                   // last, partial trip: lanes past the end take index 0 with weight 0
      const bool v0 = lane < rem, v1 = lane + 64 < rem;
      const uint32_t u0 = v0 ? ua : 0u, u1 = v1 ? ub : 0u;
      const float4 tv0 = HITL_WTGT(u0);
      const double2 a0 = s_pts[u0 & 0xffffu], n0 = s_pts[(u0 & 0xffffu) + CAP];
      stf_match<true>(acc, a0.x, a0.y, n0.x, n0.y, tv0, cR, sR, t01x, t01y, v0 ? 1.0 : 0.0);
      if (rem > 64) {
        const float4 tv1 = HITL_WTGT(u1);
        const double2 a1 = s_pts[u1 & 0xffffu], n1 = s_pts[(u1 & 0xffffu) + CAP];
        stf_match<true>(acc, a1.x, a1.y, n1.x, n1.y, tv1, cR, sR, t01x, t01y, v1 ? 1.0 : 0.0);
      }
    }
#undef HITL_WIDX
#undef HITL_WTGT
    if (live) {
      // 64 lanes x 8 sums -> 8 totals: through LDS (the wave's own 4 KB), 8 partial sums per lane, then an 8-lane DPP reduction
      red[0 * 64 + lane] = acc.S0; red[1 * 64 + lane] = acc.A0x; red[2 * 64 + lane] = acc.A0y; red[3 * 64 + lane] = acc.T01;
      red[4 * 64 + lane] = acc.S1; red[5 * 64 + lane] = acc.A1x; red[6 * 64 + lane] = acc.A1y; red[7 * 64 + lane] = acc.T10;
      __builtin_amdgcn_wave_barrier();
      const double2* rp = reinterpret_cast<const double2*>(red + lane * 8);   // value lane / 8, its lanes 8 (lane % 8) .. + 7
      const double2 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
      double tot = ((q0.x + q0.y) + (q1.x + q1.y)) + ((q2.x + q2.y) + (q3.x + q3.y));
      tot += dpp_f64<0x141>(tot);      // row_half_mirror: lane l + lane 7 - l
      tot += dpp_f64<0xB1>(tot);       // quad_perm [1,0,3,2]
      tot += dpp_f64<0x4E>(tot);       // quad_perm [2,3,0,1]
      double* st = s_tot + (par * kStfWWaves + wave) * kStfWTot;
      if ((lane & 7) == 0) st[lane >> 3] = tot;
      if (lane == 0) { st[8] = t01x; st[9] = t01y; st[10] = (double)len; }
    }
    __syncthreads();
    // the tail of all pairs of this round in lanes 0..5 of wave 0: r = k sqrt(S/n) (residual_functors.h:822-827),
    // dr/dq = k (sum e' de') / (n sqrt(S/n));  sum e0 cross(n0, w) = cross(A0, t01) - T01,  sum e1 cross(n1', t01 - p0) = cross(A1, t01) - T10
    if (wave == 0 && lane < kStfWWaves && pb + lane < p_end) {
      const double* st = s_tot + (par * kStfWWaves + lane) * kStfWTot;
      const double S0 = st[0], A0x = st[1], A0y = st[2], T01 = st[3];
      const double S1 = st[4], A1x = st[5], A1y = st[6], T10 = st[7];
      const double tx = st[8], ty = st[9];
      const double T00 = (A0x * ty - A0y * tx) - T01;
      const double T11 = (A1x * ty - A1y * tx) - T10;
      const double inv_n = 1.0 / st[10];
      const double a0 = S0 * inv_n, a1 = S1 * inv_n;
      const double i0 = (S0 != 0.0) ? rsqrt(a0) : 0.0, i1 = (S1 != 0.0) ? rsqrt(a1) : 0.0;
      const double f0 = kf * inv_n * i0, f1 = kf * inv_n * i1;
      const double g0x = c0 * A0x - s0 * A0y, g0y = s0 * A0x + c0 * A0y;   // R0 * sum e0' n0
      const double g1x = c0 * A1x - s0 * A1y, g1y = s0 * A1x + c0 * A1y;   // R0 * sum e1' n1'
      double* o = pairRJ + (size_t)kPairRJ * (pb + lane);
      o[0] = kf * a0 * i0; o[1] = kf * a1 * i1;
      o[2] = -f0 * g0x; o[3] = -f0 * g0y; o[4] = f0 * T00; o[5] = f0 * g0x; o[6] = f0 * g0y; o[7] = f0 * T01;
      o[8] = -f1 * g1x; o[9] = -f1 * g1y; o[10] = f1 * T10; o[11] = f1 * g1x; o[12] = f1 * g1y; o[13] = f1 * T11;
    }
    pb += kStfWWaves;
    if (pb >= p_end) break;
    par ^= 1;
    p = pb + wave; live = p < p_end; pc = live ? p : p_begin;
    j = pr.pair_tgt[pc]; m_begin = pr.match_off[pc]; len = live ? (int)(pr.match_off[pc + 1] - m_begin) : 0;
  }
}

// ===========================================================================================
// k_linearize_stf_p (round 4): the wave-per-pair trips above inside PERSISTENT workgroups.
//
// A workgroup of k_linearize_stf lives for one pose: descriptor loads, scan staging, ~7 trips, a reduction, a ~100-instruction
// tail -- four dependent memory rounds and two long scalar-like chains per 2 100 matches, hidden only by having nine such
// workgroups per CU in different phases (80 us of a 250 us launch remain with every trip removed).  Here a workgroup of six
// waves walks a contiguous run of STEPS (a step = one source pose and up to six of its directed pairs, one per wave; the
// table is built once per correspondence set by k_stf_steps_*), and everything a step needs is requested one step ahead:
//   * the step records (224 B: source scan, and per wave target scan, match range, pair) come by scalar loads issued before
//     the barrier of the step before -- addresses are known in advance, nothing is chased;
//   * the next pose's scan is copied global -> LDS by LDS-DMA (no registers) under the current pose's trips into an fp32
//     staging area and widened to the fp64 planes between the two barriers of a step, each wave converting the 64-point
//     pieces it fetched itself;
//   * pose states (x, cos/sin) of the next step and its first index words are requested behind the trips, land under the
//     barrier, and the pair constants (R01, t01) are formed once per wave in SGPRs;
//   * the 8 sums of a pair go through 2 KB of LDS per wave to 8 totals (16 vector instructions instead of a 45-instruction
//     swizzle butterfly), the totals into a ring in LDS, and the tail (1/n, two rsqrt, the 2x6 Jacobian: ~100 instructions
//     of one lane's arithmetic) runs for ~30 pairs at once, one pair per lane, every fifth step.
// Per pair the wave now issues ~250 vector instructions instead of ~355.
// ===========================================================================================
struct alignas(32) StfStepHdr { int pose, n_live, mi, first; int64_t src_off; int p_first, pad; };
struct alignas(32) StfStepSlot { int64_t tgt_off, m_begin; int len, j, pad0, pad1; };
struct StfStep { StfStepHdr h; StfStepSlot s[kStfWWaves]; };
static_assert(sizeof(StfStep) == 224, "step record layout");

__global__ __launch_bounds__(1024) void k_stf_step_scan(int n, const int* __restrict__ pose_pair_off, int* __restrict__ step_off) {
  __shared__ int sums[1024];
  const int t = threadIdx.x, per = (n + 1023) / 1024;
  const int a = min(n, t * per), b = min(n, a + per);
  int c = 0;
  for (int i = a; i < b; ++i) c += (pose_pair_off[i + 1] - pose_pair_off[i] + kStfWWaves - 1) / kStfWWaves;
  sums[t] = c;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = (t >= d) ? sums[t - d] : 0;
    __syncthreads();
    sums[t] += v;
    __syncthreads();
  }
  int run = sums[t] - c;
  for (int i = a; i < b; ++i) {
    step_off[i] = run;
    run += (pose_pair_off[i + 1] - pose_pair_off[i] + kStfWWaves - 1) / kStfWWaves;
  }
  if (t == 1023) step_off[n] = sums[1023];
}

__global__ void k_stf_step_fill(MapView map, PairView pr, const int* __restrict__ step_off, StfStep* __restrict__ steps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= map.n_poses) return;
  const int p0 = pr.pose_pair_off[i], p1 = pr.pose_pair_off[i + 1];
  const int64_t oi = map.obs_off[i];
  const int mi = (int)(map.obs_off[i + 1] - oi);
  int sidx = step_off[i];
  for (int pb = p0; pb < p1; pb += kStfWWaves, ++sidx) {
    StfStep r;
    r.h.pose = i; r.h.n_live = min(kStfWWaves, p1 - pb); r.h.mi = mi; r.h.first = (pb == p0) ? 1 : 0;
    r.h.src_off = oi; r.h.p_first = pb; r.h.pad = 0;
    for (int w = 0; w < kStfWWaves; ++w) {
      const int p = pb + w;
      StfStepSlot& q = r.s[w];
      if (p < p1) {
        const int j = pr.pair_tgt[p];
        q.tgt_off = map.obs_off[j]; q.m_begin = pr.match_off[p]; q.len = (int)(pr.match_off[p + 1] - pr.match_off[p]); q.j = j;
      } else {
        q.tgt_off = oi; q.m_begin = 0; q.len = 0; q.j = i;
      }
      q.pad0 = q.pad1 = 0;
    }
    steps[sidx] = r;
  }
}

void launch_build_stf_steps(hipStream_t st, const MapView& map, const PairView& pr, int* step_off, void* steps) {
  hipLaunchKernelGGL(k_stf_step_scan, dim3(1), dim3(1024), 0, st, map.n_poses, pr.pose_pair_off, step_off);
  hipLaunchKernelGGL(k_stf_step_fill, dim3((map.n_poses + 127) / 128), dim3(128), 0, st, map, pr, step_off, reinterpret_cast<StfStep*>(steps));
}
size_t stf_step_bytes() { return sizeof(StfStep); }

typedef int i8v __attribute__((ext_vector_type(8)));
typedef int i4v __attribute__((ext_vector_type(4)));
// scalar loads that stay where they are written (volatile) and that the compiler still tracks (its own s_waitcnt at first use)
__device__ __forceinline__ i8v sload8(const void* p) { return *(const volatile i8v __attribute__((address_space(4)))*)p; }
__device__ __forceinline__ i4v sload4(const void* p) { return *(const volatile i4v __attribute__((address_space(4)))*)p; }
__device__ __forceinline__ double f64_of(int lo, int hi) { return __hiloint2double(hi, lo); }
__device__ __forceinline__ int64_t i64_of(int lo, int hi) { return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo); }

// LDS-DMA: 16 bytes per lane from a per-lane global address to LDS at (wave-uniform) lds_dst + 16 * lane; no destination
// register.  Written as asm so that the compiler's vmcnt bookkeeping does not see it: with the builtin every later use of an
// ordinary load result waits for vmcnt(0), i.e. also for the index words requested a trip ahead.  The loads return in order,
// so the first gather wait behind it also covers it; reads of the staged bytes follow an explicit vmcnt wait below.
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

constexpr int kStfRing = 48;          // pairs whose totals wait in LDS for their tail
constexpr int kStfRingStride = 17;    // doubles per ring entry (16 used; odd stride: the tail's lanes read different banks)
constexpr int kStfFlushAt = 30;       // pending pairs that trigger a tail pass

__device__ __forceinline__ void lds_only_barrier() {        // orders LDS traffic, not the loads in flight
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int CAP>
__global__ __launch_bounds__(kStfWBlock, 6) void k_linearize_stf_p(const StfStep* __restrict__ steps, const int* __restrict__ n_steps_p,
                                                                   const float4* __restrict__ obs, const uint32_t* __restrict__ matches,
                                                                   const double* __restrict__ x, const double2* __restrict__ cs, double kf,
                                                                   double* __restrict__ pairRJ) {
  extern __shared__ double s_raw_p[];
  // byte offsets inside the dynamic segment (which starts at LDS address 0: no static LDS in this kernel)
  constexpr uint32_t kStage = 32u * CAP, kScratch = 48u * CAP, kRingOff = kScratch + kStfWWaves * 2048u;
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)s_raw_p != 0u) __builtin_trap();
  double2* s_pts = reinterpret_cast<double2*>(s_raw_p);
  const float4* s_stage = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_raw_p) + kStage);
  double* s_ring = reinterpret_cast<double*>(reinterpret_cast<char*>(s_raw_p) + kRingOff);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double* sc = reinterpret_cast<double*>(reinterpret_cast<char*>(s_raw_p) + kScratch) + wave * 256;
  const int S = *n_steps_p;
  // XCD-interleaved steps (see k_stf_stream_count: contiguous runs per workgroup put 4x the L2 in front of an XCD, 46 % hits):
  // XCD k = blockIdx % 8 owns an eighth of the steps, its workgroups take them round-robin, sd steps apart
  const int sd = (int)(gridDim.x >> 3);
  const int chunk8 = (S + 7) / 8;
  const int s0 = (int)(blockIdx.x & 7) * chunk8 + (int)(blockIdx.x >> 3), s1 = min(((int)(blockIdx.x & 7) + 1) * chunk8, S);
  if (sd < 1 || s0 >= s1) return;
  const int lane4 = lane << 2;

  // ---- first step of the run: records, scan, constants, all behind one another (once per workgroup) ----
  i8v H = sload8(&steps[s0].h), P = sload8(&steps[s0].s[wave]);
  i8v H1 = H, P1 = P;
  if (s0 + sd < s1) { H1 = sload8(&steps[s0 + sd].h); P1 = sload8(&steps[s0 + sd].s[wave]); }
  {
    const int mi = H[2];
    const float4* src = obs + i64_of(H[4], H[5]);
    for (int k = threadIdx.x; k < mi; k += kStfWBlock) {
      const float4 o = src[k];
      s_pts[k] = make_double2((double)o.x, (double)o.y);
      s_pts[k + CAP] = make_double2((double)o.z, (double)o.w);
    }
  }
  int ring_head = 0, ring_cnt = 0;         // workgroup-uniform: first pending entry, pending entries
  double cR, sR, t01x, t01y, c0, s0v;
  gbytes_t tgt, mb;
  int len, pidx;
  uint32_t ua, ub;
  {
    const int i = H[0], j = P[5];
    const i4v xi4 = sload4(x + 3 * (size_t)i), ci4 = sload4(cs + i), xj4 = sload4(x + 3 * (size_t)j), cj4 = sload4(cs + j);
    tgt = uniform_gptr(obs + i64_of(P[0], P[1]));
    mb = uniform_gptr(matches + i64_of(P[2], P[3]));
    len = P[4]; pidx = H[6] + wave;
    ua = *(const uint32_t __attribute__((address_space(1)))*)(mb + lane4);
    ub = *(const uint32_t __attribute__((address_space(1)))*)(mb + lane4 + 256);
    c0 = f64_of(ci4[0], ci4[1]); s0v = f64_of(ci4[2], ci4[3]);
    const double c1 = f64_of(cj4[0], cj4[1]), s1v = f64_of(cj4[2], cj4[3]);
    const double ddx = f64_of(xj4[0], xj4[1]) - f64_of(xi4[0], xi4[1]), ddy = f64_of(xj4[2], xj4[3]) - f64_of(xi4[2], xi4[3]);
    cR = uniform_f64(c0 * c1 + s0v * s1v); sR = uniform_f64(c0 * s1v - s0v * c1);
    t01x = uniform_f64(c0 * ddx + s0v * ddy); t01y = uniform_f64(c0 * ddy - s0v * ddx);
  }
  lds_only_barrier();

  for (int s = s0; s < s1; s += sd) {
    const bool has_next = s + sd < s1;                      // uniform
    const bool live = wave < H[1];
    const bool next_first = has_next;                       // the next step is another pose (or another round of one sd steps away: staged again)
    const int mi1 = H1[2];
    // ---- the next pose's scan: LDS-DMA into the staging area, this wave's 64-point pieces.  Issued BEHIND the first trip's
    //      gathers: the compiler does not know about it, and its wait for the index words in front of those gathers would
    //      otherwise cover the copy as well ----
    bool dma_due = next_first;
    const float4* src1 = obs + i64_of(H1[4], H1[5]);
#define HITL_PDMA()                                                                                         \
    if (dma_due) {                                                                                          \
      dma_due = false;                                                                                      \
      for (int k = wave; 64 * k < mi1; k += kStfWWaves)                                                     \
        if (64 * k + lane < mi1) dma16(src1 + 64 * k + lane, kStage + 1024u * (uint32_t)k);                 \
    }
    // ---- trips ----
    StfAcc acc = {0, 0, 0, 0, 0, 0, 0, 0};
    {
      const int nfull = len >> 7, rem = len & 127;
      int mo = lane4;
#define HITL_PIDX(k) (*(const uint32_t __attribute__((address_space(1)))*)(mb + (mo + 4 * (k))))
#define HITL_PTGT(u) gload_f4(tgt + (((u) >> 16) << 4))
      for (int t = 0; t < nfull; ++t) {
        const uint32_t u0 = ua, u1 = ub;
        const float4 tv0 = HITL_PTGT(u0), tv1 = HITL_PTGT(u1);
        const lds_d2_t l0 = (lds_d2_t)(uintptr_t)lo16_shl4(u0), l1 = (lds_d2_t)(uintptr_t)lo16_shl4(u1);
        const d2raw_t a0 = l0[0], n0 = l0[CAP], a1 = l1[0], n1 = l1[CAP];
        HITL_PDMA()
        mo += 512;
        ua = HITL_PIDX(0); ub = HITL_PIDX(64);
        __builtin_amdgcn_sched_barrier(0);
        stf_match<false>(acc, a0.x, a0.y, n0.x, n0.y, tv0, cR, sR, t01x, t01y, 1.0);
        stf_match<false>(acc, a1.x, a1.y, n1.x, n1.y, tv1, cR, sR, t01x, t01y, 1.0);
      }
      if (rem > 0) {                                         // last, partial trip: lanes past the end take index 0 with weight 0
        const bool v0 = lane < rem, v1 = lane + 64 < rem;
        const uint32_t u0 = v0 ? ua : 0u, u1 = v1 ? ub : 0u;
        const float4 tv0 = HITL_PTGT(u0);
        const lds_d2_t l0 = (lds_d2_t)(uintptr_t)lo16_shl4(u0);
        const d2raw_t a0 = l0[0], n0 = l0[CAP];
        HITL_PDMA()
        stf_match<true>(acc, a0.x, a0.y, n0.x, n0.y, tv0, cR, sR, t01x, t01y, v0 ? 1.0 : 0.0);
        if (rem > 64) {
          const float4 tv1 = HITL_PTGT(u1);
          const lds_d2_t l1 = (lds_d2_t)(uintptr_t)lo16_shl4(u1);
          const d2raw_t a1 = l1[0], n1 = l1[CAP];
          stf_match<true>(acc, a1.x, a1.y, n1.x, n1.y, tv1, cR, sR, t01x, t01y, v1 ? 1.0 : 0.0);
        }
      }
      HITL_PDMA()                                            // a wave without matches in this step still fetches its pieces
#undef HITL_PIDX
#undef HITL_PTGT
#undef HITL_PDMA
    }
    // ---- 64 lanes x 8 sums -> 8 totals through the wave's 2 KB of LDS, into the ring ----
    if (live) {
      sc[0 * 64 + lane] = acc.S0; sc[1 * 64 + lane] = acc.A0x; sc[2 * 64 + lane] = acc.A0y; sc[3 * 64 + lane] = acc.T01;
      __builtin_amdgcn_wave_barrier();
      const double2* rp = reinterpret_cast<const double2*>(sc + (lane >> 4) * 64 + (lane & 15) * 4);
      const double2 q0 = rp[0], q1 = rp[1];
      const double p0 = (q0.x + q0.y) + (q1.x + q1.y);       // value lane / 16, lanes 4 (lane % 16) .. + 3
      __builtin_amdgcn_wave_barrier();
      sc[0 * 64 + lane] = acc.S1; sc[1 * 64 + lane] = acc.A1x; sc[2 * 64 + lane] = acc.A1y; sc[3 * 64 + lane] = acc.T10;
      __builtin_amdgcn_wave_barrier();
      const double2 q2 = rp[0], q3 = rp[1];
      const double p1 = (q2.x + q2.y) + (q3.x + q3.y);       // value 4 + lane / 16
      __builtin_amdgcn_wave_barrier();
      sc[lane] = p0; sc[64 + lane] = p1;
      __builtin_amdgcn_wave_barrier();
      const int v = lane >> 3;                               // value 0..7: array v / 4, row v % 4 (16 partial sums), two per lane
      const double2 q4 = *reinterpret_cast<const double2*>(sc + (v >> 2) * 64 + (v & 3) * 16 + (lane & 7) * 2);
      double tot = q4.x + q4.y;
      tot += dpp_f64<0x141>(tot);      // row_half_mirror: lane l + lane 7 - l
      tot += dpp_f64<0xB1>(tot);       // quad_perm [1,0,3,2]
      tot += dpp_f64<0x4E>(tot);       // quad_perm [2,3,0,1]
      int slot = ring_head + ring_cnt + wave;
      slot = (slot >= kStfRing) ? slot - kStfRing : slot;
      double* st = s_ring + slot * kStfRingStride;
      if ((lane & 7) == 0) st[v] = tot;
      if (lane == 0) { st[8] = t01x; st[9] = t01y; st[10] = (double)len; st[11] = c0; st[12] = s0v; st[13] = (double)pidx; }
    }
    ring_cnt += H[1];
    // ---- requests for the next step (land under the barrier) and for the one after ----
    i8v H2 = H1, P2 = P1;
    i4v xi4 = {0, 0, 0, 0}, ci4 = xi4, xj4 = xi4, cj4 = xi4;
    uint32_t ua1 = 0, ub1 = 0;
    gbytes_t tgt1 = tgt, mb1 = mb;
    if (has_next) {
      const int i1 = H1[0], j1 = P1[5];
      xi4 = sload4(x + 3 * (size_t)i1); ci4 = sload4(cs + i1); xj4 = sload4(x + 3 * (size_t)j1); cj4 = sload4(cs + j1);
      if (s + 2 * sd < s1) { H2 = sload8(&steps[s + 2 * sd].h); P2 = sload8(&steps[s + 2 * sd].s[wave]); }
      tgt1 = uniform_gptr(obs + i64_of(P1[0], P1[1]));
      mb1 = uniform_gptr(matches + i64_of(P1[2], P1[3]));
      ua1 = *(const uint32_t __attribute__((address_space(1)))*)(mb1 + lane4);
      ub1 = *(const uint32_t __attribute__((address_space(1)))*)(mb1 + lane4 + 256);
    }
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");        // this wave's DMA pieces have landed (the two index words may still fly)
    lds_only_barrier();                                      // A: nobody reads the planes of this step any more; ring entries visible
    if (next_first) {
      for (int k = wave; 64 * k < mi1; k += kStfWWaves) {
        const int pt = 64 * k + lane;
        if (pt < mi1) {
          const float4 o = s_stage[pt];
          s_pts[pt] = make_double2((double)o.x, (double)o.y);
          s_pts[pt + CAP] = make_double2((double)o.z, (double)o.w);
        }
      }
    }
    lds_only_barrier();                                      // B: the planes hold the next step's scan
    // ---- tails of the pending pairs, one per lane (wave 0); the ring keeps taking entries behind them meanwhile ----
    if (ring_cnt >= kStfFlushAt || !has_next) {
      if (wave == 0 && lane < ring_cnt) {
        int slot = ring_head + lane;
        slot = (slot >= kStfRing) ? slot - kStfRing : slot;
        const double* st = s_ring + slot * kStfRingStride;
        const double S0 = st[0], A0x = st[1], A0y = st[2], T01 = st[3];
        const double S1 = st[4], A1x = st[5], A1y = st[6], T10 = st[7];
        const double tx = st[8], ty = st[9], cc = st[11], ss = st[12];
        const double T00 = (A0x * ty - A0y * tx) - T01;
        const double T11 = (A1x * ty - A1y * tx) - T10;
        const double inv_n = 1.0 / st[10];
        const double a0 = S0 * inv_n, a1 = S1 * inv_n;
        const double i0 = (S0 != 0.0) ? rsqrt(a0) : 0.0, i1 = (S1 != 0.0) ? rsqrt(a1) : 0.0;
        const double f0 = kf * inv_n * i0, f1 = kf * inv_n * i1;
        const double g0x = cc * A0x - ss * A0y, g0y = ss * A0x + cc * A0y;   // R0 * sum e0' n0
        const double g1x = cc * A1x - ss * A1y, g1y = ss * A1x + cc * A1y;   // R0 * sum e1' n1'
        double* o = pairRJ + (size_t)kPairRJ * (size_t)(int)st[13];
        o[0] = kf * a0 * i0; o[1] = kf * a1 * i1;
        o[2] = -f0 * g0x; o[3] = -f0 * g0y; o[4] = f0 * T00; o[5] = f0 * g0x; o[6] = f0 * g0y; o[7] = f0 * T01;
        o[8] = -f1 * g1x; o[9] = -f1 * g1y; o[10] = f1 * T10; o[11] = f1 * g1x; o[12] = f1 * g1y; o[13] = f1 * T11;
      }
      ring_head += ring_cnt;
      ring_head = (ring_head >= kStfRing) ? ring_head - kStfRing : ring_head;
      ring_cnt = 0;
    }
    // ---- the next step's pair constants; rotate the pipeline ----
    if (has_next) {
      c0 = f64_of(ci4[0], ci4[1]); s0v = f64_of(ci4[2], ci4[3]);
      const double c1 = f64_of(cj4[0], cj4[1]), s1v = f64_of(cj4[2], cj4[3]);
      const double ddx = f64_of(xj4[0], xj4[1]) - f64_of(xi4[0], xi4[1]), ddy = f64_of(xj4[2], xj4[3]) - f64_of(xi4[2], xi4[3]);
      cR = uniform_f64(c0 * c1 + s0v * s1v); sR = uniform_f64(c0 * s1v - s0v * c1);
      t01x = uniform_f64(c0 * ddx + s0v * ddy); t01y = uniform_f64(c0 * ddy - s0v * ddx);
      tgt = tgt1; mb = mb1; len = P1[4]; pidx = H1[6] + wave; ua = ua1; ub = ub1;
    }
    H = H1; H1 = H2; P1 = P2;
  }
}

size_t stf_p_lds_bytes(int cap) { return (size_t)48 * cap + (size_t)kStfWWaves * 2048 + (size_t)kStfRing * kStfRingStride * 8; }

// workgroups of the persistent kernel: what is resident at once, as the runtime counts it (at most 4 per CU by registers)
int stf_steps_grid(int cap, int cu_count) {
  int per_cu = 0;
  hipError_t e;
  if (cap <= 416) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_linearize_stf_p<416>, kStfWBlock, stf_p_lds_bytes(416));
  else if (cap <= 512) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_linearize_stf_p<512>, kStfWBlock, stf_p_lds_bytes(512));
  else if (cap <= 1024) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_linearize_stf_p<1024>, kStfWBlock, stf_p_lds_bytes(1024));
  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_linearize_stf_p<2048>, kStfWBlock, stf_p_lds_bytes(2048));
  if (e != hipSuccess || per_cu < 1) per_cu = 1;
  per_cu = std::min(per_cu, 4);
  if (per_cu > 1) --per_cu;            // the runtime's answer is one high for this LDS size (measured: a fourth workgroup per CU waits for a slot)
  if (dev_switches().stf_waves_per_cu > 0) per_cu = std::min(4, dev_switches().stf_waves_per_cu);     // HITL_STF_WAVES_PER_CU: workgroups per CU here
  return std::max(8, (per_cu * cu_count) & ~7);
}

void launch_linearize_stf_steps(hipStream_t st, int cap, int grid, const void* steps, const int* n_steps, const float4* obs, const uint32_t* matches,
                                const double* x, const double2* cs, float corr, float std_dev, double* pairRJ) {
  const double kf = (double)corr / (double)std_dev;
  const StfStep* sp = reinterpret_cast<const StfStep*>(steps);
  const dim3 g(grid), b(kStfWBlock);
#define HITL_STFP_GO(CAP) hipLaunchKernelGGL((k_linearize_stf_p<CAP>), g, b, stf_p_lds_bytes(CAP), st, sp, n_steps, obs, matches, x, cs, kf, pairRJ)
  if (cap <= 416) HITL_STFP_GO(416);
  else if (cap <= 512) HITL_STFP_GO(512);
  else if (cap <= 1024) HITL_STFP_GO(1024);
  else HITL_STFP_GO(2048);
#undef HITL_STFP_GO
}

// ===========================================================================================
// k_linearize_stf_s (round 4): ONE WAVE PER WORKGROUP, every wave on its own stream of pairs -- no barrier anywhere.
//
// What the two kernels above wait for is a dependent memory round trip of ~1 us under load (descriptor -> index words ->
// gathers), paid several times per pose with two gathers per lane in flight; what hides it there is the number of resident
// workgroups, and a workgroup that shares a scan must stop at a barrier per pose.  Here a wave owns a contiguous run of
// poses = a contiguous run of pairs (pairs are sorted by source pose), keeps ITS source scan in LDS (fp64, 32 B per point),
// and runs a software pipeline over the pairs in which nothing is waited for that was not requested a whole pair earlier:
//   iteration q:  gathers of pair q+1 (its index words arrived during pair q-1)  ->  arithmetic of pair q out of registers  ->
//                 index words of pair q+2, pose states of pair q+1, record of pair q+3.
// A pair of up to 511 matches is 3 full trips of 128 matches (6 index words + 6 gathered float4 per lane, two register sets
// that alternate) plus one partial trip whose gathers are issued in front of the full trips and used behind them.  Three waves
// per SIMD (<= 168 VGPRs) suffice: an fp64 instruction occupies the SIMD for 4 cycles, so the issue slots are filled by the
// arithmetic of ~1.5 waves, and the ~16 loads each wave keeps in flight are 4x what a wave of the half-wave kernel has.
// The 8 sums of a pair are reduced with v_permlane32_swap / v_permlane16_swap + DPP (no LDS), parked in LDS, and the tails of
// a pose's pairs run one per lane when the pose ends -- under the load of the next pose's scan.
// ===========================================================================================
struct alignas(32) StfPairRec { int64_t tgt_off, m_begin; int len, j, i, pad; };
constexpr int kStfSFull = 3;          // full trips of a pair held in registers
constexpr int kStfSMaxLen = 128 * kStfSFull + 127;
constexpr int kStfSSlots = 6;         // pairs whose totals wait for their tail (a pose of the 3-pose band has 6)
constexpr int kStfSSlotD = 12;        // doubles per slot: 8 totals, t01x, t01y, n, pair index

// Which poses a wave works on decides what the L2 of its XCD has to hold.  Contiguous runs per wave (first version of this
// kernel) put 384 waves x 7 scans = 17 MB of target scans in front of a 4 MB L2: 18 % hit rate, 3.2x the HBM reads, 438 us.
// Here XCD k (waves b with b % 8 == k, as the hardware hands workgroups round-robin to the XCDs) owns a contiguous eighth of
// the poses and its waves take them INTERLEAVED: wave w of the XCD works on poses base + w, base + w + W, base + w + 2 W ...
// (W waves per XCD), so that at any moment the waves of an XCD sit inside a window of a few hundred consecutive poses that
// slides through the XCD's range -- the scans i +- 3 a pose gathers from were read moments ago by its neighbours.
// The records of a wave's pairs are laid out in the order the wave visits them: the kernel walks recs[wave_off[b] .. wave_off[b + 1]).
__device__ __forceinline__ void stf_wave_poses(int g, int n_waves, int n_poses, int& first, int& last, int& stride) {
  const int xcd = g & 7, w = g >> 3, W = n_waves >> 3;                 // n_waves is a multiple of 8
  const int chunk = (n_poses + 7) / 8;
  first = xcd * chunk + w; last = min((xcd + 1) * chunk, n_poses); stride = W;
}
__global__ void k_stf_stream_count(int n_waves, int n_poses, const int* __restrict__ pose_pair_off, int* __restrict__ cnt) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_waves) return;
  int first, last, stride, c = 0;
  stf_wave_poses(g, n_waves, n_poses, first, last, stride);
  for (int i = first; i < last; i += stride) c += pose_pair_off[i + 1] - pose_pair_off[i];
  cnt[g] = c;
}
__global__ __launch_bounds__(1024) void k_excl_scan_small(int n, const int* __restrict__ in, int* __restrict__ out) {   // out[0..n], one workgroup
  __shared__ int sums[1024];
  const int t = threadIdx.x, per = (n + 1023) / 1024;
  const int a = min(n, t * per), b = min(n, a + per);
  int c = 0;
  for (int i = a; i < b; ++i) c += in[i];
  sums[t] = c;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = (t >= d) ? sums[t - d] : 0;
    __syncthreads();
    sums[t] += v;
    __syncthreads();
  }
  int run = sums[t] - c;
  for (int i = a; i < b; ++i) { const int v = in[i]; out[i] = run; run += v; }
  if (t == 1023) out[n] = sums[1023];
}
__global__ void k_stf_stream_fill(int n_waves, MapView map, PairView pr, const int* __restrict__ wave_off, StfPairRec* __restrict__ recs) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_waves) return;
  int first, last, stride;
  stf_wave_poses(g, n_waves, map.n_poses, first, last, stride);
  StfPairRec* out = recs + wave_off[g];
  for (int i = first; i < last; i += stride)
    for (int p = pr.pose_pair_off[i]; p < pr.pose_pair_off[i + 1]; ++p) {
      StfPairRec r;
      r.j = pr.pair_tgt[p]; r.i = i;
      r.tgt_off = map.obs_off[r.j]; r.m_begin = pr.match_off[p]; r.len = (int)(pr.match_off[p + 1] - pr.match_off[p]); r.pad = p;
      *out++ = r;
    }
}
// largest number of matches of a pair (the stream kernel holds <= kStfSMaxLen per pair in registers)
__global__ void k_stf_max_len(PairView pr, int* __restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  int v = (p < pr.n_pairs) ? (int)(pr.match_off[p + 1] - pr.match_off[p]) : 0;
  for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, WAVE));
  if ((threadIdx.x & 63) == 0 && v > 0) atomicMax(out, v);
}
void launch_build_stf_pair_recs(hipStream_t st, const MapView& map, const PairView& pr, int n_waves, int* wave_cnt, int* wave_off, void* recs, int* max_len) {
  if (pr.n_pairs == 0) return;
  hipLaunchKernelGGL(k_stf_stream_count, dim3((n_waves + 255) / 256), dim3(256), 0, st, n_waves, map.n_poses, pr.pose_pair_off, wave_cnt);
  hipLaunchKernelGGL(k_excl_scan_small, dim3(1), dim3(1024), 0, st, n_waves, wave_cnt, wave_off);
  hipLaunchKernelGGL(k_stf_stream_fill, dim3((n_waves + 63) / 64), dim3(64), 0, st, n_waves, map, pr, wave_off, reinterpret_cast<StfPairRec*>(recs));
  (void)hipMemsetAsync(max_len, 0, sizeof(int), st);
  hipLaunchKernelGGL(k_stf_max_len, dim3((pr.n_pairs + 255) / 256), dim3(256), 0, st, pr, max_len);
}
size_t stf_pair_rec_bytes() { return sizeof(StfPairRec); }
int stf_stream_max_len() { return kStfSMaxLen; }

// (u & 0xffff) << 5: byte offset of source point u.lo16 in the interleaved fp64 scan (32 B per point)
__device__ __forceinline__ uint32_t lo16_shl5(uint32_t u) {
  uint32_t r;
  asm("v_lshlrev_b32_sdwa %0, 5, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(u));
  return r;
}
__device__ __forceinline__ void swap32_f64(double& a, double& b) {   // a <- [a.lo32 | b.lo32], b <- [a.hi32 | b.hi32] (halves of the wave)
  const auto r0 = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  a = __hiloint2double(r1[0], r0[0]); b = __hiloint2double(r1[1], r0[1]);
}
__device__ __forceinline__ void swap16_f64(double& a, double& b) {   // odd 16-lane rows of a <-> even rows of b
  const auto r0 = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  a = __hiloint2double(r1[0], r0[0]); b = __hiloint2double(r1[1], r0[1]);
}
// 8 per-lane sums -> lane 8 v .. 8 v + 7 hold the wave total of value v (reduce-scatter: 32, 16, 8; then an 8-lane sum)
__device__ __forceinline__ double stf_reduce8(const StfAcc& a, int lane) {
  double v0 = a.S0, v1 = a.A0x, v2 = a.A0y, v3 = a.T01, v4 = a.S1, v5 = a.A1x, v6 = a.A1y, v7 = a.T10;
  swap32_f64(v0, v4); swap32_f64(v1, v5); swap32_f64(v2, v6); swap32_f64(v3, v7);
  double w0 = v0 + v4, w1 = v1 + v5, w2 = v2 + v6, w3 = v3 + v7;      // lanes < 32: values 0..3, lanes >= 32: values 4..7
  swap16_f64(w0, w2); swap16_f64(w1, w3);
  const double z0 = w0 + w2, z1 = w1 + w3;                            // row r of 16 lanes: z0 = value 2 r, z1 = value 2 r + 1
  const bool up = (lane & 8) != 0;
  const double send = up ? z0 : z1, keep = up ? z1 : z0;
  double tot = keep + dpp_f64<0x128>(send);                           // row_ror:8 = lane ^ 8 inside the row
  tot += dpp_f64<0x141>(tot);      // row_half_mirror
  tot += dpp_f64<0xB1>(tot);       // quad_perm [1,0,3,2]
  tot += dpp_f64<0x4E>(tot);       // quad_perm [2,3,0,1]
  return tot;
}

struct StfSet { uint32_t u[2 * kStfSFull]; uint32_t ru[2]; float4 g[2 * kStfSFull]; float4 rg[2]; };

__global__ __launch_bounds__(64, 3) void k_linearize_stf_s(const StfPairRec* __restrict__ recs, const int* __restrict__ wave_off,
                                                           const int64_t* __restrict__ obs_off, const float4* __restrict__ obs,
                                                           const uint32_t* __restrict__ matches, const double* __restrict__ x,
                                                           const double2* __restrict__ cs, double kf, double* __restrict__ pairRJ, int tot_off) {
  extern __shared__ double s_raw_s[];
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)s_raw_s != 0u) __builtin_trap();   // the trips address the scan from LDS address 0
  double2* s_scan = reinterpret_cast<double2*>(s_raw_s);             // point k: [2k] = (px, py), [2k + 1] = (nx, ny)
  double* s_tot = s_raw_s + tot_off;                                 // [kStfSSlots][kStfSSlotD]
  const int lane = threadIdx.x;
  const int lane4 = lane << 2;
  const int P0 = wave_off[blockIdx.x], P1 = wave_off[blockIdx.x + 1];      // this wave's records, in the order it visits its pairs
  if (P0 >= P1) return;

#define HITL_SREC(q) sload8(recs + (q))
#define HITL_SIDX(mbp, off) (*(const uint32_t __attribute__((address_space(1)))*)((mbp) + (lane4 + (off))))
#define HITL_STGT(tb, u) gload_f4((tb) + (((u) >> 16) << 4))
  // index words of a pair into a set: the partial trip first (matches [0, r)), then the full trips (128 matches each)
#define HITL_SLOAD_IDX(SET, R)                                                                                         \
  {                                                                                                                     \
    const int len_ = (R)[4];                                                                                            \
    const int F_ = len_ >> 7, r_ = len_ & 127;                                                                          \
    const gbytes_t mb_ = uniform_gptr(matches + i64_of((R)[2], (R)[3]));                                                \
    const gbytes_t mbf_ = mb_ + 4 * r_;                                                                                 \
    SET.ru[0] = HITL_SIDX(mb_, 0); SET.ru[1] = HITL_SIDX(mb_, 256);                                                     \
    _Pragma("unroll") for (int k_ = 0; k_ < kStfSFull; ++k_) {      /* unconditional (the array is padded): straight-line loads keep */ \
      SET.u[2 * k_] = HITL_SIDX(mbf_, 512 * k_); SET.u[2 * k_ + 1] = HITL_SIDX(mbf_, 512 * k_ + 256);   /* the compiler's vmcnt counts exact */ \
    }                                                                                                                   \
    (void)F_;                                                                                                           \
  }
#define HITL_SGATHER(SET, R)                                                                                           \
  {                                                                                                                     \
    const int F_ = (R)[4] >> 7, r_ = (R)[4] & 127;                                                                      \
    const gbytes_t tb_ = uniform_gptr(obs + i64_of((R)[0], (R)[1]));                                                    \
    SET.ru[0] = (lane < r_) ? SET.ru[0] : 0u; SET.ru[1] = (lane + 64 < r_) ? SET.ru[1] : 0u;   /* lanes past the partial trip: point 0, weight 0 */ \
    SET.rg[0] = HITL_STGT(tb_, SET.ru[0]); SET.rg[1] = HITL_STGT(tb_, SET.ru[1]);                                       \
    _Pragma("unroll") for (int k_ = 0; k_ < kStfSFull; ++k_) {      /* trips the pair does not have gather point 0 (one line, broadcast) */ \
      const uint32_t ua_ = (k_ < F_) ? SET.u[2 * k_] : 0u, ub_ = (k_ < F_) ? SET.u[2 * k_ + 1] : 0u;                    \
      SET.g[2 * k_] = HITL_STGT(tb_, ua_); SET.g[2 * k_ + 1] = HITL_STGT(tb_, ub_);                                     \
    }                                                                                                                   \
  }
#define HITL_SCONST(XI, CI, XJ, CJ)                                                                                    \
  {                                                                                                                     \
    c0 = f64_of((CI)[0], (CI)[1]); s0v = f64_of((CI)[2], (CI)[3]);                                                      \
    const double c1_ = f64_of((CJ)[0], (CJ)[1]), s1_ = f64_of((CJ)[2], (CJ)[3]);                                        \
    const double ddx_ = f64_of((XJ)[0], (XJ)[1]) - f64_of((XI)[0], (XI)[1]), ddy_ = f64_of((XJ)[2], (XJ)[3]) - f64_of((XI)[2], (XI)[3]); \
    cR = uniform_f64(c0 * c1_ + s0v * s1_); sR = uniform_f64(c0 * s1_ - s0v * c1_);                                     \
    t01x = uniform_f64(c0 * ddx_ + s0v * ddy_); t01y = uniform_f64(c0 * ddy_ - s0v * ddx_);                             \
  }
  // the source scan of pose I: global -> registers -> fp64 -> LDS (this wave's own copy)
#define HITL_SSCAN(I)                                                                                                  \
  {                                                                                                                     \
    const int64_t o_ = obs_off[(I)];                                                                                    \
    const int mi_ = (int)(obs_off[(I) + 1] - o_);                                                                       \
    const float4* sp_ = obs + o_;                                                                                       \
    for (int k_ = lane; k_ < mi_; k_ += 64) {                                                                           \
      const float4 o4_ = sp_[k_];                                                                                       \
      s_scan[2 * k_] = make_double2((double)o4_.x, (double)o4_.y);                                                      \
      s_scan[2 * k_ + 1] = make_double2((double)o4_.z, (double)o4_.w);                                                  \
    }                                                                                                                   \
  }

  StfSet A, B, C;
  double cR, sR, t01x, t01y, c0, s0v;
  i8v R0 = HITL_SREC(P0), R1 = HITL_SREC(min(P0 + 1, P1 - 1)), R2 = HITL_SREC(min(P0 + 2, P1 - 1)), R3 = R2;   // records of pairs q .. q + 3
  HITL_SSCAN(R0[6])
  HITL_SLOAD_IDX(A, R0)
  HITL_SGATHER(A, R0)
  HITL_SLOAD_IDX(C, R1)
  {
    const i4v xi4 = sload4(x + 3 * (size_t)R0[6]), ci4 = sload4(cs + R0[6]), xj4 = sload4(x + 3 * (size_t)R0[5]), cj4 = sload4(cs + R0[5]);
    HITL_SCONST(xi4, ci4, xj4, cj4)
  }
  int pend = 0;                           // pairs of the current pose whose totals wait in LDS

  // source points of full trip K (index words X.u[2K], X.u[2K+1]) into staging set S; a trip the pair does not have reads point 0
#define HITL_SLDS(S, X, K)                                                                                               \
  {                                                                                                                     \
    const uint32_t wa_ = ((K) < F) ? X.u[2 * (K)] : 0u, wb_ = ((K) < F) ? X.u[2 * (K) + 1] : 0u;                        \
    const lds_d2_t l0_ = (lds_d2_t)(uintptr_t)lo16_shl5(wa_), l1_ = (lds_d2_t)(uintptr_t)lo16_shl5(wb_);                \
    S##a0 = l0_[0]; S##n0 = l0_[1]; S##a1 = l1_[0]; S##n1 = l1_[1];                                                     \
  }
  // one pair: X = its set (index words + gathers in registers), Y = the set that takes the next pair's gathers
#define HITL_SITER(X, Y)                                                                                               \
  {                                                                                                                     \
    const bool has1 = q + 1 < P1;                                                                                       \
    const int len = R0[4], F = len >> 7, r = len & 127;                                                                 \
    /* every request is unconditional (pairs past the end of the run repeat its last pair): no load sits under a branch */ \
    _Pragma("unroll") for (int k_ = 0; k_ < 2 * kStfSFull; ++k_) Y.u[k_] = C.u[k_];                                     \
    Y.ru[0] = C.ru[0]; Y.ru[1] = C.ru[1];                                                                               \
    HITL_SGATHER(Y, R1)                  /* pair q + 1: all of its gathers, used one iteration from now */                \
    HITL_SLOAD_IDX(C, R2)                /* index words of pair q + 2: a whole pair's arithmetic to arrive */             \
    const bool v0 = lane < r, v1 = lane + 64 < r;                                                                       \
    StfAcc acc = {0, 0, 0, 0, 0, 0, 0, 0};                                                                              \
    /* source points out of LDS one trip ahead of the arithmetic that uses them (two staging sets, P and Q, that alternate) */ \
    d2raw_t Pa0, Pn0, Pa1, Pn1, Qa0, Qn0, Qa1, Qn1;                                                                     \
    { const lds_d2_t l0 = (lds_d2_t)(uintptr_t)lo16_shl5(X.ru[0]), l1 = (lds_d2_t)(uintptr_t)lo16_shl5(X.ru[1]);        \
      Pa0 = l0[0]; Pn0 = l0[1]; Pa1 = l1[0]; Pn1 = l1[1]; }                                                             \
    HITL_SLDS(Q, X, 0)                                                                                                     \
    stf_match<true>(acc, Pa0.x, Pa0.y, Pn0.x, Pn0.y, X.rg[0], cR, sR, t01x, t01y, v0 ? 1.0 : 0.0);                      \
    if (r > 64) stf_match<true>(acc, Pa1.x, Pa1.y, Pn1.x, Pn1.y, X.rg[1], cR, sR, t01x, t01y, v1 ? 1.0 : 0.0);          \
    if (F > 0) {                                                                                                        \
      HITL_SLDS(P, X, 1)                                                                                                   \
      stf_match<false>(acc, Qa0.x, Qa0.y, Qn0.x, Qn0.y, X.g[0], cR, sR, t01x, t01y, 1.0);                               \
      stf_match<false>(acc, Qa1.x, Qa1.y, Qn1.x, Qn1.y, X.g[1], cR, sR, t01x, t01y, 1.0);                               \
    }                                                                                                                   \
    if (F > 1) {                                                                                                        \
      HITL_SLDS(Q, X, 2)                                                                                                   \
      stf_match<false>(acc, Pa0.x, Pa0.y, Pn0.x, Pn0.y, X.g[2], cR, sR, t01x, t01y, 1.0);                               \
      stf_match<false>(acc, Pa1.x, Pa1.y, Pn1.x, Pn1.y, X.g[3], cR, sR, t01x, t01y, 1.0);                               \
    }                                                                                                                   \
    if (F > 2) {                                                                                                        \
      stf_match<false>(acc, Qa0.x, Qa0.y, Qn0.x, Qn0.y, X.g[4], cR, sR, t01x, t01y, 1.0);                               \
      stf_match<false>(acc, Qa1.x, Qa1.y, Qn1.x, Qn1.y, X.g[5], cR, sR, t01x, t01y, 1.0);                               \
    }                                                                                                                   \
    /* records and pose states for the next iterations: behind the last LDS read of this pair, in front of a reduction that */ \
    /* touches neither LDS nor memory -- the scalar loads are back before anything has to wait for them */                   \
    R3 = HITL_SREC(min(q + 3, P1 - 1));                                                                                 \
    const i4v xi4 = sload4(x + 3 * (size_t)R1[6]), ci4 = sload4(cs + R1[6]), xj4 = sload4(x + 3 * (size_t)R1[5]), cj4 = sload4(cs + R1[5]); \
    const double tot = stf_reduce8(acc, lane);                                                                          \
    {                                                                                                                   \
      double* st_ = s_tot + pend * kStfSSlotD;                                                                          \
      if ((lane & 7) == 0) st_[lane >> 3] = tot;                                                                        \
      if (lane == 0) { st_[8] = t01x; st_[9] = t01y; st_[10] = (double)len; st_[11] = (double)R0[7]; }                  \
    }                                                                                                                   \
    ++pend;                                                                                                             \
    const bool pose_ends = !has1 || R1[6] != R0[6];                                                                     \
    if (pose_ends || pend == kStfSSlots) {                                                                              \
      if (lane < pend) {                                                                                                \
        const double* st = s_tot + lane * kStfSSlotD;                                                                   \
        const double S0 = st[0], A0x = st[1], A0y = st[2], T01 = st[3];                                                 \
        const double S1 = st[4], A1x = st[5], A1y = st[6], T10 = st[7];                                                 \
        const double tx = st[8], ty = st[9];                                                                            \
        const double T00 = (A0x * ty - A0y * tx) - T01;                                                                 \
        const double T11 = (A1x * ty - A1y * tx) - T10;                                                                 \
        const double inv_n = 1.0 / st[10];                                                                              \
        const double a0 = S0 * inv_n, a1 = S1 * inv_n;                                                                  \
        const double i0_ = (S0 != 0.0) ? rsqrt(a0) : 0.0, i1_ = (S1 != 0.0) ? rsqrt(a1) : 0.0;                          \
        const double f0 = kf * inv_n * i0_, f1 = kf * inv_n * i1_;                                                      \
        const double g0x = c0 * A0x - s0v * A0y, g0y = s0v * A0x + c0 * A0y;                                            \
        const double g1x = c0 * A1x - s0v * A1y, g1y = s0v * A1x + c0 * A1y;                                            \
        double* o = pairRJ + (size_t)kPairRJ * (size_t)(int)st[11];                                                     \
        o[0] = kf * a0 * i0_; o[1] = kf * a1 * i1_;                                                                     \
        o[2] = -f0 * g0x; o[3] = -f0 * g0y; o[4] = f0 * T00; o[5] = f0 * g0x; o[6] = f0 * g0y; o[7] = f0 * T01;         \
        o[8] = -f1 * g1x; o[9] = -f1 * g1y; o[10] = f1 * T10; o[11] = f1 * g1x; o[12] = f1 * g1y; o[13] = f1 * T11;     \
      }                                                                                                                 \
      pend = 0;                                                                                                         \
    }                                                                                                                   \
    if (pose_ends && has1) HITL_SSCAN(R1[6])                                                                            \
    HITL_SCONST(xi4, ci4, xj4, cj4)                                                                                     \
    R0 = R1; R1 = R2; R2 = R3;                                                                                          \
  }

  for (int q = P0; q < P1;) {
    HITL_SITER(A, B)
    ++q;
    if (q >= P1) break;
    HITL_SITER(B, A)
    ++q;
  }
#undef HITL_SITER
#undef HITL_SLDS
#undef HITL_SSCAN
#undef HITL_SCONST
#undef HITL_SGATHER
#undef HITL_SLOAD_IDX
#undef HITL_STGT
#undef HITL_SIDX
#undef HITL_SREC
}

// waves (= workgroups) of the stream kernel: as many as fit one CU's LDS, at most 3 per SIMD (registers); a multiple of 8 (XCDs)
int stf_stream_waves(int max_scan, int cu_count) {
  const size_t lds = (size_t)32 * max_scan + (size_t)kStfSSlots * kStfSSlotD * 8;
  // every wave must be resident at once (a second round of workgroups would double the launch): ask the runtime how many
  // fit a CU with this much LDS (the allocation granule is the device's business), never more than 3 per SIMD
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_linearize_stf_s, 64, lds) != hipSuccess || per_cu < 1)
    per_cu = std::max(1, (int)(((size_t)160 * 1024) / (lds + 1024)));
  per_cu = std::max(1, std::min(per_cu, 12));
  if (dev_switches().stf_waves_per_cu > 0) per_cu = std::min(per_cu, dev_switches().stf_waves_per_cu);     // HITL_STF_WAVES_PER_CU (development)
  return std::max(8, (per_cu * cu_count) & ~7);
}
void launch_linearize_stf_stream(hipStream_t st, int max_scan, int n_waves, const void* recs, const int* wave_off, const int64_t* obs_off,
                                 const float4* obs, const uint32_t* matches, const double* x, const double2* cs, float corr, float std_dev, double* pairRJ) {
  const double kf = (double)corr / (double)std_dev;
  const int tot_off = 4 * max_scan;                                                       // doubles
  const size_t lds = (size_t)32 * max_scan + (size_t)kStfSSlots * kStfSSlotD * 8;
  hipLaunchKernelGGL(k_linearize_stf_s, dim3(n_waves), dim3(64), lds, st, reinterpret_cast<const StfPairRec*>(recs), wave_off, obs_off, obs,
                     matches, x, cs, kf, pairRJ, tot_off);
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
  // HITL_STF_ADDR64=1 (tests): take the 64-bit addressing instantiations although the arrays are smaller than 4 GB
  const bool off32 = (uint64_t)map_obs_count * 16 < ((uint64_t)1 << 32) && ((uint64_t)n_matches + kMatchPad) * 4 < ((uint64_t)1 << 32) &&
                     !dev_switches().stf_addr64;
  const double kf = (double)corr / (double)std_dev;                       // k = corr / sigma, divided here once
  const dim3 grid(8 * chunk), block(kStfBlock);
  // HITL_STF_VARIANT (development switch): 2 decimal digits "ahead minwaves", e.g. 26 = index words 2 trips ahead, 6 waves per
  // SIMD (5, 6, 7 exist; 7 only with 1 or 2 trips ahead).  Unset: the default below.
  const int variant = dev_switches().stf_variant;
  const int src = ((size_t)cap * 32 <= 64 * 1024) ? 2 : ((size_t)cap * 16 <= 64 * 1024) ? 1 : 0;
  if ((variant == 0 || variant == 700) && src == 2 && off32 && pose_recs != nullptr) {      // the half-wave kernel behind a per-pose descriptor record
    hipLaunchKernelGGL((k_linearize_stf_r<7>), grid, block, (size_t)cap * 32, st, map, pr, reinterpret_cast<const StfPoseRec*>(pose_recs), chunk, x, cs, kf,
                       pairRJ, lmsel);
    return;
  }
  if ((variant >= 600 && variant < 700) && src == 2 && off32) {      // the half-wave kernel, instruction diet of round 4 (60x: ablations)
    const int caph = max_scan <= 416 ? 416 : max_scan <= 512 ? 512 : max_scan <= 1024 ? 1024 : 2048;
    const size_t ldsh = (size_t)caph * 32 + (size_t)(kStfBlock / kStfSub) * kStfSums * 8;
#define HITL_STFH_GO(CAP, MB, FL) hipLaunchKernelGGL((k_linearize_stf_h<CAP, MB, FL>), grid, block, ldsh, st, map, pr, chunk, x, cs, kf, pairRJ, lmsel)
    if (caph == 416) {
      if (variant == 601) HITL_STFH_GO(416, 7, 1); else if (variant == 602) HITL_STFH_GO(416, 7, 2); else if (variant == 603) HITL_STFH_GO(416, 7, 3); else HITL_STFH_GO(416, 7, 0);
    } else if (caph == 512) HITL_STFH_GO(512, 7, 0); else if (caph == 1024) HITL_STFH_GO(1024, 7, 0); else HITL_STFH_GO(2048, 7, 0);
#undef HITL_STFH_GO
    return;
  }
  if (variant >= 100 && variant < 200 && src == 2) {                // one wave per directed pair (k_linearize_stf_w)
    const dim3 wblock(kStfWBlock);
    const int capw = cap <= 512 ? 512 : cap <= 1024 ? 1024 : 2048;
    const size_t ldsw = (size_t)capw * 32 + (size_t)kStfWWaves * kStfSums * 64 * 8 + 2 * kStfWWaves * kStfWTot * 8;
#define HITL_STFW_GO(CAP, MW) hipLaunchKernelGGL((k_linearize_stf_w<CAP, MW>), grid, wblock, ldsw, st, map, pr, chunk, x, cs, kf, pairRJ, lmsel)
    const int mw = variant % 10;
    if (capw == 512) { if (mw == 5) HITL_STFW_GO(512, 5); else if (mw == 7) HITL_STFW_GO(512, 7); else if (mw == 8) HITL_STFW_GO(512, 8); else HITL_STFW_GO(512, 6); }
    else if (capw == 1024) HITL_STFW_GO(1024, 6);
    else HITL_STFW_GO(2048, 6);
#undef HITL_STFW_GO
    return;
  }
  int ahead = kStfAheadDefault, minb = (src == 2 && off32) ? 7 : 5;     // the fallback layouts need > 80 VGPRs: 5 waves per SIMD instead of spilling
  if (variant > 0 && variant < 100 && src == 2 && off32) { ahead = std::min(3, std::max(1, variant / 10)); minb = (variant % 10 == 5) ? 5 : (variant % 10 == 7) ? 7 : 6; }
  const size_t lds = (src == 2) ? (size_t)cap * 32 : (src == 1) ? (size_t)cap * 16 : 0;
#define HITL_STF_GO(SRC, O32, AH, MINB) hipLaunchKernelGGL((k_linearize_stf<SRC, O32, AH, MINB>), grid, block, lds, st, map, pr, chunk, x, cs, kf, pairRJ, lmsel)
  if (src == 2 && off32 && minb == 7) {
    if (variant == 501 || variant == 0 || variant == 700) hipLaunchKernelGGL((k_linearize_stf<2, true, 1, 7, true>), grid, block, lds, st, map, pr, chunk, x, cs, kf, pairRJ, lmsel);
    else if (ahead == 1) HITL_STF_GO(2, true, 1, 7); else HITL_STF_GO(2, true, 2, 7);
  } else if (src == 2 && off32) {
    if (minb == 6) { if (ahead == 1) HITL_STF_GO(2, true, 1, 6); else if (ahead == 2) HITL_STF_GO(2, true, 2, 6); else HITL_STF_GO(2, true, 3, 6); }
    else { if (ahead == 1) HITL_STF_GO(2, true, 1, 5); else if (ahead == 2) HITL_STF_GO(2, true, 2, 5); else HITL_STF_GO(2, true, 3, 5); }
  } else if (src == 2) HITL_STF_GO(2, false, 1, 5);
  else if (src == 1) { if (off32) HITL_STF_GO(1, true, 1, 5); else HITL_STF_GO(1, false, 1, 5); }
  else { if (off32) HITL_STF_GO(0, true, 1, 5); else HITL_STF_GO(0, false, 1, 5); }
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
                                                         LinView lin, AsmLm lmsel) {
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
                     const LinView& lin, const LmDev* lm, const double* x_alt, const LinView* lin_alt) {
  const int n = map.n_poses;
  AsmLm lmsel = {lm, x_alt, lin_alt ? *lin_alt : lin};
  hipLaunchKernelGGL(k_assemble, dim3(assemble_blocks(n)), dim3(kAsmBlock), 0, st, map, pr, fv, x, lin, lmsel);
  if (fv.with_stf && fv.n_offband > 0)
    hipLaunchKernelGGL(k_assemble_offband, dim3((fv.n_offband + 127) / 128), dim3(128), 0, st, fv, lin);
}

const char* kernels_build_info() { return "gfx950 " __DATE__ " " __TIME__; }

}  // namespace hitl
