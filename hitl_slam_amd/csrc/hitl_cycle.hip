// hitl_cycle.hip -- the pose-chain stages between the user's correction and the batch re-solve
// (SURVEY.md 8f row 3), fp32 like the reference and compiled with -ffp-contract=off:
//   k_explicit_apply   == AppExpCorrect::ApplyExplicitCorrections   (ApplyExplicitCorrection.cpp:387-415)
//   k_constraint_rows  == AppExpCorrect::calculateConstraintTargets (ApplyExplicitCorrection.cpp:447-487)
//   k_backprop_rotate / k_backprop_translate / k_backprop_cov == Backprop::BackPropagateError (Backprop.cpp:98-200)
//
// Backprop in the reference is two O(span^2) loops: pivot i rotates every later pose about the CURRENT position of
// pose i, then every later pose is shifted by a share of the remaining translation.  fp32 addition is not associative,
// so the only way to return the reference's poses bit for bit is to apply the same span-many updates to each pose in
// the same order.  That is still parallel: every pose is independent once the pivots before it are final.
// k_backprop_rotate runs it as a chained scan over blocks of kBpBlock poses -- a block first applies the published
// transforms of all earlier blocks (parallel over its poses), then resolves its own pivots wave by wave (inside a wave
// the pivot position travels by v_readlane: no LDS round trip, no barrier, no branch per pivot), publishes them with
// agent-scope atomic stores and raises its flag -- so the critical path is span x one small step (~70 ns) instead of
// span^2.  All transcendental values (cos/sin of each pivot's share of the rotation) come from the host's libm.

#include "hitl_internal.h"

namespace hitl {

// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_explicit_apply(float* __restrict__ poses, ExplicitArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n || i < a.g0) return;
  float x = poses[3 * i], y = poses[3 * i + 1], th = poses[3 * i + 2];
  if (i <= a.g1) {
    // p1 = dst + R (p0 - cmA); T = p1 - p0; pose += (T, theta)
    const float qx = x - a.pivx, qy = y - a.pivy;
    const float p1x = a.dstx + (a.c * qx - a.s * qy), p1y = a.dsty + (a.s * qx + a.c * qy);
    const float tx = p1x - x, ty = p1y - y;
    x += tx; y += ty; th += a.theta;
  } else {
    th += a.lth;
    const float abx = x - a.lx, aby = y - a.ly;
    const float nx = a.cl * abx - a.sl * aby, ny = a.sl * abx + a.cl * aby;
    x = (a.lx + nx) + a.ltx; y = (a.ly + ny) + a.lty;
  }
  poses[3 * i] = x; poses[3 * i + 1] = y; poses[3 * i + 2] = th;
}

void launch_explicit_apply(hipStream_t st, float* poses, const ExplicitArgs& a) {
  hipLaunchKernelGGL(k_explicit_apply, dim3((a.n + 255) / 256), dim3(256), 0, st, poses, a);
}

// anchor table (host, libm): AnchorRow per anchor pose.  RowOut == hitl_human_constraint_t, field by field.
struct RowOut { unsigned type; int constrained, anchor; float dpar, dperp, dang, rel; };
static_assert(sizeof(RowOut) == sizeof(hitl_human_constraint_t), "row layout");

__global__ __launch_bounds__(256) void k_constraint_rows(const float* __restrict__ poses, const AnchorRow* __restrict__ anchors,
                                                         const int* __restrict__ corrected, int n_anchor, int n_corr, unsigned type,
                                                         RowOut* __restrict__ rows) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)n_anchor * n_corr) return;
  const int a = (int)(t / n_corr), j = (int)(t - (int64_t)a * n_corr);
  const AnchorRow A = anchors[a];
  const int id = corrected[j];
  const float rx = poses[3 * id] - A.x, ry = poses[3 * id + 1] - A.y;
  const float da = poses[3 * id + 2] - A.angle;
  RowOut r;
  r.type = type; r.constrained = id; r.anchor = A.id;
  r.dpar = A.c * rx + A.s * ry;
  r.dperp = -A.s * rx + A.c * ry;
  r.dang = atan2f(sinf(da), cosf(da));   // device libm, three calls deep: within 4 ulp of the host's
  r.rel = A.rel;
  rows[t] = r;
}

void launch_constraint_rows(hipStream_t st, const float* poses, const AnchorRow* anchors, const int* corrected, int n_anchor, int n_corr,
                            unsigned type, hitl_human_constraint_t* rows) {
  const int64_t tot = (int64_t)n_anchor * n_corr;
  if (tot == 0) return;
  hipLaunchKernelGGL(k_constraint_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, poses, anchors, corrected, n_anchor, n_corr,
                     type, reinterpret_cast<RowOut*>(rows));
}

// ------------------------------------------------------------------------------------------------------------
constexpr int kBpBlock = 256;

__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// piv[q] = (dtheta, cos dtheta, sin dtheta, unused) for span position q in [0, L); pose index = lo + q, q in [0, L].
// sync[0] = ticket counter, sync[1 + b] = "block b published"; xf[q] = (c, s, tx, ty) of the finished pivot q.
__global__ __launch_bounds__(kBpBlock) void k_backprop_rotate(float* __restrict__ poses, int lo, int L, const float4* __restrict__ piv,
                                                              float* xf, int* sync, float destx, float desty,
                                                              float* __restrict__ trans_out) {
  __shared__ float4 s_xf[kBpBlock], s_out[kBpBlock];
  __shared__ float s_dth[kBpBlock];
  __shared__ int s_ticket;
  if (threadIdx.x == 0) s_ticket = atomicAdd(&sync[0], 1);
  __syncthreads();
  const int b = s_ticket;
  const int q = b * kBpBlock + (int)threadIdx.x;
  const bool live = q <= L;
  float x = 0.f, y = 0.f, th = 0.f;
  if (live) { x = poses[3 * (lo + q)]; y = poses[3 * (lo + q) + 1]; th = poses[3 * (lo + q) + 2]; }

  for (int pb = 0; pb < b; ++pb) {            // earlier blocks, in order
    if (threadIdx.x == 0)
      while (__hip_atomic_load(&sync[1 + pb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    const int pq = pb * kBpBlock + (int)threadIdx.x;    // always < L here
    const float* src = xf + 4 * (size_t)pq;
    s_xf[threadIdx.x] = make_float4(ld_agent(src), ld_agent(src + 1), ld_agent(src + 2), ld_agent(src + 3));
    s_dth[threadIdx.x] = piv[pq].x;
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < kBpBlock; ++j) {
      const float4 t = s_xf[j];
      th += s_dth[j];
      const float nx = (t.x * x + -t.y * y) + t.z, ny = (t.y * x + t.x * y) + t.w;
      x = nx; y = ny;
    }
    __syncthreads();
  }

  // own pivots, one after the other
  const int npiv = min(kBpBlock, L - b * kBpBlock);       // pivots are the positions < L
  const float4 mine = (q < L) ? piv[q] : make_float4(0.f, 1.f, 0.f, 0.f);
  s_xf[threadIdx.x] = mine;                                // (dth, c, s) of this block's pivots
  __syncthreads();
  // Wave by wave: the 64 pivots of a wave are resolved inside the wave (the pivot position travels by v_readlane, no
  // LDS round trip and no barrier per pivot); the waves behind it then apply those 64 finished transforms from LDS.
  {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int wv = 0; wv * 64 < npiv; ++wv) {
      const int cnt = min(64, npiv - wv * 64);
      if (wave == wv) {
        // Branch-free inner step: (tx, ty) is the same number in every lane, lane p keeps it for the block's output,
        // lanes behind p take the update through selects.  Only the pivot position is on the dependent chain
        // (two v_readlane); the pivot's (dth, c, s) come from LDS.
        float keep_tx = 0.f, keep_ty = 0.f;
        float4 pr_next = s_xf[wv * 64];
        for (int p = 0; p < cnt; ++p) {
          const float4 pr = pr_next;
          pr_next = s_xf[wv * 64 + min(p + 1, 63)];          // next pivot's parameters: off the dependent chain
          const float dth = pr.x, c = pr.y, s = pr.z;
          const float Px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), p));
          const float Py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), p));
          // Translation(P) * Rotation(dth) * Translation(-P):  translation = P + R (-P)
          const float tx = Px + (c * -Px + -s * -Py), ty = Py + (s * -Px + c * -Py);
          const float nx = (c * x + -s * y) + tx, ny = (s * x + c * y) + ty;
          const bool behind = lane > p;
          x = behind ? nx : x; y = behind ? ny : y;
          th = (lane >= p) ? th + dth : th;
          keep_tx = (lane == p) ? tx : keep_tx; keep_ty = (lane == p) ? ty : keep_ty;
        }
        if (lane < cnt) s_out[wv * 64 + lane] = make_float4(mine.y, mine.z, keep_tx, keep_ty);
      }
      __syncthreads();
      if (wave > wv && live) {
        for (int p = 0; p < cnt; ++p) {
          const float4 t = s_out[wv * 64 + p];
          th += s_xf[wv * 64 + p].x;
          const float nx = (t.x * x + -t.y * y) + t.z, ny = (t.y * x + t.x * y) + t.w;
          x = nx; y = ny;
        }
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < npiv) {
    const float4 t = s_out[threadIdx.x];
    // Publication without a release fence (an agent-scope fence writes back the whole L2 of this XCD, ~25 us per
    // block here): the transforms go out as agent-scope atomic stores, which are performed at the device-wide
    // coherence point; every wave then waits for ITS stores explicitly (s_waitcnt vmcnt(0) below -- the barrier alone
    // does not: the compiler emits no vmcnt wait for relaxed atomic stores before s_barrier), and only after the
    // barrier is the flag raised.  Readers use agent-scope atomic loads for flag and data alike.
    float* dst = xf + 4 * (size_t)(b * kBpBlock + (int)threadIdx.x);
    __hip_atomic_store(dst + 0, t.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 1, t.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 2, t.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 3, t.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0) expcnt(7) lgkmcnt(0): this wave's four stores above have been performed
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&sync[1 + b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (live) { poses[3 * (lo + q)] = x; poses[3 * (lo + q) + 1] = y; poses[3 * (lo + q) + 2] = th; }
  if (q == L) { trans_out[0] = destx - x; trans_out[1] = desty - y; }   // trans = destination - poses_[max].translation
}

// translation stage: position q receives tw[i] * trans for i = 0 .. q-1, added one at a time in that order
__global__ __launch_bounds__(kBpBlock) void k_backprop_translate(float* __restrict__ poses, int lo, int L, const float* __restrict__ tw,
                                                                 const float* __restrict__ trans) {
  __shared__ float s_tw[kBpBlock];
  const int q = blockIdx.x * kBpBlock + (int)threadIdx.x;
  const bool live = q >= 1 && q <= L;
  const float trx = trans[0], try_ = trans[1];
  float x = 0.f, y = 0.f;
  if (live) { x = poses[3 * (lo + q)]; y = poses[3 * (lo + q) + 1]; }
  const int last = min(L, (int)(blockIdx.x + 1) * kBpBlock - 1);      // largest position in this block
  for (int i0 = 0; i0 < last; i0 += kBpBlock) {
    __syncthreads();
    s_tw[threadIdx.x] = (i0 + (int)threadIdx.x < L) ? tw[i0 + threadIdx.x] : 0.f;
    __syncthreads();
    const int cnt = min(kBpBlock, q - i0);                             // this pose takes pivots i < q
    for (int j = 0; j < cnt; ++j) { const float w = s_tw[j]; x += w * trx; y += w * try_; }
  }
  if (live) { poses[3 * (lo + q)] = x; poses[3 * (lo + q) + 1] = y; }
}

// covariance shrink, Backprop.cpp:156-169 (entry (0,2) is scaled twice and (1,2) never, as there)
__global__ __launch_bounds__(256) void k_backprop_cov(float* __restrict__ cov9, int lo, int hi, float trans_beta, float rot_beta) {
  const int i = lo + blockIdx.x * 256 + (int)threadIdx.x;
  if (i >= hi) return;
  float* c = cov9 + 9 * (size_t)i;
  c[0] *= trans_beta; c[1] *= trans_beta; c[3] *= trans_beta; c[4] *= trans_beta;
  float v = c[2]; v *= rot_beta; v *= rot_beta; c[2] = v;
  c[6] *= rot_beta; c[7] *= rot_beta; c[8] *= rot_beta;
}

int backprop_blocks(int L) { return (L + 1 + kBpBlock - 1) / kBpBlock; }

void launch_backprop(hipStream_t st, float* poses, float* cov9, int lo, int hi, const float* piv4, const float* tw, float* xf, int* sync,
                     float destx, float desty, float trans_beta, float rot_beta, float* trans2) {
  const int L = hi - lo, nb = backprop_blocks(L);
  (void)hipMemsetAsync(sync, 0, sizeof(int) * (size_t)(nb + 1), st);
  hipLaunchKernelGGL(k_backprop_cov, dim3((L + 255) / 256), dim3(256), 0, st, cov9, lo, hi, trans_beta, rot_beta);
  hipLaunchKernelGGL(k_backprop_rotate, dim3(nb), dim3(kBpBlock), 0, st, poses, lo, L, (const float4*)piv4, xf, sync, destx, desty, trans2);
  hipLaunchKernelGGL(k_backprop_translate, dim3(nb), dim3(kBpBlock), 0, st, poses, lo, L, tw, trans2);
}

}  // namespace hitl
