// hitl_device.h -- device-side helpers shared by the .hip files of libhitlgpu.so (wavefront = 64 everywhere).
#pragma once
#include "hitl_internal.h"

namespace hitl {

#define WAVE 64

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
  return v;
}

// a value that is the same in every lane, moved to scalar registers
__device__ __forceinline__ double uniform_f64(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
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

// ===========================================================================================
// The decision of one Levenberg-Marquardt iteration on the device: Ceres' trust-region rules as restated in SURVEY.md A.4 /
// DESIGN.md section 2, the same statements in the same order as the host loop of lm_solve (hitl_lm.cpp), compiled without
// contraction like that loop (1 - q*q*q must round the same way).  One thread.
// ===========================================================================================
__device__ __forceinline__ void lm_decide_step(LmDev& s, const LmOpts& o, double rr, double bb, double dt_rhs, double dt_q, double step2, double xn2,
                                               double bad, double cand_cost, double cand_gmax) {
#pragma clang fp contract(off)
  ++s.iter;
  const bool ok = (rr == rr) && !(rr > o.guard2 * bb);          // the factorisation did not break down
  const double mcc = dt_rhs - 0.5 * dt_q;                        // model_cost_change; rhs = -S g
  if (!ok || bad > 0.0 || !(mcc == mcc) || mcc < 0.0) {         // invalid step
    ++s.unsuccessful;
    if (++s.invalid_run >= 5) { s.term = HITL_FAILURE; s.done = 1; }
    else s.radius *= 0.5;                                        // LevenbergMarquardtStrategy::StepIsInvalid
  } else {
    s.invalid_run = 0;
    ++s.lin_calls;
    const bool par_tol = sqrt(step2) <= o.parameter_tolerance * (sqrt(xn2) + o.parameter_tolerance);
    const double cost_change = s.cost - cand_cost;
    const bool fun_tol = fabs(cost_change) <= o.function_tolerance * s.cost;
    if (!o.late && (par_tol || fun_tol)) { s.term = HITL_CONVERGENCE; s.done = 1; }
    else {
      const double rho = cost_change / mcc;
      if (rho > o.min_relative_decrease) {
        ++s.successful;
        s.cur ^= 1;
        s.cost = cand_cost; s.gmax = cand_gmax;
        if (s.gmax <= o.gradient_tolerance) { s.term = HITL_CONVERGENCE; s.done = 1; }
        else if (o.late && (par_tol || fun_tol)) { s.term = HITL_CONVERGENCE; s.done = 1; }
        else {
          const double q = 2.0 * rho - 1.0;
          s.radius = fmin(o.max_radius, s.radius / fmax(1.0 / 3.0, 1.0 - q * q * q));
          s.decrease_factor = 2.0;
        }
      } else {
        ++s.unsuccessful;
        s.radius /= s.decrease_factor; s.decrease_factor *= 2.0;
        if (s.radius < o.min_radius) { s.term = HITL_CONVERGENCE; s.done = 1; }
      }
    }
  }
  if (!s.done && s.iter >= o.max_iter) { s.term = HITL_NO_CONVERGENCE; s.done = 1; }
}

}  // namespace hitl
