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

}  // namespace hitl
