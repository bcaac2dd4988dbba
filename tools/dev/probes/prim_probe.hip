// Development probe: what the primitives of a persistent kernel cost on this GPU (one workgroup of 512 threads, timings by
// s_memrealtime, 100 MHz): workgroup barriers, dependent LDS round trips, dependent fp64 FMA chains, a level-like step
// (12 LDS reads + 12 FMAs + 1 LDS write + barrier).   hipcc --offload-arch=gfx950 -O3 prim_probe.hip -o prim_probe && ./prim_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__global__ __launch_bounds__(512) void k_probe(unsigned long long* out, double* sink, int n) {
  __shared__ double buf[2048];
  const int tid = threadIdx.x;
  for (int e = tid; e < 2048; e += 512) buf[e] = 1.0 + 1e-9 * e;
  __syncthreads();
  unsigned long long t[8];
  double acc = 0.0;
  t[0] = __builtin_amdgcn_s_memrealtime();
  for (int k = 0; k < n; ++k) lds_barrier();
  t[1] = __builtin_amdgcn_s_memrealtime();
  // dependent LDS round trips (one wave's view): read -> add -> write -> read ...
  double v = buf[tid];
  for (int k = 0; k < n; ++k) { buf[(tid + 64) & 2047] = v + 1.0; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); v = buf[(tid + 64 * (k & 1)) & 2047]; }
  acc += v;
  t[2] = __builtin_amdgcn_s_memrealtime();
  // dependent fp64 FMA chain
  double f = v;
  for (int k = 0; k < n; ++k) { f = f * 1.0000001 + 0.5; f = f * 0.9999999 - 0.5; f = f * 1.0000001 + 0.25; f = f * 0.9999999 - 0.25; }
  acc += f;
  t[3] = __builtin_amdgcn_s_memrealtime();
  // a level-like step: 12 LDS reads (broadcast-ish), 12 FMAs in two chains, one LDS write, one barrier
  for (int k = 0; k < n; ++k) {
    const double* w = buf + ((tid / 24) * 13 + 64 * (k & 3));
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int r = 0; r < 12; r += 2) { d0 += w[r] * f; d1 += w[r + 1] * v; }
    buf[1024 + tid] = d0 + d1;
    lds_barrier();
  }
  acc += buf[1024 + ((tid + 1) & 511)];
  t[4] = __builtin_amdgcn_s_memrealtime();
  // the same with __syncthreads()
  for (int k = 0; k < n; ++k) {
    const double* w = buf + ((tid / 24) * 13 + 64 * (k & 3));
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int r = 0; r < 12; r += 2) { d0 += w[r] * f; d1 += w[r + 1] * v; }
    buf[1024 + tid] = d0 + d1;
    __syncthreads();
  }
  acc += buf[1024 + ((tid + 1) & 511)];
  t[5] = __builtin_amdgcn_s_memrealtime();
  // one wave only does the work, the other seven just meet it at the barrier
  for (int k = 0; k < n; ++k) {
    if (tid < 64) {
      const double* w = buf + ((tid / 24) * 13 + 64 * (k & 3));
      double d0 = 0.0, d1 = 0.0;
#pragma unroll
      for (int r = 0; r < 12; r += 2) { d0 += w[r] * f; d1 += w[r + 1] * v; }
      buf[1024 + tid] = d0 + d1;
    }
    lds_barrier();
  }
  t[6] = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x * 512 + tid] = acc;
  if (tid == 0) for (int k = 0; k < 7; ++k) out[blockIdx.x * 8 + k] = t[k];
}
int main() {
  unsigned long long* out; double* sink;
  hipMalloc(&out, 8 * 8 * 4); hipMalloc(&sink, 8 * 512 * 4);
  const int n = 1000;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_probe, dim3(1), dim3(512), 0, 0, out, sink, n); hipDeviceSynchronize(); }
  unsigned long long h[8];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[6] = {"lds_barrier (8 waves)", "dependent LDS write+read", "4 dependent fp64 FMAs", "level step + lds_barrier", "level step + __syncthreads", "one wave works + lds_barrier"};
  for (int k = 0; k < 6; ++k) printf("%-32s %8.1f ns each\n", names[k], (double)(h[k + 1] - h[k]) * 10.0 / n);
  return 0;
}
