// Development probe: the preconditioner sweep of k_pcg_fused (hitl_slam_amd/csrc/hitl_pcg_sweep.h) timed in isolation: one workgroup,
// made-up factors, n repetitions of [forward levels + boundary sums + back levels]; 100 MHz timestamps.
//   hipcc --offload-arch=gfx950 -O3 -I../../../hitl_slam_amd/csrc sweep_probe.hip -o sweep_probe && ./sweep_probe [M]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hitl_pcg_sweep.h"
using namespace hitl;

template <int B, int NS>
__global__ __launch_bounds__(kPcgThreads) void k_sweep(CrView cr, int M, int n, unsigned long long* out, double* sink, void* gran) {
  extern __shared__ double lds[];
  constexpr int BP = B + 1;
  const int tid = threadIdx.x;
  double* w = lds;
  double* xs = w + (size_t)(M + 1) * BP;
  double* ca = xs + (size_t)(M + 1) * BP;
  double* cb = ca + (size_t)(M + 1) * BP;
  double* acc0 = cb + (size_t)(M + 1) * BP;
  double* accM = acc0 + B;
  PcgItems<B, NS> items;
  pcg_items_init<B, NS>(items, cr, M, M - 1, 0, 1);
  for (int e = tid; e < 4 * (M + 1) * BP + 2 * B; e += kPcgThreads) w[e] = 0.0;
  lds_barrier();
  for (int e = tid; e < (M + 1) * BP; e += kPcgThreads) w[e] = 1.0 + 1e-3 * e;
  lds_barrier();
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), t1 = 0, t2 = 0, tf = 0, tb = 0, tp = 0;
  __shared__ double sred[8];
  double acc = 0.0;
  for (int k = 0; k < n; ++k) {
    const unsigned long long a0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 1; s < M; s *= 2) {
      pcg_fwd_products<B, NS>(items, s, w, ca, cb);
      lds_barrier();
      pcg_fwd_survivors<B>(s, M, M - 1, w, ca, cb, acc0, accM);
      lds_barrier();
    }
    const unsigned long long a1 = __builtin_amdgcn_s_memrealtime();
    if (tid < 2 * B) {
      const int q = tid % B;
      if (tid < B) { xs[q] = 1e-3 * acc0[q]; acc0[q] = 0.0; }
      else { xs[M * BP + q] = 1e-3 * accM[q]; accM[q] = 0.0; }
    }
    lds_barrier();
    const unsigned long long a2 = __builtin_amdgcn_s_memrealtime();
    for (int s = M / 2; s >= 1; s /= 2) { pcg_back_level<B, NS>(items, s, w, xs); lds_barrier(); }
    const unsigned long long a3 = __builtin_amdgcn_s_memrealtime();
    double gz = 0.0;
    if (tid < M * B) {
      const double z = xs[(tid / B) * BP + tid % B]; acc += z; w[(tid / B) * BP + tid % B] = 1.0 + 1e-6 * z;
      gz = z * 0.5;
      typedef double v2 __attribute__((ext_vector_type(2)));
      typedef __attribute__((address_space(1))) v2 gv2;
      v2 g; g.x = z; g.y = (double)k;
      *(volatile gv2*)((double*)gran + 2 * tid) = g;                      // the hand-off store of the real kernel (16 bytes, sc0 sc1)
    }
    gz = wave_sum(gz);
    if ((tid & 63) == 0) sred[tid >> 6] = gz;
    lds_barrier();
    if (tid == 0) {
      double t = 0.0;
      for (int q = 0; q < 8; ++q) t += sred[q];
      typedef double v2 __attribute__((ext_vector_type(2)));
      typedef __attribute__((address_space(1))) v2 gv2;
      v2 g; g.x = t; g.y = (double)k;
      *(volatile gv2*)((double*)gran + 2048) = g;
    }
    const unsigned long long a4 = __builtin_amdgcn_s_memrealtime();
    lds_barrier();
    tf += a1 - a0; t1 += a2 - a1; tb += a3 - a2; tp += a4 - a3;
  }
  t2 = __builtin_amdgcn_s_memrealtime();
  sink[tid] = acc;
  if (tid == 0) { out[0] = t2 - t0; out[1] = tf; out[2] = t1; out[3] = tb; out[4] = tp; }
}

int main(int argc, char** argv) {
  constexpr int B = 12;
  const int M = argc > 1 ? atoi(argv[1]) : 16, n = 2000;
  const size_t BB = B * B;
  std::vector<double> h(M * BB);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01 * ((i * 2654435761u) % 97) / 97.0;
  CrView cr;
  cr.nsb = M; cr.B = B;
  hipMalloc(&cr.Gl, h.size() * 8); hipMalloc(&cr.Gr, h.size() * 8); hipMalloc(&cr.Dinv, h.size() * 8);
  hipMemcpy(cr.Gl, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemcpy(cr.Gr, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(cr.Dinv, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  unsigned long long* out; double* sink; void* gran;
  hipMalloc(&gran, 32768);
  hipMalloc(&out, 64); hipMalloc(&sink, 512 * 8);
  const size_t lds = sizeof(double) * (4 * (M + 1) * (B + 1) + 2 * B + 8);
  for (int rep = 0; rep < 2; ++rep) {
    if (6 * B * (M - 1) + 4 <= kPcgThreads * 3) hipLaunchKernelGGL((k_sweep<B, 3>), dim3(1), dim3(kPcgThreads), lds, 0, cr, M, n, out, sink, gran);
    else hipLaunchKernelGGL((k_sweep<B, 5>), dim3(1), dim3(kPcgThreads), lds, 0, cr, M, n, out, sink, gran);
    hipDeviceSynchronize();
  }
  unsigned long long t[5];
  hipMemcpy(t, out, sizeof(t), hipMemcpyDeviceToHost);
  printf("B %d, M %d: sweep %.2f us = forward levels %.2f + boundary hand-over %.2f + back levels %.2f + publish and reduce %.2f\n", B, M, t[0] * 0.01 / n, t[1] * 0.01 / n, t[2] * 0.01 / n,
         t[3] * 0.01 / n, t[4] * 0.01 / n);
  return 0;
}
