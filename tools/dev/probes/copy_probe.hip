// Development probe: device-to-device copy rates of a few float4 copy kernels (which form reaches the guide's 6.3 TB/s ceiling?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k_one(size_t n, const float4* __restrict__ s, float4* __restrict__ d) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_stride4(size_t n, const float4* __restrict__ s, float4* __restrict__ d) {
  const size_t st = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * st < n; i += 4 * st) { const float4 a = s[i], b = s[i + st], c = s[i + 2 * st], e = s[i + 3 * st]; d[i] = a; d[i + st] = b; d[i + 2 * st] = c; d[i + 3 * st] = e; }
  for (; i < n; i += st) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_chunk4(size_t n, const float4* __restrict__ s, float4* __restrict__ d) {   // a workgroup copies 4 KB x 4 contiguous
  size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
  if (i + 768 < n) { const float4 a = s[i], b = s[i + 256], c = s[i + 512], e = s[i + 768]; d[i] = a; d[i + 256] = b; d[i + 512] = c; d[i + 768] = e; }
  else for (int k = 0; k < 4; ++k) if (i + 256 * k < n) d[i + 256 * k] = s[i + 256 * k];
}
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_chunk4_nt(size_t n, const float4* __restrict__ s_, float4* __restrict__ d_) {
  const v4f* s = reinterpret_cast<const v4f*>(s_); v4f* d = reinterpret_cast<v4f*>(d_);
  size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
  if (i + 768 < n) {
    const v4f a = __builtin_nontemporal_load(s + i), b = __builtin_nontemporal_load(s + i + 256), c = __builtin_nontemporal_load(s + i + 512), e = __builtin_nontemporal_load(s + i + 768);
    __builtin_nontemporal_store(a, d + i); __builtin_nontemporal_store(b, d + i + 256); __builtin_nontemporal_store(c, d + i + 512); __builtin_nontemporal_store(e, d + i + 768);
  }
}
template <typename F> static double run(F launch, size_t bytes) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int k = 0; k < 3; ++k) launch();
  hipEventRecord(e0, 0);
  for (int k = 0; k < 10; ++k) launch();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return 2.0 * bytes * 10 / (ms * 1e-3) / 1e9;
}
int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? atoll(argv[1]) : 1024) * (size_t)(1 << 20), n = bytes / 16;
  float4 *a, *b;
  hipMalloc(&a, bytes + 65536); hipMalloc(&b, bytes + 65536);
  hipMemset(a, 1, bytes);
  int cus = 256;
  printf("%zu MiB per buffer\n", bytes >> 20);
  printf("one float4 per thread          %7.0f GB/s\n", run([&] { hipLaunchKernelGGL(k_one, dim3((n + 255) / 256), dim3(256), 0, 0, n, a, b); }, bytes));
  for (int w : {4, 8, 16, 32}) printf("grid-stride x4, %2d WG per CU    %7.0f GB/s\n", w, run([&] { hipLaunchKernelGGL(k_stride4, dim3(cus * w), dim3(256), 0, 0, n, a, b); }, bytes));
  printf("16 KB chunk per workgroup      %7.0f GB/s\n", run([&] { hipLaunchKernelGGL(k_chunk4, dim3((n + 1023) / 1024), dim3(256), 0, 0, n, a, b); }, bytes));
  printf("16 KB chunk, nontemporal       %7.0f GB/s\n", run([&] { hipLaunchKernelGGL(k_chunk4_nt, dim3((n + 1023) / 1024), dim3(256), 0, 0, n, a, b); }, bytes));
  printf("16 KB chunk, dst + 4 KB offset %7.0f GB/s\n", run([&] { hipLaunchKernelGGL(k_chunk4, dim3((n + 1023) / 1024), dim3(256), 0, 0, n, a, b + 256); }, bytes));
  printf("hipMemcpyDtoD                  %7.0f GB/s\n", run([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, bytes));
  return 0;
}
