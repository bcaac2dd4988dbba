// hitl_host.h -- host-side helpers shared by the .cpp files of libhitlgpu.so (error slot, HIP call checks,
// device arrays, a narrow view of the solver object for the files that do not own it).
#pragma once
#include <algorithm>
#include <string>
#include <vector>

#include "hitl_internal.h"

struct hitl_solver;

// Test hook, exported by libhitlgpu.so but NOT part of the C ABI (include/hitl_gpu.h does not declare it and no caller needs it):
// sets one of the DevSwitches test fields (hitl_internal.h: "cr_legacy", "cr_unfused_build", "stf_addr64", "device_lm", "lm_ahead",
// "kd_cap", "kd_redo_cap", "cand_tf_check", "cand_tile_bytes") for every solver this process creates afterwards;
// name == NULL clears them all.  Returns 0, or -1 for an unknown name.  tests/ drive it through ctypes.
extern "C" int hitl_test_set_switch(const char* name, long long value);

namespace hitl {

int fail(int code, const std::string& msg);   // records the message behind hitl_last_error(); returns code

// roctx ranges around the phases of a call (the counterpart of the reference's FunctionTimer scopes, shared/util/timer.h:135-161;
// SURVEY.md 5.1): with HITL_ROCTX=1 in the environment of hitl_solver_create the library resolves roctxRangePushA / roctxRangePop
// from librocprofiler-sdk-roctx.so (rocprofv3 --marker-trace shows them) at run time; without it, or without that library, a
// range is two untaken branches.  Nothing links against the profiler.
struct RoctxApi { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };
const RoctxApi& roctx_api();
void roctx_resolve(bool enable);            // hitl_solver_create only
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char* name) : on(roctx_api().push != nullptr) { if (on) roctx_api().push(name); }
  ~RoctxRange() { if (on) roctx_api().pop(); }
  RoctxRange(const RoctxRange&) = delete;
  RoctxRange& operator=(const RoctxRange&) = delete;
};

#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail(e_ == hipErrorOutOfMemory ? HITL_ERR_OUT_OF_MEMORY : HITL_ERR_HIP,             \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                           \
  } while (0)

#define TRY(expr)            \
  do {                       \
    int s_ = (expr);         \
    if (s_ != HITL_OK) return s_; \
  } while (0)

template <typename T>
struct DevArray {
  T* p = nullptr;
  size_t cap = 0;
  size_t n = 0;
  DevArray() = default;
  DevArray(const DevArray&) = delete;              // owns device memory: no copies
  DevArray& operator=(const DevArray&) = delete;
  ~DevArray() { release(); }                       // whatever a code path forgot to release() goes with the object
  int resize(size_t count) {
    n = count;
    if (count <= cap && p) return HITL_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = std::max<size_t>(count, 1);
    HIP_TRY(hipMalloc((void**)&p, want * sizeof(T)));
    cap = want;
    return HITL_OK;
  }
  int upload(const T* src, size_t count, hipStream_t st) {
    TRY(resize(count));
    if (count) HIP_TRY(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, st));
    return HITL_OK;
  }
  int upload(const std::vector<T>& v, hipStream_t st) { return upload(v.data(), v.size(), st); }
  int zero(hipStream_t st) {
    if (n) HIP_TRY(hipMemsetAsync(p, 0, n * sizeof(T), st));
    return HITL_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = n = 0;
  }
};

// page-locked host staging buffer that lives as long as its owner (uploads from it are real DMA, queued without a copy through
// a bounce buffer, and need no synchronisation before the function that filled it returns)
template <typename T>
struct PinnedArray {
  T* p = nullptr;
  size_t cap = 0;
  PinnedArray() = default;
  PinnedArray(const PinnedArray&) = delete;
  PinnedArray& operator=(const PinnedArray&) = delete;
  ~PinnedArray() { release(); }
  int resize(size_t count) {
    if (count <= cap && p) return HITL_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = std::max<size_t>(count, 1);
    HIP_TRY(hipHostMalloc((void**)&p, want * sizeof(T)));
    cap = want;
    return HITL_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

// what hitl_cycle.cpp needs from a solver (defined in hitl_solver.cpp)
struct SolverCtx {
  int device;
  hipStream_t stream;
  int n;             // poses of the resident map
  int64_t n_obs;
  int world;
  MapView map;
  int own_lo, own_hi;        // owned local pose range (the whole map on a single GPU)
  int64_t pose_base;         // GLOBAL index of local pose 0 (submap ranks take global pose arrays)
};
int solver_ctx(hitl_solver* s, SolverCtx* out);

// The fixed summation order of the per-workgroup partial sums, shared by the host-driven LM loop and the device-side decision in k_assemble (so that
// both take bit-identical decisions): 32 accumulators take the terms b = lane, lane + 32, ... in ascending order, then the
// 32 accumulators are added from lane 0 up.  v[b * stride] is term b.
static inline double ordered_sum32(const double* v, int n, int stride) {
  double p[32];
  for (int q = 0; q < 32; ++q) p[q] = 0.0;
  for (int b = 0; b < n; ++b) p[b & 31] += v[(size_t)b * stride];
  double t = 0.0;
  for (int q = 0; q < 32; ++q) t += p[q];
  return t;
}
// K sums over records of `stride` doubles (term b of sum k is v[b * stride + k]) in ONE pass over the records: the same order per
// sum as ordered_sum32 (same bits), each cache line of the pinned partials touched once instead of K times
template <int K>
static inline void ordered_sum32_records(const double* v, int n, int stride, double* out) {
  double p[32][K];
  for (int q = 0; q < 32; ++q)
    for (int k = 0; k < K; ++k) p[q][k] = 0.0;
  for (int b = 0; b < n; ++b) {
    const double* r = v + (size_t)b * stride;
    double* pq = p[b & 31];
    for (int k = 0; k < K; ++k) pq[k] += r[k];
  }
  for (int k = 0; k < K; ++k) {
    double t = 0.0;
    for (int q = 0; q < 32; ++q) t += p[q][k];
    out[k] = t;
  }
}
int solver_allgather(hitl_solver* s, const void* send, void* recv, int64_t bytes_per_rank);   // the submap transport, host buffers
// (cos, sin, x, y) of every resident pose in device memory, for the observation scans: the entry points of one correction
// (verify, endpoint fit, observation sets) pass the SAME poses one after the other; the table -- 2 x 50 000 fp32 libm calls on the
// host + an upload -- is rebuilt only when the poses differ from the last call's.  `poses_local` = the rank's resident poses.
int solver_pose_rt(hitl_solver* s, const float* poses_local, const float** T4_dev);

}  // namespace hitl
