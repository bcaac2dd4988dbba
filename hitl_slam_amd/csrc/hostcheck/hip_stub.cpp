// hostcheck/hip_stub.cpp -- a STUB DEVICE LAYER for `make hostcheck` (CPU only, AddressSanitizer + UBSan over the host halves of
// libhitlgpu.so).  TEST INFRASTRUCTURE: never compiled into or linked with libhitlgpu.so, never on a product path.
//  * the HIP runtime entry points the host code calls: "device" memory is host memory (calloc), copies are memcpy, streams and
//    events are tokens, one device is reported -- so the host control flow (argument checks, table building, staging buffers, the
//    LM loop's bookkeeping, the file formats) runs end to end under the sanitizers;
//  * the RCCL entry points: refuse;
//  * every kernel launcher: an empty function (gen_launch_stubs.py), except the two that raise the flag the host waits for;
//  * the few size helpers that live next to the kernels: generous upper bounds.
// Numerical results under this layer are meaningless (no kernel runs): the target only reports memory errors and undefined
// behaviour of the HOST code.
#include <cstdlib>
#include <cstring>

#include "../hitl_internal.h"
#include <rccl/rccl.h>

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) { *v = (a == hipDeviceAttributeMultiprocessorCount) ? 256 : 65536; return hipSuccess; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)64 << 30; *t = (size_t)288 << 30; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(calloc(1, 8)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(calloc(1, 8)); return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { memset(a, 0, sizeof(*a)); a->type = hipMemoryTypeUnregistered; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(calloc(1, 8)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "hostcheck stub"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 0, sizeof(*id)); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int) { return ncclSystemError; }
ncclResult_t ncclCommDestroy(ncclComm_t) { return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t, int* n) { *n = 1; return ncclSuccess; }
ncclResult_t ncclAllGather(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) { return ncclSystemError; }
const char* ncclGetErrorString(ncclResult_t) { return "hostcheck stub"; }
}

namespace hitl {
// the two launchers whose kernels raise the pinned flag the host polls (everything else: _gen_launch_stubs.inc)
void launch_host_flag(hipStream_t, double* flag_pinned, double seq) { *flag_pinned = seq; }
// (the linearisation also fakes a slowly falling cost and a gradient that never vanishes, so that the LM loop's accept / reject /
// trust-region bookkeeping iterates to its cap instead of stopping at iteration 0 on an all-zero problem)
void launch_assemble(hipStream_t, const MapView&, const PairView&, const FactorView&, const double*, const LinView& lin, const LmDev*, const double*, const LinView*,
                     const LmDecide* decide) {
  static int calls = 0;
  ++calls;
  if (lin.partials != nullptr) { lin.partials[0] = 1.0 / (1.0 + calls); lin.partials[1] = 1.0; }
  if (decide != nullptr && decide->lm == nullptr && decide->host_flag != nullptr) *decide->host_flag = decide->flag_seq;
}
// size helpers that live next to the kernels: upper bounds are enough here
int cr_contrib_slots(int nsb) { return nsb / 2 + 3; }
int spmv_blocks(int npad) { return ((npad + 2) / 3 + 127) / 128; }
int pair_scan_blocks(int n) { return (n + 255) / 256; }
int fit_eval_blocks(int64_t m) { const int64_t w = (m + 1023) / 1024; return (int)(w < 1 ? 1 : (w > 256 ? 256 : w)); }
int backprop_blocks(int L) { return (L + 1 + 63) / 64 + 1; }
size_t kd_build_lds_bytes(int max_scan) { return (size_t)((max_scan + 3) & ~3) * 26; }
size_t match_lds_bytes(int max_scan) { return (size_t)(max_scan + 8) * 24; }
size_t match_kd_lds_bytes(int max_scan) { return (size_t)(max_scan + 8) * 24 + 64 * 1024; }
size_t match_kd_tgt_lds_bytes(int max_scan, int block, bool) { return (size_t)max_scan * 20 + (size_t)6 * block * 12; }
bool match_kd_tgt_applies(int slots, int window, const MatchParams& mp) { return window >= 1 && 2 * window <= 16 && slots <= 2 * window && slots <= mp.cap && mp.skip >= 1; }
size_t stf_pose_rec_bytes() { return 128; }
bool cr_build_factor_solve_is_fused(const CrView&, const CrSource&, double*, double*) { return true; }
bool cr_sep_applies(const CrView& chain, const CrSource&, double*, double*) { return chain.rsep && chain.nsb >= 2; }
bool pcg_fused_plan(int, int, int, int, PcgPlan*) { return false; }      // the stub device runs no kernels: the host keeps the launch-per-stage loop
size_t pcg_fused_sync_words() { return 32; }
size_t kd_grid_words() { return 8 + 1024; }
bool match_kd_takes_order(int) { return true; }
bool launch_pcg_fused(hipStream_t, const PcgPlan&, int, int, const FactorView&, const LinView&, const CrView&, const double*, double, double, double, double, int,
                      double*, const double*, const double*, int, const int*, int, const unsigned char*, double*, unsigned*, uint64_t, double*, unsigned, bool, unsigned long long*) { return false; }
const char* kernels_build_info() { return "hostcheck stub (no device code)"; }
}  // namespace hitl

#include "_gen_launch_stubs.inc"
