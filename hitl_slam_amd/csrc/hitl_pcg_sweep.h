// hitl_pcg_sweep.h -- the preconditioner sweep of the persistent PCG launch (hitl_pcg.hip): z = M^-1 r through the stored cyclic-reduction
// factors, by the 512 threads of a chunk / top workgroup.  A header of its own so that tools/dev/probes/sweep_probe.hip can time exactly
// this code in isolation.
#pragma once
#include "hitl_device.h"

namespace hitl {

constexpr int kPcgThreads = 512;
constexpr int kPcgWaves = kPcgThreads / WAVE;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global store in flight (s_waitcnt vmcnt(0)) --
// here the write-through hand-off stores, 1-2 us each: nothing inside a workgroup depends on them.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// ---- the preconditioner's sweep: coefficients in REGISTERS --------------------------------------------------------------------
// z = M^-1 r through the stored factors is a chain of 2 log2(nodes) dependent levels of tiny products.  Measured (in-kernel probe):
// with the tiles in LDS a level costs ~1 us whatever reads them -- 512 threads with barriers, one wave without; plain layout (5-7 LDS
// passes per read: the nodes of a level lie 1152 bytes apart) or a conflict-free one (its address arithmetic costs what the conflicts
// did).  The factors do not change during a solve, so every thread of a chunk / top workgroup OWNS a few fixed items of the sweep and
// keeps their B coefficients in registers for the whole solve:
//   forward item (level s, node u, which, c): column c of Gl_u (which 0) or Gr_u (1): its dot product with w_u is what node u hands
//                  to its neighbour u - s / u + s;
//   back item (level s, node u, row r, part): row r of Dinv_u (part 0), Gl_u (1) or Gr_u (2) -- the three partial dots of
//                  x_u[r] = Dinv_u[r,:] w_u - Gl_u[r,:] x_{u-s} - Gr_u[r,:] x_{u+s} sit in neighbouring lanes (quads; lane 3 idle) and are
//                  combined by two shuffles.
// Item g lives in thread g mod 512, slot g / 512 (kPcgSlots slots).  Only the small vectors travel through LDS, one barrier per level.
// A forward level is two short passes (products by the items, then the survivors' update by plain thread index) with a barrier each:
// measured faster than correcting a node's right-hand side lazily when it is eliminated (every product lane then re-reads 2 l + 1 vectors).
constexpr int kPcgSlotsMax = 5;     // NS, the slots a thread really carries, is a template parameter: 3 where every item fits (fewer registers)

template <int B, int NS>
struct PcgItems {
  int kind[NS];          // 0 none, 1 forward, 2 back
  int s[NS], u[NS], aux[NS];     // level stride, local node, forward: which * B + c; back: r * 4 + part
  double cf[NS][B];
};

// local node u is global node gbase + u * gstride and exists for u <= u_max; Mloc (a power of two) local nodes 0 .. Mloc - 1
template <int B, int NS>
__device__ __forceinline__ void pcg_items_init(PcgItems<B, NS>& it, const CrView& cr, int Mloc, int u_max, size_t gbase, size_t gstride) {
  constexpr int BB = B * B;
  const int total_f = (Mloc - 1) * 2 * B, g0b = (total_f + 3) & ~3, total_b = (Mloc - 1) * 4 * B;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int g = (int)threadIdx.x + kPcgThreads * j;
    it.kind[j] = 0; it.s[j] = 0; it.u[j] = 0; it.aux[j] = 0;
#pragma unroll
    for (int q = 0; q < B; ++q) it.cf[j][q] = 0.0;
    if (g < total_f) {
      int rem = g, s = 1;
      while (rem >= (Mloc / (2 * s)) * 2 * B) { rem -= (Mloc / (2 * s)) * 2 * B; s *= 2; }
      const int k = rem / (2 * B), wc = rem % (2 * B), u = s * (2 * k + 1);
      it.s[j] = s; it.u[j] = u; it.aux[j] = wc;
      if (u <= u_max) {
        it.kind[j] = 1;
        const double* src = ((wc >= B) ? cr.Gr : cr.Gl) + (gbase + (size_t)u * gstride) * BB + (wc % B);
#pragma unroll
        for (int r = 0; r < B; ++r) it.cf[j][r] = src[r * B];
      }
    } else if (g >= g0b && g < g0b + total_b) {
      int rem = g - g0b, s = 1;
      while (rem >= (Mloc / (2 * s)) * 4 * B) { rem -= (Mloc / (2 * s)) * 4 * B; s *= 2; }
      const int k = rem / (4 * B), rp = rem % (4 * B), r = rp / 4, part = rp % 4, u = s * (2 * k + 1);
      it.s[j] = s; it.u[j] = u; it.aux[j] = rp;
      if (u <= u_max && part < 3) {
        it.kind[j] = 2;
        const double* src = ((part == 0) ? cr.Dinv : (part == 1) ? cr.Gl : cr.Gr) + (gbase + (size_t)u * gstride) * BB + r * B;
#pragma unroll
        for (int c = 0; c < B; ++c) it.cf[j][c] = src[c];
      }
    }
    // the coefficients are loaded HERE, once: without this the compiler sinks the loads into the level loops (it sees loop-invariant
    // loads from read-only memory and saves the registers): every level then waited for memory -- ~0.8 us per level, measured
#pragma unroll
    for (int q = 0; q < B; ++q) asm volatile("" : "+v"(it.cf[j][q]));
  }
}

// the forward products of the level of stride s: every forward item of that level hands ca[u] = Gl_u^T w_u (for node u - s) or
// cb[u] = Gr_u^T w_u (for node u + s) over; a barrier follows, then pcg_fwd_survivors
template <int B, int NS>
__device__ __forceinline__ void pcg_fwd_products(const PcgItems<B, NS>& it, int s, const double* w, double* ca, double* cb) {
  constexpr int BP = B + 1;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    if (it.kind[j] != 1 || it.s[j] != s) continue;
    const int u = it.u[j], which = it.aux[j] / B, c = it.aux[j] % B;
    const double* wu = w + u * BP;
    double d[4] = {0.0, 0.0, 0.0, 0.0};              // four chains: a dependent fp64 FMA chain of B links is what a level waits for
#pragma unroll
    for (int r = 0; r < B; ++r) d[r & 3] += it.cf[j][r] * wu[r];
    (which ? cb : ca)[u * BP + c] = (d[0] + d[1]) + (d[2] + d[3]);
  }
}

// the survivors of the level of stride s (nodes j = 0, 2 s, 4 s, ...) take what their eliminated neighbours handed over:
// w_j = (w_j - cb[j - s]) - ca[j + s].  Node 0 and node Mloc (chunk boundaries: they belong to the top) accumulate into acc0 / accM
// when those are given; with acc0 == nullptr node 0 is an ordinary survivor (the top's root) and there is no node Mloc.
template <int B>
__device__ __forceinline__ void pcg_fwd_survivors(int s, int Mloc, int u_max, double* w, const double* ca, const double* cb, double* acc0, double* accM) {
  constexpr int BP = B + 1;
  const int nsv = Mloc / (2 * s) + (accM != nullptr ? 1 : 0);
  for (int e = threadIdx.x; e < nsv * B; e += kPcgThreads) {
    const int j = 2 * s * (e / B), q = e % B;
    const double pl = (j - s >= 1 && j - s <= u_max) ? cb[(j - s) * BP + q] : 0.0;
    const double pr = (j + s <= u_max && j + s < Mloc) ? ca[(j + s) * BP + q] : 0.0;
    if (j == 0 && acc0 != nullptr) acc0[q] += pr;
    else if (j == Mloc) accM[q] += pl;
    else if (j <= u_max) w[j * BP + q] = (w[j * BP + q] - pl) - pr;
  }
}

// value of the lane `off` (1 or 2) further up in the same quad (lanes 4 k .. 4 k + 3), by DPP: no LDS crossbar
template <int OFF>
__device__ __forceinline__ double quad_down(double v) {
  constexpr int ctrl = (OFF == 1) ? (1 | (2 << 2) | (3 << 4) | (3 << 6)) : (2 | (3 << 2) | (3 << 4) | (3 << 6));
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// the back-substitution level of stride s: x_u[r] = Dinv_u[r,:] w_u - Gl_u[r,:] x_{u-s} - Gr_u[r,:] x_{u+s} (xs holds zeros at nodes that
// do not exist; a barrier follows)
template <int B, int NS>
__device__ __forceinline__ void pcg_back_level(const PcgItems<B, NS>& it, int s, const double* w, double* xs) {
  constexpr int BP = B + 1;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const bool act = (it.kind[j] == 2 && it.s[j] == s);
    if (__builtin_amdgcn_ballot_w64(act) == 0ull) continue;          // wave-uniform: the quad shuffles below are executed by whole quads
    const int u = it.u[j], r = it.aux[j] / 4, part = it.aux[j] % 4;
    double d[4] = {0.0, 0.0, 0.0, 0.0};
    if (act) {
      const double* vec = (part == 0) ? w + u * BP : (part == 1) ? xs + (u - s) * BP : xs + (u + s) * BP;
#pragma unroll
      for (int c = 0; c < B; ++c) d[c & 3] += it.cf[j][c] * vec[c];
    }
    const double a = (d[0] + d[1]) + (d[2] + d[3]);
    const double p1 = quad_down<1>(a), p2 = quad_down<2>(a);
    if (act && part == 0) xs[u * BP + r] = (a - p1) - p2;
  }
}

}  // namespace hitl
