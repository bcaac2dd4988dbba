// hitl_pcg.hip -- the sparse PCG solve of the LM step as ONE persistent launch ("fused single-kernel
// persistent CG for small N to dodge launch latency").
//
// Which systems: pose pairs outside the band (the reference's all-pairs search, JointOptimization.cpp:561-642, on a map that revisits
// places: every real hitl-slam dataset).  The matrix is A = M + O: M the block-tridiagonal band part with the LM diagonal, factorised
// exactly by the block cyclic reduction in front of this launch (hitl_cr.hip: Dinv / Gl / Gr per node of the reduction tree), O the 3x3
// blocks of the off-band pose pairs.  CG on A preconditioned by M: what Ceres' SPARSE_SCHUR + the reference's options solve exactly
// (JointOptimization.cpp:145-162, 1208) is here solved to hitl_options.linear_solver_tolerance (1e-12).
//
// Rounds 2-5 ran one CG iteration as EIGHT launches (four preconditioner stages, r.z + direction, off-band product, band product,
// p.q + updates + r.r: ~58 us per CG iteration on a 1 000-pose Figure-8 map, nearly all of it launch boundaries and cold loads of the
// same factor tiles) and the host looked at r.r every fourth iteration.  Here the whole solve is one launch of G <= #CU workgroups
// with fixed roles that hand small vectors to each other through device memory:
//
//   * chunk workgroups c = 0 .. chunks-1: each owns M consecutive nodes of the reduction tree (M * Wb poses).  Their Dinv / Gl / Gr
//     tiles are loaded into LDS ONCE and stay there for every CG iteration of the solve; x, r, p, s = A p of the owned rows live in LDS
//     too.  Per iteration: the vector updates, the local levels of the preconditioner's forward sweep, later its back-substitution.
//   * the top workgroup (G - 1): the nodes 0, M, 2M, ... (one per chunk), tiles resident in LDS: the upper levels of the sweep.
//   * every workgroup but the top: its share of the product w = A z, one wave per pose over the pose's band and off-band blocks.
//
// The recurrence is Chronopoulos & Gear's single-reduction form of PCG (z = M^-1 r, w = A z, gamma = r.z, delta = w.z,
// beta = gamma / gamma_old, alpha = gamma / (delta - beta gamma / alpha_old), p = z + beta p, s = w + beta s, x += alpha p, r -= alpha s):
// the same iterates as textbook PCG in exact arithmetic, with ONE all-to-all reduction per iteration instead of two.  A CG iteration is
// four hand-offs: chunks -> top (boundary contributions), top -> chunks (z at the chunk boundaries), chunks -> everybody (z, gamma, r.r),
// everybody -> chunks (w, delta).  Every handed-off double is written and read with agent-scope atomic accesses (sc1: past the
// per-XCD L2s, which are not coherent with each other) and announced through monotonic counters -- no fences, no L2 write-backs.
// Sums across workgroups are formed by every reader in the same fixed order: all workgroups take identical decisions, and a solve
// is bit-reproducible for a given launch geometry.  All waits are bounded; a wait that runs out (the workgroups were not all
// resident: a device shared with another process) marks the solve as void and the host repeats it on the launch-per-stage path.
//
// fp64 throughout.  No CPU fallback, no oracle: this is the product path.

#include <algorithm>

#include "hitl_device.h"

namespace hitl {

constexpr int kPcgThreads = 512;
constexpr int kPcgWaves = kPcgThreads / WAVE;

// ---- hand-off granules ------------------------------------------------------------------------------------------------------
// Every double that crosses from one workgroup to another travels as a 16-byte granule {value, tag}: ONE 16-byte store, ONE 16-byte
// load, both past every cache level (volatile: sc0 sc1).  tag = the solve's epoch base + the CG iteration: a reader spins on the
// granule itself until it carries the tag it is waiting for -- no counter, no drain, no barrier on the producer's side, one memory
// round trip from store to use.  (A 16-byte aligned access is one transaction of one cache line; observed untorn on gfx950.)  Tags grow
// monotonically over the life of the solver (the base advances by 2^21 per launch), so buffers are zeroed once, when allocated.
typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) v2f64 gv2f64;       // device memory, said so in the type: global_ instructions, not flat_ ones
__device__ __forceinline__ v2f64 gran_ld(const v2f64* p) { return *(const volatile gv2f64*)p; }
__device__ __forceinline__ void gran_st(v2f64* p, double v, double tag) {
  v2f64 g;
  g.x = v; g.y = tag;
  *(volatile gv2f64*)p = g;
}

struct PcgPoll { unsigned* abort_w; unsigned spins; unsigned launch; };   // abort word = launch << 2 | code: a word left by an earlier launch means nothing
__device__ __forceinline__ unsigned pcg_abort_code(const PcgPoll& pp) {
  const unsigned v = __hip_atomic_load(pp.abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return ((v >> 2) == pp.launch) ? (v & 3u) : 0u;
}
__device__ __forceinline__ void pcg_abort_set(const PcgPoll& pp, unsigned code) {
  __hip_atomic_store(pp.abort_w, (pp.launch << 2) | code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// spin until the granule carries `tag`.  Every wait is bounded; the abort word ends all of them: 1 = the chunk workgroups have ended the
// solve (breakdown), 2 = somebody's wait ran out.  SLEEP: polls of data that is usually a while away back off between loads.
template <bool SLEEP>
__device__ __forceinline__ double gran_take(const v2f64* p, double tag, const PcgPoll& pp, bool& bad) {
  for (unsigned k = 1;; ++k) {
    const v2f64 g = gran_ld(p);
    if (g.y == tag) return g.x;
    if ((k & 63u) == 0u) {
      if (k >= pp.spins) { pcg_abort_set(pp, 2u); bad = true; return 0.0; }
      if (pcg_abort_code(pp) != 0u) { bad = true; return 0.0; }
    }
    if (SLEEP) __builtin_amdgcn_s_sleep(2);
  }
}

// sum over the workgroup of one value per thread, fixed order; result in every thread.  sm: kPcgWaves doubles of LDS.
__device__ __forceinline__ double pcg_block_sum(double v, double* sm) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < kPcgWaves; ++k) t += sm[k];
  return t;
}

// One level of the forward sweep over local nodes held in LDS (node u at index u; tiles of node u at (u - toff) * BB).
// Eliminated: u with lowbit(u) == s, 1 <= u <= u_max.  ca[u] = Gl_u^T w_u goes to node u - s, cb[u] = Gr_u^T w_u to node u + s.
template <int B>
__device__ __forceinline__ void pcg_fwd_products(const double* sGl, const double* sGr, const double* w, double* ca, double* cb, int s, int Mloc, int u_max,
                                                 int toff) {
  constexpr int BB = B * B;
  const int nel = Mloc / (2 * s);
  for (int e = threadIdx.x; e < nel * 2 * B; e += kPcgThreads) {
    const int u = s * (2 * (e / (2 * B)) + 1), which = (e / B) & 1, c = e % B;
    if (u > u_max) continue;
    const double* G = (which ? sGr : sGl) + (size_t)(u - toff) * BB;
    const double* wu = w + (size_t)u * B;
    double cw = 0.0;
#pragma unroll
    for (int r = 0; r < B; ++r) cw += G[r * B + c] * wu[r];
    (which ? cb : ca)[(size_t)u * B + c] = cw;
  }
}

// One level of the back-substitution: x_u = Dinv_u w_u - Gl_u x_{u-s} - Gr_u x_{u+s} for lowbit(u) == s, 1 <= u <= u_max.  xs holds
// zeros at nodes that do not exist.
template <int B>
__device__ __forceinline__ void pcg_back_level(const double* sDi, const double* sGl, const double* sGr, const double* w, double* xs, int s, int Mloc, int u_max,
                                               int toff) {
  constexpr int BB = B * B;
  const int nel = Mloc / (2 * s);
  for (int e = threadIdx.x; e < nel * B; e += kPcgThreads) {
    const int u = s * (2 * (e / B) + 1), r = e % B;
    if (u > u_max) continue;
    const double* Di = sDi + (size_t)(u - toff) * BB + r * B;
    const double* Gl = sGl + (size_t)(u - toff) * BB + r * B;
    const double* Gr = sGr + (size_t)(u - toff) * BB + r * B;
    const double* wu = w + (size_t)u * B;
    const double* xl = xs + (size_t)(u - s) * B;
    const double* xr = xs + (size_t)(u + s) * B;
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < B; ++c) acc += Di[c] * wu[c];
#pragma unroll
    for (int c = 0; c < B; ++c) acc -= Gl[c] * xl[c];
#pragma unroll
    for (int c = 0; c < B; ++c) acc -= Gr[c] * xr[c];
    xs[(size_t)u * B + r] = acc;
  }
}

struct PcgArgs {
  int n_poses, npad, nsb, Wb;
  int M, chunks, Tp, G;          // chunk size (a power of two), chunks = top nodes, Tp = the power of two >= chunks, workgroups
  double mu, min_diag, max_diag, tol2;
  double tag0;                   // epoch base of this launch: the granules of CG iteration `it` carry tag0 + it + 1
  int max_it;
  unsigned spins;
  double* x;                     // in: x0 = M^-1 b; out: the solution
  const double* r0;              // b - A x0
  const double* partials0;       // (r0.r0, b.b, ...) per workgroup of the residual pass, 8 doubles apart
  int nblk0;
  // granule arrays (device memory, zero when allocated)
  v2f64* zsG;                    // npad: scale * z, gathered by the products
  v2f64* wG;                     // npad: w = A z, taken by the rows' chunk workgroups
  v2f64* ctrG;                   // chunks x 3 B: contribution to the chunk's own first node | to the next chunk's first node | r at the first node
  v2f64* topL;                   // chunks x B: z at the chunk's first node
  v2f64* topR;                   // chunks x B: z at the NEXT chunk's first node (the chunk's right boundary)
  v2f64* gamG; v2f64* rrG;       // chunks each: r.z and r.r of the chunk's rows
  v2f64* delG;                   // G: w.z of the workgroup's product rows
  unsigned* abort_w;
  unsigned launch;               // low 30 bits of the solver's launch counter (see PcgPoll)
  double* result;                // pinned host memory: iterations, r.r, status (0 ok, 1 breakdown, 2 a wait ran out, 3 not finite), b.b
  unsigned long long* probe;     // HITL_TIMING: 16 timestamps (100 MHz) per workgroup around the phases of CG iteration 2, or null
};

#define PCG_PROBE(slot) do { if (a.probe != nullptr && it == 2 && tid == 0) a.probe[(size_t)wg * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)

template <int B>
__global__ __launch_bounds__(kPcgThreads) void k_pcg_fused(PcgArgs a, FactorView fv, LinView lin, CrView cr, const double* __restrict__ scale) {
  constexpr int BB = B * B;
  extern __shared__ double lds[];
  __shared__ double s_red[kPcgWaves];
  __shared__ double s_sc[8];               // [0] rr, [1] gamma, [2] delta, [6] bb
  const int tid = threadIdx.x, wg = blockIdx.x, wave = tid >> 6, lane = tid & 63;
  const int M = a.M, T = a.chunks, Tp = a.Tp, H = a.G - 1;
  const bool is_top = (wg == a.G - 1), is_chunk = (wg < a.chunks);
  const PcgPoll pp = {a.abort_w, a.spins, a.launch};

  // ---- LDS carve-up (one dynamic block sized for the larger role) ----
  // chunk: tiles of the interior nodes 1 .. M-1 | w, xs (M + 1 nodes) | ca, cb (M nodes) | xv rv pv sv zl (M nodes) | accL0, accRM
  // top:   tiles of the nodes 0 .. T-1          | w (Tp nodes), xs (2 Tp + 1 nodes: zeros beyond T - 1) | ca, cb (Tp nodes)
  double *sGl, *sGr, *sDi, *w, *xs, *ca, *cb, *xv = nullptr, *rv = nullptr, *pv = nullptr, *sv = nullptr, *zl = nullptr, *accL0 = nullptr, *accRM = nullptr;
  int nv = 0, nvB = 0;
  if (is_top) {
    sGl = lds; sGr = sGl + (size_t)T * BB; sDi = sGr + (size_t)T * BB;
    w = sDi + (size_t)T * BB; xs = w + (size_t)Tp * B; ca = xs + (size_t)(2 * Tp + 1) * B; cb = ca + (size_t)Tp * B;
  } else {
    sGl = lds; sGr = sGl + (size_t)(M - 1) * BB; sDi = sGr + (size_t)(M - 1) * BB;
    w = sDi + (size_t)(M - 1) * BB; xs = w + (size_t)(M + 1) * B; ca = xs + (size_t)(M + 1) * B; cb = ca + (size_t)M * B;
    xv = cb + (size_t)M * B; rv = xv + (size_t)M * B; pv = rv + (size_t)M * B; sv = pv + (size_t)M * B; zl = sv + (size_t)M * B;
    accL0 = zl + (size_t)M * B; accRM = accL0 + B;
    if (is_chunk) { nv = min(M, a.nsb - wg * M); nvB = nv * B; }
  }
  const size_t row0 = (size_t)wg * M * B;           // first owned row of a chunk workgroup

  // ---- set-up: tiles into LDS (they stay for the whole solve), the owned rows of x0 and r0 ----
  if (is_top) {
    for (int e = tid; e < T * BB; e += kPcgThreads) {
      const size_t g = (size_t)(e / BB) * M * BB + (e % BB);
      sGl[e] = cr.Gl[g]; sGr[e] = cr.Gr[g]; sDi[e] = cr.Dinv[g];
    }
    for (int e = tid; e < (2 * Tp + 1) * B; e += kPcgThreads) xs[e] = 0.0;
  } else if (is_chunk) {
    for (int e = tid; e < (nv - 1) * BB; e += kPcgThreads) {
      const size_t g = ((size_t)wg * M + 1) * BB + e;
      sGl[e] = cr.Gl[g]; sGr[e] = cr.Gr[g]; sDi[e] = cr.Dinv[g];
    }
    for (int e = tid; e < (M + 1) * B; e += kPcgThreads) { xs[e] = 0.0; w[e] = 0.0; }
    for (int t = tid; t < M * B; t += kPcgThreads) {
      const bool own = t < nvB;
      xv[t] = own ? a.x[row0 + t] : 0.0; rv[t] = own ? a.r0[row0 + t] : 0.0; pv[t] = 0.0; sv[t] = 0.0; zl[t] = 0.0;
    }
  }
  // ---- the product rows of this wave: pose i0 first, then every (H * kPcgWaves)-th.  The matrix does not change during the solve: the
  //      first pose's first entry per lane (its 3x3 block, its column) and the row constants stay in registers for every iteration ----
  const int W1 = a.Wb + 1, nband = 2 * a.Wb + 1, n_rows = a.npad / 3;
  const int i0 = wg * kPcgWaves + wave;
  const bool have_i0 = !is_top && i0 < n_rows;
  const bool ident0 = have_i0 && (i0 >= a.n_poses || fv.fixed[i0]);
  int tot0 = 0, e00 = 0, j0 = 0, tr0 = 0;
  bool ent0 = false;
  double bk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double sa0 = 1.0, dd0 = 0.0;             // lanes 0..2: scale of row 3 i0 + lane; clamp(sa^2 H_aa) / mu / sa
  if (have_i0 && !ident0) {
    e00 = (fv.n_offband > 0) ? fv.obadj_off[i0] : 0;
    tot0 = nband + ((fv.n_offband > 0) ? fv.obadj_off[i0 + 1] - e00 : 0);
    if (lane < tot0) {
      const double* blk = nullptr;
      if (lane < nband) {
        if (lane <= a.Wb) { j0 = i0 + lane; tr0 = 0; if (j0 < a.n_poses) blk = lin.Hband + ((size_t)i0 * W1 + lane) * 9; }
        else { const int d = lane - a.Wb; j0 = i0 - d; tr0 = 1; if (j0 >= 0) blk = lin.Hband + ((size_t)j0 * W1 + d) * 9; }
      } else {
        const int e = e00 + lane - nband, ent = fv.obadj_ent[e];
        j0 = fv.obadj_col[e]; tr0 = ent & 1; blk = lin.obblk + (size_t)(ent >> 1) * 9;
      }
      if (blk != nullptr) {
        ent0 = true;
#pragma unroll
        for (int q = 0; q < 9; ++q) bk[q] = blk[q];
      }
    }
    if (lane < 3) {
      sa0 = scale[3 * i0 + lane];
      const double dg = sa0 * sa0 * lin.Hband[(size_t)i0 * W1 * 9 + lane * 4];
      dd0 = fmin(fmax(dg, a.min_diag), a.max_diag) / a.mu / sa0;
    }
  }
  {  // r0.r0 and b.b from the residual pass in front of this launch (a kernel boundary: plain data), every workgroup alike
    double r_ = 0.0, b_ = 0.0;
    if (tid < WAVE) {
      for (int k = tid; k < a.nblk0; k += WAVE) { r_ += a.partials0[(size_t)k * 8]; b_ += a.partials0[(size_t)k * 8 + 1]; }
      r_ = wave_sum(r_); b_ = wave_sum(b_);
    }
    if (tid == 0) { s_sc[0] = r_; s_sc[6] = b_; }
  }
  __syncthreads();
  const double bb = s_sc[6];
  const double target = a.tol2 * bb;
  double rr = s_sc[0];
  double alpha = 0.0, alpha_old = 0.0, gamma_old = 0.0;
  int status = 0;
  int it = 0;
  if (!(rr == rr)) status = 3;
  const bool run = (status == 0) && (rr > target) && a.max_it > 0;

  while (run) {
    const double tag = a.tag0 + (double)(it + 1);
    bool bad = false;
    double rr_c = 0.0;
    PCG_PROBE(0);
    if (is_chunk) {
      // ---- P1: x += alpha p, r -= alpha s; r.r; the local levels of the forward sweep on r ----
      double rr_t = 0.0;
      for (int t = tid; t < nvB; t += kPcgThreads) {
        if (it > 0) { xv[t] += alpha * pv[t]; rv[t] -= alpha * sv[t]; }
        const double rt = rv[t];
        w[t] = rt;
        rr_t += rt * rt;
      }
      if (tid < 2 * B) accL0[tid] = 0.0;                 // accL0 and accRM are adjacent
      rr_c = pcg_block_sum(rr_t, s_red);                 // (its barriers also publish w and the cleared accumulators)
      for (int s = 1; s < M; s *= 2) {
        pcg_fwd_products<B>(sGl, sGr, w, ca, cb, s, M, nv - 1, 1);
        __syncthreads();
        const int nsv = M / (2 * s) + 1;
        for (int e = tid; e < nsv * B; e += kPcgThreads) {
          const int j = 2 * s * (e / B), q = e % B;
          const double pl = (j - s >= 1 && j - s <= nv - 1) ? cb[(size_t)(j - s) * B + q] : 0.0;
          const double pr = (j + s <= nv - 1 && j + s < M) ? ca[(size_t)(j + s) * B + q] : 0.0;
          if (j == 0) accL0[q] += pr;
          else if (j == M) accRM[q] += pl;
          else if (j <= nv - 1) w[(size_t)j * B + q] = (w[(size_t)j * B + q] - pl) - pr;
        }
        __syncthreads();
      }
      if (tid < 3 * B) {
        const double v = (tid < B) ? accL0[tid] : (tid < 2 * B) ? accRM[tid - B] : rv[tid - 2 * B];
        gran_st(a.ctrG + (size_t)wg * 3 * B + tid, v, tag);
      }
      PCG_PROBE(1);
      // ---- P3: z at the two boundary nodes comes back from the top; back-substitution; r.z ----
      if (tid < 2 * B) {
        const int q = tid % B;
        if (tid < B) xs[q] = gran_take<true>(a.topL + (size_t)wg * B + q, tag, pp, bad);
        else xs[(size_t)M * B + q] = (wg + 1 < T) ? gran_take<true>(a.topR + (size_t)wg * B + q, tag, pp, bad) : 0.0;
      }
      if (__syncthreads_or(bad ? 1 : 0)) { status = 5; break; }
      PCG_PROBE(2);
      for (int s = M / 2; s >= 1; s /= 2) {
        pcg_back_level<B>(sDi, sGl, sGr, w, xs, s, M, nv - 1, 1);
        __syncthreads();
      }
      double gz = 0.0;
      for (int t = tid; t < nvB; t += kPcgThreads) {
        const double z = xs[t];
        zl[t] = z;
        gz += rv[t] * z;
        gran_st(a.zsG + row0 + t, scale[row0 + t] * z, tag);
      }
      const double gamma_c = pcg_block_sum(gz, s_red);
      if (tid == 0) { gran_st(a.gamG + wg, gamma_c, tag); gran_st(a.rrG + wg, rr_c, tag); }
      PCG_PROBE(3);
    } else if (is_top) {
      // ---- P2: the upper levels: nodes 0, M, 2M, ... ----
      for (int e = tid; e < T * B; e += kPcgThreads) {
        const int t = e / B, q = e % B;
        double v = gran_take<true>(a.ctrG + (size_t)t * 3 * B + 2 * B + q, tag, pp, bad);
        v -= gran_take<false>(a.ctrG + (size_t)t * 3 * B + q, tag, pp, bad);
        if (t >= 1) v -= gran_take<true>(a.ctrG + (size_t)(t - 1) * 3 * B + B + q, tag, pp, bad);
        w[e] = v;
      }
      if (__syncthreads_or(bad ? 1 : 0)) { status = 5; break; }
      PCG_PROBE(1);
      for (int s = 1; s < Tp; s *= 2) {
        pcg_fwd_products<B>(sGl, sGr, w, ca, cb, s, Tp, T - 1, 0);
        __syncthreads();
        const int nsv = Tp / (2 * s);
        for (int e = tid; e < nsv * B; e += kPcgThreads) {
          const int j = 2 * s * (e / B), q = e % B;
          if (j > T - 1) continue;
          const double pl = (j - s >= 1) ? cb[(size_t)(j - s) * B + q] : 0.0;
          const double pr = (j + s <= T - 1) ? ca[(size_t)(j + s) * B + q] : 0.0;
          w[(size_t)j * B + q] = (w[(size_t)j * B + q] - pl) - pr;
        }
        __syncthreads();
      }
      if (tid < B) {
        double v = 0.0;
#pragma unroll
        for (int c = 0; c < B; ++c) v += sDi[tid * B + c] * w[c];
        xs[tid] = v;
      }
      __syncthreads();
      for (int s = Tp / 2; s >= 1; s /= 2) {
        pcg_back_level<B>(sDi, sGl, sGr, w, xs, s, Tp, T - 1, 0);
        __syncthreads();
      }
      PCG_PROBE(2);
      for (int e = tid; e < T * B; e += kPcgThreads) {
        const double v = xs[e];
        gran_st(a.topL + e, v, tag);
        if (e >= B) gran_st(a.topR + (e - B), v, tag);
      }
      PCG_PROBE(3);
    }
    // ---- everybody: r.r is out with z; is the solve over? ----
    if (tid < WAVE) {
      double v = 0.0;
      if (tid == 0) v = gran_take<true>(a.rrG, tag, pp, bad);               // one poller per workgroup until the first chunk has published ...
      for (int c = tid + (tid == 0 ? WAVE : 0); c < T; c += WAVE) v += gran_take<false>(a.rrG + c, tag, pp, bad);   // ... then the others' (on their way)
      v = wave_sum(v);
      if (tid == 0) s_sc[0] = v;
    }
    if (__syncthreads_or(bad ? 1 : 0)) { status = 5; break; }
    rr = s_sc[0];
    PCG_PROBE(4);
    if (!(rr == rr)) { status = 3; break; }
    if (!(rr > target) || it >= a.max_it) break;
    if (!is_top) {
      // ---- P4: w = A z over this wave's poses, delta = w.z ----
      double dz = 0.0;
      for (int i = i0; i < n_rows; i += H * kPcgWaves) {
        const bool first = (i == i0);
        if (first ? ident0 : (i >= a.n_poses || fv.fixed[i])) {             // identity rows (padding, constant poses; their scale is 1): w = z
          if (lane < 3) { const double z = gran_take<false>(a.zsG + 3 * i + lane, tag, pp, bad); gran_st(a.wG + 3 * i + lane, z, tag); dz += z * z; }
          continue;
        }
        const int e0 = first ? e00 : ((fv.n_offband > 0) ? fv.obadj_off[i] : 0);
        const int total = first ? tot0 : nband + ((fv.n_offband > 0) ? fv.obadj_off[i + 1] - e0 : 0);
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
        for (int idx = lane; idx < total; idx += WAVE) {
          double b9[9];
          int j, tr;
          if (first && idx == lane) {                   // the resident entry
            if (!ent0) continue;
            j = j0; tr = tr0;
#pragma unroll
            for (int q = 0; q < 9; ++q) b9[q] = bk[q];
          } else {
            const double* blk;
            if (idx < nband) {
              if (idx <= a.Wb) { j = i + idx; tr = 0; blk = lin.Hband + ((size_t)i * W1 + idx) * 9; if (j >= a.n_poses) continue; }
              else { const int d = idx - a.Wb; j = i - d; tr = 1; if (j < 0) continue; blk = lin.Hband + ((size_t)j * W1 + d) * 9; }
            } else {
              const int e = e0 + idx - nband, ent = fv.obadj_ent[e];
              j = fv.obadj_col[e]; tr = ent & 1; blk = lin.obblk + (size_t)(ent >> 1) * 9;
            }
#pragma unroll
            for (int q = 0; q < 9; ++q) b9[q] = blk[q];
          }
          const double v0 = gran_take<false>(a.zsG + 3 * j, tag, pp, bad), v1 = gran_take<false>(a.zsG + 3 * j + 1, tag, pp, bad),
                       v2 = gran_take<false>(a.zsG + 3 * j + 2, tag, pp, bad);
          if (!tr) {
            acc0 += b9[0] * v0 + b9[1] * v1 + b9[2] * v2;
            acc1 += b9[3] * v0 + b9[4] * v1 + b9[5] * v2;
            acc2 += b9[6] * v0 + b9[7] * v1 + b9[8] * v2;
          } else {
            acc0 += b9[0] * v0 + b9[3] * v1 + b9[6] * v2;
            acc1 += b9[1] * v0 + b9[4] * v1 + b9[7] * v2;
            acc2 += b9[2] * v0 + b9[5] * v1 + b9[8] * v2;
          }
        }
        acc0 = wave_sum(acc0); acc1 = wave_sum(acc1); acc2 = wave_sum(acc2);
        if (lane < 3) {
          const int t = 3 * i + lane;
          double sa = sa0, dd = dd0;
          if (!first) {
            sa = scale[t];
            const double dg = sa * sa * lin.Hband[(size_t)i * W1 * 9 + lane * 4];
            dd = fmin(fmax(dg, a.min_diag), a.max_diag) / a.mu / sa;
          }
          const double sz = gran_take<false>(a.zsG + t, tag, pp, bad);
          const double accv = (lane == 0) ? acc0 : (lane == 1) ? acc1 : acc2;
          const double y = sa * accv + dd * sz;            // (S H S + D^2) z, with z = sz / sa
          gran_st(a.wG + t, y, tag);
          dz += y * (sz / sa);
        }
      }
      const double delta_h = pcg_block_sum(dz, s_red);
      if (tid == 0) gran_st(a.delG + wg, delta_h, tag);
      PCG_PROBE(5);
    }
    if (is_chunk) {
      // ---- P5: alpha, beta from the global sums; p = z + beta p, s = w + beta s ----
      if (tid < WAVE) {
        double g = 0.0, d = 0.0;
        for (int c = tid; c < T; c += WAVE) g += gran_take<false>(a.gamG + c, tag, pp, bad);
        for (int h = tid; h < H; h += WAVE) d += gran_take<true>(a.delG + h, tag, pp, bad);
        g = wave_sum(g); d = wave_sum(d);
        if (tid == 0) { s_sc[1] = g; s_sc[2] = d; }
      }
      // the rows' products (their owners' waves publish them as they go): one row per thread (M * B <= kPcgThreads: pcg_fused_plan)
      double my_w = 0.0;
      if (tid < nvB) my_w = gran_take<true>(a.wG + row0 + tid, tag, pp, bad);
      if (__syncthreads_or(bad ? 1 : 0)) { status = 5; break; }
      PCG_PROBE(6);
      const double gamma = s_sc[1], delta = s_sc[2];
      const double beta = (it == 0 || gamma_old == 0.0) ? 0.0 : gamma / gamma_old;
      const double denom = (it == 0) ? delta : delta - beta * gamma / alpha_old;
      if (!(denom > 0.0)) {
        // p.A p <= 0 (or not finite): with r.z == 0 the residual is exactly zero and nothing is left to do; otherwise a breakdown
        if (gamma != 0.0) status = 1;
        if (tid == 0) pcg_abort_set(pp, 1u);      // the helpers and the top leave at their next wait
        break;
      }
      alpha = gamma / denom;
      if (tid < nvB) {
        pv[tid] = zl[tid] + beta * pv[tid];
        sv[tid] = my_w + beta * sv[tid];
      }
      gamma_old = gamma; alpha_old = alpha;
      __syncthreads();
    }
    PCG_PROBE(7);
    ++it;
  }
  // ---- the solution back to memory; one record for the host ----
  if (is_chunk) {
    for (int t = tid; t < nvB; t += kPcgThreads) a.x[row0 + t] = xv[t];
  }
  // status here: 0 = converged or iteration cap, 1 = breakdown, 3 = not finite (every workgroup alike); 5 = a wait was given up: because it ran
  // out (here or elsewhere: the abort word is 2 and the solve is void) or because the chunk workgroups ended the solve (abort word 1: helpers
  // and the top only).  Workgroup 0 (a chunk workgroup: one of those that decide) reports; a wait that ran out is reported by everybody who saw it.
  if (tid == 0) {
    const unsigned ab = (status == 5) ? pcg_abort_code(pp) : 0u;
    if (status == 5 && ab != 1u) __hip_atomic_store(a.result + 2, 2.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (wg == 0) {
      __hip_atomic_store(a.result + 0, (double)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(a.result + 1, rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(a.result + 3, bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(a.result + 2, (double)(status == 5 ? 2 : status), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static int pcg_pow2_at_least(int v) { int p = 1; while (p < v) p *= 2; return p; }

static size_t pcg_lds_doubles(int B, int M, int T) {
  const int BB = B * B, Tp = pcg_pow2_at_least(T);
  const size_t chunk = (size_t)3 * (M - 1) * BB + (size_t)(2 * (M + 1) + 7 * M) * B + 2 * B;
  const size_t top = (size_t)3 * T * BB + (size_t)(Tp + 2 * Tp + 1 + 2 * Tp) * B;
  return std::max(chunk, top) + 8;
}

// The launch geometry for a chain of nsb super-blocks of B x B, or false: chunk size M (a power of two), chunks = top nodes <= 64,
// both roles' tiles within the LDS budget, a chunk's rows one per thread, every workgroup resident at once (one per compute unit).
bool pcg_fused_plan(int nsb, int B, int n_rows, int cus, PcgPlan* out) {
  if (B != 3 && B != 6 && B != 9 && B != 12) return false;
  if (nsb < 4 || cus < 4) return false;
  const size_t lds_cap = (size_t)120 * 1024 / sizeof(double);
  int best = 0;
  for (int M = 4; M <= 256; M *= 2) {
    const int T = (nsb + M - 1) / M;
    if (T < 2 || T > 64 || T + 1 > cus || M * B > kPcgThreads) continue;
    if (pcg_lds_doubles(B, M, T) > lds_cap) continue;
    // the sweep's depth is log2(M) + log2(T) whatever the split; fewer chunks make the fan-ins cheaper, more chunks spread the
    // local levels over more workgroups: take the split closest to M == T
    if (best == 0 || std::abs(M - T) < std::abs(best - (nsb + best - 1) / best)) best = M;
  }
  if (best == 0) return false;
  PcgPlan p;
  p.M = best; p.chunks = (nsb + best - 1) / best; p.Tp = pcg_pow2_at_least(p.chunks);
  const int want = (n_rows + kPcgWaves - 1) / kPcgWaves + 1;        // one wave per pose for the products, plus the top
  p.G = std::max(p.chunks + 1, std::min(cus, want));
  p.lds_bytes = pcg_lds_doubles(B, p.M, p.chunks) * sizeof(double);
  // granules (16 bytes each): zs, w (nsb * B each) | ctr (chunks * 3 B) | topL, topR (chunks * B each) | gamma, r.r (chunks each) | delta (G)
  p.work_doubles = 2 * ((size_t)2 * nsb * B + (size_t)p.chunks * 5 * B + (size_t)p.chunks * 2 + (size_t)p.G + 8);
  if (out) *out = p;
  return true;
}

size_t pcg_fused_sync_words() { return 32; }

template <int B>
static bool pcg_launch_t(hipStream_t st, const PcgPlan& pl, const PcgArgs& a, const FactorView& fv, const LinView& lin, const CrView& cr, const double* scale, int groups) {
  static bool raised = false;
  if (!raised) {
    raised = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pcg_fused<B>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
    if (!raised) { (void)hipGetLastError(); return false; }
  }
  hipLaunchKernelGGL(k_pcg_fused<B>, dim3(groups), dim3(kPcgThreads), pl.lds_bytes, st, a, fv, lin, cr, scale);
  return true;
}

bool launch_pcg_fused(hipStream_t st, const PcgPlan& pl, int n_poses, int npad, const FactorView& fv, const LinView& lin, const CrView& cr, const double* scale,
                      double mu, double min_diag, double max_diag, double tol, int max_it, double* x, const double* r0, const double* partials0, int nblk0,
                      double* work, unsigned* sync, uint64_t launch_index, double* result_pinned, unsigned spins, bool starve, unsigned long long* probe) {
  const int B = cr.B;
  const int groups = pl.G - (starve ? 1 : 0);       // test hook: the top workgroup (the last one) is never started
  PcgArgs a;
  a.n_poses = n_poses; a.npad = npad; a.nsb = cr.nsb; a.Wb = fv.band_poses;
  a.M = pl.M; a.chunks = pl.chunks; a.Tp = pl.Tp; a.G = pl.G;
  a.mu = mu; a.min_diag = min_diag; a.max_diag = max_diag; a.tol2 = tol * tol;
  a.tag0 = (double)launch_index * 2097152.0;        // 2^21 tags per launch (max_it is capped below it); exact in a double for 2^32 launches
  a.max_it = std::min(max_it, (1 << 21) - 2); a.spins = spins;
  a.x = x; a.r0 = r0; a.partials0 = partials0; a.nblk0 = nblk0;
  v2f64* g = reinterpret_cast<v2f64*>(work);
  a.zsG = g; g += (size_t)cr.nsb * B;
  a.wG = g; g += (size_t)cr.nsb * B;
  a.ctrG = g; g += (size_t)pl.chunks * 3 * B;
  a.topL = g; g += (size_t)pl.chunks * B;
  a.topR = g; g += (size_t)pl.chunks * B;
  a.gamG = g; g += pl.chunks;
  a.rrG = g; g += pl.chunks;
  a.delG = g;
  a.abort_w = sync; a.launch = (unsigned)(launch_index & 0x3fffffffu);
  a.result = result_pinned; a.probe = probe;
  switch (B) {
    case 3: return pcg_launch_t<3>(st, pl, a, fv, lin, cr, scale, groups);
    case 6: return pcg_launch_t<6>(st, pl, a, fv, lin, cr, scale, groups);
    case 9: return pcg_launch_t<9>(st, pl, a, fv, lin, cr, scale, groups);
    default: return pcg_launch_t<12>(st, pl, a, fv, lin, cr, scale, groups);
  }
}

}  // namespace hitl
