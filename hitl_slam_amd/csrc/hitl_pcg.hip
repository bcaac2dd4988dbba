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
//     too.  Per iteration: the vector updates, the local levels of the preconditioner's forward sweep, later its back-substitution --
//     all by ONE wave (the sweep is a chain of small dependent levels: a second wave would only add barriers: measured, 4 x slower).
//   * the top workgroup (the last one): the nodes 0, M, 2M, ... (one per chunk), tiles resident in LDS: the upper levels, one wave.
//   * product workgroups (the ones between): w = A z, one wave per pose over the pose's band and off-band blocks.  The matrix does
//     not change during a solve: every wave keeps its poses' 3x3 blocks (already oriented) and column indices in LDS.
//
// The recurrence is Chronopoulos & Gear's single-reduction form of PCG (z = M^-1 r, w = A z, gamma = r.z, delta = w.z,
// beta = gamma / gamma_old, alpha = gamma / (delta - beta gamma / alpha_old), p = z + beta p, s = w + beta s, x += alpha p, r -= alpha s):
// the same iterates as textbook PCG in exact arithmetic, with ONE all-to-all reduction per iteration instead of two.  A CG iteration is
// four hand-offs: chunks -> top (boundary contributions), top -> chunks (z at the chunk boundaries), chunks -> product waves (z; r.z and
// r.r ride along), product waves -> chunks (w, delta).  Every handed-off double travels as a tagged 16-byte granule (below).
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

// Every wait is bounded and the abort word ends all of them: 1 = the chunk workgroups have ended the solve (breakdown), 2 = somebody's
// wait ran out.  Called every 64th poll of a spin loop; true = give up.
__device__ __forceinline__ bool pcg_give_up(unsigned k, const PcgPoll& pp) {
  if (k >= pp.spins) { pcg_abort_set(pp, 2u); return true; }
  return pcg_abort_code(pp) != 0u;
}

// spin until the granule carries `tag`.  SLEEP: polls of data that is usually a while away back off between loads.
template <bool SLEEP>
__device__ __forceinline__ double gran_take(const v2f64* p, double tag, const PcgPoll& pp, bool& bad) {
  for (unsigned k = 1;; ++k) {
    const v2f64 g = gran_ld(p);
    if (g.y == tag) return g.x;
    if ((k & 63u) == 0u && pcg_give_up(k, pp)) { bad = true; return 0.0; }
    if (SLEEP) __builtin_amdgcn_s_sleep(2);
  }
}
// three granules at once (the three scalars of a pose): all loads of a round are in flight together
__device__ __forceinline__ void gran_take3(const v2f64* p, double tag, const PcgPoll& pp, bool& bad, double& v0, double& v1, double& v2) {
  for (unsigned k = 1;; ++k) {
    const v2f64 g0 = gran_ld(p), g1 = gran_ld(p + 1), g2 = gran_ld(p + 2);
    if (g0.y == tag && g1.y == tag && g2.y == tag) { v0 = g0.x; v1 = g1.x; v2 = g2.x; return; }
    if ((k & 63u) == 0u && pcg_give_up(k, pp)) { bad = true; v0 = v1 = v2 = 0.0; return; }
  }
}
// count granules into LDS by one wave (entry e = lane, lane + 64, ...): every pending load of a round is in flight together
__device__ __forceinline__ void gran_take_rows(const v2f64* p, int count, double tag, const PcgPoll& pp, bool& bad, double* out, int lane) {
  unsigned pend = 0u;
  for (int k = 0; lane + WAVE * k < count; ++k) pend |= 1u << k;         // count <= 16 * 64 (pcg_fused_plan: a chunk's rows <= 512)
  for (unsigned spin = 1; pend != 0u; ++spin) {
    for (unsigned m = pend; m != 0u; m &= m - 1u) {
      const int k = __builtin_ctz(m), e = lane + WAVE * k;
      const v2f64 g = gran_ld(p + e);
      if (g.y == tag) { out[e] = g.x; pend &= ~(1u << k); }
    }
    if (pend != 0u) {
      if ((spin & 63u) == 0u && pcg_give_up(spin, pp)) { bad = true; return; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
}

// LDS written by some lanes of a wave and read by others of the SAME wave: the hardware keeps a wave's LDS operations in order;
// the compiler must not move a read above the write it depends on
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---- the sweep's coefficients in LDS -----------------------------------------------------------------------------------------
// One wave works through a level: an LDS read instruction takes as many passes as lanes fall on one bank, and the plain tile
// layout [node][row][column] put the nodes of a level (1152 bytes apart) on the same banks (measured: 5-7 passes per read, the sweep of a
// 16-node chunk 9.5 us per CG iteration).  The tiles are therefore stored LEVEL by level with the node index k fastest (node
// u = s (2 k + 1) of the level of stride s, nel = Mloc / (2 s) nodes), so that the lanes of a read touch consecutive words:
//   GL, GR (used by columns in the forward sweep, by rows in the back-substitution):  [row r][slot (c + r) mod B][k]  -- the rotation by
//          r makes BOTH uses conflict-free (columns: lanes (c, k) cover a contiguous range; rows: lanes (r, k) of a half-wave differ
//          in (c + r) nel mod 32);
//   DI (rows only):  [column c][row r][k].
// Levels follow each other: level of stride s starts (Mloc - Mloc / s) tiles in.  Vectors (w, xs, ca, cb) keep B + 1 doubles per node.
template <int B>
__device__ __forceinline__ size_t pcg_level_off(int Mloc, int s) { return (size_t)(Mloc - Mloc / s) * B * B; }

// forward products of the level of stride s: item e = (which B + c) nel + k; ca[u] = Gl_u^T w_u goes to node u - s, cb[u] = Gr_u^T w_u to u + s
template <int B>
__device__ __forceinline__ void pcg_fwd_products(const double* GL, const double* GR, const double* w, double* ca, double* cb, int s, int Mloc, int u_max, int lane) {
  constexpr int BP = B + 1;
  const int nel = Mloc / (2 * s);
  const size_t off = pcg_level_off<B>(Mloc, s);
  for (int e = lane; e < nel * 2 * B; e += WAVE) {
    const int k = e % nel, wc = e / nel, which = wc / B, c = wc % B, u = s * (2 * k + 1);
    if (u > u_max) continue;
    const double* G = (which ? GR : GL) + off + k;
    const double* wu = w + (size_t)u * BP;
    double c0 = 0.0, c1 = 0.0;                      // two chains: a dependent fp64 FMA chain of B links is what a level waits for
#pragma unroll
    for (int r = 0; r < B; ++r) {
      const int slot = (c + r >= B) ? c + r - B : c + r;
      const double g = G[(size_t)(r * B + slot) * nel];
      if (r & 1) c1 += g * wu[r]; else c0 += g * wu[r];
    }
    (which ? cb : ca)[(size_t)u * BP + c] = c0 + c1;
  }
}

// back-substitution of the level of stride s: item e = r nel + k: x_u[r] = Dinv_u[r,:] w_u - Gl_u[r,:] x_{u-s} - Gr_u[r,:] x_{u+s}
// (xs holds zeros at nodes that do not exist)
template <int B>
__device__ __forceinline__ void pcg_back_level(const double* DI, const double* GL, const double* GR, const double* w, double* xs, int s, int Mloc, int u_max, int lane) {
  constexpr int BP = B + 1;
  const int nel = Mloc / (2 * s);
  const size_t off = pcg_level_off<B>(Mloc, s);
  for (int e = lane; e < nel * B; e += WAVE) {
    const int k = e % nel, r = e / nel, u = s * (2 * k + 1);
    if (u > u_max) continue;
    const double* Di = DI + off + e;
    const double* Gl = GL + off + (size_t)r * B * nel + k;
    const double* Gr = GR + off + (size_t)r * B * nel + k;
    const double* wu = w + (size_t)u * BP;
    const double* xl = xs + (size_t)(u - s) * BP;
    const double* xr = xs + (size_t)(u + s) * BP;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;            // three independent chains
#pragma unroll
    for (int c = 0; c < B; ++c) {
      const int slot = (c + r >= B) ? c + r - B : c + r;
      a0 += Di[(size_t)c * B * nel] * wu[c];
      a1 += Gl[(size_t)slot * nel] * xl[c];
      a2 += Gr[(size_t)slot * nel] * xr[c];
    }
    xs[(size_t)u * BP + r] = (a0 - a1) - a2;
  }
}

// the tiles of the local nodes 1 .. Mloc - 1 into that layout, by the whole workgroup; local node u is global node gbase + u * gstride and exists for u <= u_max
template <int B>
__device__ __forceinline__ void pcg_fill_tiles(const CrView& cr, double* GL, double* GR, double* DI, int Mloc, int u_max, size_t gbase, size_t gstride) {
  constexpr int BB = B * B;
  for (int t = threadIdx.x; t < (Mloc - 1) * BB; t += kPcgThreads) {
    const int u = 1 + t / BB, e = t % BB, r = e / B, c = e % B;
    const int s = u & -u, nel = Mloc / (2 * s), k = (u / s - 1) / 2;
    const size_t off = pcg_level_off<B>(Mloc, s);
    const bool ex = u <= u_max;
    const size_t g = (gbase + (size_t)u * gstride) * BB + e;
    const int slot = (c + r >= B) ? c + r - B : c + r;
    GL[off + (size_t)(r * B + slot) * nel + k] = ex ? cr.Gl[g] : 0.0;
    GR[off + (size_t)(r * B + slot) * nel + k] = ex ? cr.Gr[g] : 0.0;
    DI[off + (size_t)(c * B + r) * nel + k] = ex ? cr.Dinv[g] : 0.0;
  }
}

struct PcgArgs {
  int n_poses, npad, nsb, Wb;
  int M, chunks, Tp, G;          // chunk size (a power of two), chunks = top nodes, Tp = the power of two >= chunks, workgroups:
                                 // [0, chunks) the chunks, [chunks, G - 1) the product workgroups, G - 1 the top
  int capw;                      // product entries a wave keeps in LDS
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
  v2f64* delG;                   // product workgroups: w.z of the workgroup's rows
  unsigned* abort_w;
  unsigned launch;               // low 30 bits of the solver's launch counter (see PcgPoll)
  double* result;                // pinned host memory: iterations, r.r, status (0 ok, 1 breakdown, 2 a wait ran out, 3 not finite), b.b
  unsigned long long* probe;     // HITL_TIMING: 16 timestamps (100 MHz) per workgroup around the phases of CG iteration 2, or null
};

#define PCG_PROBE(slot) do { if (a.probe != nullptr && it == 2 && tid == 0) { a.probe[(size_t)wg * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    if ((slot) == 0 || (slot) == 7) a.probe[(size_t)wg * 16 + 8 + (slot)] = __builtin_readcyclecounter(); } } while (0)

// sum of the chunks' r.r granules (every workgroup: is the solve over?): lane 0 waits for the first chunk's, then the others' are on their way
__device__ __forceinline__ double pcg_rr_sum(const v2f64* rrG, int T, double tag, const PcgPoll& pp, bool& bad, int lane) {
  double v = 0.0;
  if (lane == 0) v = gran_take<true>(rrG, tag, pp, bad);
  for (int c = lane + (lane == 0 ? WAVE : 0); c < T; c += WAVE) v += gran_take<false>(rrG + c, tag, pp, bad);
  return wave_sum(v);
}

template <int B>
__global__ __launch_bounds__(kPcgThreads) void k_pcg_fused(PcgArgs a, FactorView fv, LinView lin, CrView cr, const double* __restrict__ scale) {
  constexpr int BB = B * B;
  extern __shared__ double lds[];
  __shared__ double s_red[kPcgWaves];
  __shared__ double s_sc[8];               // [0] rr (product workgroups), [6] bb, [7] rr0
  const int tid = threadIdx.x, wg = blockIdx.x, wave = tid >> 6, lane = tid & 63;
  const int M = a.M, T = a.chunks, Tp = a.Tp, Hh = a.G - 1 - a.chunks;
  const bool is_top = (wg == a.G - 1), is_chunk = (wg < a.chunks);
  const PcgPoll pp = {a.abort_w, a.spins, a.launch};
  {  // r0.r0 and b.b from the residual pass in front of this launch (a kernel boundary: plain data), every workgroup alike
    double r_ = 0.0, b_ = 0.0;
    if (tid < WAVE) {
      for (int k = tid; k < a.nblk0; k += WAVE) { r_ += a.partials0[(size_t)k * 8]; b_ += a.partials0[(size_t)k * 8 + 1]; }
      r_ = wave_sum(r_); b_ = wave_sum(b_);
    }
    if (tid == 0) { s_sc[7] = r_; s_sc[6] = b_; }
  }
  int it = 0;
  int status = 0;
  double rr = 0.0, bb = 0.0;

  if (!is_chunk && !is_top) {
    // =============================== product workgroup: w = A z over its waves' poses ===============================
    // wave gw takes poses gw, gw + Hh * 8, ...; their entries (band blocks both ways + off-band blocks), oriented, sit in LDS:
    // SoA blkS[q][capw] + colS[capw] per wave; entries beyond capw are read from memory every iteration (never on maps of
    // hitl-slam's size).  Per pose: sum over its entries of block * (scale z)[column], reduced over the wave, finished by lanes 0..2.
    const int capw = a.capw;
    double* blkS = lds + (size_t)wave * capw * 10;
    int* colS = reinterpret_cast<int*>(blkS + (size_t)capw * 9);
    const int W1 = a.Wb + 1, nband = 2 * a.Wb + 1, n_rows = a.npad / 3;
    const int gw = (wg - a.chunks) * kPcgWaves + wave, TW = Hh * kPcgWaves;
    auto entry = [&](int i, int e0, int idx, const double*& blk, int& j, int& tr) -> bool {       // entry idx of pose i; false: no such block
      if (idx < nband) {
        if (idx <= a.Wb) { j = i + idx; tr = 0; if (j >= a.n_poses) return false; blk = lin.Hband + ((size_t)i * W1 + idx) * 9; }
        else { const int d = idx - a.Wb; j = i - d; tr = 1; if (j < 0) return false; blk = lin.Hband + ((size_t)j * W1 + d) * 9; }
      } else {
        const int e = e0 + idx - nband, ent = fv.obadj_ent[e];
        j = fv.obadj_col[e]; tr = ent & 1; blk = lin.obblk + (size_t)(ent >> 1) * 9;
      }
      return true;
    };
    {
      int o = 0;
      for (int i = gw; i < n_rows; i += TW) {
        if (i >= a.n_poses || fv.fixed[i]) continue;
        const int e0 = (fv.n_offband > 0) ? fv.obadj_off[i] : 0;
        const int total = nband + ((fv.n_offband > 0) ? fv.obadj_off[i + 1] - e0 : 0);
        for (int idx = lane; idx < total && o + idx < capw; idx += WAVE) {
          const double* blk = nullptr;
          int j = -1, tr = 0;
          const bool ok = entry(i, e0, idx, blk, j, tr);
          colS[o + idx] = ok ? j : -1;
          if (ok) {
#pragma unroll
            for (int q = 0; q < 9; ++q) blkS[(size_t)q * capw + o + idx] = tr ? blk[(q % 3) * 3 + q / 3] : blk[q];
          }
        }
        o += total;
      }
    }
    __syncthreads();
    bb = s_sc[6]; rr = s_sc[7];
    const double target = a.tol2 * bb;
    if (!(rr == rr)) status = 3;
    const bool run = (status == 0) && (rr > target) && a.max_it > 0;
    while (run) {
      const double tag = a.tag0 + (double)(it + 1);
      bool bad = false;
      PCG_PROBE(0);
      if (wave == 0) { const double v = pcg_rr_sum(a.rrG, T, tag, pp, bad, lane); if (lane == 0) s_sc[0] = v; }
      if (__syncthreads_or(bad ? 1 : 0)) { status = 5; break; }
      rr = s_sc[0];
      PCG_PROBE(4);
      if (!(rr == rr)) { status = 3; break; }
      if (!(rr > target) || it >= a.max_it) break;
      double dz = 0.0;
      int o = 0;
      for (int i = gw; i < n_rows; i += TW) {
        if (i >= a.n_poses || fv.fixed[i]) {                                  // identity rows (padding, constant poses; their scale is 1): w = z
          if (lane < 3) { const double z = gran_take<false>(a.zsG + 3 * i + lane, tag, pp, bad); gran_st(a.wG + 3 * i + lane, z, tag); dz += z * z; }
          continue;
        }
        const int e0 = (fv.n_offband > 0) ? fv.obadj_off[i] : 0;
        const int total = nband + ((fv.n_offband > 0) ? fv.obadj_off[i + 1] - e0 : 0);
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, d0 = 0.0, d1 = 0.0, d2 = 0.0;      // d: (scale z) of the pose's own column (entry 0 = the diagonal block)
        for (int idx = lane; idx < total; idx += WAVE) {
          double b9[9];
          int j;
          if (o + idx < capw) {
            j = colS[o + idx];
            if (j < 0) continue;
#pragma unroll
            for (int q = 0; q < 9; ++q) b9[q] = blkS[(size_t)q * capw + o + idx];
          } else {
            const double* blk = nullptr;
            int tr = 0;
            if (!entry(i, e0, idx, blk, j, tr)) continue;
#pragma unroll
            for (int q = 0; q < 9; ++q) b9[q] = tr ? blk[(q % 3) * 3 + q / 3] : blk[q];
          }
          double v0, v1, v2;
          gran_take3(a.zsG + 3 * j, tag, pp, bad, v0, v1, v2);
          if (idx == 0) { d0 = v0; d1 = v1; d2 = v2; }
          acc0 += b9[0] * v0 + b9[1] * v1 + b9[2] * v2;
          acc1 += b9[3] * v0 + b9[4] * v1 + b9[5] * v2;
          acc2 += b9[6] * v0 + b9[7] * v1 + b9[8] * v2;
        }
        acc0 = wave_sum(acc0); acc1 = wave_sum(acc1); acc2 = wave_sum(acc2);
        d1 = __shfl(d1, 0, WAVE); d2 = __shfl(d2, 0, WAVE);
        if (lane < 3) {
          const int t = 3 * i + lane;
          const double sa = scale[t];
          const double dg = sa * sa * lin.Hband[(size_t)i * W1 * 9 + lane * 4];
          const double dd = fmin(fmax(dg, a.min_diag), a.max_diag) / a.mu / sa;
          const double sz = (lane == 0) ? d0 : (lane == 1) ? d1 : d2;       // (lane 0 took the diagonal block's column itself)
          const double accv = (lane == 0) ? acc0 : (lane == 1) ? acc1 : acc2;
          const double y = sa * accv + dd * sz;            // (S H S + D^2) z, with z = sz / sa
          gran_st(a.wG + t, y, tag);
          dz += y * (sz / sa);
        }
        o += total;
      }
      {
        dz = wave_sum(dz);
        __syncthreads();
        if (lane == 0) s_red[wave] = dz;
        __syncthreads();
        if (tid == 0) {
          double t = 0.0;
#pragma unroll
          for (int k = 0; k < kPcgWaves; ++k) t += s_red[k];
          gran_st(a.delG + (wg - a.chunks), t, tag);
        }
      }
      PCG_PROBE(5);
      PCG_PROBE(7);
      ++it;
    }
    if (tid == 0 && status == 5 && pcg_abort_code(pp) != 1u) __hip_atomic_store(a.result + 2, 2.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }

  // =============================== chunk workgroups and the top: one wave each ===============================
  // ---- LDS carve-up (vectors keep BP = B + 1 doubles per node) ----
  // chunk: GL GR DI of the interior nodes 1 .. M-1 (level layout) | w, xs (M + 1 nodes) | ca, cb (M nodes) | xv rv pv sv zl wl scl (M B rows) | accL0, accRM
  // top:   GL GR DI of the nodes 1 .. Tp-1 | Dinv of node 0 | w (Tp nodes), xs (2 Tp + 1 nodes: zeros beyond T - 1) | ca, cb (Tp nodes)
  constexpr int BP = B + 1;
  double *sGl, *sGr, *sDi, *sD0 = nullptr, *w, *xs, *ca, *cb, *xv = nullptr, *rv = nullptr, *pv = nullptr, *sv = nullptr, *zl = nullptr, *wl = nullptr, *scl = nullptr,
         *accL0 = nullptr, *accRM = nullptr;
  int nv = 0, nvB = 0;
  const size_t row0 = (size_t)wg * M * B;           // first owned row of a chunk workgroup
  if (is_top) {
    sGl = lds; sGr = sGl + (size_t)(Tp - 1) * BB; sDi = sGr + (size_t)(Tp - 1) * BB; sD0 = sDi + (size_t)(Tp - 1) * BB;
    w = sD0 + BB; xs = w + (size_t)Tp * BP; ca = xs + (size_t)(2 * Tp + 1) * BP; cb = ca + (size_t)Tp * BP;
    pcg_fill_tiles<B>(cr, sGl, sGr, sDi, Tp, T - 1, 0, (size_t)M);
    for (int e = tid; e < BB; e += kPcgThreads) sD0[e] = cr.Dinv[e];
    for (int e = tid; e < (2 * Tp + 1) * BP; e += kPcgThreads) xs[e] = 0.0;
    for (int e = tid; e < Tp * BP; e += kPcgThreads) w[e] = 0.0;
  } else {
    sGl = lds; sGr = sGl + (size_t)(M - 1) * BB; sDi = sGr + (size_t)(M - 1) * BB;
    w = sDi + (size_t)(M - 1) * BB; xs = w + (size_t)(M + 1) * BP; ca = xs + (size_t)(M + 1) * BP; cb = ca + (size_t)M * BP;
    xv = cb + (size_t)M * BP; rv = xv + (size_t)M * B; pv = rv + (size_t)M * B; sv = pv + (size_t)M * B; zl = sv + (size_t)M * B;
    wl = zl + (size_t)M * B; scl = wl + (size_t)M * B;
    accL0 = scl + (size_t)M * B; accRM = accL0 + B;
    nv = min(M, a.nsb - wg * M); nvB = nv * B;
    pcg_fill_tiles<B>(cr, sGl, sGr, sDi, M, nv - 1, (size_t)wg * M, 1);
    for (int e = tid; e < (M + 1) * BP; e += kPcgThreads) { xs[e] = 0.0; w[e] = 0.0; }
    for (int t = tid; t < M * B; t += kPcgThreads) {
      const bool own = t < nvB;
      xv[t] = own ? a.x[row0 + t] : 0.0; rv[t] = own ? a.r0[row0 + t] : 0.0; pv[t] = 0.0; sv[t] = 0.0; zl[t] = 0.0; wl[t] = 0.0;
      scl[t] = own ? scale[row0 + t] : 1.0;
    }
  }
  __syncthreads();
  bb = s_sc[6]; rr = s_sc[7];
  if (wave != 0) return;                              // from here on ONE wave per workgroup: no barriers below this line
  const double target = a.tol2 * bb;
  double alpha = 0.0, alpha_old = 0.0, gamma_old = 0.0;
  if (!(rr == rr)) status = 3;
  const bool run = (status == 0) && (rr > target) && a.max_it > 0;

  while (run) {
    const double tag = a.tag0 + (double)(it + 1);
    bool bad = false;
    PCG_PROBE(0);
    if (is_chunk) {
      // ---- P1: x += alpha p, r -= alpha s; r.r; the local levels of the forward sweep on r ----
      double rr_t = 0.0;
      for (int t = lane; t < nvB; t += WAVE) {
        if (it > 0) { xv[t] += alpha * pv[t]; rv[t] -= alpha * sv[t]; }
        const double rt = rv[t];
        w[(t / B) * BP + t % B] = rt;
        rr_t += rt * rt;
      }
      if (lane < 2 * B) accL0[lane] = 0.0;               // accL0 and accRM are adjacent
      const double rr_c = wave_sum(rr_t);
      wave_lds_sync();
      PCG_PROBE(9);
      for (int s = 1; s < M; s *= 2) {
        pcg_fwd_products<B>(sGl, sGr, w, ca, cb, s, M, nv - 1, lane);
        wave_lds_sync();
        if (s == 1) PCG_PROBE(10);
        const int nsv = M / (2 * s) + 1;
        for (int e = lane; e < nsv * B; e += WAVE) {
          const int j = 2 * s * (e / B), q = e % B;
          const double pl = (j - s >= 1 && j - s <= nv - 1) ? cb[(size_t)(j - s) * BP + q] : 0.0;
          const double pr = (j + s <= nv - 1 && j + s < M) ? ca[(size_t)(j + s) * BP + q] : 0.0;
          if (j == 0) accL0[q] += pr;
          else if (j == M) accRM[q] += pl;
          else if (j <= nv - 1) w[(size_t)j * BP + q] = (w[(size_t)j * BP + q] - pl) - pr;
        }
        wave_lds_sync();
        if (s == 1) PCG_PROBE(11);
        if (s == 2) PCG_PROBE(12);
        if (s == 4) PCG_PROBE(13);
        if (s == 8) PCG_PROBE(14);
      }
      if (lane < 3 * B) {
        const double v = (lane < B) ? accL0[lane] : (lane < 2 * B) ? accRM[lane - B] : rv[lane - 2 * B];
        gran_st(a.ctrG + (size_t)wg * 3 * B + lane, v, tag);
      }
      PCG_PROBE(1);
      // ---- P3: z at the two boundary nodes comes back from the top; back-substitution; r.z ----
      if (lane < 2 * B) {
        const int q = lane % B;
        if (lane < B) xs[q] = gran_take<true>(a.topL + (size_t)wg * B + q, tag, pp, bad);
        else xs[(size_t)M * BP + q] = (wg + 1 < T) ? gran_take<true>(a.topR + (size_t)wg * B + q, tag, pp, bad) : 0.0;
      }
      if (__any(bad ? 1 : 0)) { status = 5; break; }
      wave_lds_sync();
      PCG_PROBE(2);
      for (int s = M / 2; s >= 1; s /= 2) {
        pcg_back_level<B>(sDi, sGl, sGr, w, xs, s, M, nv - 1, lane);
        wave_lds_sync();
      }
      double gz = 0.0;
      for (int t = lane; t < nvB; t += WAVE) {
        const double z = xs[(t / B) * BP + t % B];
        zl[t] = z;
        gz += rv[t] * z;
        gran_st(a.zsG + row0 + t, scl[t] * z, tag);
      }
      const double gamma_c = wave_sum(gz);
      if (lane == 0) { gran_st(a.gamG + wg, gamma_c, tag); gran_st(a.rrG + wg, rr_c, tag); }
      PCG_PROBE(3);
    } else {
      // ---- P2 (the top): the upper levels: nodes 0, M, 2M, ... ----
      {
        unsigned pend = 0u;
        for (int k = 0; lane + WAVE * k < T * B; ++k) pend |= 1u << k;          // T * B <= 64 * 12
        for (unsigned spin = 1; pend != 0u; ++spin) {
          for (unsigned m = pend; m != 0u; m &= m - 1u) {
            const int k = __builtin_ctz(m), e = lane + WAVE * k, t = e / B, q = e % B;
            const v2f64 g0 = gran_ld(a.ctrG + (size_t)t * 3 * B + 2 * B + q), g1 = gran_ld(a.ctrG + (size_t)t * 3 * B + q);
            v2f64 g2; g2.x = 0.0; g2.y = tag;
            if (t >= 1) g2 = gran_ld(a.ctrG + (size_t)(t - 1) * 3 * B + B + q);
            if (g0.y == tag && g1.y == tag && g2.y == tag) { w[t * BP + q] = (g0.x - g1.x) - g2.x; pend &= ~(1u << k); }
          }
          if (pend != 0u) {
            if ((spin & 63u) == 0u && pcg_give_up(spin, pp)) { bad = true; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
      }
      if (__any(bad ? 1 : 0)) { status = 5; break; }
      wave_lds_sync();
      PCG_PROBE(1);
      for (int s = 1; s < Tp; s *= 2) {
        pcg_fwd_products<B>(sGl, sGr, w, ca, cb, s, Tp, T - 1, lane);
        wave_lds_sync();
        const int nsv = Tp / (2 * s);
        for (int e = lane; e < nsv * B; e += WAVE) {
          const int j = 2 * s * (e / B), q = e % B;
          if (j > T - 1) continue;
          const double pl = (j - s >= 1) ? cb[(size_t)(j - s) * BP + q] : 0.0;
          const double pr = (j + s <= T - 1) ? ca[(size_t)(j + s) * BP + q] : 0.0;
          w[(size_t)j * BP + q] = (w[(size_t)j * BP + q] - pl) - pr;
        }
        wave_lds_sync();
      }
      if (lane < B) {
        double v = 0.0;
#pragma unroll
        for (int c = 0; c < B; ++c) v += sD0[lane * B + c] * w[c];
        xs[lane] = v;
      }
      wave_lds_sync();
      for (int s = Tp / 2; s >= 1; s /= 2) {
        pcg_back_level<B>(sDi, sGl, sGr, w, xs, s, Tp, T - 1, lane);
        wave_lds_sync();
      }
      PCG_PROBE(2);
      for (int e = lane; e < T * B; e += WAVE) {
        const double v = xs[(e / B) * BP + e % B];
        gran_st(a.topL + e, v, tag);
        if (e >= B) gran_st(a.topR + (e - B), v, tag);
      }
      PCG_PROBE(3);
    }
    // ---- r.r is out with z; is the solve over? ----
    rr = pcg_rr_sum(a.rrG, T, tag, pp, bad, lane);
    if (__any(bad ? 1 : 0)) { status = 5; break; }
    PCG_PROBE(4);
    if (!(rr == rr)) { status = 3; break; }
    if (!(rr > target) || it >= a.max_it) break;
    if (is_chunk) {
      // ---- P5: w of the owned rows and the global sums come in; alpha, beta; p = z + beta p, s = w + beta s ----
      double g = 0.0, d = 0.0;
      for (int c = lane; c < T; c += WAVE) g += gran_take<false>(a.gamG + c, tag, pp, bad);
      for (int h = lane; h < Hh; h += WAVE) d += gran_take<true>(a.delG + h, tag, pp, bad);
      gran_take_rows(a.wG + row0, nvB, tag, pp, bad, wl, lane);
      if (__any(bad ? 1 : 0)) { status = 5; break; }
      const double gamma = wave_sum(g), delta = wave_sum(d);
      PCG_PROBE(6);
      const double beta = (it == 0 || gamma_old == 0.0) ? 0.0 : gamma / gamma_old;
      const double denom = (it == 0) ? delta : delta - beta * gamma / alpha_old;
      if (!(denom > 0.0)) {
        // p.A p <= 0 (or not finite): with r.z == 0 the residual is exactly zero and nothing is left to do; otherwise a breakdown
        if (gamma != 0.0) status = 1;
        if (lane == 0) pcg_abort_set(pp, 1u);      // the product workgroups and the top leave at their next wait
        break;
      }
      alpha = gamma / denom;
      for (int t = lane; t < nvB; t += WAVE) {
        pv[t] = zl[t] + beta * pv[t];
        sv[t] = wl[t] + beta * sv[t];
      }
      gamma_old = gamma; alpha_old = alpha;
    }
    PCG_PROBE(7);
    ++it;
  }
  // ---- the solution back to memory; one record for the host ----
  if (is_chunk) {
    for (int t = lane; t < nvB; t += WAVE) a.x[row0 + t] = xv[t];
  }
  // status here: 0 = converged or iteration cap, 1 = breakdown, 3 = not finite (every workgroup alike); 5 = a wait was given up: because it ran
  // out (here or elsewhere: the abort word is 2 and the solve is void) or because the chunk workgroups ended the solve (abort word 1: the product
  // workgroups and the top only).  Workgroup 0 (a chunk: one of those that decide) reports; a wait that ran out is reported by everybody who saw it.
  if (lane == 0) {
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
  const size_t chunk = (size_t)3 * (M - 1) * BB + (size_t)(2 * (M + 1) + 2 * M) * (B + 1) + (size_t)7 * M * B + 2 * B;
  const size_t top = (size_t)3 * (Tp - 1) * BB + BB + (size_t)(Tp + 2 * Tp + 1 + 2 * Tp) * (B + 1);
  return std::max(chunk, top) + 8;
}

// The launch geometry for a chain of nsb super-blocks of B x B, or false: chunk size M (a power of two), chunks = top nodes <= 64,
// both roles' tiles within the LDS budget, a chunk's rows at most 512, every workgroup resident at once (one per compute unit).
bool pcg_fused_plan(int nsb, int B, int n_rows, int cus, PcgPlan* out) {
  if (B != 3 && B != 6 && B != 9 && B != 12) return false;
  if (nsb < 4 || cus < 4) return false;
  const size_t lds_cap = (size_t)156 * 1024 / sizeof(double);      // of the CU's 160 KB
  int best = 0;
  for (int M = 4; M <= 256; M *= 2) {
    const int T = (nsb + M - 1) / M;
    if (T < 2 || T > 64 || T + 2 > cus || M * B > kPcgThreads) continue;
    if (pcg_lds_doubles(B, M, T) > lds_cap) continue;
    // the sweep's depth is log2(M) + log2(T) whatever the split; fewer chunks make the fan-ins cheaper, more chunks spread the
    // local levels over more workgroups: take the split closest to M == T
    if (best == 0 || std::abs(M - T) < std::abs(best - (nsb + best - 1) / best)) best = M;
  }
  if (best == 0) return false;
  PcgPlan p;
  p.M = best; p.chunks = (nsb + best - 1) / best; p.Tp = pcg_pow2_at_least(p.chunks);
  const int want = (n_rows + kPcgWaves - 1) / kPcgWaves;            // one wave per pose for the products
  const int helpers = std::max(1, std::min(cus - p.chunks - 1, want));
  p.G = p.chunks + helpers + 1;
  // product workgroups keep their poses' blocks in LDS (80 bytes per entry and wave): at least 96 KB whatever the tiles need
  p.lds_bytes = std::max(pcg_lds_doubles(B, p.M, p.chunks) * sizeof(double), (size_t)96 * 1024);
  p.capw = (int)(p.lds_bytes / (kPcgWaves * 80));
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
    raised = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pcg_fused<B>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) == hipSuccess;
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
  a.M = pl.M; a.chunks = pl.chunks; a.Tp = pl.Tp; a.G = pl.G; a.capw = pl.capw;
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
