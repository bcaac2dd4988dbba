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
//   * chunk workgroups c = 0 .. chunks-1: each owns M consecutive nodes of the reduction tree (M * Wb poses).  Every thread keeps the
//     coefficients of a few fixed items of the preconditioner's sweep in REGISTERS for the whole solve (hitl_pcg_sweep.h) and -- one row
//     per thread -- x, r, p, s = A p of the owned rows; only the sweep's small vectors go through LDS.  Per iteration: the vector updates,
//     the local levels of the forward sweep, later the back-substitution.
//   * the top workgroup (the last one): the nodes 0, M, 2M, ... (one per chunk): the upper levels of the sweep, same engine.
//   * product workgroups (the ones between): w = A z, one wave per pose over the pose's band and off-band blocks.  The matrix does
//     not change during a solve: a workgroup keeps its poses' 3x3 blocks (already oriented) and column indices in LDS; consecutive
//     poses go to different workgroups (a map's first poses can have ten times the off-band partners of the others).
//
// Where a CG iteration's time goes (in-kernel probe, HITL_TIMING=1; 1 000-pose Figure-8 map, 16 chunks of 16 nodes, 125 product
// workgroups; profiles/r06_pcg.md): chunk forward sweep 3.4 us, hand-off 1.8, top 4.3 + 1.0, hand-off 0.7, chunk back-substitution +
// publishing z 3.7, products 1.6 after z, their hand-off back 3.5 (192 rows from 64 producers per chunk: the last one counts) -- ~20 us,
// against ~58 us for the eight launches.  What did NOT help, each measured: the sweep's tiles in LDS (plain layout: 5-7 bank passes per
// read; conflict-free level-major layout: its address arithmetic cost the same), one wave per chunk without barriers, a "lazy" forward
// sweep (every product lane re-reading 2 l + 1 vectors), several loads in flight per watcher (more polling traffic slowed every
// hand-off), one watcher per workgroup in front of the gathers (a second memory round trip).
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
#include "hitl_pcg_sweep.h"

namespace hitl {

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

// A reader waits for EXACTLY the tag of its iteration, so no producer may publish iteration it + 1 into a granule before every reader
// of iteration it has taken it.  That holds by data dependence, not by timing: a chunk hands its boundary record over (it + 1) only after
// it has taken the top's z (it), which the top published after reading every record (it); the top publishes (it + 1) after every
// chunk's record (it + 1), i.e. after they took its z (it); a chunk publishes z (it + 1) after it has taken every product workgroup's
// w.z (it), each published after ALL that workgroup's gathers (it); a product item publishes w (it + 1) only after it has taken z (it + 1)
// of its OWN pose (every item does, also the ones that do not need the value), which that pose's chunk published after taking w (it).
// The one value without such a chain is r.r (read by the product workgroups AFTER they have published w.z): it alternates between two
// buffers, and the buffer of iteration it is rewritten at it + 2, which needs the product workgroups' w.z (it + 1), published after
// their look at r.r (it).
struct PcgPoll { unsigned* abort_w; unsigned spins; unsigned launch; double* dbg; };      // dbg (pinned): workgroup * 1000 + thread + site / 100 of a wait that ran out   // abort word = launch << 2 | code: a word left by an earlier launch means nothing
__device__ __forceinline__ unsigned pcg_abort_code(const PcgPoll& pp) {
  const unsigned v = __hip_atomic_load(pp.abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return ((v >> 2) == pp.launch) ? (v & 3u) : 0u;
}
__device__ __forceinline__ void pcg_abort_set(const PcgPoll& pp, unsigned code) {
  __hip_atomic_store(pp.abort_w, (pp.launch << 2) | code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every wait is bounded and the abort word ends all of them: 1 = the chunk workgroups have ended the solve (breakdown), 2 = somebody's
// wait ran out.  Called every 64th poll of a spin loop; true = give up.
__device__ __forceinline__ bool pcg_give_up(unsigned k, const PcgPoll& pp, int site) {
  if (k >= pp.spins) {
    pcg_abort_set(pp, 2u);
    if (pp.dbg != nullptr) __hip_atomic_store(pp.dbg, 1000.0 * (double)blockIdx.x + (double)threadIdx.x + 0.01 * (double)site, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return true;
  }
  return pcg_abort_code(pp) != 0u;
}

// spin until the granule carries `tag`.  SLEEP: polls of data that is usually a while away back off between loads.
template <bool SLEEP>
__device__ __forceinline__ double gran_take(const v2f64* p, double tag, const PcgPoll& pp, bool& bad) {
  for (unsigned k = 1;; ++k) {
    const v2f64 g = gran_ld(p);
    if (g.y == tag) return g.x;
    if ((k & 63u) == 0u && pcg_give_up(k, pp, 1)) { bad = true; return 0.0; }
    if (SLEEP) __builtin_amdgcn_s_sleep(2);
  }
}
// three granules at once (the three scalars of a pose): all loads of a round are in flight together
__device__ __forceinline__ void gran_take3(const v2f64* p, double tag, const PcgPoll& pp, bool& bad, double& v0, double& v1, double& v2) {
  for (unsigned k = 1;; ++k) {
    const v2f64 g0 = gran_ld(p), g1 = gran_ld(p + 1), g2 = gran_ld(p + 2);
    if (g0.y == tag && g1.y == tag && g2.y == tag) { v0 = g0.x; v1 = g1.x; v2 = g2.x; return; }
    if ((k & 63u) == 0u && pcg_give_up(k, pp, 2)) { bad = true; v0 = v1 = v2 = 0.0; return; }
    __builtin_amdgcn_s_sleep(4);                    // a thousand product waves wait here through a whole sweep: poll politely
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
      if ((spin & 63u) == 0u && pcg_give_up(spin, pp, 3)) { bad = true; return; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
}

// sum over the wave by DPP (quad permutes, row mirrors, row broadcasts: ALU moves, no LDS crossbar as in __shfl_xor), the same in
// every lane.  A fixed order of additions, not the butterfly's: used inside this file only.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get(double v) {      // the DPP source lane's value; 0 where the row is masked or there is no source
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_get<0xB1, 0xf>(v);       // quad_perm [1,0,3,2]
  v += dpp_get<0x4E, 0xf>(v);       // quad_perm [2,3,0,1]
  v += dpp_get<0x141, 0xf>(v);      // row_half_mirror: the other quad pair of the 8
  v += dpp_get<0x140, 0xf>(v);      // row_mirror: the other half of the row of 16 -- every lane of a row now holds the row's sum
  v += dpp_get<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
  v += dpp_get<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return readlane_f64(v, 63);
}

// has any thread of the workgroup given a wait up?  (s_bad: LDS, zero at the start, only ever set)
__device__ __forceinline__ bool pcg_any_bad(bool bad, int* s_bad) {
  if (bad) *s_bad = 1;
  lds_barrier();
  return *s_bad != 0;
}

// LDS written by some lanes of a wave and read by others of the SAME wave: the hardware keeps a wave's LDS operations in order;
// the compiler must not move a read above the write it depends on
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Granules written by DIFFERENT workgroups never share a 128-byte line (write-through partial writes from several CUs to one line
// serialise at the memory side: the products' w took 2.8-3.5 us to reach the chunks while the coalesced records took 0.7-1.8):
//   w: the three rows of one PART of a pose's product (hitl_lm.cpp cuts a pose's entries into parts of 128) in a 64-byte slot of their own;
//   per-workgroup scalars (r.z and r.r of a chunk, w.z of a product workgroup): one 128-byte slot each (8 granules);
//   a chunk's boundary record: its 3 B granules rounded up to whole lines.
__host__ __device__ inline size_t pcg_w_slot(size_t pose, int part, int comp) { return ((pose * kPcgMaxParts + part) * 4) + comp; }
constexpr int kPcgLine = 8;                       // granules per 128-byte line
__host__ __device__ inline int pcg_ctr_stride(int B) { return ((3 * B + kPcgLine - 1) / kPcgLine) * kPcgLine; }

struct PcgArgs {
  int n_poses, npad, nsb, Wb;
  int M, chunks, Tp, G;          // chunk size (a power of two), chunks = top nodes, Tp = the power of two >= chunks, workgroups:
                                 // [0, chunks) the chunks, [chunks, G - 1) the product workgroups, G - 1 the top
  int capw;                      // product entries a workgroup keeps in LDS (a pool shared by its waves)
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
  const int* items;              // the products cut into items: (pose, first entry, entries, part) x n_items
  int n_items;
  const unsigned char* parts;    // parts of every pose row
  v2f64* wG;                     // (npad / 3) x kPcgMaxParts x 4: the parts of w = A z, summed by the rows' chunk workgroups
  v2f64* ctrG;                   // chunks x 3 B: contribution to the chunk's own first node | to the next chunk's first node | r at the first node
  v2f64* topL;                   // chunks x B: z at the chunk's first node
  v2f64* topR;                   // chunks x B: z at the NEXT chunk's first node (the chunk's right boundary)
  v2f64* gamG; v2f64* rrG;       // r.z (chunks lines) and r.r (2 x chunks lines: two buffers taken in turn) of the chunks' rows
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
  for (int c = lane + (lane == 0 ? WAVE : 0); c < T; c += WAVE) v += gran_take<false>(rrG + (size_t)c * kPcgLine, tag, pp, bad);
  return wave_sum_dpp(v);
}

template <int B, int NS>
__global__ __launch_bounds__(kPcgThreads) void k_pcg_fused(PcgArgs a, FactorView fv, LinView lin, CrView cr, const double* __restrict__ scale) {
  constexpr int BB = B * B;
  extern __shared__ double lds[];
  __shared__ int s_bad;
  __shared__ double s_red[2 * kPcgWaves];
  __shared__ double s_sc[8];               // [0] rr, [1] gamma, [2] delta, [6] bb, [7] rr0
  const int tid = threadIdx.x, wg = blockIdx.x, wave = tid >> 6, lane = tid & 63;
  const int M = a.M, T = a.chunks, Tp = a.Tp, Hh = a.G - 1 - a.chunks;
  const bool is_top = (wg == a.G - 1), is_chunk = (wg < a.chunks);
  const PcgPoll pp = {a.abort_w, a.spins, a.launch, a.result + 1};
  {  // r0.r0 and b.b from the residual pass in front of this launch (a kernel boundary: plain data), every workgroup alike
    double r_ = 0.0, b_ = 0.0;
    if (tid < WAVE) {
      for (int k = tid; k < a.nblk0; k += WAVE) { r_ += a.partials0[(size_t)k * 8]; b_ += a.partials0[(size_t)k * 8 + 1]; }
      r_ = wave_sum(r_); b_ = wave_sum(b_);
    }
    if (tid == 0) { s_sc[7] = r_; s_sc[6] = b_; s_bad = 0; }
  }
  int it = 0;
  int status = 0;
  double rr = 0.0, bb = 0.0;

  if (!is_chunk && !is_top) {
    // =============================== product workgroup: w = A z over its waves' poses ===============================
    // A wave's poses' entries (band blocks both ways + off-band blocks), oriented, sit in LDS: SoA blkS[q][capw] + colS[capw]; entries
    // beyond the pool are read from memory every iteration (never on maps of hitl-slam's size).  Per pose: sum over its entries of
    // block * (scale z)[column], reduced over the wave, finished by lanes 0..2.
    // (capw: the WORKGROUP's pool of LDS entries, shared by its waves in proportion to what their poses need: a map's first poses can
    // have ten times the off-band partners of the others)
    const int capw = a.capw;
    double* blkS = lds;
    int* colS = reinterpret_cast<int*>(blkS + (size_t)capw * 9);
    const int W1 = a.Wb + 1, nband = 2 * a.Wb + 1;
    // wave `wave` of product workgroup h takes poses wave * Hh + h, + Hh * 8, ...: CONSECUTIVE poses go to different workgroups
    const int gw = wave * Hh + (wg - a.chunks), TW = Hh * kPcgWaves;
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
    // The products are cut into ITEMS on the host (hitl_lm.cpp): (pose, first entry, entries <= 128 except a last part, part number); item t
    // goes to wave (t mod Hh * 8): wave `wave` of product workgroup h takes items wave * Hh + h, + Hh * 8, ... -- consecutive items (the
    // parts of one pose, neighbouring poses) go to DIFFERENT workgroups.
    int o_wave = 0;                                     // this wave's first entry in the pool
    {
      int cnt = 0;
      for (int t = gw; t < a.n_items; t += TW) {
        const int i = a.items[4 * t];
        if (i >= a.n_poses || fv.fixed[i]) continue;
        cnt += a.items[4 * t + 2];
      }
      if (lane == 0) s_red[wave] = (double)cnt;
      lds_barrier();
      for (int k = 0; k < wave; ++k) o_wave += (int)s_red[k];
      lds_barrier();
      int o = o_wave;
      for (int t = gw; t < a.n_items; t += TW) {
        const int i = a.items[4 * t], first = a.items[4 * t + 1], cnt_t = a.items[4 * t + 2];
        if (i >= a.n_poses || fv.fixed[i]) continue;
        const int e0 = (fv.n_offband > 0) ? fv.obadj_off[i] : 0;
        for (int q = lane; q < cnt_t && o + q < capw; q += WAVE) {
          const double* blk = nullptr;
          int j = -1, tr = 0;
          const bool ok = entry(i, e0, first + q, blk, j, tr);
          colS[o + q] = ok ? j : -1;
          if (ok) {
#pragma unroll
            for (int c = 0; c < 9; ++c) blkS[(size_t)c * capw + o + q] = tr ? blk[(c % 3) * 3 + c / 3] : blk[c];
          }
        }
        o += cnt_t;
      }
    }
    lds_barrier();
    bb = s_sc[6]; rr = s_sc[7];
    const double target = a.tol2 * bb;
    if (!(rr == rr)) status = 3;
    const bool run = (status == 0) && (rr > target) && a.max_it > 0;
    while (run) {
      const double tag = a.tag0 + (double)(it + 1);
      bool bad = false;
      PCG_PROBE(0);
      // The products come FIRST, the question "is the solve over?" after them: z arrives tagged, so a wave starts on it the moment it is
      // there, and r.r (published right behind z) is looked at while the chunk workgroups take w in.  When the solve IS over the last
      // products were for nothing (nobody takes them).
      double dz = 0.0;
      int o = o_wave;
      for (int t = gw; t < a.n_items; t += TW) {
        const int i = a.items[4 * t], first = a.items[4 * t + 1], cnt_t = a.items[4 * t + 2], part = a.items[4 * t + 3];
        if (i >= a.n_poses || fv.fixed[i]) {                                  // identity rows (padding, constant poses; their scale is 1): w = z
          // (the host cuts items by the entry count alone: a constant pose with many partners has several; part 0 carries z, the others 0)
          if (lane < 3) {
            // EVERY item waits for z of its own pose before it publishes (here too, where the value is not needed): the row's chunk
            // publishes that z only after it has taken the previous iteration's w in -- a product that ran ahead would overwrite a
            // granule whose tag its reader is still waiting for
            const double zt = gran_take<false>(a.zsG + 3 * i + lane, tag, pp, bad);
            const double z = (part == 0) ? zt : 0.0;
            gran_st(a.wG + pcg_w_slot(i, part, lane), z, tag);
            dz += z * z;
          }
          continue;
        }
        const int e0 = (fv.n_offband > 0) ? fv.obadj_off[i] : 0;
        // the row's constants (lanes 0..2), asked for before the gather so that their latency hides under it
        double sa = 1.0, dd = 0.0;
        if (lane < 3) {
          sa = scale[3 * i + lane];
          const double dg = sa * sa * lin.Hband[(size_t)i * W1 * 9 + lane * 4];
          dd = (part == 0) ? fmin(fmax(dg, a.min_diag), a.max_diag) / a.mu / sa : 0.0;       // the LM diagonal rides with the part that holds the diagonal block
        }
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, d0 = 0.0, d1 = 0.0, d2 = 0.0;      // d: (scale z) of the pose's own column
        if (part != 0 && lane < 3) {                    // parts without the diagonal block ask for the own column too (for w.z)
          const double v = gran_take<false>(a.zsG + 3 * i + lane, tag, pp, bad);
          d0 = v;
        }
        for (int base = 0; base < cnt_t; base += 5 * WAVE) {
          // up to five entries per lane and round (an item has 128 entries at most, except the last part of a pose with more than 1024):
          // every granule load of a round is in flight before the first tag is looked at; entries whose z has not arrived are asked for
          // again TOGETHER
          int jj[5];
          v2f64 g[5][3];
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            const int q = base + lane + WAVE * k;
            jj[k] = -1;
            if (q < cnt_t) {
              if (o + q < capw) jj[k] = colS[o + q];
              else { const double* blk = nullptr; int tr = 0, j = -1; if (entry(i, e0, first + q, blk, j, tr)) jj[k] = j; }
            }
          }
          for (unsigned spin = 1;; ++spin) {
            bool all = true;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
              if (jj[k] < 0) continue;
              if (spin > 1 && g[k][0].y == tag && g[k][1].y == tag && g[k][2].y == tag) continue;
              g[k][0] = gran_ld(a.zsG + 3 * jj[k]); g[k][1] = gran_ld(a.zsG + 3 * jj[k] + 1); g[k][2] = gran_ld(a.zsG + 3 * jj[k] + 2);
            }
#pragma unroll
            for (int k = 0; k < 5; ++k)
              if (jj[k] >= 0 && !(g[k][0].y == tag && g[k][1].y == tag && g[k][2].y == tag)) all = false;
            if (all) break;
            if ((spin & 63u) == 0u && pcg_give_up(spin, pp, 4)) { bad = true; break; }
            __builtin_amdgcn_s_sleep(4);
          }
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            if (jj[k] < 0) continue;
            const int q = base + lane + WAVE * k;
            const double v0 = g[k][0].x, v1 = g[k][1].x, v2 = g[k][2].x;
            double b9[9];
            if (o + q < capw) {
#pragma unroll
              for (int c = 0; c < 9; ++c) b9[c] = blkS[(size_t)c * capw + o + q];
            } else {
              const double* blk = nullptr;
              int tr = 0, j = -1;
              (void)entry(i, e0, first + q, blk, j, tr);
#pragma unroll
              for (int c = 0; c < 9; ++c) b9[c] = tr ? blk[(c % 3) * 3 + c / 3] : blk[c];
            }
            if (first + q == 0) { d0 = v0; d1 = v1; d2 = v2; }
            acc0 += b9[0] * v0 + b9[1] * v1 + b9[2] * v2;
            acc1 += b9[3] * v0 + b9[4] * v1 + b9[5] * v2;
            acc2 += b9[6] * v0 + b9[7] * v1 + b9[8] * v2;
          }
        }
        acc0 = wave_sum_dpp(acc0); acc1 = wave_sum_dpp(acc1); acc2 = wave_sum_dpp(acc2);
        if (part == 0) { d1 = readlane_f64(d1, 0); d2 = readlane_f64(d2, 0); }
        if (lane < 3) {
          const double sz = (part != 0 || lane == 0) ? d0 : (lane == 1) ? d1 : d2;       // part 0: lane 0 took the diagonal block's column itself
          const double accv = (lane == 0) ? acc0 : (lane == 1) ? acc1 : acc2;
          const double y = sa * accv + dd * sz;            // this part of (S H S + D^2) z, with z = sz / sa
          gran_st(a.wG + pcg_w_slot(i, part, lane), y, tag);
          dz += y * (sz / sa);
        }
        o += cnt_t;
      }
      {
        dz = wave_sum_dpp(dz);
        if (lane == 0) s_red[wave] = dz;
        lds_barrier();
        if (tid == 0) {
          double t = 0.0;
#pragma unroll
          for (int k = 0; k < kPcgWaves; ++k) t += s_red[k];
          gran_st(a.delG + (size_t)(wg - a.chunks) * kPcgLine, t, tag);
        }
      }
      PCG_PROBE(5);
      if (wave == 0) { const double v = pcg_rr_sum(a.rrG + (size_t)(it & 1) * T * kPcgLine, T, tag, pp, bad, lane); if (lane == 0) s_sc[0] = v; }
      if (pcg_any_bad(bad, &s_bad)) { status = 5; break; }
      rr = s_sc[0];
      PCG_PROBE(4);
      if (!(rr == rr)) { status = 3; break; }
      if (!(rr > target) || it >= a.max_it) break;
      PCG_PROBE(7);
      ++it;
    }
    if (tid == 0 && status == 5 && pcg_abort_code(pp) != 1u) __hip_atomic_store(a.result + 2, 2.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }

  // =============================== chunk workgroups and the top ===============================
  // Small vectors in LDS, B + 1 doubles per node: w0 (the sweep's right-hand side, updated in place), xs (the solution; zeros at nodes that
  // do not exist), ca / cb (what an eliminated node hands to its left / right neighbour); the top also keeps Dinv of node 0.  Everything
  // else lives in registers.
  constexpr int BP = B + 1;
  const int Mloc = is_top ? Tp : M;
  double* w0 = lds;
  double* xs = w0 + (size_t)(Mloc + 1) * BP;
  double* ca = xs + (size_t)(Mloc + 1) * BP;
  double* cb = ca + (size_t)(Mloc + 1) * BP;
  double* acc0 = cb + (size_t)(Mloc + 1) * BP;          // chunk: what the interior hands to its first node ...
  double* accM = acc0 + B;                              // ... and to the next chunk's first node
  double* sD0 = accM + B;                               // top only: BB doubles
  const int nv = is_top ? 0 : min(M, a.nsb - wg * M), nvB = nv * B;       // chunk: owned nodes / rows (row t in thread t)
  const size_t row0 = (size_t)wg * M * B;
  PcgItems<B, NS> items;
  if (is_top) pcg_items_init<B, NS>(items, cr, Tp, T - 1, 0, (size_t)M);
  else pcg_items_init<B, NS>(items, cr, M, nv - 1, (size_t)wg * M, 1);
  for (int e = tid; e < 4 * (Mloc + 1) * BP + 2 * B; e += kPcgThreads) w0[e] = 0.0;
  if (is_top) for (int e = tid; e < BB; e += kPcgThreads) sD0[e] = cr.Dinv[e];
  // the owned row of a chunk thread: x, r, p, s = A p, z, scale
  const bool own = is_chunk && tid < nvB;
  double xr = own ? a.x[row0 + tid] : 0.0, rr_ = own ? a.r0[row0 + tid] : 0.0, pr = 0.0, sr = 0.0, zr = 0.0;
  const double scr = own ? scale[row0 + tid] : 1.0;
  const int own_slot = (tid / B) * BP + tid % B;
  const int own_parts = own ? (int)a.parts[(row0 + tid) / 3] : 0;
  lds_barrier();
  bb = s_sc[6]; rr = s_sc[7];
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
      double rsq = 0.0;
      if (own) {
        if (it > 0) { xr += alpha * pr; rr_ -= alpha * sr; }
        w0[own_slot] = rr_;
        rsq = rr_ * rr_;
      }
      rsq = wave_sum_dpp(rsq);
      if (lane == 0) s_red[wave] = rsq;
      lds_barrier();
      PCG_PROBE(9);
      for (int s = 1; s < M; s *= 2) {
        pcg_fwd_products<B, NS>(items, s, w0, ca, cb);
        lds_barrier();
        pcg_fwd_survivors<B>(s, M, nv - 1, w0, ca, cb, acc0, accM);
        lds_barrier();
      }
      PCG_PROBE(10);
      if (tid < 3 * B) {                                 // what the chunk's interior hands to its two boundary nodes, and r at its first node
        const int q = tid % B;
        double v;
        if (tid < 2 * B) { v = acc0[tid]; acc0[tid] = 0.0; }          // (acc0 and accM are adjacent; cleared for the next iteration)
        else v = w0[q];
        gran_st(a.ctrG + (size_t)wg * pcg_ctr_stride(B) + tid, v, tag);
      }
      PCG_PROBE(1);
      // ---- P3: z at the two boundary nodes comes back from the top; back-substitution; r.z ----
      if (tid < 2 * B) {
        const int q = tid % B;
        if (tid < B) xs[q] = gran_take<true>(a.topL + (size_t)wg * B + q, tag, pp, bad);
        else xs[M * BP + q] = (wg + 1 < T) ? gran_take<true>(a.topR + (size_t)wg * B + q, tag, pp, bad) : 0.0;
      }
      if (pcg_any_bad(bad, &s_bad)) { status = 5; break; }
      PCG_PROBE(2);
      for (int s = M / 2; s >= 1; s /= 2) {
        pcg_back_level<B, NS>(items, s, w0, xs);
        lds_barrier();
      }
      PCG_PROBE(11);
      double gz = 0.0;
      if (own) {
        zr = xs[own_slot];
        gz = rr_ * zr;
        gran_st(a.zsG + row0 + tid, scr * zr, tag);
      }
      gz = wave_sum_dpp(gz);
      if (lane == 0) s_red[kPcgWaves + wave] = gz;
      lds_barrier();
      if (tid == 0) {
        double rr_c = 0.0, gamma_c = 0.0;
#pragma unroll
        for (int k = 0; k < kPcgWaves; ++k) { rr_c += s_red[k]; gamma_c += s_red[kPcgWaves + k]; }
        gran_st(a.gamG + (size_t)wg * kPcgLine, gamma_c, tag);
        gran_st(a.rrG + ((size_t)(it & 1) * T + wg) * kPcgLine, rr_c, tag);
      }
      PCG_PROBE(3);
    } else {
      // ---- P2 (the top): the upper levels: nodes 0, M, 2M, ... ----
      {
        unsigned pend = 0u;
        for (int k = 0; tid + kPcgThreads * k < T * B; ++k) pend |= 1u << k;          // T * B <= 64 * 12: at most two per thread
        for (unsigned spin = 1; pend != 0u; ++spin) {
          for (unsigned m = pend; m != 0u; m &= m - 1u) {
            const int k = __builtin_ctz(m), e = tid + kPcgThreads * k, t = e / B, q = e % B;
            const v2f64 g0 = gran_ld(a.ctrG + (size_t)t * pcg_ctr_stride(B) + 2 * B + q), g1 = gran_ld(a.ctrG + (size_t)t * pcg_ctr_stride(B) + q);
            v2f64 g2; g2.x = 0.0; g2.y = tag;
            if (t >= 1) g2 = gran_ld(a.ctrG + (size_t)(t - 1) * pcg_ctr_stride(B) + B + q);
            if (g0.y == tag && g1.y == tag && g2.y == tag) { w0[t * BP + q] = (g0.x - g1.x) - g2.x; pend &= ~(1u << k); }
          }
          if (pend != 0u) {
            if ((spin & 63u) == 0u && pcg_give_up(spin, pp, 5)) { bad = true; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
      }
      if (pcg_any_bad(bad, &s_bad)) { status = 5; break; }
      PCG_PROBE(1);
      for (int s = 1; s < Tp; s *= 2) {
        pcg_fwd_products<B, NS>(items, s, w0, ca, cb);
        lds_barrier();
        pcg_fwd_survivors<B>(s, Tp, T - 1, w0, ca, cb, nullptr, nullptr);
        lds_barrier();
      }
      if (tid < B) {                                     // the root: x_0 = Dinv_0 w_0
        double v = 0.0;
#pragma unroll
        for (int c = 0; c < B; ++c) v += sD0[tid * B + c] * w0[c];
        xs[tid] = v;
      }
      lds_barrier();
      for (int s = Tp / 2; s >= 1; s /= 2) {
        pcg_back_level<B, NS>(items, s, w0, xs);
        lds_barrier();
      }
      PCG_PROBE(2);
      for (int e = tid; e < T * B; e += kPcgThreads) {
        const double v = xs[(e / B) * BP + e % B];
        gran_st(a.topL + e, v, tag);
        if (e >= B) gran_st(a.topR + (e - B), v, tag);
      }
      PCG_PROBE(3);
    }
    // ---- r.r is out with z; is the solve over? ----
    if (wave == 0) { const double v = pcg_rr_sum(a.rrG + (size_t)(it & 1) * T * kPcgLine, T, tag, pp, bad, lane); if (lane == 0) s_sc[0] = v; }
    if (pcg_any_bad(bad, &s_bad)) { status = 5; break; }
    rr = s_sc[0];
    PCG_PROBE(4);
    if (!(rr == rr)) { status = 3; break; }
    if (!(rr > target) || it >= a.max_it) break;
    if (is_chunk) {
      // ---- P5: w of the owned rows and the global sums come in; alpha, beta; p = z + beta p, s = w + beta s ----
      if (wave == 0) {
        double g = 0.0, d = 0.0;
        for (int c = lane; c < T; c += WAVE) g += gran_take<false>(a.gamG + (size_t)c * kPcgLine, tag, pp, bad);
        {                                                // the product workgroups' w.z: every pending load of a round in flight together
          unsigned pend = 0u;
          for (int k = 0; lane + WAVE * k < Hh; ++k) pend |= 1u << k;
          for (unsigned spin = 1; pend != 0u; ++spin) {
            for (unsigned m = pend; m != 0u; m &= m - 1u) {
              const int k = __builtin_ctz(m);
              const v2f64 gq = gran_ld(a.delG + (size_t)(lane + WAVE * k) * kPcgLine);
              if (gq.y == tag) { d += gq.x; pend &= ~(1u << k); }
            }
            if (pend != 0u && (spin & 63u) == 0u && pcg_give_up(spin, pp, 6)) { bad = true; break; }
          }
        }
        g = wave_sum_dpp(g); d = wave_sum_dpp(d);
        if (lane == 0) { s_sc[1] = g; s_sc[2] = d; }
      }
      double wr = 0.0;
      if (own) {                                       // the row's product arrives in own_parts parts: every pending load of a round in flight together
        const size_t pose = (row0 + tid) / 3;
        const int comp = (int)((row0 + tid) % 3);
        double pv_[kPcgMaxParts];
        unsigned pend = (1u << own_parts) - 1u;
        for (unsigned spin = 1; pend != 0u; ++spin) {
#pragma unroll
          for (int q = 0; q < kPcgMaxParts; ++q) {
            if (!(pend & (1u << q))) continue;
            const v2f64 gq = gran_ld(a.wG + pcg_w_slot(pose, q, comp));
            if (gq.y == tag) { pv_[q] = gq.x; pend &= ~(1u << q); }
          }
          if (pend != 0u) {
            if ((spin & 63u) == 0u && pcg_give_up(spin, pp, 7)) { bad = true; break; }
            __builtin_amdgcn_s_sleep(2);
          }
        }
#pragma unroll
        for (int q = 0; q < kPcgMaxParts; ++q) if (q < own_parts) wr += pv_[q];          // part order: the same sum on every run
      }
      if (pcg_any_bad(bad, &s_bad)) { status = 5; break; }
      const double gamma = s_sc[1], delta = s_sc[2];
      PCG_PROBE(6);
      const double beta = (it == 0 || gamma_old == 0.0) ? 0.0 : gamma / gamma_old;
      const double denom = (it == 0) ? delta : delta - beta * gamma / alpha_old;
      if (!(denom > 0.0)) {
        // p.A p <= 0 (or not finite): with r.z == 0 the residual is exactly zero and nothing is left to do; otherwise a breakdown
        if (gamma != 0.0) status = 1;
        if (tid == 0) pcg_abort_set(pp, 1u);       // the product workgroups and the top leave at their next wait
        break;
      }
      alpha = gamma / denom;
      pr = zr + beta * pr;
      sr = wr + beta * sr;
      gamma_old = gamma; alpha_old = alpha;
    }
    PCG_PROBE(7);
    ++it;
  }
  // ---- the solution back to memory; one record for the host ----
  if (own) a.x[row0 + tid] = xr;
  // status here: 0 = converged or iteration cap, 1 = breakdown, 3 = not finite (every workgroup alike); 5 = a wait was given up: because it ran
  // out (here or elsewhere: the abort word is 2 and the solve is void) or because the chunk workgroups ended the solve (abort word 1: the product
  // workgroups and the top only).  Workgroup 0 (a chunk: one of those that decide) reports; a wait that ran out is reported by everybody who saw it.
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
  const int Tp = pcg_pow2_at_least(T), Mloc = std::max(M, Tp);
  return (size_t)4 * (Mloc + 1) * (B + 1) + 2 * (size_t)B + (size_t)B * B + 8;            // the sweep's vectors (+ Dinv of node 0 in the top)
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
    // every item of a sweep has a slot: 6 B (nodes - 1) items over 512 threads x 5 slots (the top has Tp nodes); a chunk's rows one per thread
    if (T < 2 || T > 64 || T + 2 > cus || M * B > kPcgThreads) continue;
    if (6 * B * (M - 1) + 4 > kPcgThreads * kPcgSlotsMax || 6 * B * (pcg_pow2_at_least(T) - 1) + 4 > kPcgThreads * kPcgSlotsMax) continue;
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
  p.capw = (int)((p.lds_bytes - 64) / 76);                           // the workgroup's pool: 9 doubles + 1 int per entry
  // granules (16 bytes each): zs, w (nsb * B each) | ctr (chunks * 3 B) | topL, topR (chunks * B each) | gamma, r.r (chunks each) | delta (G)
  p.work_doubles = 2 * ((size_t)nsb * B + pcg_w_slot((size_t)(nsb * B + 2) / 3, 0, 0) + 8 + 8 + (size_t)p.chunks * pcg_ctr_stride(B) + 2 * ((size_t)p.chunks * B + kPcgLine) +
                        (size_t)(3 * p.chunks + p.G) * kPcgLine + 16);
  if (out) *out = p;
  return true;
}

size_t pcg_fused_sync_words() { return 32; }

template <int B, int NS>
static bool pcg_launch_tn(hipStream_t st, const PcgPlan& pl, const PcgArgs& a, const FactorView& fv, const LinView& lin, const CrView& cr, const double* scale, int groups) {
  static bool raised = false;
  if (!raised) {
    raised = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pcg_fused<B, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) == hipSuccess;
    if (!raised) { (void)hipGetLastError(); return false; }
  }
  hipLaunchKernelGGL((k_pcg_fused<B, NS>), dim3(groups), dim3(kPcgThreads), pl.lds_bytes, st, a, fv, lin, cr, scale);
  return true;
}
template <int B>
static bool pcg_launch_t(hipStream_t st, const PcgPlan& pl, const PcgArgs& a, const FactorView& fv, const LinView& lin, const CrView& cr, const double* scale, int groups) {
  const int nodes = std::max(pl.M, pl.Tp);
  if (6 * B * (nodes - 1) + 4 <= kPcgThreads * 3) return pcg_launch_tn<B, 3>(st, pl, a, fv, lin, cr, scale, groups);
  return pcg_launch_tn<B, kPcgSlotsMax>(st, pl, a, fv, lin, cr, scale, groups);
}

bool launch_pcg_fused(hipStream_t st, const PcgPlan& pl, int n_poses, int npad, const FactorView& fv, const LinView& lin, const CrView& cr, const double* scale,
                      double mu, double min_diag, double max_diag, double tol, int max_it, double* x, const double* r0, const double* partials0, int nblk0,
                      const int* items, int n_items, const unsigned char* parts,
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
  a.items = items; a.n_items = n_items; a.parts = parts;
  v2f64* g = reinterpret_cast<v2f64*>(work);
  a.zsG = g; g += (size_t)cr.nsb * B;
  a.wG = g; g += pcg_w_slot((size_t)(cr.nsb * B + 2) / 3, 0, 0) + 8;
  g = reinterpret_cast<v2f64*>(((uintptr_t)g + 127) & ~(uintptr_t)127);       // whole lines from here on
  a.ctrG = g; g += (size_t)pl.chunks * pcg_ctr_stride(B);
  a.topL = g; g += (((size_t)pl.chunks * B + kPcgLine - 1) / kPcgLine) * kPcgLine;
  a.topR = g; g += (((size_t)pl.chunks * B + kPcgLine - 1) / kPcgLine) * kPcgLine;
  a.gamG = g; g += (size_t)pl.chunks * kPcgLine;
  a.rrG = g; g += (size_t)2 * pl.chunks * kPcgLine;
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
