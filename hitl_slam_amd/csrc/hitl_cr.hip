// hitl_cr.hip -- the linear solve of the LM step on gfx950:
//   k_jacobi_scale, k_build_system : Jacobi scaling + LM diagonal -> block-tridiagonal super-block system.
//   k_cr_*           : K7b block cyclic reduction = recursive red/black Schur complement of the
//                      banded normal equations (what Ceres SPARSE_SCHUR does on a chain,
//                      JointOptimization.cpp:148), exact preconditioner of the PCG.
//   k_cr_selinv_level, k_cov_extract : diagonal blocks of the inverse from the same factors (pose covariances).
// All arithmetic is fp64.

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hitl_device.h"

namespace hitl {

__global__ void k_jacobi_scale(int n_poses, int npad, int Wb, const double* __restrict__ Hband, int enable,
                               const unsigned char* __restrict__ fixed, double* __restrict__ scale) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= npad) return;
  double s = 1.0;
  if (enable && a < 3 * n_poses && !fixed[a / 3]) {
    const int i = a / 3, c = a % 3;
    s = 1.0 / (1.0 + sqrt(Hband[(size_t)i * (Wb + 1) * 9 + c * 3 + c]));
  }
  scale[a] = s;
}

void launch_jacobi_scale(hipStream_t st, int n_poses, int npad, int Wb, const double* Hband, int enable,
                         const unsigned char* fixed, double* scale) {
  hipLaunchKernelGGL(k_jacobi_scale, dim3((npad + 255) / 256), dim3(256), 0, st, n_poses, npad, Wb, Hband, enable, fixed, scale);
}

// ===========================================================================================
// build_system: banded 3x3-block Hessian -> block-tridiagonal system of B x B super-blocks
// (Wb consecutive poses each) in the Jacobi-scaled space with the LM diagonal
//   D^2 = clamp(diag(S H S), min, max) / mu     
// ===========================================================================================
// entry e = a * B + b of super-block k: D (with the LM diagonal) and L (coupling to super-block k - 1)
__device__ __forceinline__ void cr_source_tile(const CrSource& sc, int B, int k, int e, double& dv, double& lv) {
  const int a = e / B, b = e % B;
  const int Wb = sc.Wb, i = k * Wb + a / 3, ca = a % 3;
  if (i >= sc.n_poses || sc.fixed[i]) { dv = (b == a) ? 1.0 : 0.0; lv = 0.0; return; }     // padding rows and constant poses: identity
  const int W1 = Wb + 1;
  const double sa = sc.scale[3 * i + ca];
  const int i2 = k * Wb + b / 3, cb = b % 3;
  double v = 0.0;
  if (i2 < sc.n_poses && !sc.fixed[i2]) {
    v = (i2 >= i) ? sc.Hband[((size_t)i * W1 + (i2 - i)) * 9 + ca * 3 + cb]
                  : sc.Hband[((size_t)i2 * W1 + (i - i2)) * 9 + cb * 3 + ca];
    v *= sa * sc.scale[3 * i2 + cb];
  }
  if (b == a) v += fmin(fmax(v, sc.min_diag), sc.max_diag) / sc.mu;
  dv = v;
  double l = 0.0;
  if (k > 0) {
    const int i3 = (k - 1) * Wb + b / 3;
    const int d = i - i3;
    if (d <= Wb && !sc.fixed[i3]) l = sc.Hband[((size_t)i3 * W1 + d) * 9 + cb * 3 + ca] * sa * sc.scale[3 * i3 + cb];
  }
  lv = l;
}

// The same entry from a staged copy of the band rows: pose i's record (rl = (Wb + 1) * 9 + 4 doubles: band blocks, 3 scales,
// constant flag) sits at rec + (i - p0) * rl.  Same operations in the same order as cr_source_tile.
__device__ __forceinline__ void cr_source_tile_rec(const double* rec, int rl, int p0, int n_poses, int Wb, double mu, double min_diag,
                                                   double max_diag, int B, int k, int e, double& dv, double& lv) {
  const int a = e / B, b = e % B;
  const int i = k * Wb + a / 3, ca = a % 3;
  const int so = (Wb + 1) * 9;
  const double* ri = rec + (size_t)(i - p0) * rl;
  if (i >= n_poses || ri[so + 3] != 0.0) { dv = (b == a) ? 1.0 : 0.0; lv = 0.0; return; }
  const double sa = ri[so + ca];
  const int i2 = k * Wb + b / 3, cb = b % 3;
  const double* r2 = rec + (size_t)(i2 - p0) * rl;
  double v = 0.0;
  if (i2 < n_poses && r2[so + 3] == 0.0) {
    v = (i2 >= i) ? ri[(i2 - i) * 9 + ca * 3 + cb] : r2[(i - i2) * 9 + cb * 3 + ca];
    v *= sa * r2[so + cb];
  }
  if (b == a) v += fmin(fmax(v, min_diag), max_diag) / mu;
  dv = v;
  double l = 0.0;
  if (k > 0) {
    const int i3 = (k - 1) * Wb + b / 3;
    const int d = i - i3;
    const double* r3 = rec + (size_t)(i3 - p0) * rl;
    if (d <= Wb && r3[so + 3] == 0.0) l = r3[d * 9 + cb * 3 + ca] * sa * r3[so + cb];
  }
  lv = l;
}

__device__ __forceinline__ double cr_source_rhs(const CrSource& sc, int k, int a) {
  const int i = k * sc.Wb + a / 3, ca = a % 3;
  if (i >= sc.n_poses || sc.fixed[i]) return 0.0;
  return -sc.scale[3 * i + ca] * sc.g[3 * i + ca];
}

// one thread per (super-block k, row a, column b): coalesced stores of D and L
__global__ void k_build_system(CrSource sc, CrView cr, double* __restrict__ rhs_work) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int B = cr.B, BB = B * B;
  if (t >= cr.nsb * BB) return;
  const int k = t / BB, e = t % BB;
  double dv, lv;
  cr_source_tile(sc, B, k, e, dv, lv);
  cr.D[t] = dv;
  cr.L[t] = lv;
  if (e % B == 0) {
    const int a = e / B;
    const double v = cr_source_rhs(sc, k, a);
    sc.rhs[k * B + a] = v;
    if (rhs_work) rhs_work[k * B + a] = v;      // the copy the factor-solve consumes in place
  }
}

void launch_build_system(hipStream_t st, int n_poses, int Wb, const LinView& lin, const double* scale,
                         const unsigned char* fixed, double mu, double min_diag, double max_diag, const CrView& cr,
                         double* rhs, double* rhs_work) {
  const int tot = cr.nsb * cr.B * cr.B;
  const CrSource sc = {lin.Hband, lin.g, scale, fixed, mu, min_diag, max_diag, n_poses, Wb, rhs, nullptr, 0};
  hipLaunchKernelGGL(k_build_system, dim3((tot + 255) / 256), dim3(256), 0, st, sc, cr, rhs_work);
}

// ===========================================================================================
// block cyclic reduction.  One 64-lane workgroup per super-block; B x B tiles live in LDS.
// Level stride s: nodes s, 3s, 5s.. are eliminated, nodes 0, 2s, 4s.. survive and absorb the
// Schur complement.  k_cr_invert inverts the level-0 odd nodes; k_cr_level then does one whole
// level per launch (survivor update + rhs sweep + inverse of the next level's odd nodes) and
// k_cr_top finishes the last kCrTopNodes nodes, the root and their back-substitution.
// ===========================================================================================
// Register-resident Gauss-Jordan sweep of one SPD super-block: lane (grp, c) owns column c of
// the augmented matrix [D | I | L_k | L_{k+s}^T]; for pivot p the pivot column is broadcast from
// lane p with v_readlane (no LDS, no barriers, no pivoting needed for SPD tiles).  Result:
// [I | Dinv | Gl = Dinv L_k | Gr = Dinv L_{k+s}^T].
// 1 / x for a pivot of an SPD tile (positive, well scaled after Jacobi scaling + LM diagonal): v_rcp_f64 seed + two Newton
// steps (full double precision for normal inputs) instead of the 11-instruction IEEE division sequence, 9 times per tile
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

template <int B>
__device__ __forceinline__ void gauss_jordan_cols(double (&col)[B]) {
#pragma unroll
  for (int p = 0; p < B; ++p) {
    double m[B];
#pragma unroll
    for (int r = 0; r < B; ++r) m[r] = readlane_f64(col[r], p);   // current column p (lane p, group 0)
    const double inv = fast_rcp(m[p]);
    const double vk = col[p] * inv;
#pragma unroll
    for (int r = 0; r < B; ++r) col[r] = (r == p) ? vk : col[r] - m[r] * vk;
  }
}

template <int B>
__device__ __forceinline__ void cr_eliminate_node(const CrView& cr, int s, int k, int lane) {
  static_assert(4 * B <= WAVE, "four column groups must fit one wave");
  const size_t off = (size_t)k * B * B;
  const bool has_right = (k + s) < cr.nsb;
  const int grp = lane / B, c = lane % B;
  const double* Dk = cr.D + off;
  const double* Lk = cr.L + off;
  const double* Lr = cr.L + (size_t)(has_right ? (k + s) : k) * B * B;
  double col[B];
#pragma unroll
  for (int r = 0; r < B; ++r) {
    double v = 0.0;
    if (grp == 0) v = Dk[r * B + c];
    else if (grp == 1) v = (r == c) ? 1.0 : 0.0;
    else if (grp == 2) v = Lk[r * B + c];
    else if (grp == 3 && has_right) v = Lr[c * B + r];      // column c of L_{k+s}^T
    col[r] = v;
  }
  gauss_jordan_cols<B>(col);
  double* dst = (grp == 1) ? cr.Dinv + off : (grp == 2) ? cr.Gl + off : cr.Gr + off;
  if (grp >= 1 && grp <= 3) {
#pragma unroll
    for (int r = 0; r < B; ++r) dst[r * B + c] = col[r];
  }
}

// Schur-complement update of survivor j at level stride s (LDS-free):
//   D_j  -= L_j Gr[k1] + L_{k2}^T Gl[k2]        L'_j = -L_j Gl[k1]
// Lane (grp, c) owns column c of the right operands (grp 0: Gr[k1], Gl[k2] -> D_j; grp 1: Gl[k1]
// -> L'_j); the left matrices L_j and L_{k2}^T are the same for every lane (uniform loads).
// With w != nullptr the right-hand side rides along (w_j -= Gr[k1]^T w_k1 + Gl[k2]^T w_k2), which
// saves the separate forward sweep when a factorisation is followed by exactly one solve.
template <int B>
__device__ __forceinline__ void cr_update_node(const CrView& cr, int s, int j, int lane, double* w, int nrhs, size_t stride) {
  const size_t off = (size_t)j * B * B;
  const int k1 = j - s, k2 = j + s;
  const bool has1 = k1 >= 0, has2 = k2 < cr.nsb, has_ll = (k1 - s) >= 0;
  const int grp = lane / B, c = lane % B;
  if (grp > 1) return;
  const double* Lj = cr.L + off;                                   // overwritten below: no __restrict__
  const double* L2 = cr.L + (size_t)(has2 ? k2 : j) * B * B;
  double b1[B], b2[B], acc[B];
  const double* B1 = (grp == 0 ? cr.Gr : cr.Gl) + (size_t)(has1 ? k1 : j) * B * B;
  const double* B2 = cr.Gl + (size_t)(has2 ? k2 : j) * B * B;
#pragma unroll
  for (int r = 0; r < B; ++r) {
    b1[r] = has1 ? B1[r * B + c] : 0.0;
    b2[r] = (has2 && grp == 0) ? B2[r * B + c] : 0.0;
    acc[r] = 0.0;
  }
  if (has1) {
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int q = 0; q < B; ++q) acc[r] += Lj[r * B + q] * b1[q];          // (L_j * B1)[r][c]
  }
  if (has2) {
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int q = 0; q < B; ++q) acc[r] += L2[q * B + r] * b2[q];          // (L_{k2}^T * Gl[k2])[r][c]
  }
  if (grp == 0) {
    if (w != nullptr) {   // rhs forward sweep: w_j[c] -= Gr[k1][:,c] . w_k1 + Gl[k2][:,c] . w_k2
      for (int v = 0; v < nrhs; ++v) {
        double* wv = w + stride * v;
        double wacc = wv[(size_t)j * B + c];
        if (has1) {
          const double* wk = wv + (size_t)k1 * B;
#pragma unroll
          for (int r = 0; r < B; ++r) wacc -= b1[r] * wk[r];
        }
        if (has2) {
          const double* wk = wv + (size_t)k2 * B;
#pragma unroll
          for (int r = 0; r < B; ++r) wacc -= b2[r] * wk[r];
        }
        wv[(size_t)j * B + c] = wacc;
      }
    }
#pragma unroll
    for (int r = 0; r < B; ++r) cr.D[off + r * B + c] -= acc[r];
  } else if (has1) {
    // every read of the old L_j happened above (same wave, in order); now it is replaced
#pragma unroll
    for (int r = 0; r < B; ++r) cr.L[off + r * B + c] = has_ll ? -acc[r] : 0.0;
  }
}

// One launch per big level (instead of eliminate + update): the workgroup of survivor j
//  1. forms, from the inverses Dinv[k1], Dinv[k2] of its two eliminated neighbours (computed by
//     the PREVIOUS level's launch, step 3 below), the products
//        Gr[k1] = Dinv_k1 L_j^T     Gl[k1] = Dinv_k1 L_k1     Gl[k2] = Dinv_k2 L_k2
//     (Gr/Gl of node k are stored by its right-hand survivor; the back substitution reads them),
//  2. applies the Schur complement  D_j -= L_j Gr[k1] + L_k2^T Gl[k2],  L'_j = -L_j Gl[k1]  and the
//     right-hand-side sweep  w_j -= Gr[k1]^T w_k1 + Gl[k2]^T w_k2,
//  3. if j itself is eliminated at the next level, inverts its new D_j (register-resident
//     Gauss-Jordan) so that level needs no separate inversion launch either.
template <int B>
__device__ __forceinline__ void lds_load(double* dst, const double* src) {
  for (int e = threadIdx.x; e < B * B; e += blockDim.x) dst[e] = src[e];
}

// C = A * Bm (TA: use A^T, TB: use Bm^T), all B x B tiles in LDS, one wave
template <int B, bool TA, bool TB>
__device__ __forceinline__ void lds_matmul(double* C, const double* A, const double* Bm) {
  for (int e = threadIdx.x; e < B * B; e += blockDim.x) {
    const int r = e / B, c = e % B;
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < B; ++q) v += (TA ? A[q * B + r] : A[r * B + q]) * (TB ? Bm[c * B + q] : Bm[q * B + c]);
    C[e] = v;
  }
}

// two waves per survivor: the 81-element tile loops (loads, three products, Schur update) take one pass instead of two
constexpr int kCrLevelThreads = 128;
template <int B>
__global__ __launch_bounds__(kCrLevelThreads) void k_cr_level(CrView cr, int s, int invert_next, double* w, int nrhs, size_t stride) {
  __shared__ double sLj[B * B], sL1[B * B], sL2[B * B], sI1[B * B], sI2[B * B], sGr1[B * B], sGl1[B * B], sGl2[B * B], sD[B * B];
  const int j = 2 * s * blockIdx.x;
  if (j >= cr.nsb) return;
  const size_t BB = (size_t)B * B, off = (size_t)j * BB;
  const int k1 = j - s, k2 = j + s;
  const bool has1 = k1 >= 0, has2 = k2 < cr.nsb, has_ll = (k1 - s) >= 0;
  lds_load<B>(sD, cr.D + off);
  if (has1) { lds_load<B>(sLj, cr.L + off); lds_load<B>(sL1, cr.L + (size_t)k1 * BB); lds_load<B>(sI1, cr.Dinv + (size_t)k1 * BB); }
  if (has2) { lds_load<B>(sL2, cr.L + (size_t)k2 * BB); lds_load<B>(sI2, cr.Dinv + (size_t)k2 * BB); }
  __syncthreads();
  if (has1) {
    lds_matmul<B, false, true>(sGr1, sI1, sLj);       // Dinv_k1 * L_j^T
    if (has_ll) lds_matmul<B, false, false>(sGl1, sI1, sL1);   // Dinv_k1 * L_k1
    else for (int e = threadIdx.x; e < B * B; e += kCrLevelThreads) sGl1[e] = 0.0;
  }
  if (has2) lds_matmul<B, false, false>(sGl2, sI2, sL2);       // Dinv_k2 * L_k2
  __syncthreads();
  // factors of the eliminated neighbours, for the back substitution
  if (has1) for (int e = threadIdx.x; e < B * B; e += kCrLevelThreads) { cr.Gr[(size_t)k1 * BB + e] = sGr1[e]; cr.Gl[(size_t)k1 * BB + e] = sGl1[e]; }
  if (has2 && (k2 + s) >= cr.nsb)      // k2 has no right-hand survivor: its Gl comes from here, its Gr is zero
    for (int e = threadIdx.x; e < B * B; e += kCrLevelThreads) { cr.Gl[(size_t)k2 * BB + e] = sGl2[e]; cr.Gr[(size_t)k2 * BB + e] = 0.0; }
  // right-hand-side sweep
  if (w != nullptr && (int)threadIdx.x < B) {
    for (int v = 0; v < nrhs; ++v) {
      double* wv = w + stride * v;
      double acc_w = wv[(size_t)j * B + threadIdx.x];
      if (has1) { const double* wk = wv + (size_t)k1 * B;
#pragma unroll
        for (int r = 0; r < B; ++r) acc_w -= sGr1[r * B + threadIdx.x] * wk[r]; }
      if (has2) { const double* wk = wv + (size_t)k2 * B;
#pragma unroll
        for (int r = 0; r < B; ++r) acc_w -= sGl2[r * B + threadIdx.x] * wk[r]; }
      wv[(size_t)j * B + threadIdx.x] = acc_w;
    }
  }
  // Schur complement
  for (int e = threadIdx.x; e < B * B; e += kCrLevelThreads) {
    const int r = e / B, c = e % B;
    double d = sD[e], l = 0.0;
    if (has1) {
#pragma unroll
      for (int q = 0; q < B; ++q) { d -= sLj[r * B + q] * sGr1[q * B + c]; l -= sLj[r * B + q] * sGl1[q * B + c]; }
    }
    if (has2) {
#pragma unroll
      for (int q = 0; q < B; ++q) d -= sL2[q * B + r] * sGl2[q * B + c];
    }
    cr.D[off + e] = d;
    if (has1) cr.L[off + e] = l;
    sI1[e] = d;                       // reuse: new D_j for the inversion below (sI1 is dead after the matmuls... see barrier)
  }
  // inversion of the new D_j when j is eliminated at the next level
  if (invert_next && (blockIdx.x & 1)) {
    __syncthreads();
    if (threadIdx.x >= WAVE) return;             // the register-resident inverse lives in the first wave
    const int grp = threadIdx.x / B, c = threadIdx.x % B;
    double col[B];
#pragma unroll
    for (int r = 0; r < B; ++r) col[r] = (grp == 0) ? sI1[r * B + c] : ((grp == 1 && r == c) ? 1.0 : 0.0);
    gauss_jordan_cols<B>(col);
    if (grp == 1) {
#pragma unroll
      for (int r = 0; r < B; ++r) cr.Dinv[off + r * B + c] = col[r];
    }
  }
}

// Dinv of the nodes eliminated at the first level (one wave per node)
template <int B>
__global__ __launch_bounds__(WAVE) void k_cr_invert(CrView cr, int s) {
  const int k = s * (2 * blockIdx.x + 1);
  if (k >= cr.nsb) return;
  const size_t off = (size_t)k * B * B;
  const int grp = threadIdx.x / B, c = threadIdx.x % B;
  double col[B];
#pragma unroll
  for (int r = 0; r < B; ++r) col[r] = (grp == 0) ? cr.D[off + r * B + c] : ((grp == 1 && r == c) ? 1.0 : 0.0);
  gauss_jordan_cols<B>(col);
  if (grp == 1) {
#pragma unroll
    for (int r = 0; r < B; ++r) cr.Dinv[off + r * B + c] = col[r];
  }
}

template <int B>
__device__ __forceinline__ void cr_root_node(const CrView& cr, int lane) {
  const int grp = lane / B, c = lane % B;
  double col[B];
#pragma unroll
  for (int r = 0; r < B; ++r) col[r] = (grp == 0) ? cr.D[r * B + c] : ((grp == 1 && r == c) ? 1.0 : 0.0);
  gauss_jordan_cols<B>(col);
  if (grp == 1) {
#pragma unroll
    for (int r = 0; r < B; ++r) cr.Dinv[r * B + c] = col[r];
  }
}

template <int B>
__global__ __launch_bounds__(WAVE) void k_cr_root(CrView cr) { cr_root_node<B>(cr, threadIdx.x); }

// back substitution of one row: x_k[r] = (Dinv_k w_k - Gl_k x_{k-s} - Gr_k x_{k+s})[r]
template <int B>
__device__ __forceinline__ void cr_backward_row(const CrView& cr, int s, int k, int r, const double* w, double* x) {
  const size_t off = (size_t)k * B * B;
  const double* wk = w + (size_t)k * B;
  const double* xl = x + (size_t)(k - s) * B;
  const bool has_right = (k + s) < cr.nsb;
  const double* xr = x + (size_t)(has_right ? (k + s) : k) * B;
  double v = 0.0;
#pragma unroll
  for (int c = 0; c < B; ++c) {
    v += cr.Dinv[off + r * B + c] * wk[c];
    v -= cr.Gl[off + r * B + c] * xl[c];
    if (has_right) v -= cr.Gr[off + r * B + c] * xr[c];
  }
  x[(size_t)k * B + r] = v;
}

// All levels from stride s_start up to the root, the root itself and (x != nullptr) the
// matching back-substitution levels in ONE single-workgroup launch: these levels hold at most a
// few hundred super-blocks, so they are pure launch latency when issued as 3 kernels per level.
// 16 waves, one super-block per wave per step, __syncthreads between phases (all traffic is
// workgroup-local global memory).
constexpr int kCrTopThreads = 1024;
template <int B>
__global__ __launch_bounds__(kCrTopThreads) void k_cr_top(CrView cr, int s_start, double* w, double* x, int nrhs, size_t stride) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NW = kCrTopThreads / WAVE;
  int s_last = 0;
  for (int s = s_start; s < cr.nsb; s *= 2) {
    for (int q = wave;; q += NW) {
      const int k = s * (2 * q + 1);
      if (k >= cr.nsb) break;
      cr_eliminate_node<B>(cr, s, k, lane);
    }
    __syncthreads();
    for (int q = wave;; q += NW) {
      const int j = 2 * s * q;
      if (j >= cr.nsb) break;
      cr_update_node<B>(cr, s, j, lane, w, nrhs, stride);
    }
    __syncthreads();
    s_last = s;
  }
  if (wave == 0) cr_root_node<B>(cr, lane);
  __syncthreads();
  if (x == nullptr) return;
  if ((int)threadIdx.x < B * nrhs) {
    const int vv = threadIdx.x / B, r = threadIdx.x % B;
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < B; ++c) v += cr.Dinv[r * B + c] * w[stride * vv + c];
    x[stride * vv + r] = v;
  }
  __syncthreads();
  for (int s = s_last; s >= s_start && s >= 1; s /= 2) {
    for (int vv = 0; vv < nrhs; ++vv)
      for (int t = threadIdx.x;; t += kCrTopThreads) {
        const int k = s * (2 * (t / B) + 1);
        if (k >= cr.nsb) break;
        cr_backward_row<B>(cr, s, k, t % B, w + stride * vv, x + stride * vv);
      }
    __syncthreads();
  }
}

// The same top of the tree with every tile resident in LDS: the M = ceil(nsb / s_start) nodes alive at stride s_start
// are copied in once (D, L, right-hand sides), all levels, the root and the back-substitution run on a LOCAL system of
// M nodes with local strides 1, 2, 4 .. through the very same node routines (they only see a CrView whose arrays
// happen to live in LDS), and Dinv / Gl / Gr / w / x go back out once.  Global memory is touched twice instead of at
// every one of the ~20 dependent phases.
// 512 threads: with 1024 the 128-VGPR cap makes the node routines spill (7 VGPRs at B = 9, 55 at B = 12)
constexpr int kCrTopLdsThreads = 512;
template <int B>
__global__ __launch_bounds__(kCrTopLdsThreads) void k_cr_top_lds(CrView cr, int s_start, int M, double* w, double* x, int nrhs, size_t stride) {
  extern __shared__ double lds[];
  constexpr int BB = B * B;
  double* sD = lds;
  double* sL = sD + (size_t)M * BB;
  double* sDinv = sL + (size_t)M * BB;
  double* sGl = sDinv + (size_t)M * BB;
  double* sGr = sGl + (size_t)M * BB;
  double* sw = sGr + (size_t)M * BB;                 // nrhs vectors of M * B, lstride apart
  double* sx = sw + (size_t)M * B * nrhs;
  const size_t lstride = (size_t)M * B;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NW = kCrTopLdsThreads / WAVE;
  for (int e = threadIdx.x; e < M * BB; e += kCrTopLdsThreads) {
    const int m = e / BB, q = e % BB;
    const size_t g = (size_t)m * s_start * BB + q;
    sD[e] = cr.D[g]; sL[e] = cr.L[g]; sGl[e] = 0.0; sGr[e] = 0.0; sDinv[e] = 0.0;
  }
  if (w != nullptr)
    for (int e = threadIdx.x; e < M * B * nrhs; e += kCrTopLdsThreads) {
      const int v = e / (M * B), r = e % (M * B);
      sw[(size_t)v * lstride + r] = w[stride * v + (size_t)(r / B) * s_start * B + (r % B)];
    }
  __syncthreads();
  CrView lc;
  lc.nsb = M; lc.B = B; lc.D = sD; lc.L = sL; lc.Dinv = sDinv; lc.Gl = sGl; lc.Gr = sGr;
  double* lw = (w != nullptr) ? sw : nullptr;
  int s_last = 0;
  for (int s = 1; s < M; s *= 2) {
    for (int q = wave;; q += NW) {
      const int k = s * (2 * q + 1);
      if (k >= M) break;
      cr_eliminate_node<B>(lc, s, k, lane);
    }
    __syncthreads();
    for (int q = wave;; q += NW) {
      const int j = 2 * s * q;
      if (j >= M) break;
      cr_update_node<B>(lc, s, j, lane, lw, nrhs, lstride);
    }
    __syncthreads();
    s_last = s;
  }
  if (wave == 0) cr_root_node<B>(lc, lane);
  __syncthreads();
  if (x != nullptr) {
    if ((int)threadIdx.x < B * nrhs) {
      const int vv = threadIdx.x / B, r = threadIdx.x % B;
      double v = 0.0;
#pragma unroll
      for (int c = 0; c < B; ++c) v += sDinv[r * B + c] * sw[lstride * vv + c];
      sx[lstride * vv + r] = v;
    }
    __syncthreads();
    for (int s = s_last; s >= 1; s /= 2) {
      for (int vv = 0; vv < nrhs; ++vv)
        for (int t = threadIdx.x;; t += kCrTopLdsThreads) {
          const int k = s * (2 * (t / B) + 1);
          if (k >= M) break;
          cr_backward_row<B>(lc, s, k, t % B, sw + lstride * vv, sx + lstride * vv);
        }
      __syncthreads();
    }
  }
  for (int e = threadIdx.x; e < M * BB; e += kCrTopLdsThreads) {
    const int m = e / BB, q = e % BB;
    const size_t g = (size_t)m * s_start * BB + q;
    cr.Dinv[g] = sDinv[e]; cr.Gl[g] = sGl[e]; cr.Gr[g] = sGr[e];
  }
  if (w != nullptr)
    for (int e = threadIdx.x; e < M * B * nrhs; e += kCrTopLdsThreads) {
      const int v = e / (M * B), r = e % (M * B);
      const size_t g = stride * v + (size_t)(r / B) * s_start * B + (r % B);
      w[g] = sw[(size_t)v * lstride + r];
      if (x != nullptr) x[g] = sx[(size_t)v * lstride + r];
    }
}

// forward sweep on right-hand sides (blockIdx.y selects the rhs, `stride` doubles apart):
// w_j -= Gr[k1]^T w_k1 + Gl[k2]^T w_k2
template <int B>
__global__ void k_cr_forward(CrView cr, int s, double* __restrict__ w, size_t stride) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int q = t / B, c = t % B;
  const int j = 2 * s * q;
  if (j >= cr.nsb) return;
  w += stride * blockIdx.y;
  const int k1 = j - s, k2 = j + s;
  double v = w[(size_t)j * B + c];
  if (k1 >= 0) {
    const double* G = cr.Gr + (size_t)k1 * B * B;
    const double* wk = w + (size_t)k1 * B;
#pragma unroll
    for (int r = 0; r < B; ++r) v -= G[r * B + c] * wk[r];
  }
  if (k2 < cr.nsb) {
    const double* G = cr.Gl + (size_t)k2 * B * B;
    const double* wk = w + (size_t)k2 * B;
#pragma unroll
    for (int r = 0; r < B; ++r) v -= G[r * B + c] * wk[r];
  }
  w[(size_t)j * B + c] = v;
}

// back substitution: x_k = Dinv_k w_k - Gl_k x_{k-s} - Gr_k x_{k+s}
template <int B>
__global__ void k_cr_backward(CrView cr, int s, const double* __restrict__ w, double* __restrict__ x, size_t stride) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int q = t / B, r = t % B;
  const int k = s * (2 * q + 1);
  if (k >= cr.nsb) return;
  cr_backward_row<B>(cr, s, k, r, w + stride * blockIdx.y, x + stride * blockIdx.y);
}

// The back-substitution is top-down, so once x is known at every multiple of S the S - 1 nodes between two such
// neighbours form an independent subtree: one workgroup walks it through strides S/2 .. 1 with a barrier per level.
// Replaces log2(S) latency-bound launches by one.
constexpr int kCrTreeThreads = 256;
template <int B>
__global__ __launch_bounds__(kCrTreeThreads) void k_cr_backward_tree(CrView cr, int S, const double* w, double* x, int nrhs, size_t stride) {
  // x of the interval's S + 1 nodes lives in LDS between the levels (one right-hand side at a time), so a level only
  // waits for its own tile loads, not for the previous level's stores to come back from L2
  extern __shared__ double sx[];                       // (S + 1) * B
  const int base = blockIdx.x * S;
  {
    const int vv = blockIdx.y;                          // one right-hand side per workgroup row
    const double* wv = w + stride * vv;
    double* xv = x + stride * vv;
    for (int t = threadIdx.x; t < 2 * B; t += kCrTreeThreads) {            // the two boundary nodes (known)
      const int node = (t < B) ? base : base + S, r = t % B;
      sx[(size_t)(node - base) * B + r] = (node < cr.nsb) ? xv[(size_t)node * B + r] : 0.0;
    }
    __syncthreads();
    for (int s = S / 2; s >= 1; s /= 2) {
      const int per = S / (2 * s);
      for (int t = threadIdx.x; t < per * B; t += kCrTreeThreads) {
        const int k = base + s * (2 * (t / B) + 1), r = t % B;
        if (k < cr.nsb) {
          const size_t off = (size_t)k * B * B;
          const double* wk = wv + (size_t)k * B;
          const double* xl = sx + (size_t)(k - s - base) * B;
          const double* xr = sx + (size_t)(k + s - base) * B;             // zero-filled beyond the last node
          double v = 0.0;
#pragma unroll
          for (int c = 0; c < B; ++c) {
            v += cr.Dinv[off + r * B + c] * wk[c];
            v -= cr.Gl[off + r * B + c] * xl[c];
            if (k + s < cr.nsb) v -= cr.Gr[off + r * B + c] * xr[c];
          }
          sx[(size_t)(k - base) * B + r] = v;
        }
      }
      __syncthreads();
    }
    for (int t = threadIdx.x; t < (S - 1) * B; t += kCrTreeThreads) {       // interior nodes go out once
      const int node = base + 1 + t / B;
      if (node < cr.nsb) xv[(size_t)node * B + t % B] = sx[(size_t)(node - base) * B + t % B];
    }
  }
}

template <int B>
static void cr_backward_levels(hipStream_t st, const CrView& cr, int s_from, const double* w, double* x, int nrhs, size_t stride) {
  // strides s_from, s_from / 2 .. 1: wide levels one launch each, the subtrees below stride kTree in one launch
  constexpr int kTree = 64;
  int s = s_from;
  for (; s >= kTree; s /= 2) {
    const int n_elim = (cr.nsb - s + 2 * s - 1) / (2 * s);
    hipLaunchKernelGGL(k_cr_backward<B>, dim3((n_elim * B + 127) / 128, nrhs), dim3(128), 0, st, cr, s, w, x, stride);
  }
  if (s >= 1) {
    const int S = 2 * s;
    hipLaunchKernelGGL(k_cr_backward_tree<B>, dim3((cr.nsb + S - 1) / S, nrhs), dim3(kCrTreeThreads), sizeof(double) * (S + 1) * B, st, cr, S, w, x,
                       nrhs, stride);
  }
}

template <int B>
__global__ void k_cr_root_solve(CrView cr, const double* __restrict__ w, double* __restrict__ x, size_t stride) {
  const int r = threadIdx.x;
  if (r >= B) return;
  w += stride * blockIdx.x;
  x += stride * blockIdx.x;
  double v = 0.0;
#pragma unroll
  for (int c = 0; c < B; ++c) v += cr.Dinv[r * B + c] * w[c];
  x[r] = v;
}

// first stride handled by the fused single-workgroup kernel: at most kCrTopNodes eliminations at that stride (one
// per wave) and every tile of the nodes alive there resident in LDS
constexpr int kCrTopNodes = 16;
constexpr size_t kCrTopLdsBytes = 144 * 1024;
static size_t cr_top_lds_bytes(int M, int B, int nrhs) { return sizeof(double) * ((size_t)5 * M * B * B + (size_t)2 * M * B * nrhs); }
static int cr_top_start(int nsb, int B, int nrhs) {
  int s = 1;
  while (s < nsb && ((nsb - s + 2 * s - 1) / (2 * s) > kCrTopNodes || cr_top_lds_bytes((nsb + s - 1) / s, B, nrhs) > kCrTopLdsBytes)) s *= 2;
  return s;
}

// ===========================================================================================
// Fused cyclic reduction (round 2): the same elimination tree, Dinv / Gl / Gr per node and arithmetic as the level-per-launch
// kernels above, but a whole run of levels per launch.  The nodes alive at stride S are cut into chunks of M = 2^q
// consecutive ones; a workgroup keeps its chunk's tiles in LDS and eliminates the M - 1 interior nodes through q local
// levels.  The two boundary nodes (local 0 and M) are NOT eliminated in this stage and -- because the Schur complement of
// the interior onto a boundary never reads the boundary's own diagonal block -- need nothing from the neighbouring
// chunks: a chunk only ACCUMULATES what its interior contributes to them (accR to node 0 from the right, accL to node M
// from the left, the new coupling M <-> 0, and the same for the right-hand side).  The next stage (stride S*M) folds the two
// contributions of every node into D and w in a fixed order when it loads them (left chunk's first): bit-reproducible,
// no atomics.  C4 (16 667 super-blocks): stage (S=1, M=16) -> stage (S=16, M=32) -> top (33 nodes, root included):
// 3 launches instead of 11, then 3 back-substitution launches instead of 5.
//
// Work split inside a level (local stride s): one wave per eliminated node u (neighbours a = u - s, b = u + s) does the
// register-resident Gauss-Jordan [D_u | I | L_u | L_b^T] -> Dinv, Gl, Gr, stores them for the back-substitution, and forms
// the three products its neighbours need while Gl / Gr are still in registers:
//     P_a = L_u^T Gl   (into the slot of L_u)     P_b = L_b Gr   (into the slot of D_u)     L_b <- -L_b Gl
// plus Gl^T w_u, Gr^T w_u; after one barrier the survivors subtract what their two neighbours left for them.
// ===========================================================================================
// where stage node sn lives in D / L / w / x: at sn * S -- except the right separator of a submap rank's chain (cr.rsep), which is
// stage node nst - 1 of every stage and stays at its true index nsb - 1  (needs `cr`, `S` and `rsep_sn` in scope)
#define CR_ADDR(sn) ((RS && (sn) == rsep_sn) ? (int64_t)cr.nsb - 1 : (int64_t)(sn) * S)

struct CrStage {
  int S;          // global stride of this stage's nodes
  int M;          // nodes per chunk (power of two); local node u <-> global node (chunk * M + u) * S
  int top;        // 1: single chunk holding every remaining node, local 0 is the root (no right boundary)
  int has_prev;   // contributions of the previous stage are pending in P / Pw (parity prev_par)
  int prev_par, next_par;
  int np;         // slots per contribution array
  int nsl;        // LDS slots (local nodes 0 .. nsl-1 are addressable)
  int nrhs;       // right-hand sides riding through the factorisation (0: factorise only)
  size_t wstride; // doubles between consecutive right-hand sides in w / x
};

// workgroup barrier that orders LDS traffic only: the Dinv / Gl / Gr stores of a level are read by LATER launches, never by
// this one, so a level must not wait for them (__syncthreads() would: s_waitcnt vmcnt(0) costs 1-2 us per level here)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

typedef double v4f64 __attribute__((ext_vector_type(4)));

// Development build (-DHITL_CR_PROBE): thread 0 of one workgroup per stage records the shader clock at the phase boundaries of
// k_cr_chunk; hitl_cr_probe_dump() prints the differences.  Not compiled into the shipped library.
#ifdef HITL_CR_PROBE
__device__ unsigned long long g_seq_probe[4 * 4096];       // k_cr_seq: per workgroup (start, after staging, wave 0 done, hardware id), 100 MHz clock
#define SEQ_PROBE(slot, val) do { if (lane == 0 && wave == 0 && blockIdx.x < 4096) g_seq_probe[4 * blockIdx.x + (slot)] = (val); } while (0)
__device__ unsigned long long g_seq_node_probe[16];       // shader clock at the phase boundaries of ONE node (workgroup 500, wave 0, u = 3)
#define SEQ_T(slot) do { if (probe_on) { __builtin_amdgcn_s_waitcnt(0); if (lane == 0) g_seq_node_probe[slot] = clock64(); } } while (0)
__device__ unsigned long long g_cr_probe[3 * 64];
#define CR_PROBE(slot) do { if (tid == 0 && blockIdx.x == (gridDim.x > 600 ? 600 : gridDim.x / 2)) g_cr_probe[(sg.top ? 2 : (sg.S == 1 ? 0 : 1)) * 64 + (slot)] = clock64(); } while (0)
#define CR_PROBE_IN(slot) do { if (s == 4 && q == wave) { __builtin_amdgcn_s_waitcnt(0); CR_PROBE(slot); } } while (0)
#else
#define CR_PROBE(slot) do { } while (0)
#define CR_PROBE_IN(slot) do { } while (0)
#define SEQ_PROBE(slot, val) do { } while (0)
#define SEQ_T(slot) do { } while (0)
#endif

// KIND 1: the first stage of the LM path's factor-solve (tiles cut from the band rows, one right-hand side, no previous stage,
// not the top) with those facts known at compile time: fewer live values, a register budget that lets more workgroups share a CU.
#ifndef HITL_CR_WAVES
#define HITL_CR_WAVES 8
#endif
// RS: the chain's last node is a separator (CrView::rsep), known at compile time: the plain instantiations keep the code -- and the
// registers -- they had before submap ranks learnt the separator form (a runtime flag cost the first stage 3 % and the 3 x 3
// tiles of the one-pose band 8 %: same-box A/B, tools/dev/lib_ab.sh).
// (the body of a chunk stage as a device function: k_cr_chunk launches it one chunk per workgroup; k_cr_tail runs the last chunk stage,
// the top and both back-substitutions in ONE launch)
template <int B, int KIND, bool RS>
__device__ __forceinline__ void cr_chunk_body(CrView cr, CrStage sg_in, CrSource src, double* w, double* x, double* lds, int chunk_, int tid_, int nthr_) {
  CrStage sg = sg_in;
  if (KIND == 1) { sg.top = 0; sg.has_prev = 0; sg.nrhs = 1; sg.nsl = sg.M + 1; sg.S = 1; }
  if (cr.lm != nullptr) {                                  // device-driven LM loop (uniform branch)
    if (cr.lm->done) return;
    if (src.Hband != nullptr && src.lm != nullptr) {       // the first stage: the CURRENT buffers, today's trust region
      if (src.lm->cur) { src.Hband = src.Hband_alt; src.g = src.g_alt; }
      src.mu = src.lm->radius;
    }
  }
  constexpr int BB = B * B;
  const int M = sg.M, S = sg.S;
  const int nsl = sg.nsl;                                 // local slots: 0 .. M (a chunk) / 0 .. alive-1 (the top)
  double* sD = lds;                                       // nsl tiles: D_u, later P_a of an eliminated u
  double* sL = sD + (size_t)nsl * BB;                     // nsl tiles: L_u (coupling u <-> u - s), later P_b of an eliminated u
  double* sG = sL + (size_t)nsl * BB;                     // per wave: Gl, Gr of the node it is eliminating (2 tiles)
  const int NR = sg.nrhs;                                 // right-hand sides; vector v of local node u lives at [(v * nsl + u) * B]
  double* sW = sG + (size_t)(nthr_ >> 6) * 2 * BB;   // NR x nsl vectors: right-hand sides
  double* sCa = sW + (size_t)NR * nsl * B;                // NR x nsl vectors: Gl_u^T w_u (for a)
  double* sCb = sCa + (size_t)NR * nsl * B;               // NR x nsl vectors: Gr_u^T w_u (for b)
  double* accR = sCb + (size_t)NR * nsl * B;              // tile: contributions to local node 0
  double* accL = accR + BB;                               // tile: contributions to local node M
  double* accWR = accL + BB;                              // NR vectors
  double* accWL = accWR + (size_t)NR * B;                 // NR vectors
  const int tid = tid_, nthr = nthr_;
  const int wave = tid >> 6, lane = tid & 63, NW = nthr >> 6;
  CR_PROBE(0);
  const int64_t base = (int64_t)chunk_ * M;            // in stage nodes
  const int64_t nst = RS ? ((int64_t)cr.nsb - 2) / S + 2 : ((int64_t)cr.nsb + S - 1) / S;   // stage nodes alive: 0 .. nst-1 (cr_stage_nodes)
  // local slots that exist: u with base + u < nst
  const int u_last = (int)((nst - 1 - base < M) ? (nst - 1 - base) : M);     // >= 0
  // The chunk's right boundary (never eliminated here): local node M of a full chunk -- or, on a chain whose last node is a
  // separator (cr.rsep), the last node wherever it falls: it sits in slot u_last of the last chunk, lives at its true index
  // nsb - 1, and is the right neighbour of every eliminated node whose regular one (u + s) would lie beyond it.
  const bool has_rb = (u_last == M) || RS;
  const int rbs = RS ? u_last : M;                         // slot of the right boundary (when there is one)
  const bool last_is_rb = has_rb && (base + rbs == nst - 1);
  const int64_t rsep_sn = RS ? nst - 1 : -1;
  (void)rsep_sn;
  const bool has_w = KIND == 1 || ((w != nullptr) && NR > 0);
  const bool from_src = KIND == 1 || src.Hband != nullptr;   // host: only with S == 1, no previous stage
  const size_t pws = (size_t)4 * sg.np * B;               // doubles of one right-hand side's contribution arrays
  const double* PLp = cr.P + (size_t)(sg.prev_par * 2 + 0) * sg.np * BB;
  const double* PRp = cr.P + (size_t)(sg.prev_par * 2 + 1) * sg.np * BB;
  const double* WLp = cr.Pw + (size_t)(sg.prev_par * 2 + 0) * sg.np * B;
  const double* WRp = cr.Pw + (size_t)(sg.prev_par * 2 + 1) * sg.np * B;
  // ---- first stage straight from the band blocks: stage the band rows of this chunk's poses (and of the super-block on its
  //      left, for L of local 0) with coalesced loads; the tiles are then cut from LDS.  The staging area lies over everything
  //      behind the tiles (sG .. accWL, idle until the tiles are cut; cr_chunk_src_extra extends the allocation when it is longer) ----
  const int Wb = (KIND == 1) ? B / 3 : src.Wb;            // a super-block is Wb poses: a constant of the first-stage instantiation
  const int src_rl = (Wb + 1) * 9 + 4;
  const int src_p0 = from_src ? (int)((base > 0 ? base - 1 : 0) * Wb) : 0;
  double* sRec = nullptr;
  // KIND 1: this thread's right-hand-side entry (-S g) is requested before anything else, so that its memory round trip
  // overlaps the staging of the band rows instead of following the tile cut
  double rhs_v = 0.0;
  const bool rhs_mine = KIND == 1 && tid < (u_last + 1) * B;
  if (rhs_mine) rhs_v = cr_source_rhs(src, (int)(base + tid / B), tid % B);
  if (from_src) {
    const int p1 = min(src.n_poses, (int)(base + u_last + 1) * Wb);
    const int np_st = max(p1 - src_p0, 0);
    sRec = sG;
    const int hl = (Wb + 1) * 9;
    const double* Hg = src.Hband + (size_t)src_p0 * hl;
    if (KIND == 1 && (hl & 1) == 0) {                       // 16-byte loads (hl = 36 doubles per pose, records of 40)
      const double2* Hg2 = reinterpret_cast<const double2*>(Hg);
      const int h2 = hl / 2;
      for (int e = tid; e < np_st * h2; e += nthr) *reinterpret_cast<double2*>(sRec + (size_t)(e / h2) * src_rl + 2 * (e % h2)) = Hg2[e];
    } else
    for (int e = tid; e < np_st * hl; e += nthr) sRec[(size_t)(e / hl) * src_rl + e % hl] = Hg[e];
    for (int e = tid; e < np_st * 4; e += nthr) {
      const int pp = e >> 2, c = e & 3;
      sRec[(size_t)pp * src_rl + hl + c] = (c < 3) ? src.scale[3 * (src_p0 + pp) + c] : (src.fixed[src_p0 + pp] ? 1.0 : 0.0);
    }
    lds_barrier();
    CR_PROBE(1);
  }
  // ---- load: D (folded with the previous stage's contributions), L, w of local 0 .. u_last ----
  for (int e = tid; e < (u_last + 1) * BB; e += nthr) {
    const int u = e / BB, q = e % BB;
    const int64_t sn = base + u;                           // stage node index == slot of the previous stage's arrays
    const size_t g = (size_t)CR_ADDR(sn) * BB + q;
    double d, l;
    if (from_src) {
      // first stage straight from the band blocks (k_build_system's arithmetic): only the nodes that outlive this stage
      // -- the chunk boundaries -- need their D in memory, where the next stage folds the contributions into it
      cr_source_tile_rec(sRec, src_rl, src_p0, src.n_poses, Wb, src.mu, src.min_diag, src.max_diag, B, (int)sn, q, d, l);
      if (u == 0 || (u == rbs && sn == nst - 1)) cr.D[g] = d;
    } else {
      d = cr.D[g];
      l = cr.L[g];
    }
    if (sg.has_prev) d = (d - PLp[(size_t)sn * BB + q]) - PRp[(size_t)sn * BB + q];
    sD[e] = d;
    sL[e] = l;
    // A boundary node stays alive: keep its folded value where the next stage reads it.  Every boundary is local node 0 of
    // one chunk -- except the very last node when it is local node M of the last chunk (no chunk on its right).
    if ((u == 0 || (u == rbs && sn == nst - 1)) && sg.has_prev && !sg.top) cr.D[g] = d;
  }
  CR_PROBE(2);
  if (from_src) lds_barrier();                            // the staging area is overwritten from here on
  if (KIND == 1) {
    if (rhs_mine) {
      const size_t g = (size_t)(base * B) + tid;            // S == 1: node (base + u), row q  <->  base * B + u * B + q
      src.rhs[g] = rhs_v;
      w[g] = rhs_v;
      sW[tid] = rhs_v;
    }
  } else if (has_w)
    for (int e = tid; e < NR * (u_last + 1) * B; e += nthr) {
      const int rv = e / ((u_last + 1) * B), e2 = e % ((u_last + 1) * B);
      const int u = e2 / B, q = e2 % B;
      const int64_t sn = base + u;
      const size_t g = sg.wstride * rv + (size_t)CR_ADDR(sn) * B + q;
      double v;
      if (from_src) {
        // right-hand side 0: -S g, kept for the back-substitution and the caller; the others: unit vectors (submap interface columns)
        if (rv == 0) { v = cr_source_rhs(src, (int)sn, q); src.rhs[(size_t)(sn * S) * B + q] = v; }
        else v = ((int)(sn * S) * B + q == src.unit_cols[rv - 1]) ? 1.0 : 0.0;
        w[g] = v;
      } else v = w[g];
      if (sg.has_prev) v = (v - WLp[pws * rv + (size_t)sn * B + q]) - WRp[pws * rv + (size_t)sn * B + q];
      sW[((size_t)rv * nsl + u) * B + q] = v;
      if (sg.has_prev && (u == 0 || u == rbs) == false) w[g] = v;          // interior: the back-substitution reads the folded w
      if ((u == 0 || (u == rbs && sn == nst - 1)) && sg.has_prev && !sg.top) w[g] = v;
    }
  for (int e = tid; e < 2 * BB + 2 * NR * B; e += nthr) accR[e] = 0.0;   // accR, accL, accWR, accWL are contiguous
  lds_barrier();
  CR_PROBE(3);
  int probe_slot = 4; (void)probe_slot;
  const int u_hi = RS ? ((!sg.top) ? u_last - 1 : u_last)
                      : (sg.top ? u_last : ((u_last < M) ? u_last : M - 1));   // largest local node that may be eliminated
  for (int s = 1; s < M; s *= 2) {
    // ---- eliminate u = s, 3s, 5s .. <= u_hi ----
    for (int q = wave;; q += NW) {
      const int u = s * (2 * q + 1);
      if (u > u_hi) break;
      const int b = (RS && u + s > u_last) ? u_last : u + s;
      const bool has_b = RS ? true : (b <= u_last);
      const int grp = lane / B, c = lane % B;
      double* Du = sD + (size_t)u * BB;
      double* Lu = sL + (size_t)u * BB;
      double* Lb = sL + (size_t)(has_b ? b : u) * BB;
      // column c of [D_u | I | L_u | L_b^T]: one LDS address and stride per lane, every lane loads (idle lanes read D_u) and the
      // choice is a select -- the empty asm keeps the compiler from turning the select back into four divergent load blocks
      double col[B];
      int c_id = (grp == 1) ? c : -1;                       // row of the 1 in this lane's identity column
      asm volatile("" : "+v"(c_id));                        // formed here, not kept as B loop-invariant doubles across the whole kernel
      {
        const double* sp = (grp == 2) ? Lu + c : (grp == 3) ? Lb + c * B : Du + c;
        const int st = (grp == 3) ? 1 : B;
        const bool rd = (grp == 0) || (grp == 2) || (grp == 3 && has_b);
        // all B loads are issued before anything waits for one of them (an opaque point per load put an LDS round trip
        // behind every single read: nine dependent ~130-cycle trips per eliminated node, a quarter of its time)
        double vv[B];
#pragma unroll
        for (int r = 0; r < B; ++r) vv[r] = sp[r * st];
#pragma unroll
        for (int r = 0; r < B; ++r) asm volatile("" : "+v"(vv[r]));
#pragma unroll
        for (int r = 0; r < B; ++r) col[r] = rd ? vv[r] : ((r == c_id) ? 1.0 : 0.0);
      }
      CR_PROBE_IN(40);
      gauss_jordan_cols<B>(col);
      CR_PROBE_IN(41);
      const size_t gn = (size_t)((base + u) * S) * BB;
      if (grp >= 1 && grp <= 3) {
        double* dst = (grp == 1) ? cr.Dinv + gn : (grp == 2) ? cr.Gl + gn : cr.Gr + gn;
#pragma unroll
        for (int r = 0; r < B; ++r) dst[r * B + c] = col[r];
      }
      // The three products its neighbours need, on three lane groups at once: Gl / Gr go to this wave's scratch tiles, then
      // lane (g, c) forms column c of   g = 0: P_a = L_u^T Gl -> slot of D_u (dead since the load above)
      //                                  g = 1: P_b = L_b Gr   -> slot of L_u (dead once the products are formed)
      //                                  g = 2: -L_b Gl        -> L_b in place (new coupling b <-> a)
      // The left operands are wave-uniform LDS reads per group.  In-place safety: every q-step reads column / row data of
      // L_u and L_b, all writes happen after the last q-step (program order within the wave).
      if (has_w && (grp == 2 || grp == 3)) {
        for (int rv = 0; rv < NR; ++rv) {
          const double* wu = sW + ((size_t)rv * nsl + u) * B;
          double cw = 0.0;
#pragma unroll
          for (int r = 0; r < B; ++r) cw += col[r] * wu[r];                        // (G^T w_u)[c]
          if (grp == 2) sCa[((size_t)rv * nsl + u) * B + c] = cw; else sCb[((size_t)rv * nsl + u) * B + c] = cw;
        }
      }
      CR_PROBE_IN(42);
      double* Gs = sG + (size_t)wave * 2 * BB;
      if (grp == 2 || grp == 3) {
        double* dstg = Gs + (grp == 3 ? BB : 0);
#pragma unroll
        for (int r = 0; r < B; ++r) dstg[r * B + c] = col[r];
      }
      // (same wave: LDS operations complete in order, no barrier needed)
      // The three B x B products on the matrix pipe: v_mfma_f64_16x16x4_f64 holds a whole (zero-padded) 16 x 16 result per wave,
      // lane l feeds A[l & 15][4 step + (l >> 4)] and B[4 step + (l >> 4)][l & 15] and receives rows (l >> 4) + 4 reg of column
      // l & 15: ceil(B / 4) steps per product instead of B^2 broadcast reads and FMAs on a third of the lanes.
      CR_PROBE_IN(43);
      {
        const int mr = lane & 15, mk = lane >> 4;
        const int mrc = (mr < B) ? mr : B - 1;                             // lanes outside the tile read a valid address and feed zeros
        v4f64 c1 = {0.0, 0.0, 0.0, 0.0}, c2 = c1, c3 = c1;
#pragma unroll
        for (int ks = 0; ks < (B + 3) / 4; ++ks) {
          const int k = 4 * ks + mk;
          const int kc = (k < B) ? k : B - 1;
          const bool in = (k < B) && (mr < B);
          double a_lut = Lu[kc * B + mrc];                                 // L_u^T[mr][k]
          double a_lb = Lb[mrc * B + kc];                                  // L_b[mr][k]
          double b_gl = Gs[kc * B + mrc];                                  // Gl[k][mr]
          double b_gr = Gs[BB + kc * B + mrc];                             // Gr[k][mr]
          asm volatile("" : "+v"(a_lut), "+v"(a_lb), "+v"(b_gl), "+v"(b_gr));      // unconditional loads, then selects (see the column load)
          a_lut = in ? a_lut : 0.0;
          a_lb = (in && has_b) ? a_lb : 0.0;
          b_gl = in ? b_gl : 0.0;
          b_gr = in ? b_gr : 0.0;
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_lut, b_gl, c1, 0, 0, 0);    // P_a = L_u^T Gl
          c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_lb, b_gr, c2, 0, 0, 0);     // P_b = L_b Gr
          c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_lb, b_gl, c3, 0, 0, 0);     // L_b Gl
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int row = mk + 4 * rg;
          if (row < B && mr < B) {
            Du[row * B + mr] = c1[rg];                                     // slot of D_u (dead since the load above)
            Lu[row * B + mr] = c2[rg];                                     // slot of L_u (every read of it is behind us: program order within the wave)
            if (has_b) Lb[row * B + mr] = -c3[rg];                         // new coupling b <-> a, in place
          }
        }
      }
      CR_PROBE_IN(44);
    }
    CR_PROBE(probe_slot++);
    lds_barrier();
    CR_PROBE(probe_slot++);
    // ---- survivors j = 0, 2s, 4s .. subtract what their eliminated neighbours left (left one first) ----
    // (separator chain: the last entry of the enumeration is the chunk's right boundary, slot u_last; the node that left something
    // for it at this level -- if any -- is the last multiple of s below it when that is an ODD multiple: M - s in a full chunk)
    const int nsv = M / (2 * s) + 1;
    const int rb_t = RS ? (u_last - 1) / s : 0;
    const int rb_u = rb_t * s;                                                     // >= 1 whenever rb_t is odd
    for (int e = tid; e < nsv * BB; e += nthr) {
      if (RS) {
        const int idx = e / BB, q = e % BB;
        const bool isb = idx == nsv - 1;
        const int j = isb ? u_last : 2 * s * idx;
        if (!isb && j >= u_last) continue;
        const bool hl_ = isb ? ((rb_t & 1) != 0) : (j - s >= 1);
        const bool hr_ = !isb && (j + s <= u_hi);
        const double pl = hl_ ? sL[(size_t)(isb ? rb_u : j - s) * BB + q] : 0.0;   // P_b of the left neighbour
        const double pr = hr_ ? sD[(size_t)(j + s) * BB + q] : 0.0;                // P_a of the right neighbour
        if (isb) accL[q] += pl;
        else if (j == 0) accR[q] += pr;
        else sD[(size_t)j * BB + q] = (sD[(size_t)j * BB + q] - pl) - pr;
      } else {
        const int j = 2 * s * (e / BB), q = e % BB;
        if (j > u_last) continue;
        const bool hl_ = (j - s >= 1), hr_ = (j + s <= u_hi) && (j + s < M || sg.top);
        const double pl = hl_ ? sL[(size_t)(j - s) * BB + q] : 0.0;                // P_b of the left neighbour
        const double pr = hr_ ? sD[(size_t)(j + s) * BB + q] : 0.0;                // P_a of the right neighbour
        if (j == 0) accR[q] += pr;
        else if (j == M) accL[q] += pl;
        else sD[(size_t)j * BB + q] = (sD[(size_t)j * BB + q] - pl) - pr;
      }
    }
    if (has_w)
      for (int e = tid; e < NR * nsv * B; e += nthr) {
        const int rv = e / (nsv * B), e2 = e % (nsv * B);
        const size_t vb = (size_t)rv * nsl;
        if (RS) {
          const int idx = e2 / B, q = e2 % B;
          const bool isb = idx == nsv - 1;
          const int j = isb ? u_last : 2 * s * idx;
          if (!isb && j >= u_last) continue;
          const bool hl_ = isb ? ((rb_t & 1) != 0) : (j - s >= 1);
          const bool hr_ = !isb && (j + s <= u_hi);
          const double pl = hl_ ? sCb[(vb + (isb ? rb_u : j - s)) * B + q] : 0.0;
          const double pr = hr_ ? sCa[(vb + (j + s)) * B + q] : 0.0;
          if (isb) accWL[(size_t)rv * B + q] += pl;
          else if (j == 0) accWR[(size_t)rv * B + q] += pr;
          else sW[(vb + j) * B + q] = (sW[(vb + j) * B + q] - pl) - pr;
        } else {
          const int j = 2 * s * (e2 / B), q = e2 % B;
          if (j > u_last) continue;
          const bool hl_ = (j - s >= 1), hr_ = (j + s <= u_hi) && (j + s < M || sg.top);
          const double pl = hl_ ? sCb[(vb + (j - s)) * B + q] : 0.0;
          const double pr = hr_ ? sCa[(vb + (j + s)) * B + q] : 0.0;
          if (j == 0) accWR[(size_t)rv * B + q] += pr;
          else if (j == M) accWL[(size_t)rv * B + q] += pl;
          else sW[(vb + j) * B + q] = (sW[(vb + j) * B + q] - pl) - pr;
        }
      }
    lds_barrier();
    CR_PROBE(probe_slot++);
    // the right-hand side of a node is final once it is eliminated: the back-substitution reads it from global memory
    if (has_w)
      for (int e = tid; e < NR * (M / (2 * s)) * B; e += nthr) {
        const int rv = e / ((M / (2 * s)) * B), e2 = e % ((M / (2 * s)) * B);
        const int u2 = 2 * s * (2 * (e2 / B) + 1), q = e2 % B;                     // nodes eliminated at the NEXT level
        if (u2 <= u_hi && 2 * s < M) w[sg.wstride * rv + (size_t)((base + u2) * S) * B + q] = sW[((size_t)rv * nsl + u2) * B + q];
      }
  }
  if (!sg.top) {
    // ---- hand the boundary contributions to the next stage (its node index = this chunk index (+1)) ----
    double* PLn = cr.P + (size_t)(sg.next_par * 2 + 0) * sg.np * BB;
    double* PRn = cr.P + (size_t)(sg.next_par * 2 + 1) * sg.np * BB;
    double* WLn = cr.Pw + (size_t)(sg.next_par * 2 + 0) * sg.np * B;
    double* WRn = cr.Pw + (size_t)(sg.next_par * 2 + 1) * sg.np * B;
    const size_t c0 = (size_t)chunk_, c1 = c0 + 1;
    for (int e = tid; e < BB; e += nthr) {
      PRn[c0 * BB + e] = accR[e];
      if (chunk_ == 0) PLn[e] = 0.0;                                           // node 0 has no chunk on its left
      if (has_rb) { PLn[c1 * BB + e] = accL[e]; cr.L[(size_t)CR_ADDR(base + rbs) * BB + e] = sL[(size_t)rbs * BB + e]; }
      if (last_is_rb) PRn[c1 * BB + e] = 0.0;                                      // the last node has no chunk on its right
    }
    if (has_w)
      for (int e2 = tid; e2 < NR * B; e2 += nthr) {
        const int rv = e2 / B, e = e2 % B;
        const size_t vo = pws * rv;
        WRn[vo + c0 * B + e] = accWR[e2];
        if (chunk_ == 0) WLn[vo + e] = 0.0;
        if (has_rb) WLn[vo + c1 * B + e] = accWL[e2];
        if (last_is_rb) WRn[vo + c1 * B + e] = 0.0;
      }
    CR_PROBE(probe_slot++);
    return;
  }
  // ---- top: the root ----
  if (wave == 0) {
    const int grp = lane / B, c = lane % B;
    double col[B];
#pragma unroll
    for (int r = 0; r < B; ++r) col[r] = (grp == 0) ? (sD[r * B + c] - accR[r * B + c]) : ((grp == 1 && r == c) ? 1.0 : 0.0);
    gauss_jordan_cols<B>(col);
    if (grp == 1) {
#pragma unroll
      for (int r = 0; r < B; ++r) { cr.Dinv[r * B + c] = col[r]; sL[r * B + c] = col[r]; }   // slot of L_0 (unused: node 0 has no left)
    }
  }
  lds_barrier();
  if (has_w)
    for (int e = tid; e < NR * B; e += nthr) {
      const int rv = e / B, q = e % B;
      const double wr = sW[(size_t)rv * nsl * B + q] - accWR[e];
      w[sg.wstride * rv + q] = wr;                                                // folded root right-hand side (cr_back reads Dinv_0 w_0)
      sW[(size_t)rv * nsl * B + q] = wr;
    }
  lds_barrier();
  if (has_w && x != nullptr)
    for (int e = tid; e < NR * B; e += nthr) {
      const int rv = e / B, q = e % B;
      double v = 0.0;
#pragma unroll
      for (int c = 0; c < B; ++c) v += sL[q * B + c] * sW[(size_t)rv * nsl * B + c];
      x[sg.wstride * rv + q] = v;
    }
  CR_PROBE(probe_slot++);
}


template <int B, int KIND, bool RS = false>
__global__ __launch_bounds__((B >= 12 || KIND == 1) ? 512 : 1024) __attribute__((amdgpu_waves_per_eu((KIND == 1 && B == 9) ? HITL_CR_WAVES : 4)))
void k_cr_chunk(CrView cr, CrStage sg_in, CrSource src, double* w, double* x) {
  extern __shared__ double lds[];
  cr_chunk_body<B, KIND, RS>(cr, sg_in, src, w, x, lds, (int)blockIdx.x, (int)threadIdx.x, (int)blockDim.x);
}

// Back-substitution of one stage: x is known at the chunk's two boundary nodes; thread (u, r) owns row r of interior
// node u, fetches its three coefficient rows ONCE (all loads of the launch are in flight together), forms
// v = (Dinv_u w_u)[r] and then only touches LDS: x_u[r] = v - Gl_u[r,:] x_{u-s} - Gr_u[r,:] x_{u+s}, level by level.
// nrhs > 1 (submap ranks: the interface columns): one workgroup per (chunk, right-hand side), numbered so that the nrhs
// workgroups of a chunk land on the SAME XCD one after the other (workgroups go to the 8 XCDs round-robin by index): the
// chunk's 3 x 81 x (M - 1) coefficient doubles come from HBM once and from that XCD's L2 for the other right-hand sides,
// instead of nrhs times from HBM (a (chunks, nrhs) grid puts the workgroups of a chunk ~1000 dispatches apart).
template <int B, bool RS>
__device__ __forceinline__ void cr_chunk_back_body(CrView cr, int S, int M, int top, const double* __restrict__ w, double* __restrict__ x, size_t wstride,
                                                   int chunks, int nrhs, double* sx, int block_, int tid_, int nthr_) {
  const int tid = tid_;
  int chunk = block_, rv = 0;
  if (nrhs > 1) {
    const int xcd = block_ & 7, t = block_ >> 3;
    rv = t % nrhs;
    chunk = (t / nrhs) * 8 + xcd;
    if (chunk >= chunks) return;
  }
  w += wstride * rv;
  x += wstride * rv;
  const int64_t base = (int64_t)chunk * M;
  const int64_t nst = RS ? ((int64_t)cr.nsb - 2) / S + 2 : ((int64_t)cr.nsb + S - 1) / S;
  const int u_last = (int)((nst - 1 - base < M) ? (nst - 1 - base) : M);
  const bool has_rb = !top && ((u_last == M) || RS);     // see k_cr_chunk: on a separator chain the right boundary sits in slot u_last
  const int64_t rsep_sn = RS ? nst - 1 : -1;
  (void)rsep_sn;
  const int u = 1 + tid / B, r = tid % B;
  const bool mine = RS ? (u <= (has_rb ? u_last - 1 : u_last) && (tid < (M - (top ? 0 : 1)) * B) && u <= M)
                       : ((u < M || top) && u <= u_last && (tid < (M - (top ? 0 : 1)) * B) && u <= M);
  double gl[B], gr[B], v = 0.0;
#pragma unroll
  for (int c = 0; c < B; ++c) { gl[c] = 0.0; gr[c] = 0.0; }
  if (mine) {
    const size_t gn = (size_t)((base + u) * S);
    const double* Di = cr.Dinv + gn * B * B + (size_t)r * B;
    const double* Gl = cr.Gl + gn * B * B + (size_t)r * B;
    const double* Gr = cr.Gr + gn * B * B + (size_t)r * B;
    const double* wk = w + gn * B;
#pragma unroll
    for (int c = 0; c < B; ++c) { v += Di[c] * wk[c]; gl[c] = Gl[c]; gr[c] = Gr[c]; }
  }
  for (int e = tid; e < (M + 1) * B; e += nthr_) sx[e] = 0.0;
  __syncthreads();
  if (tid < 2 * B) {                                      // the two boundary nodes (known)
    const int ub = (tid < B) ? 0 : (RS ? u_last : M), rr = tid % B;
    if (RS ? (tid < B || has_rb) : (ub <= u_last && !(top && ub == M))) sx[(size_t)ub * B + rr] = x[(size_t)CR_ADDR(base + ub) * B + rr];
  }
  __syncthreads();
  for (int s = M / 2; s >= 1; s /= 2) {
    if (mine && (u & (2 * s - 1)) == s) {                 // u is eliminated at local stride s
      const int b = (RS && has_rb && u + s > u_last) ? u_last : u + s;
      const double* xl = sx + (size_t)(u - s) * B;
      const double* xr = sx + (size_t)((b <= M) ? b : u) * B;
      double acc = v;
#pragma unroll
      for (int c = 0; c < B; ++c) acc -= gl[c] * xl[c];
      if (b <= M) {
#pragma unroll
        for (int c = 0; c < B; ++c) acc -= gr[c] * xr[c];   // zero-filled beyond the last node; Gr is zero there too
      }
      sx[(size_t)u * B + r] = acc;
      x[(size_t)((base + u) * S) * B + r] = acc;
    }
    __syncthreads();
  }
}


template <int B, bool RS = false>
__global__ __launch_bounds__(1024) void k_cr_chunk_back(CrView cr, int S, int M, int top, const double* __restrict__ w, double* __restrict__ x, size_t wstride,
                                                        int chunks, int nrhs) {
  if (cr.lm != nullptr && cr.lm->done) return;
  extern __shared__ double sx[];                         // (M + 1) * B
  cr_chunk_back_body<B, RS>(cr, S, M, top, w, x, wstride, chunks, nrhs, sx, (int)blockIdx.x, (int)threadIdx.x, (int)blockDim.x);
}

// ===========================================================================================
// k_cr_tail (round 5): the LAST chunk stage, the top, the top's back-substitution and that stage's own back-substitution in ONE
// launch.  The upper stages are chains of dependent latency on a handful of workgroups (C4: 33 chunks, then one workgroup, then the
// same again backwards): as four launches they pay four launch boundaries (~4 us each in the kernel trace, whatever the work) for
// ~60 us of work.  Here workgroups 0 .. C-1 run the chunk stage and announce themselves; workgroup C waits for all of them, runs the
// top (forward, root, back-substitution: x at every node the chunk stage kept) and releases them; they then back-substitute their
// interiors from the tiles they have just written.  All C + 1 workgroups are resident at once (one per compute unit; the host only
// takes this path for C + 1 <= kCrTailMaxGroups), so the waits cannot deadlock; they are bounded anyway (a wait that runs out flags
// the launch as failed -- ticket[2] -- instead of hanging the device).
// Ordering: every thread's stores have reached the L2 (s_waitcnt) before the workgroup barrier; then ONE thread makes them visible
// device-wide (release fence: one L2 write-back per workgroup, not per wave) and counts the workgroup in with a device-scope atomic.
// The waiting side polls with device-scope loads, then ONE thread invalidates (acquire fence) before the workgroup proceeds.
// Tickets: [0] arrivals, [1] released, [2] a wait ran out, [3] departures; the last chunk workgroup to leave clears them for the next
// launch (launches of a stream do not overlap), so the host keeps no state and a launch skipped by lm->done leaves them untouched.
// KIND2: 0 = the chunk stage folds a previous stage's contributions (C4: the stage above the first one); 1 = it IS the first stage and
// cuts its tiles from the band rows (short chains -- C2, Figure-8-size maps: the whole linear solve is this one launch).
// ===========================================================================================
constexpr int kCrTailMaxGroups = 128;
__device__ __forceinline__ bool cr_tail_wait(const unsigned* word, unsigned target, unsigned max_spins) {
  for (unsigned spins = 0; spins < max_spins; ++spins) {
    if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    __builtin_amdgcn_s_sleep(4);
  }
  return false;
}

__device__ __forceinline__ void cr_tail_report(const CrView& cr, unsigned* ticket) {
  __hip_atomic_store(&ticket[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (cr.fail_host != nullptr) __hip_atomic_store(cr.fail_host, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int B, int KIND2>
__global__ __launch_bounds__(1024) void k_cr_tail(CrView cr, CrStage sg2, CrStage sgt, CrSource src, double* w, double* x, int chunks2, unsigned* ticket) {
  if (cr.lm != nullptr && cr.lm->done) return;
  extern __shared__ double lds[];
  __shared__ int s_ok;
  const int tid = threadIdx.x, nthr = blockDim.x;
  if ((int)blockIdx.x < chunks2) {
    cr_chunk_body<B, KIND2, false>(cr, sg2, src, w, x, lds, (int)blockIdx.x, tid, nthr);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(&ticket[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_ok = cr_tail_wait(&ticket[1], 1u, cr.tail_spins) ? 1 : 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (s_ok) cr_chunk_back_body<B, false>(cr, sg2.S, sg2.M, 0, w, x, (size_t)0, chunks2, 1, lds, (int)blockIdx.x, tid, nthr);
    if (tid == 0) {
      if (!s_ok) cr_tail_report(cr, ticket);
      if (__hip_atomic_fetch_add(&ticket[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)chunks2 - 1u) {     // the last one out
        __hip_atomic_store(&ticket[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ticket[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ticket[3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else {
    if (tid == 0) {
      s_ok = cr_tail_wait(&ticket[0], (unsigned)chunks2, cr.tail_spins) ? 1 : 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (s_ok) {
      CrSource none = {};
      cr_chunk_body<B, 0, false>(cr, sgt, none, w, x, lds, 0, tid, nthr);
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      cr_chunk_back_body<B, false>(cr, sgt.S, sgt.M, 1, w, x, (size_t)0, 1, 1, lds, 0, tid, nthr);
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
    }
    if (tid == 0) {
      if (!s_ok) cr_tail_report(cr, ticket);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(&ticket[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         // (also after a failed wait: nobody is left waiting)
    }
  }
}

// Forward sweep of one stage on STORED factors (a new right-hand side against an existing factorisation: the PCG
// preconditioner): the chunk's Gl / Gr tiles and right-hand side come into LDS in one round of loads, then every local
// level is   c_a = Gl_u^T w_u -> node a,   c_b = Gr_u^T w_u -> node b   for the eliminated u, a barrier, and the survivors'
// subtraction (left neighbour first) -- the same order of operations as the factorising kernel, so a solve through
// k_cr_chunk and one through this kernel agree bit for bit.  Boundary contributions travel through Pw like there.
template <int B>
__global__ __launch_bounds__(512) void k_cr_chunk_fwd(CrView cr, CrStage sg, const double* w_in, double* w, double* x) {
  extern __shared__ double lds[];
  constexpr int BB = B * B;
  const int M = sg.M, S = sg.S, nsl = sg.nsl;
  double* sGl = lds;                                      // nsl tiles
  double* sGr = sGl + (size_t)nsl * BB;                   // nsl tiles
  double* sW = sGr + (size_t)nsl * BB;                    // nsl vectors
  double* sCa = sW + (size_t)nsl * B;
  double* sCb = sCa + (size_t)nsl * B;
  double* accWR = sCb + (size_t)nsl * B;
  double* accWL = accWR + B;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int64_t base = (int64_t)blockIdx.x * M;
  const int64_t nst = ((int64_t)cr.nsb + S - 1) / S;
  const int u_last = (int)((nst - 1 - base < M) ? (nst - 1 - base) : M);
  const int u_hi = sg.top ? u_last : ((u_last < M) ? u_last : M - 1);
  const double* WLp = cr.Pw + (size_t)(sg.prev_par * 2 + 0) * sg.np * B;
  const double* WRp = cr.Pw + (size_t)(sg.prev_par * 2 + 1) * sg.np * B;
  for (int e = tid; e < u_hi * BB; e += nthr) {            // interior nodes 1 .. u_hi
    const int u = 1 + e / BB, q = e % BB;
    const size_t g = (size_t)((base + u) * S) * BB + q;
    sGl[(size_t)u * BB + q] = cr.Gl[g];
    sGr[(size_t)u * BB + q] = cr.Gr[g];
  }
  for (int e = tid; e < (u_last + 1) * B; e += nthr) {
    const int u = e / B, q = e % B;
    const int64_t sn = base + u;
    const size_t g = (size_t)(sn * S) * B + q;
    double v = sg.has_prev ? w[g] : w_in[g];     // the first stage may read the right-hand side from another vector (no copy in front)
    if (sg.has_prev) v = (v - WLp[(size_t)sn * B + q]) - WRp[(size_t)sn * B + q];
    else if (w_in != w) w[g] = v;                // ... and then leaves every node's value where the backward stages read it
    sW[e] = v;
    if (sg.has_prev && (u == 0 || u == M) == false) w[g] = v;
    if ((u == 0 || (u == M && sn == nst - 1)) && sg.has_prev && !sg.top) w[g] = v;     // see k_cr_chunk: the last node may be nobody's local 0
  }
  for (int e = tid; e < 2 * B; e += nthr) accWR[e] = 0.0;
  lds_barrier();
  for (int s = 1; s < M; s *= 2) {
    const int nel = M / (2 * s);                           // eliminated nodes u = s (2 k + 1)
    for (int e = tid; e < nel * 2 * B; e += nthr) {
      const int u = s * (2 * (e / (2 * B)) + 1), c = e % B, which = (e / B) & 1;
      if (u > u_hi) continue;
      const double* G = (which ? sGr : sGl) + (size_t)u * BB;
      const double* wu = sW + (size_t)u * B;
      double cw = 0.0;
#pragma unroll
      for (int r = 0; r < B; ++r) cw += G[r * B + c] * wu[r];
      (which ? sCb : sCa)[(size_t)u * B + c] = cw;
    }
    lds_barrier();
    const int nsv = M / (2 * s) + 1;
    for (int e = tid; e < nsv * B; e += nthr) {
      const int j = 2 * s * (e / B), q = e % B;
      if (j > u_last) continue;
      const bool hl_ = (j - s >= 1), hr_ = (j + s <= u_hi) && (j + s < M || sg.top);
      const double pl = hl_ ? sCb[(size_t)(j - s) * B + q] : 0.0;
      const double pr = hr_ ? sCa[(size_t)(j + s) * B + q] : 0.0;
      if (j == 0) accWR[q] += pr;
      else if (j == M) accWL[q] += pl;
      else sW[(size_t)j * B + q] = (sW[(size_t)j * B + q] - pl) - pr;
    }
    lds_barrier();
    for (int e = tid; e < (M / (2 * s)) * B; e += nthr) {
      const int u2 = 2 * s * (2 * (e / B) + 1), q = e % B;
      if (u2 <= u_hi && 2 * s < M) w[(size_t)((base + u2) * S) * B + q] = sW[(size_t)u2 * B + q];
    }
  }
  if (!sg.top) {
    double* WLn = cr.Pw + (size_t)(sg.next_par * 2 + 0) * sg.np * B;
    double* WRn = cr.Pw + (size_t)(sg.next_par * 2 + 1) * sg.np * B;
    const size_t c0 = (size_t)blockIdx.x, c1 = c0 + 1;
    for (int e = tid; e < B; e += nthr) {
      WRn[c0 * B + e] = accWR[e];
      if (blockIdx.x == 0) WLn[e] = 0.0;
      if (u_last == M) WLn[c1 * B + e] = accWL[e];
      if (u_last == M && base + M == nst - 1) WRn[c1 * B + e] = 0.0;
    }
    return;
  }
  for (int e = tid; e < B; e += nthr) { const double wr = sW[e] - accWR[e]; w[e] = wr; sW[e] = wr; }
  lds_barrier();
  if (x != nullptr)
    for (int e = tid; e < B; e += nthr) {
      double v = 0.0;
#pragma unroll
      for (int c = 0; c < B; ++c) v += cr.Dinv[e * B + c] * sW[c];
      x[e] = v;
    }
}

// ===========================================================================================
// First stage of the LM path's factor-solve in TWO-WAY SEQUENTIAL order (round 5).
//
// The cyclic-reduction stage above eliminates a chunk's M - 1 interior nodes level by level: log2 M levels, a workgroup
// barrier and a survivor pass per level, and from the second level on most of the workgroup's waves wait.  At the FIRST stage
// (16 667 super-blocks on 1024 SIMDs) that is not a latency problem but a throughput one: 49 us for 15 600 node eliminations.
// Here the same eliminations -- the same [D | I | L_u | L_b^T] Gauss-Jordan, the same three products on the matrix pipe, the same
// Dinv / Gl / Gr per node, the same contributions handed to the next stage -- run in another ORDER: the chunk's interior nodes
// 1 .. U are a chain; wave 0 eliminates it from the left end (node u with neighbours a = the chunk's left boundary and
// b = u + 1: the fill F_u = coupling u <-> left boundary travels along), wave 1 from the right end (neighbours a = u - 1 and
// b = the right boundary, fill E_u), and wave 0 finishes with the node in the middle, whose two neighbours are the boundaries.
// No barrier but the one before the middle node, no idle wave, products accumulate in the matrix pipe's registers, and
// a workgroup is two waves: every chunk of the map is resident at once.  Tiles are cut from the staged band rows one node
// ahead of their use instead of all up front (LDS: the rows + seven tiles per wave).
// Back-substitution (k_cr_seq_back): x_mid from the two boundaries, then the two half-chains outwards, one step per node.
// The elimination tree differs from the balanced one, so Dinv / Gl / Gr of a node mean "towards ITS neighbours a and b" of this
// order: only k_cr_seq_back reads them (selected inversion and the PCG preconditioner factorise through the stages above).
// ===========================================================================================
// (every LDS pointer of this kernel carries its address space in its type: a pointer picked at run time -- tD[cur] -- would otherwise
// decay to a generic one, whose loads are FLAT instructions that wait for the node's global stores as well)
typedef __attribute__((address_space(3))) double ldsd;

template <int B>
struct SeqNodeOut { v4f64 c1, c2, c3; double cw; };

// Tiles are cut from the staged band rows through per-lane DESCRIPTORS formed once per kernel: which words of the records of a
// node's own poses (and of the super-block on its left) an entry needs do not depend on the node, only their common base does.  A
// cut is then a handful of independent LDS reads and multiplications per entry -- k_build_system's arithmetic in its order (value
// * (scale_a * scale_b) for D, (value * scale_a) * scale_b for L, the LM diagonal on top), no index arithmetic, no branches.
// Records of poses beyond the end of the map are staged as constant poses, so there is no bounds test either.
// D entries are cut in the matrix pipe's result layout (lane l: rows (l >> 4) + 4 rg of column l & 15): the product that is
// subtracted from them sits in exactly those registers.  L entries in linear order (entry e = lane + 64 t).
template <int B>
struct SeqCutDesc {
  static constexpr int NT = (B * B + WAVE - 1) / WAVE;
  int dh[3], dsa[3], dsb[3];      // D, rg = 0..2: value word, scale words of the row / column pose (flag word = scale word - comp + 3 -> dfa, dfb)
  int dfa[3], dfb[3];
  int lh[NT], lsa[NT], ls3[NT], lfa[NT], lf3[NT];
  unsigned dmask;                 // bit rg: entry exists; bit 4 + rg: it is a diagonal entry
  unsigned lmask;                 // bit t: entry exists; bit 8 + t: inside the band (d <= Wb)
};

template <int B>
__device__ __forceinline__ void seq_cut_desc(SeqCutDesc<B>& q, int lane) {
  constexpr int Wb = B / 3, rl = (Wb + 1) * 9 + 4, so = (Wb + 1) * 9, BB = B * B;
  const int mr = lane & 15, mk = lane >> 4;
  q.dmask = 0; q.lmask = 0;
#pragma unroll
  for (int rg = 0; rg < 3; ++rg) {
    const int a = mk + 4 * rg, b = mr;
    const bool ok = a < B && b < B;
    const int ac = ok ? a : 0, bc = ok ? b : 0;
    const int pa = ac / 3, ca = ac % 3, pb = bc / 3, cb = bc % 3;
    q.dh[rg] = (pb >= pa) ? pa * rl + (pb - pa) * 9 + ca * 3 + cb : pb * rl + (pa - pb) * 9 + cb * 3 + ca;
    q.dsa[rg] = pa * rl + so + ca; q.dsb[rg] = pb * rl + so + cb;
    q.dfa[rg] = pa * rl + so + 3; q.dfb[rg] = pb * rl + so + 3;
    if (ok) q.dmask |= 1u << rg;
    if (ok && a == b) q.dmask |= 16u << rg;
  }
#pragma unroll
  for (int t = 0; t < SeqCutDesc<B>::NT; ++t) {
    const int e = lane + WAVE * t;
    const bool ok = e < BB;
    const int ec = ok ? e : 0;
    const int a = ec / B, b = ec % B, pa = a / 3, ca = a % 3, pb = b / 3, cb = b % 3;
    const int d = Wb + pa - pb;                                   // pose distance row pose (this node) - column pose (the node on the left)
    const bool band = d <= Wb;
    q.lh[t] = (pb - Wb) * rl + (band ? d : 0) * 9 + cb * 3 + ca;
    q.lsa[t] = pa * rl + so + ca; q.lfa[t] = pa * rl + so + 3;
    q.ls3[t] = (pb - Wb) * rl + so + cb; q.lf3[t] = (pb - Wb) * rl + so + 3;
    if (ok) q.lmask |= 1u << t;
    if (ok && band) q.lmask |= 256u << t;
  }
}

// D of the node whose first pose record sits at nb, in the result layout: d[rg] (0 where the lane has no entry)
template <int B>
__device__ __forceinline__ void seq_cut_d(const SeqCutDesc<B>& q, const ldsd* nb, double inv_mu, double min_diag, double max_diag, double (&d)[3]) {
  double h[3], sa[3], sb[3], fa[3], fb[3];
#pragma unroll
  for (int rg = 0; rg < 3; ++rg) { h[rg] = nb[q.dh[rg]]; sa[rg] = nb[q.dsa[rg]]; sb[rg] = nb[q.dsb[rg]]; fa[rg] = nb[q.dfa[rg]]; fb[rg] = nb[q.dfb[rg]]; }
#pragma unroll
  for (int rg = 0; rg < 3; ++rg) {
    const bool diag = (q.dmask >> (4 + rg)) & 1u;
    double v = (fb[rg] != 0.0) ? 0.0 : h[rg] * (sa[rg] * sb[rg]);
    const double dg = v + fmin(fmax(v, min_diag), max_diag) * inv_mu;
    v = diag ? dg : v;
    d[rg] = (fa[rg] != 0.0) ? (diag ? 1.0 : 0.0) : v;
  }
}

// L (coupling to the node on the left) of the node at nb into an LDS tile, linear order
template <int B>
__device__ __forceinline__ void seq_cut_l(const SeqCutDesc<B>& q, const ldsd* nb, ldsd* tile, int lane) {
  constexpr int NT = SeqCutDesc<B>::NT;
  double h[NT], sa[NT], s3[NT], fa[NT], f3[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { h[t] = nb[q.lh[t]]; sa[t] = nb[q.lsa[t]]; s3[t] = nb[q.ls3[t]]; fa[t] = nb[q.lfa[t]]; f3[t] = nb[q.lf3[t]]; }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const bool band = (q.lmask >> (8 + t)) & 1u;
    const double v = (band && fa[t] == 0.0 && f3[t] == 0.0) ? h[t] * sa[t] * s3[t] : 0.0;
    if ((q.lmask >> t) & 1u) tile[lane + WAVE * t] = v;
  }
}

// tile[row][col] = d - acc (result layout -> row-major LDS tile); ZERO_D: the tile receives -acc
template <int B>
__device__ __forceinline__ void seq_put_d(ldsd* tile, const double (&d)[3], const v4f64& acc, unsigned dmask, int lane) {
  const int mr = lane & 15, mk = lane >> 4;
#pragma unroll
  for (int rg = 0; rg < 3; ++rg)
    if ((dmask >> rg) & 1u) tile[(mk + 4 * rg) * B + mr] = d[rg] - acc[rg];
}
template <int B>
__device__ __forceinline__ void seq_put_neg(ldsd* tile, const v4f64& acc, double sign, unsigned dmask, int lane) {
  const int mr = lane & 15, mk = lane >> 4;
#pragma unroll
  for (int rg = 0; rg < 3; ++rg)
    if ((dmask >> rg) & 1u) tile[(mk + 4 * rg) * B + mr] = sign * acc[rg];
}

// one node: columns from LDS tiles, Gauss-Jordan, factors to global memory, (G^T w)[c] in the lanes of groups 2 / 3, the three
// products accumulated INTO o.c1 / o.c2 / o.c3 (the caller zeroes what must start fresh).  add_d: D = Du + Dadd (the middle node)
template <int B>
__device__ __forceinline__ void seq_node(double* __restrict__ gDinv, double* __restrict__ gGl, double* __restrict__ gGr, const ldsd* Du, const ldsd* Dadd,
                                         bool add_d, const ldsd* Lu, const ldsd* Lb, ldsd* Gs, const ldsd* wu, SeqNodeOut<B>& o, int lane, bool probe_on = false) {
  constexpr int BB = B * B;
  const int grp = lane / B, c = lane % B;
  (void)probe_on;
  SEQ_T(2);
  double col[B];
  int c_id = (grp == 1) ? c : -1;
  asm volatile("" : "+v"(c_id));
  {
    const ldsd* sp = (grp == 2) ? Lu + c : (grp == 3) ? Lb + c * B : Du + c;
    const int st = (grp == 3) ? 1 : B;
    const bool rd = (grp == 0) || (grp == 2) || (grp == 3);
    double vv[B];
#pragma unroll
    for (int r = 0; r < B; ++r) vv[r] = sp[r * st];
    if (add_d) {                                             // (wave-uniform) the middle node: what the left sweep left + what the right sweep left
      double va[B];
#pragma unroll
      for (int r = 0; r < B; ++r) va[r] = Dadd[r * B + c];
#pragma unroll
      for (int r = 0; r < B; ++r) vv[r] = (grp == 0) ? vv[r] + va[r] : vv[r];
    }
#pragma unroll
    for (int r = 0; r < B; ++r) asm volatile("" : "+v"(vv[r]));
#pragma unroll
    for (int r = 0; r < B; ++r) col[r] = rd ? vv[r] : ((r == c_id) ? 1.0 : 0.0);
  }
  SEQ_T(3);
  gauss_jordan_cols<B>(col);
  SEQ_T(4);
  if (grp >= 1 && grp <= 3) {
    double* dst = (grp == 1) ? gDinv : (grp == 2) ? gGl : gGr;
#pragma unroll
    for (int r = 0; r < B; ++r) dst[r * B + c] = col[r];
  }
  double cw = 0.0;
  if (grp == 2 || grp == 3) {
    double wr[B];
#pragma unroll
    for (int r = 0; r < B; ++r) wr[r] = wu[r];
#pragma unroll
    for (int r = 0; r < B; ++r) cw += col[r] * wr[r];                              // (G^T w_u)[c]
    ldsd* dstg = Gs + (grp == 3 ? BB : 0);
#pragma unroll
    for (int r = 0; r < B; ++r) dstg[r * B + c] = col[r];
  }
  o.cw = cw;
  SEQ_T(5);
  {
    const int mr = lane & 15, mk = lane >> 4;
    const int mrc = (mr < B) ? mr : B - 1;
#pragma unroll
    for (int ks = 0; ks < (B + 3) / 4; ++ks) {
      const int k = 4 * ks + mk;
      const int kc = (k < B) ? k : B - 1;
      const bool in = (k < B) && (mr < B);
      double a_lut = Lu[kc * B + mrc];                                 // L_u^T[mr][k]
      double a_lb = Lb[mrc * B + kc];                                  // L_b[mr][k]
      double b_gl = Gs[kc * B + mrc];                                  // Gl[k][mr]
      double b_gr = Gs[BB + kc * B + mrc];                             // Gr[k][mr]
      asm volatile("" : "+v"(a_lut), "+v"(a_lb), "+v"(b_gl), "+v"(b_gr));
      a_lut = in ? a_lut : 0.0;
      a_lb = in ? a_lb : 0.0;
      b_gl = in ? b_gl : 0.0;
      b_gr = in ? b_gr : 0.0;
      o.c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_lut, b_gl, o.c1, 0, 0, 0);    // += L_u^T Gl   (onto neighbour a)
      o.c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_lb, b_gr, o.c2, 0, 0, 0);     // += L_b Gr     (onto neighbour b)
      o.c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_lb, b_gl, o.c3, 0, 0, 0);     // += L_b Gl     (minus the new coupling b <-> a)
    }
  }
  SEQ_T(6);
}

constexpr int kCrSeqThreads = 128;
// Chunk size of the two-way stage.  A workgroup is two waves and every chunk is resident from the start, so the stage ends when the
// fullest compute unit does: with 1042 chunks of 16 on 256 CUs, 18 CUs hold five workgroups and finish 10 us after the rest
// (per-workgroup timeline, tools/dev/seq_probe.sh).  Chunks are therefore sized so that no CU holds more than four; short chains keep
// chunks of 8 (fewer nodes in sequence per wave; the stages above are latency-bound either way).
static int cr_seq_chunk(int nsb, int B, int cus) {
  const int top = (B == 3 ? 128 : 40) - 1;                 // what the top stage takes in one workgroup (cr_plan)
  if (cus < 1) cus = 256;
  const int cap = std::min(24, 256 / B + 1);               // k_cr_seq_back: (M - 1) * B <= 256 threads
  int m = (nsb - 1 + 4 * cus - 1) / (4 * cus);
  if (m < 8) {
    m = 8;
    for (int k = 8; k <= cap; ++k)                           // short chains: the smallest chunk that hands the top stage everything at once
      if ((nsb + k - 1) / k <= top) { m = k; break; }
  }
  return m > cap ? cap : m;
}
template <int B>
static size_t cr_seq_lds_bytes(int M) {
  const int Wb = B / 3, rl = (Wb + 1) * 9 + 4;
  return sizeof(double) * ((size_t)(M + 2) * Wb * rl + (size_t)(M + 1) * B + 2 * (size_t)(7 * B * B + 3 * B));
}

template <int B>
__global__ __launch_bounds__(kCrSeqThreads) void k_cr_seq(CrView cr, CrStage sg, CrSource src, double* w) {
  if (cr.lm != nullptr) {                                  // device-driven LM loop (uniform branch): the CURRENT buffers, today's trust region
    if (cr.lm->done) return;
    if (src.lm != nullptr) {
      if (src.lm->cur) { src.Hband = src.Hband_alt; src.g = src.g_alt; }
      src.mu = src.lm->radius;
    }
  }
  extern __shared__ double lds_raw[];
  ldsd* lds = (ldsd*)lds_raw;
  constexpr int BB = B * B, Wb = B / 3;
  constexpr int src_rl = (Wb + 1) * 9 + 4;
  const int M = sg.M;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int grp = lane / B, c = lane % B;
  const int64_t base = (int64_t)blockIdx.x * M;
  const int64_t nst = cr.nsb;                                  // first stage: stride 1, every node alive
  const int u_last = (int)((nst - 1 - base < M) ? (nst - 1 - base) : M);
  const bool has_rb = (u_last == M);                           // the chunk ends in a boundary node (kept for the next stage)
  const int U = has_rb ? M - 1 : u_last;                       // interior nodes 1 .. U  (>= 1)
  const int mid = (U + 1) / 2;                                 // left sweep 1 .. mid-1, right sweep U .. mid+1, then mid
  const bool last_is_rb = has_rb && (base + M == nst - 1);
  double* const gDinv = cr.Dinv + (size_t)base * BB;           // this chunk's factor tiles
  double* const gGl = cr.Gl + (size_t)base * BB;
  double* const gGr = cr.Gr + (size_t)base * BB;
  double* const gw = w + (size_t)base * B;
  const int n_poses = src.n_poses;
  const double inv_mu = 1.0 / src.mu, min_diag = src.min_diag, max_diag = src.max_diag;
  SEQ_PROBE(0, wall_clock64());
  SEQ_PROBE(3, (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32));
  ldsd* sRec = lds;
  ldsd* sW = sRec + (M + 2) * Wb * src_rl;                     // raw right-hand sides of local nodes 0 .. u_last
  constexpr int kWaveLds = 7 * BB + 3 * B;
  ldsd* T = sW + (M + 1) * B + wave * kWaveLds;
  // per wave: tD[2] (D of the current / next node), tL[2], tF (the travelling fill: F_u on the left sweep, E_u on the right),
  // tG (Gl, Gr of the node being eliminated; at the end wave 1's accL), tw[2] (w'_u of the current / next node), tacc (wave 1's accWL)
  ldsd* const tF = T + 4 * BB;
  ldsd* const tG = T + 5 * BB;
  ldsd* const tacc = T + 7 * BB + 2 * B;
#define SEQ_TD(i) (T + (i) * BB)
#define SEQ_TL(i) (T + (2 + (i)) * BB)
#define SEQ_TW(i) (T + 7 * BB + (i) * B)
  const ldsd* T1 = sW + (M + 1) * B + kWaveLds;                // wave 1's block, as wave 0 sees it
  // ---- stage the band rows of this chunk's poses and of the super-block on its left (k_cr_chunk's prologue); poses beyond the
  //      end of the map are staged as constant poses ----
  const int src_p0 = (int)((base > 0 ? base - 1 : 0) * Wb);
  {
    const int p_end = (int)(base + u_last + 1) * Wb;           // one past the last pose of the chunk's last node
    const int p1 = min(n_poses, p_end);
    const int np_st = max(p1 - src_p0, 0);
    constexpr int hl = (Wb + 1) * 9;
    const double* Hg = src.Hband + (size_t)src_p0 * hl;
    if ((hl & 1) == 0) {
      const double2* Hg2 = reinterpret_cast<const double2*>(Hg);
      constexpr int h2 = hl / 2;
      for (int e = tid; e < np_st * h2; e += kCrSeqThreads) {
        const double2 v = Hg2[e];
        ldsd* d = sRec + (e / h2) * src_rl + 2 * (e % h2);
        d[0] = v.x; d[1] = v.y;
      }
    } else
      for (int e = tid; e < np_st * hl; e += kCrSeqThreads) sRec[(e / hl) * src_rl + e % hl] = Hg[e];
    for (int e = tid; e < np_st * 4; e += kCrSeqThreads) {
      const int pp = e >> 2, cc = e & 3;
      sRec[pp * src_rl + hl + cc] = (cc < 3) ? src.scale[3 * (src_p0 + pp) + cc] : (src.fixed[src_p0 + pp] ? 1.0 : 0.0);
    }
    for (int e = tid; e < (p_end - src_p0 - np_st) * src_rl; e += kCrSeqThreads)      // padding poses of the map's last super-block
      sRec[np_st * src_rl + e] = (e % src_rl == hl + 3) ? 1.0 : 0.0;
    for (int e = tid; e < (u_last + 1) * B; e += kCrSeqThreads) {
      const int u = e / B;
      const double v = cr_source_rhs(src, (int)(base + u), e % B);
      src.rhs[(size_t)(base * B) + e] = v;                     // the copy the residual check and the step bookkeeping read
      sW[e] = v;
      if (u == 0 || (u == M && last_is_rb)) gw[e] = v;         // boundary nodes: raw (the next stage folds the contributions in);
    }                                                          // interior nodes: their final value, once, when they are eliminated
  }
  SeqCutDesc<B> q;
  seq_cut_desc<B>(q, lane);
  const ldsd* nb0 = sRec + ((int)base * Wb - src_p0) * src_rl;  // first pose record of local node 0; node u: + u * Wb * src_rl
  constexpr int kNodeRec = Wb * src_rl;
  lds_barrier();
  SEQ_PROBE(1, wall_clock64());
  SeqNodeOut<B> o;
  const v4f64 zero4 = {0.0, 0.0, 0.0, 0.0};
  o.c1 = zero4; o.c2 = zero4; o.c3 = zero4; o.cw = 0.0;
  double accw = 0.0;                                           // lanes of group 2 (wave 0: accWR) / group 3 (wave 1: accWL)
  double dn[3];
  const int mr_ = lane & 15, mk_ = lane >> 4;
  int cur = 0;
  if (wave == 0) {
    // boundary node 0 keeps its raw D in memory; node 1 opens the chain: D_1, F_1 = L_1
    seq_cut_d<B>(q, nb0, inv_mu, min_diag, max_diag, dn);
#pragma unroll
    for (int rg = 0; rg < 3; ++rg)
      if ((q.dmask >> rg) & 1u) cr.D[(size_t)base * BB + (mk_ + 4 * rg) * B + mr_] = dn[rg];
    seq_cut_d<B>(q, nb0 + kNodeRec, inv_mu, min_diag, max_diag, dn);
    seq_put_d<B>(SEQ_TD(0), dn, zero4, q.dmask, lane);
    seq_cut_l<B>(q, nb0 + kNodeRec, tF, lane);
    if (lane < B) SEQ_TW(0)[lane] = sW[B + lane];
    for (int u = 1; u < mid; ++u) {
      const int nxt = cur ^ 1;
#ifdef HITL_CR_PROBE
      const bool probe_on = (blockIdx.x == 500 && (u == 3 || u == 4));
      if (u == 4) { __builtin_amdgcn_s_waitcnt(0); if (probe_on && lane == 0) g_seq_node_probe[8] = clock64(); }
      else SEQ_T(0);
#else
      const bool probe_on = false;
#endif
      const ldsd* nbn = nb0 + (u + 1) * kNodeRec;
      seq_cut_d<B>(q, nbn, inv_mu, min_diag, max_diag, dn);                                            // D_{u+1} (registers), L_{u+1}
      seq_cut_l<B>(q, nbn, SEQ_TL(0), lane);
      if (u == 3) SEQ_T(1);
      if (lane < B) gw[u * B + lane] = SEQ_TW(cur)[lane];                                          // final right-hand side of node u
      o.c2 = zero4; o.c3 = zero4;
      seq_node<B>(gDinv + (size_t)u * BB, gGl + (size_t)u * BB, gGr + (size_t)u * BB, SEQ_TD(cur), SEQ_TD(cur), false, tF, SEQ_TL(0), tG, SEQ_TW(cur), o, lane,
                  probe_on && u == 3);
      seq_put_d<B>(SEQ_TD(nxt), dn, o.c2, q.dmask, lane);                                          // D'_{u+1} = D_{u+1} - L_{u+1} Gr
      seq_put_neg<B>(tF, o.c3, -1.0, q.dmask, lane);                                               // F_{u+1} = -L_{u+1} Gl
      if (u == 3) SEQ_T(7);
      if (grp == 2) accw += o.cw;
      if (grp == 3) SEQ_TW(nxt)[c] = sW[(u + 1) * B + c] - o.cw;
      cur = nxt;
    }
  } else {
    // right sweep; E_U = L_{U+1} when the chunk ends in a boundary node (whose raw D stays in memory if nobody else owns it)
    if (has_rb) {
      seq_cut_l<B>(q, nb0 + M * kNodeRec, tF, lane);
      if (last_is_rb) {
        seq_cut_d<B>(q, nb0 + M * kNodeRec, inv_mu, min_diag, max_diag, dn);
#pragma unroll
        for (int rg = 0; rg < 3; ++rg)
          if ((q.dmask >> rg) & 1u) cr.D[(size_t)(base + M) * BB + (mk_ + 4 * rg) * B + mr_] = dn[rg];
      }
    } else
      for (int e = lane; e < BB; e += WAVE) tF[e] = 0.0;
    if (U > mid) {
      seq_cut_d<B>(q, nb0 + U * kNodeRec, inv_mu, min_diag, max_diag, dn);
      seq_put_d<B>(SEQ_TD(0), dn, zero4, q.dmask, lane);
      seq_cut_l<B>(q, nb0 + U * kNodeRec, SEQ_TL(0), lane);
      if (lane < B) SEQ_TW(0)[lane] = sW[U * B + lane];
    } else {                                                   // nothing to sweep: nothing pending for the middle node
      for (int e = lane; e < BB; e += WAVE) SEQ_TD(0)[e] = 0.0;
      if (lane < B) SEQ_TW(0)[lane] = 0.0;
    }
    for (int u = U; u > mid; --u) {
      const int nxt = cur ^ 1;
      const bool to_mid = (u - 1 == mid);                      // what this node leaves for the middle node is handed over as a difference
      const ldsd* nbn = nb0 + (u - 1) * kNodeRec;
      seq_cut_d<B>(q, nbn, inv_mu, min_diag, max_diag, dn);                                            // D_{u-1} (or 0), L_{u-1}
      if (to_mid) { dn[0] = 0.0; dn[1] = 0.0; dn[2] = 0.0; }
      seq_cut_l<B>(q, nbn, SEQ_TL(nxt), lane);
      if (lane < B) gw[u * B + lane] = SEQ_TW(cur)[lane];
      o.c1 = zero4; o.c3 = zero4;
      seq_node<B>(gDinv + (size_t)u * BB, gGl + (size_t)u * BB, gGr + (size_t)u * BB, SEQ_TD(cur), SEQ_TD(cur), false, SEQ_TL(cur), tF, tG, SEQ_TW(cur), o, lane);
      seq_put_d<B>(SEQ_TD(nxt), dn, o.c1, q.dmask, lane);                                          // D''_{u-1} = D_{u-1} - L_u^T Gl
      seq_put_neg<B>(tF, o.c3, -1.0, q.dmask, lane);                                               // E_{u-1} = -E_u Gl
      if (grp == 3) accw += o.cw;
      if (grp == 2) SEQ_TW(nxt)[c] = (to_mid ? 0.0 : sW[(u - 1) * B + c]) - o.cw;
      cur = nxt;
    }
    // hand-over for the middle node: pending D and w differences (slot 0), E_mid (tF), accL (-> tG), accWL (-> tacc)
    seq_put_neg<B>(tG, o.c2, 1.0, q.dmask, lane);
    if (grp == 3) tacc[c] = accw;
    if (cur != 0) {                                            // wave 0 looks at slot 0
      for (int e = lane; e < BB; e += WAVE) SEQ_TD(0)[e] = SEQ_TD(1)[e];
      if (lane < B) SEQ_TW(0)[lane] = SEQ_TW(1)[lane];
    }
  }
  lds_barrier();
  if (wave != 0) return;
  // ---- the middle node: neighbours = the chunk's two boundaries ----
  const ldsd* bD = T1, *bF = T1 + 4 * BB, *bAccL = T1 + 5 * BB, *bw = T1 + 7 * BB, *bAccW = T1 + 7 * BB + 2 * B;
  if (lane < B) { const double wm = SEQ_TW(cur)[lane] + bw[lane]; SEQ_TW(cur)[lane] = wm; gw[mid * B + lane] = wm; }
  o.c2 = zero4; o.c3 = zero4;
  seq_node<B>(gDinv + (size_t)mid * BB, gGl + (size_t)mid * BB, gGr + (size_t)mid * BB, SEQ_TD(cur), bD, true, tF, bF, tG, SEQ_TW(cur), o, lane);
#undef SEQ_TD
#undef SEQ_TL
#undef SEQ_TW
  // ---- hand the boundary contributions to the next stage (its node index = this chunk index (+1)) ----
  {
    double* PLn = cr.P + (size_t)(sg.next_par * 2 + 0) * sg.np * BB;
    double* PRn = cr.P + (size_t)(sg.next_par * 2 + 1) * sg.np * BB;
    double* WLn = cr.Pw + (size_t)(sg.next_par * 2 + 0) * sg.np * B;
    double* WRn = cr.Pw + (size_t)(sg.next_par * 2 + 1) * sg.np * B;
    const size_t c0 = (size_t)blockIdx.x, c1 = c0 + 1;
    const int mr = lane & 15, mk = lane >> 4;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int row = mk + 4 * rg;
      if (row < B && mr < B) {
        const int e = row * B + mr;
        PRn[c0 * BB + e] = o.c1[rg];
        if (blockIdx.x == 0) PLn[e] = 0.0;                                          // node 0 has no chunk on its left
        if (has_rb) { PLn[c1 * BB + e] = o.c2[rg] + bAccL[e]; cr.L[(size_t)(base + M) * BB + e] = -o.c3[rg]; }
        if (last_is_rb) PRn[c1 * BB + e] = 0.0;                                     // the last node has no chunk on its right
      }
    }
    if (grp == 2) {
      WRn[c0 * B + c] = accw + o.cw;
      if (blockIdx.x == 0) WLn[c] = 0.0;
      if (last_is_rb) WRn[c1 * B + c] = 0.0;
    }
    if (grp == 3 && has_rb) WLn[c1 * B + c] = bAccW[c] + o.cw;
  }
  SEQ_PROBE(2, wall_clock64());
}

// Back-substitution of the two-way stage: thread (u, r) owns row r of interior node u and fetches its three coefficient rows
// once; x_mid from the two boundaries, then one node per step outwards on both half-chains (left: neighbours (0, u + 1), right:
// (u - 1, M)).  The part of a row that only needs the boundary values is taken off before the chain starts.
template <int B>
__global__ __launch_bounds__(256) void k_cr_seq_back(CrView cr, int M, const double* __restrict__ w, double* __restrict__ x) {
  if (cr.lm != nullptr && cr.lm->done) return;
  extern __shared__ double sx[];                         // (M + 1) * B
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * M;
  const int64_t nst = cr.nsb;
  const int u_last = (int)((nst - 1 - base < M) ? (nst - 1 - base) : M);
  const bool has_rb = (u_last == M);
  const int U = has_rb ? M - 1 : u_last;
  const int mid = (U + 1) / 2;
  const int u = 1 + tid / B, r = tid % B;
  const bool mine = u <= U;
  double gl[B], gr[B], v = 0.0;
#pragma unroll
  for (int c = 0; c < B; ++c) { gl[c] = 0.0; gr[c] = 0.0; }
  if (mine) {
    const size_t gn = (size_t)(base + u);
    const double* Di = cr.Dinv + gn * B * B + (size_t)r * B;
    const double* Gl = cr.Gl + gn * B * B + (size_t)r * B;
    const double* Gr = cr.Gr + gn * B * B + (size_t)r * B;
    const double* wk = w + gn * B;
#pragma unroll
    for (int c = 0; c < B; ++c) { v += Di[c] * wk[c]; gl[c] = Gl[c]; gr[c] = Gr[c]; }
  }
  if (tid < 2 * B) {
    const int ub = (tid < B) ? 0 : M, rr = tid % B;
    sx[(size_t)ub * B + rr] = (ub == 0 || has_rb) ? x[(size_t)(base + ub) * B + rr] : 0.0;
  }
  __syncthreads();
  if (mine) {
    const double* x0 = sx, *xM = sx + (size_t)M * B;
    if (u <= mid) {
#pragma unroll
      for (int c = 0; c < B; ++c) v -= gl[c] * x0[c];
    }
    if (u >= mid) {
#pragma unroll
      for (int c = 0; c < B; ++c) v -= gr[c] * xM[c];
    }
    if (u == mid) { sx[(size_t)u * B + r] = v; x[(size_t)(base + u) * B + r] = v; }
  }
  __syncthreads();
  const int steps = max(mid - 1, U - mid);
  for (int k = 1; k <= steps; ++k) {
    if (mine && (u == mid - k || u == mid + k)) {
      const double* xn = sx + (size_t)((u < mid) ? u + 1 : u - 1) * B;
      double acc = v;
      if (u < mid) {
#pragma unroll
        for (int c = 0; c < B; ++c) acc -= gr[c] * xn[c];
      } else {
#pragma unroll
        for (int c = 0; c < B; ++c) acc -= gl[c] * xn[c];
      }
      sx[(size_t)u * B + r] = acc;
      x[(size_t)(base + u) * B + r] = acc;
    }
    __syncthreads();
  }
}

struct CrPlan { CrStage st[8]; int n; };

static size_t cr_chunk_lds_bytes(int M, int B, int threads, int nrhs) {
  const int nr = nrhs > 0 ? nrhs : 1;
  return sizeof(double) * ((size_t)(M + 1) * (2 * B * B + 3 * B * nr) + (size_t)(threads / 64) * 2 * B * B + 2 * B * B + 2 * B * nr);
}
// extra dynamic LDS of a first stage that stages its band rows (k_cr_chunk, from_src) when they are longer than what lies behind the tiles
static size_t cr_chunk_src_extra(int nsl, int B, int threads, int Wb, int nrhs) {
  const int nr = nrhs > 0 ? nrhs : 1;
  const size_t need = (size_t)(nsl + 1) * Wb * ((Wb + 1) * 9 + 4);
  const size_t avail = (size_t)nsl * 3 * B * nr + (size_t)(threads / 64) * 2 * B * B + 2 * B * B + 2 * B * nr;
  return (need <= avail) ? 0 : (need - avail) * sizeof(double);
}
static int cr_chunk_threads(int M, int B, int stage = 1) {
  // first stage of the narrow bands: a node costs a wave whatever the tile size, and thousands of chunks queue for the CUs' 32
  // wave slots -- fewer waves per workgroup (the first level in several passes) puts more chunks on a CU at once.  Measured at
  // 40 000 - 50 000 poses: B = 3: 128 threads 4.9 ms per mode-A solve against 5.6 with 512; B = 6: 256 threads 5.5 against 5.9;
  // B = 9 and 12: 512 stays best (workgroup-size sweeps of round 3)
  if (stage == 0 && M == 16 && B == 3) return 128;
  if (stage == 0 && M == 16 && B == 6) return 256;
  return (M <= 2) ? 64 : (M <= 4) ? 128 : (M <= 8) ? 256 : (M >= 32 && B < 12) ? 1024 : 512;
}

// stages of the fused factorisation for nsb super-blocks (host): chunks of 16 at stride 1, chunks of 32 above, then the top.
// (Chunks of M = 2 -- one wave per eliminated node, no barrier -- were measured for the wide first levels and are not
// faster: a node costs a whole wave's instruction stream either way; chunks of 4 or 8 at 2 waves per workgroup: 56 / 68 us for
// the first stage against 54 us for chunks of 16.)
// m0 > 0: the first stage is the two-way sequential one (k_cr_seq) with chunks of m0 nodes (any size >= 2, not only powers of two)
static CrPlan cr_plan(const CrView& crv, int B, int np, int nrhs, size_t wstride, int m0 = 0) {
  const int nsb = crv.nsb; (void)nsb;
  CrPlan pl;
  pl.n = 0;
  const size_t lds_cap = 128 * 1024;
  int S = 1, par = 0;
  bool prev = false;
  for (;;) {
    const int64_t alive = cr_stage_nodes(crv, S);
    int Mtop = 2;
    while (Mtop < alive) Mtop *= 2;                        // interior u = 1 .. alive-1 < Mtop
    // the top takes every remaining node in one workgroup: up to 40 of them (128 of the 3 x 3 tiles of the one-pose band: seven
    // levels in one launch instead of a stage of its own in front of a two-level top)
    const bool fits_top = alive <= (B == 3 ? 128 : 40) && cr_chunk_lds_bytes((int)alive, B, cr_chunk_threads(64, B), nrhs) <= lds_cap;
    CrStage& g = pl.st[pl.n++];
    g.S = S; g.has_prev = prev ? 1 : 0; g.prev_par = par; g.next_par = par ^ 1; g.np = np; g.nrhs = nrhs; g.wstride = wstride;
    // (a separator chain ends in an ordinary chunk -- both ends stay -- that holds every remaining node)
    if (fits_top || pl.n == 8) { g.M = Mtop; g.top = crv.rsep ? 0 : 1; g.nsl = (int)alive + 1; break; }
    int M = (pl.n == 1) ? 16 : 32;       // measured on C4 (tools/cr_plans.sh): 16,32,top = 111 us; 2,2,16,.. = 117; 2,2,2,32,.. = 125
    if (cr_chunk_lds_bytes(M, B, cr_chunk_threads(M, B), nrhs) > lds_cap) M = 16;      // many right-hand sides (submap solve): smaller chunks
    if (cr_chunk_lds_bytes(M, B, cr_chunk_threads(M, B), nrhs) > lds_cap) M = 8;
    if (pl.n == 1 && m0 >= 2) M = m0;
    g.M = M; g.top = 0; g.nsl = g.M + 1;
    S *= g.M; par ^= 1; prev = true;
  }
  return pl;
}

int cr_contrib_slots(int nsb) { return nsb / 2 + 3; }     // boundary nodes of the finest chunk stage (M = 2)

// validate == true: nothing is launched, the return value says whether the fused stages apply.  Every check runs BEFORE the
// first launch, so a `false` never leaves half a factorisation behind for the level-per-launch fallback.
template <int B>
static bool cr_factor_fused(hipStream_t st, const CrView& cr, double* w, double* x, int nrhs, size_t wstride, const CrSource* src = nullptr,
                            bool validate = false, int phase = 3 /* 1: the forward stages only, 2: the back-substitution only */) {
  if (cr.P == nullptr || cr.Pw == nullptr) return false;
  if (w == nullptr) nrhs = 0;
  if (src != nullptr && (x == nullptr || nrhs != 1 + src->n_unit)) return false;
  if (nrhs > cr.pw_rhs) return false;                     // contribution arrays are sized for cr.pw_rhs right-hand sides
  static bool raised = false;
  if (!raised) {
    raised = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cr_chunk<B, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess &&
             hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cr_chunk<B, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess &&
             hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cr_chunk<B, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess &&
             hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cr_chunk<B, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
    if (!raised) { (void)hipGetLastError(); return false; }
  }
  // the LM path's first stage (tiles from the band rows, one right-hand side, a single GPU's chain): two-way sequential order
  // (long chains only: the stage is about instruction throughput; on a short chain -- C2: 334 super-blocks -- a wave's 4-5 node
  // eliminations in sequence cost more than the 4 levels of the cyclic-reduction stage: 1.14 against 1.03 ms per re-solve)
  const bool seq_ok = src != nullptr && nrhs == 1 && 3 * src->Wb == B && !cr.rsep && dev_switches().cr_seq &&
                      (cr.nsb >= 4096 || dev_switches().cr_seq_m >= 2);
  const int m0 = !seq_ok ? 0 : dev_switches().cr_seq_m >= 2 ? dev_switches().cr_seq_m : cr_seq_chunk(cr.nsb, B, cr.cus);
  const CrPlan pl = cr_plan(cr, B, cr.np, nrhs, wstride, m0);
  const bool seq0 = seq_ok && pl.n >= 2 && !pl.st[0].top && pl.st[0].M == m0 && cr_seq_lds_bytes<B>(m0) <= 64 * 1024;
  if (seq_ok && !seq0 && dev_switches().cr_seq_m >= 2) return false;           // (a forced chunk size that cannot run)
  for (int t = 0; t < pl.n; ++t) {                        // every stage must fit its LDS budget, else the level-per-launch path
    const CrStage& g = pl.st[t];
    if (t == 0 && seq0) { if ((g.M - 1) * B > 256) return false; continue; }
    if (g.M < 2 || g.M > (B == 3 ? 128 : 64) || (g.M & (g.M - 1)) != 0) return false;
    const int thr = cr_chunk_threads(g.M, B, t);
    const bool use_src = (src != nullptr && t == 0);
    if (cr_chunk_lds_bytes(g.nsl - 1, B, thr, nrhs) + (use_src ? cr_chunk_src_extra(g.nsl, B, thr, src->Wb, nrhs) : 0) > 128 * 1024) return false;
    int bthr = (g.M - (g.top ? 0 : 1)) * B;
    bthr = (bthr + 63) & ~63;
    if (bthr > 1024) return false;                                           // never with M <= 64, B <= 12 (or M = 128, B = 3)
  }
  if (validate) return true;
  // the last chunk stage + the top + both their back-substitutions as ONE launch (k_cr_tail)
  const int t2 = pl.n - 2;
  int tail_chunks = 0;
  bool tail = phase == 3 && w != nullptr && x != nullptr && nrhs == 1 && !cr.rsep && pl.n >= 2 && B <= 9 && cr.ticket != nullptr && !cr.no_tail && dev_switches().cr_tail &&
              !(t2 == 0 && seq0);
  if (tail) {
    const CrStage& g2 = pl.st[t2];
    tail_chunks = (int)((cr_stage_nodes(cr, g2.S) - 1 + g2.M - 1) / g2.M);
    // every workgroup of the launch must be resident at once: 1024 threads and up to 128 KB of LDS = one per compute unit
    if (tail_chunks < 1 || tail_chunks + 1 > std::min(kCrTailMaxGroups, cr.cus) || (g2.M + 1) * B > 1024) tail = false;
  }
  for (int t = 0; t < pl.n && (phase & 1); ++t) {
    if (tail && t >= t2) {
      if (t > t2) continue;
      const CrStage& g2 = pl.st[t2];
      const CrStage& gt = pl.st[t2 + 1];
      const bool use_src2 = (src != nullptr && t2 == 0);
      const bool kind1 = use_src2 && !g2.has_prev && g2.S == 1 && 3 * src->Wb == B;
      const int thr = 1024;
      size_t lds = cr_chunk_lds_bytes(g2.nsl - 1, B, thr, 1) + (use_src2 ? cr_chunk_src_extra(g2.nsl, B, thr, src->Wb, 1) : 0);
      lds = std::max(lds, cr_chunk_lds_bytes(gt.nsl - 1, B, thr, 1));
      lds = std::max(lds, sizeof(double) * (size_t)(std::max(g2.M, gt.M) + 1) * B);
      if (lds > 128 * 1024) { tail = false; }
      else {
        static bool raised_tail = false;
        if (!raised_tail) {
          raised_tail = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cr_tail<B, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cr_tail<B, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
          if (!raised_tail) (void)hipGetLastError();
        }
        if (!raised_tail) tail = false;
        else {
          CrSource none2 = {};
          CrView crt = cr;
          const bool starve = dev_switches().cr_tail_starve;      // test hook: no top workgroup, a short wait -- the chunk workgroups give up
          if (starve) crt.tail_spins = 1u << 10;
          const int groups = tail_chunks + (starve ? 0 : 1);
          if (kind1) hipLaunchKernelGGL((k_cr_tail<B, 1>), dim3(groups), dim3(thr), lds, st, crt, g2, gt, *src, w, x, tail_chunks, cr.ticket);
          else hipLaunchKernelGGL((k_cr_tail<B, 0>), dim3(groups), dim3(thr), lds, st, crt, g2, gt, use_src2 ? *src : none2, w, x, tail_chunks, cr.ticket);
          continue;
        }
      }
    }
    const CrStage& g = pl.st[t];
    const int64_t alive = cr_stage_nodes(cr, g.S);
    const int chunks = g.top ? 1 : (int)((alive - 1 + g.M - 1) / g.M);       // chunks with at least one interior or boundary-M node
    const int thr = cr_chunk_threads(g.M, B, t);
    const bool use_src = (src != nullptr && t == 0);
    const size_t lds = cr_chunk_lds_bytes(g.nsl - 1, B, thr, nrhs) + (use_src ? cr_chunk_src_extra(g.nsl, B, thr, src->Wb, nrhs) : 0);
    CrSource none = {};
    if (t == 0 && seq0) {
      hipLaunchKernelGGL((k_cr_seq<B>), dim3(chunks > 0 ? chunks : 1), dim3(kCrSeqThreads), cr_seq_lds_bytes<B>(g.M), st, cr, g, *src, w);
      continue;
    }
    if (use_src && nrhs == 1 && !g.top && !g.has_prev && g.S == 1 && thr <= 512 && 3 * src->Wb == B && (g.M + 1) * B <= thr &&
        !(cr.rsep && t == pl.n - 1)) {
      if (cr.rsep) hipLaunchKernelGGL((k_cr_chunk<B, 1, true>), dim3(chunks > 0 ? chunks : 1), dim3(thr), lds, st, cr, g, *src, w, x);
      else hipLaunchKernelGGL((k_cr_chunk<B, 1>), dim3(chunks > 0 ? chunks : 1), dim3(thr), lds, st, cr, g, *src, w, x);
    } else if (cr.rsep)
      hipLaunchKernelGGL((k_cr_chunk<B, 0, true>), dim3(chunks > 0 ? chunks : 1), dim3(thr), lds, st, cr, g, use_src ? *src : none, w, x);
    else
      hipLaunchKernelGGL((k_cr_chunk<B, 0>), dim3(chunks > 0 ? chunks : 1), dim3(thr), lds, st, cr, g, use_src ? *src : none, w, x);
  }
  if (w != nullptr && x != nullptr && (phase & 2)) {
    for (int t = pl.n - 1; t >= 0; --t) {
      if (tail && t >= t2) continue;                       // k_cr_tail has back-substituted these two stages already
      const CrStage& g = pl.st[t];
      const int64_t alive = cr_stage_nodes(cr, g.S);
      if (alive <= 1) continue;
      const int chunks = g.top ? 1 : (int)((alive - 1 + g.M - 1) / g.M);
      int thr = (g.M - (g.top ? 0 : 1)) * B;
      thr = (thr + 63) & ~63;
      const int grid = (nrhs > 1) ? 8 * nrhs * ((chunks + 7) / 8) : chunks;
      if (t == 0 && seq0) {
        hipLaunchKernelGGL((k_cr_seq_back<B>), dim3(chunks), dim3(thr), sizeof(double) * (g.M + 1) * B, st, cr, g.M, w, x);
        continue;
      }
      if (cr.rsep)
        hipLaunchKernelGGL((k_cr_chunk_back<B, true>), dim3(grid), dim3(thr), sizeof(double) * (g.M + 1) * B, st, cr, g.S, g.M, g.top, w, x, wstride, chunks, nrhs);
      else
        hipLaunchKernelGGL((k_cr_chunk_back<B, false>), dim3(grid), dim3(thr), sizeof(double) * (g.M + 1) * B, st, cr, g.S, g.M, g.top, w, x, wstride, chunks, nrhs);
    }
  }
  return true;
}

// x = M^-1 w through the stored factors, one right-hand side, fused stages (3 + 3 launches instead of ~30)
template <int B>
static bool cr_solve_fused(hipStream_t st, const CrView& cr, const double* w_in, double* w, double* x) {
  if (cr.P == nullptr || cr.Pw == nullptr) return false;
  static bool raised = false;
  if (!raised) {
    raised = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cr_chunk_fwd<B>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
    if (!raised) { (void)hipGetLastError(); return false; }
  }
  if (cr.rsep) return false;
  const CrPlan pl = cr_plan(cr, B, cr.np, 1, 0);
  for (int t = 0; t < pl.n; ++t) {
    const CrStage& g = pl.st[t];
    const int64_t alive = cr_stage_nodes(cr, g.S);
    const int chunks = g.top ? 1 : (int)((alive - 1 + g.M - 1) / g.M);
    const size_t lds = sizeof(double) * ((size_t)g.nsl * (2 * B * B + 3 * B) + 2 * B);
    if (lds > 128 * 1024) return false;
    hipLaunchKernelGGL(k_cr_chunk_fwd<B>, dim3(chunks > 0 ? chunks : 1), dim3(256), lds, st, cr, g, w_in, w, x);
  }
  for (int t = pl.n - 1; t >= 0; --t) {
    const CrStage& g = pl.st[t];
    const int64_t alive = ((int64_t)cr.nsb + g.S - 1) / g.S;
    if (alive <= 1) continue;
    const int chunks = g.top ? 1 : (int)((alive - 1 + g.M - 1) / g.M);
    int thr = (g.M - (g.top ? 0 : 1)) * B;
    thr = (thr + 63) & ~63;
    if (thr > 1024) return false;
    hipLaunchKernelGGL((k_cr_chunk_back<B, false>), dim3(chunks), dim3(thr), sizeof(double) * (g.M + 1) * B, st, cr, g.S, g.M, g.top, w, x, (size_t)0, chunks, 1);
  }
  return true;
}

// x == nullptr: factorise only (w, if given, still receives the forward sweep)
template <int B>
static void cr_factor_t(hipStream_t st, const CrView& cr, double* w, double* x, int nrhs, size_t stride) {
  const int nr = (w != nullptr) ? nrhs : 0;
  const bool legacy = dev_switches().cr_legacy;       // development switch: the level-per-launch path
  if (!legacy && cr_factor_fused<B>(st, cr, nr ? w : nullptr, nr ? x : nullptr, nr, stride)) return;
  const int s_top = cr_top_start(cr.nsb, B, nr);
  if (s_top > 1) {
    const int n_elim = (cr.nsb - 1 + 1) / 2;
    hipLaunchKernelGGL(k_cr_invert<B>, dim3(n_elim), dim3(WAVE), 0, st, cr, 1);
  }
  for (int s = 1; s < s_top; s *= 2) {
    const int n_surv = (cr.nsb + 2 * s - 1) / (2 * s);
    const int invert_next = (2 * s < s_top) ? 1 : 0;    // the fused top kernel inverts its own nodes
    hipLaunchKernelGGL(k_cr_level<B>, dim3(n_surv), dim3(kCrLevelThreads), 0, st, cr, s, invert_next, w, nrhs, stride);
  }
  {
    const int M = (cr.nsb + s_top - 1) / s_top;
    const size_t lds_bytes = cr_top_lds_bytes(M, B, nr);
    static bool raised = false;      // more than the default 64 KB of dynamic LDS needs to be asked for once
    if (!raised) {
      raised = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cr_top_lds<B>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kCrTopLdsBytes) == hipSuccess;
      if (!raised) (void)hipGetLastError();
    }
    if (lds_bytes <= (raised ? kCrTopLdsBytes : (size_t)64 * 1024))
      hipLaunchKernelGGL(k_cr_top_lds<B>, dim3(1), dim3(kCrTopLdsThreads), lds_bytes, st, cr, s_top, M, w, x, nr, stride);
    else
      hipLaunchKernelGGL(k_cr_top<B>, dim3(1), dim3(kCrTopThreads), 0, st, cr, s_top, w, x, nrhs, stride);
  }
  if (x != nullptr && s_top >= 2) cr_backward_levels<B>(st, cr, s_top / 2, w, x, nrhs, stride);
}

// back substitution only, any number of right-hand sides (after cr_fwd_t)
template <int B>
static void cr_back_t(hipStream_t st, const CrView& cr, const double* w, double* x, int nrhs, size_t stride) {
  int s_last = 0;
  for (int s = 1; s < cr.nsb; s *= 2) s_last = s;
  hipLaunchKernelGGL(k_cr_root_solve<B>, dim3(nrhs), dim3(WAVE), 0, st, cr, w, x, stride);
  if (s_last >= 1) cr_backward_levels<B>(st, cr, s_last, w, x, nrhs, stride);
}

template <int B>
static void cr_fwd_t(hipStream_t st, const CrView& cr, double* w, int nrhs, size_t stride) {
  for (int s = 1; s < cr.nsb; s *= 2) {
    const int n_surv = (cr.nsb + 2 * s - 1) / (2 * s);
    hipLaunchKernelGGL(k_cr_forward<B>, dim3((n_surv * B + 127) / 128, nrhs), dim3(128), 0, st, cr, s, w, stride);
  }
}

template <int B>
static void cr_solve_t(hipStream_t st, const CrView& cr, double* w, double* x, int nrhs, size_t stride) {
  cr_fwd_t<B>(st, cr, w, nrhs, stride);
  cr_back_t<B>(st, cr, w, x, nrhs, stride);
}

#ifdef HITL_CR_PROBE
extern "C" void hitl_seq_probe_dump(unsigned long long* out, int n) {      // n * 4 words
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_seq_probe), sizeof(unsigned long long) * 4 * (size_t)(n < 4096 ? n : 4096));
  unsigned long long h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_seq_node_probe), sizeof(h)) != hipSuccess) return;
  fprintf(stderr, "seq node probe (shader cycles): cut %llu, column load %llu, Gauss-Jordan %llu, stores + G^T w + G to LDS %llu, products %llu, puts %llu; whole step %llu\n",
          h[1] - h[0], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[8] - h[0]);
}
extern "C" void hitl_cr_probe_dump() {
  unsigned long long h[3 * 64];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cr_probe), sizeof(h)) != hipSuccess) return;
  for (int st = 0; st < 3; ++st) {
    fprintf(stderr, "cr probe stage %d:", st);
    unsigned long long prev = h[st * 64];
    for (int k = 1; k < 64; ++k)
      if (h[st * 64 + k] != 0) { fprintf(stderr, " [%d]%llu", k, h[st * 64 + k] - prev); prev = h[st * 64 + k]; }
    fprintf(stderr, "\n");
  }
}
#endif

void launch_cr_factor(hipStream_t st, const CrView& cr) {
  switch (cr.B) {
    case 3: cr_factor_t<3>(st, cr, nullptr, nullptr, 0, 0); break;
    case 6: cr_factor_t<6>(st, cr, nullptr, nullptr, 0, 0); break;
    case 9: cr_factor_t<9>(st, cr, nullptr, nullptr, 0, 0); break;
    default: cr_factor_t<12>(st, cr, nullptr, nullptr, 0, 0); break;
  }
}

void launch_cr_factor_solve(hipStream_t st, const CrView& cr, double* w, double* x, int nrhs, size_t stride) {
  // nrhs right-hand sides `stride` doubles apart ride through the factorisation
  switch (cr.B) {
    case 3: cr_factor_t<3>(st, cr, w, x, nrhs, stride); break;
    case 6: cr_factor_t<6>(st, cr, w, x, nrhs, stride); break;
    case 9: cr_factor_t<9>(st, cr, w, x, nrhs, stride); break;
    default: cr_factor_t<12>(st, cr, w, x, nrhs, stride); break;
  }
}

void launch_cr_build_factor_solve_multi(hipStream_t st, const CrView& cr, const CrSource& src, double* w, double* x, size_t stride) {
  const bool legacy = dev_switches().cr_legacy || dev_switches().cr_unfused_build;
  const int nrhs = 1 + src.n_unit;
  bool done = false;
  if (!legacy) {
    switch (cr.B) {
      case 3: done = cr_factor_fused<3>(st, cr, w, x, nrhs, stride, &src); break;
      case 6: done = cr_factor_fused<6>(st, cr, w, x, nrhs, stride, &src); break;
      case 9: done = cr_factor_fused<9>(st, cr, w, x, nrhs, stride, &src); break;
      default: done = cr_factor_fused<12>(st, cr, w, x, nrhs, stride, &src); break;
    }
  }
  if (done) return;
  const int tot = cr.nsb * cr.B * cr.B;
  hipLaunchKernelGGL(k_build_system, dim3((tot + 255) / 256), dim3(256), 0, st, src, cr, w);
  launch_unit_rhs(st, cr.nsb * cr.B, w + stride, stride, src.unit_cols, src.n_unit);
  launch_cr_factor_solve(st, cr, w, x, nrhs, stride);
}

// whether launch_cr_build_factor_solve would run as fused stages only (the device-driven LM loop needs that: the level-per-
// launch fallback takes its buffers and the trust region from the host)
bool cr_build_factor_solve_is_fused(const CrView& cr, const CrSource& src, double* w, double* x) {
  if (dev_switches().cr_legacy || dev_switches().cr_unfused_build) return false;
  switch (cr.B) {
    case 3: return cr_factor_fused<3>(nullptr, cr, w, x, 1, 0, &src, true);
    case 6: return cr_factor_fused<6>(nullptr, cr, w, x, 1, 0, &src, true);
    case 9: return cr_factor_fused<9>(nullptr, cr, w, x, 1, 0, &src, true);
    default: return cr_factor_fused<12>(nullptr, cr, w, x, 1, 0, &src, true);
  }
}

void launch_cr_build_factor_solve(hipStream_t st, const CrView& cr, const CrSource& src, double* w, double* x) {
  const bool legacy = dev_switches().cr_legacy || dev_switches().cr_unfused_build;
  bool done = false;
  if (!legacy) {
    switch (cr.B) {
      case 3: done = cr_factor_fused<3>(st, cr, w, x, 1, 0, &src); break;
      case 6: done = cr_factor_fused<6>(st, cr, w, x, 1, 0, &src); break;
      case 9: done = cr_factor_fused<9>(st, cr, w, x, 1, 0, &src); break;
      default: done = cr_factor_fused<12>(st, cr, w, x, 1, 0, &src); break;
    }
  }
  if (done) return;
  const int tot = cr.nsb * cr.B * cr.B;
  hipLaunchKernelGGL(k_build_system, dim3((tot + 255) / 256), dim3(256), 0, st, src, cr, w);
  launch_cr_factor_solve(st, cr, w, x, 1, 0);
}

void launch_cr_solve_multi(hipStream_t st, const CrView& cr, double* w, double* x, int nrhs, size_t stride) {
  switch (cr.B) {
    case 3: cr_solve_t<3>(st, cr, w, x, nrhs, stride); break;
    case 6: cr_solve_t<6>(st, cr, w, x, nrhs, stride); break;
    case 9: cr_solve_t<9>(st, cr, w, x, nrhs, stride); break;
    default: cr_solve_t<12>(st, cr, w, x, nrhs, stride); break;
  }
}

void launch_cr_solve(hipStream_t st, const CrView& cr, const double* rhs, double* w, double* x) {
  const bool legacy = dev_switches().cr_legacy;
  bool done = false;
  if (!legacy) {
    switch (cr.B) {
      case 3: done = cr_solve_fused<3>(st, cr, rhs, w, x); break;
      case 6: done = cr_solve_fused<6>(st, cr, rhs, w, x); break;
      case 9: done = cr_solve_fused<9>(st, cr, rhs, w, x); break;
      default: done = cr_solve_fused<12>(st, cr, rhs, w, x); break;
    }
  }
  if (!done) {
    if (rhs != w) (void)hipMemcpyAsync(w, rhs, sizeof(double) * (size_t)cr.nsb * cr.B, hipMemcpyDeviceToDevice, st);
    launch_cr_solve_multi(st, cr, w, x, 1, 0);
  }
}

// ===========================================================================================
// Submap ranks, separator form: see hitl_internal.h.  The chain's fused stages run with chain.rsep = 1 (k_cr_chunk keeps the last
// node as the right boundary of the last chunk of every stage; the last stage is an ordinary chunk holding every remaining node).
// ===========================================================================================
template <int B>
static int cr_sep_last_parity(const CrView& chain) {
  const CrPlan pl = cr_plan(chain, B, chain.np, 1, 0);
  return pl.st[pl.n - 1].next_par;
}

bool cr_sep_applies(const CrView& chain, const CrSource& src, double* w, double* x) {
  if (!chain.rsep || chain.nsb < 2 || dev_switches().cr_legacy || dev_switches().cr_unfused_build) return false;
  switch (chain.B) {
    case 3: return cr_factor_fused<3>(nullptr, chain, w, x, 1, 0, &src, true);
    case 6: return cr_factor_fused<6>(nullptr, chain, w, x, 1, 0, &src, true);
    case 9: return cr_factor_fused<9>(nullptr, chain, w, x, 1, 0, &src, true);
    default: return cr_factor_fused<12>(nullptr, chain, w, x, 1, 0, &src, true);
  }
}

void launch_cr_sep_forward(hipStream_t st, const CrView& chain, const CrSource& src, double* w, double* x) {
  switch (chain.B) {
    case 3: (void)cr_factor_fused<3>(st, chain, w, x, 1, 0, &src, false, 1); break;
    case 6: (void)cr_factor_fused<6>(st, chain, w, x, 1, 0, &src, false, 1); break;
    case 9: (void)cr_factor_fused<9>(st, chain, w, x, 1, 0, &src, false, 1); break;
    default: (void)cr_factor_fused<12>(st, chain, w, x, 1, 0, &src, false, 1); break;
  }
}

void launch_cr_sep_backward(hipStream_t st, const CrView& chain, double* w, double* x) {
  switch (chain.B) {
    case 3: (void)cr_factor_fused<3>(st, chain, w, x, 1, 0, nullptr, false, 2); break;
    case 6: (void)cr_factor_fused<6>(st, chain, w, x, 1, 0, nullptr, false, 2); break;
    case 9: (void)cr_factor_fused<9>(st, chain, w, x, 1, 0, nullptr, false, 2); break;
    default: (void)cr_factor_fused<12>(st, chain, w, x, 1, 0, nullptr, false, 2); break;
  }
}

// the rank's exchange record from what the forward stages left: D / w of F and R folded with the last stage's contributions (the
// order every stage folds in: left chunk's first), the coupling R <-> F, the raw cross-cut block F x (the Wb poses below the cut)
// and the Jacobi scales of the separators' poses (0 for a constant pose and for the padding rows of a short last block)
__global__ void k_sep_record(CrView cr, int par, const double* __restrict__ w, const double* __restrict__ Hband, const double* __restrict__ scale,
                             const unsigned char* __restrict__ fixed, SepGeom gm, double* __restrict__ rec) {
  const int B = cr.B, BB = B * B, Wb = gm.Wb, W1 = Wb + 1;
  const size_t np = (size_t)cr.np;
  const double* PL = cr.P + (size_t)(par * 2 + 0) * np * BB;
  const double* PR = cr.P + (size_t)(par * 2 + 1) * np * BB;
  const double* WL = cr.Pw + (size_t)(par * 2 + 0) * np * B;
  const double* WR = cr.Pw + (size_t)(par * 2 + 1) * np * B;
  const size_t rn = (size_t)cr.nsb - 1;
  for (int e = threadIdx.x; e < BB; e += blockDim.x) {
    rec[e] = (cr.D[e] - PL[e]) - PR[e];
    rec[BB + e] = (cr.D[rn * BB + e] - PL[BB + e]) - PR[BB + e];
    rec[2 * BB + e] = cr.L[rn * BB + e];
    const int a = e / B, b = e % B, ap = a / 3, r = a % 3, bp = b / 3, c = b % 3;
    double v = 0.0;
    const int i3 = gm.own_lo - Wb + bp, d = (Wb + ap) - bp;        // row pose own_lo + ap, column pose own_lo - Wb + bp
    if (gm.rank > 0 && i3 >= 0 && d <= Wb && gm.own_lo + ap < gm.own_hi) v = Hband[((size_t)i3 * W1 + d) * 9 + c * 3 + r];
    rec[3 * BB + e] = v;
  }
  for (int e = threadIdx.x; e < B; e += blockDim.x) {
    rec[4 * BB + e] = (w[e] - WL[e]) - WR[e];
    rec[4 * BB + B + e] = (w[rn * B + e] - WL[B + e]) - WR[B + e];
    const int pf = gm.own_lo + e / 3, pr = gm.r_pose0 + e / 3;
    rec[4 * BB + 2 * B + e] = (pf < gm.own_hi && !fixed[pf]) ? scale[3 * pf + e % 3] : 0.0;
    rec[4 * BB + 3 * B + e] = (pr < gm.own_hi && !fixed[pr]) ? scale[3 * pr + e % 3] : 0.0;
  }
}

void launch_sep_record(hipStream_t st, const CrView& chain, const double* w, const double* Hband_local, const double* scale_local,
                       const unsigned char* fixed_local, const SepGeom& gm, double* rec) {
  int par = 0;
  switch (chain.B) {
    case 3: par = cr_sep_last_parity<3>(chain); break;
    case 6: par = cr_sep_last_parity<6>(chain); break;
    case 9: par = cr_sep_last_parity<9>(chain); break;
    default: par = cr_sep_last_parity<12>(chain); break;
  }
  hipLaunchKernelGGL(k_sep_record, dim3(1), dim3(192), 0, st, chain, par, w, Hband_local, scale_local, fixed_local, gm, rec);
}

// the chain of all ranks' separators: node 2g = F of rank g (coupled to R of rank g - 1 by its scaled cross-cut block), node
// 2g + 1 = R of rank g (coupled to F by K)
__global__ void k_sep_build(CrView red, const double* __restrict__ rec_all, int rec_len, double* __restrict__ w2) {
  const int B = red.B, BB = B * B;
  const int node = blockIdx.x, g = node >> 1, isR = node & 1;
  const double* rc = rec_all + (size_t)rec_len * g;
  const double* rp = rec_all + (size_t)rec_len * (g > 0 ? g - 1 : 0);
  for (int e = threadIdx.x; e < BB; e += blockDim.x) {
    red.D[(size_t)node * BB + e] = rc[isR ? BB + e : e];
    double l;
    if (isR) l = rc[2 * BB + e];
    else l = (g > 0) ? rc[4 * BB + 2 * B + e / B] * rc[3 * BB + e] * rp[4 * BB + 3 * B + e % B] : 0.0;
    red.L[(size_t)node * BB + e] = l;
  }
  for (int e = threadIdx.x; e < B; e += blockDim.x) w2[(size_t)node * B + e] = rc[4 * BB + isR * B + e];
}

__global__ void k_sep_scatter(int B, const double* __restrict__ rec_all, int rec_len, const double* __restrict__ x2, SepGeom gm,
                              double* __restrict__ dt, double* __restrict__ scale) {
  const int BB = B * B, me = gm.rank, Wb = gm.Wb;
  // A breakdown of the separator chain (or of a rank's Schur complement feeding it) shows as a non-finite entry SOMEWHERE in x2 --
  // maybe only at another rank's separators.  Every rank solves the same chain: each plants a NaN in an owned row when any entry is
  // not finite, so the step scalars of ALL ranks report the step as invalid (slot `bad`) and the LM loop raises the damping, as
  // the interface-column form does when its dense solve fails
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * gm.world * B; t += blockDim.x) if (!isfinite(x2[t])) s_bad = 1;
  __syncthreads();
  const double poison = s_bad ? __builtin_nan("") : 0.0;
  for (int t = threadIdx.x; t < B; t += blockDim.x) {
    dt[3 * gm.own_lo + t] = x2[(size_t)(2 * me) * B + t] + poison;
    dt[3 * gm.r_pose0 + t] = x2[(size_t)(2 * me + 1) * B + t];
  }
  for (int t = threadIdx.x; t < 3 * gm.halo_lo; t += blockDim.x) {          // the last halo_lo poses of the previous rank's R
    const int sc = 3 * (Wb - gm.halo_lo) + t, dst = 3 * (gm.own_lo - gm.halo_lo) + t;
    dt[dst] = x2[(size_t)(2 * (me - 1) + 1) * B + sc];
    scale[dst] = rec_all[(size_t)rec_len * (me - 1) + 4 * BB + 3 * B + sc];
  }
  for (int t = threadIdx.x; t < 3 * gm.halo_hi; t += blockDim.x) {          // the first halo_hi poses of the next rank's F
    const int dst = 3 * gm.own_hi + t;
    dt[dst] = x2[(size_t)(2 * (me + 1)) * B + t];
    scale[dst] = rec_all[(size_t)rec_len * (me + 1) + 4 * BB + 2 * B + t];
  }
}

// Selected inversion in separator form: the chain of all ranks' separators is built and factorised as in launch_sep_solve, its
// own selected inversion gives the blocks of the GLOBAL inverse at this rank's two separators (diagonal blocks of F and R and
// the block between them), and the rank's chain is then walked down from those instead of from a root.  Sd / So: nsb tiles
// each, indexed by chain node; Sd2 / So2: scratch of 2 * world tiles each.
__global__ void k_sep_selinv_seed(int B, int me, int rnode, const double* __restrict__ Sd2, const double* __restrict__ So2,
                                  double* __restrict__ Sd, double* __restrict__ So) {
  const int BB = B * B;
  for (int e = threadIdx.x; e < BB; e += blockDim.x) {
    Sd[e] = Sd2[(size_t)(2 * me) * BB + e];
    Sd[(size_t)rnode * BB + e] = Sd2[(size_t)(2 * me + 1) * BB + e];
    So[e] = So2[(size_t)(2 * me) * BB + e];
  }
}

void launch_sep_selinv(hipStream_t st, const CrView& red, const double* rec_all, double* w2, double* x2, const SepGeom& gm, const CrView& chain,
                       double* Sd2, double* So2, double* Sd, double* So) {
  hipLaunchKernelGGL(k_sep_build, dim3(red.nsb), dim3(192), 0, st, red, rec_all, sep_rec_len(red.B), w2);
  launch_cr_factor_solve(st, red, w2, x2, 1, 0);
  launch_cr_selinv(st, red, Sd2, So2);
  hipLaunchKernelGGL(k_sep_selinv_seed, dim3(1), dim3(192), 0, st, red.B, gm.rank, chain.nsb - 1, Sd2, So2, Sd, So);
  launch_cr_selinv(st, chain, Sd, So);
}

void launch_sep_solve(hipStream_t st, const CrView& red, const double* rec_all, double* w2, double* x2, const SepGeom& gm, double* dt_local,
                      double* scale_local) {
  const int rl = sep_rec_len(red.B);
  hipLaunchKernelGGL(k_sep_build, dim3(red.nsb), dim3(192), 0, st, red, rec_all, rl, w2);
  launch_cr_factor_solve(st, red, w2, x2, 1, 0);
  hipLaunchKernelGGL(k_sep_scatter, dim3(1), dim3(64), 0, st, red.B, rec_all, rl, x2, gm, dt_local, scale_local);
}

// ===========================================================================================
// Selected inversion through the cyclic-reduction factors: the diagonal super-blocks of A^-1.
// A node k eliminated at stride s between survivors a = k - s and b = k + s satisfies
//     x_k = Dinv_k w_k - Gl_k x_a - Gr_k x_b ,
// so with Sigma = A^-1 and the blocks of the coarser level (Sigma_aa, Sigma_bb, Sigma_ab) known,
//     M1 = Gl Sigma_aa + Gr Sigma_ab^T        M2 = Gl Sigma_ab + Gr Sigma_bb
//     Sigma_kk = Dinv_k + M1 Gl^T + M2 Gr^T    Sigma_ak = -M1^T      Sigma_kb = -M2
// which are exactly the blocks the next finer level needs (a,k and k,b are adjacent there).
// Sd[node] = Sigma_node,node; So[node] = Sigma between node and its right neighbour at the level
// processed last.  One launch per level, one wave per eliminated node; ~log2(nsb) launches.
// ===========================================================================================
template <int B>
__global__ __launch_bounds__(WAVE) void k_cr_selinv_level(CrView cr, int s, double* Sd, double* So) {
  __shared__ double sGl[B * B], sGr[B * B], sAA[B * B], sBB[B * B], sAB[B * B], sM1[B * B], sM2[B * B];
  const int k = s * (2 * blockIdx.x + 1);
  if (k >= cr.nsb - (cr.rsep ? 1 : 0)) return;
  const size_t BB = (size_t)B * B;
  // (a chain whose last node is a separator: it is never eliminated, and it is the right neighbour of every node whose regular
  // one would lie beyond the last regular node -- the tree the fused stages of k_cr_chunk build with cr.rsep)
  const int a = k - s, b = (cr.rsep && k + s > cr.nsb - 2) ? cr.nsb - 1 : k + s;
  const bool has_b = b < cr.nsb;
  lds_load<B>(sGl, cr.Gl + (size_t)k * BB);
  lds_load<B>(sAA, Sd + (size_t)a * BB);
  if (has_b) { lds_load<B>(sGr, cr.Gr + (size_t)k * BB); lds_load<B>(sBB, Sd + (size_t)b * BB); lds_load<B>(sAB, So + (size_t)a * BB); }
  else for (int e = threadIdx.x; e < B * B; e += WAVE) { sGr[e] = 0.0; sBB[e] = 0.0; sAB[e] = 0.0; }
  __syncthreads();
  for (int e = threadIdx.x; e < B * B; e += WAVE) {
    const int r = e / B, c = e % B;
    double m1 = 0.0, m2 = 0.0;
#pragma unroll
    for (int q = 0; q < B; ++q) {
      m1 += sGl[r * B + q] * sAA[q * B + c] + sGr[r * B + q] * sAB[c * B + q];
      m2 += sGl[r * B + q] * sAB[q * B + c] + sGr[r * B + q] * sBB[q * B + c];
    }
    sM1[e] = m1; sM2[e] = m2;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < B * B; e += WAVE) {
    const int r = e / B, c = e % B;
    double v = cr.Dinv[(size_t)k * BB + e];
#pragma unroll
    for (int q = 0; q < B; ++q) v += sM1[r * B + q] * sGl[c * B + q] + sM2[r * B + q] * sGr[c * B + q];
    Sd[(size_t)k * BB + e] = v;
    So[(size_t)a * BB + e] = -sM1[c * B + r];
    if (has_b) So[(size_t)k * BB + e] = -sM2[e];
  }
}

template <int B>
static void cr_selinv_t(hipStream_t st, const CrView& cr, double* Sd, double* So) {
  // root; on a separator chain the caller has put the blocks of its two ends (Sd[0], Sd[nsb - 1], So[0]) there already
  if (!cr.rsep) (void)hipMemcpyAsync(Sd, cr.Dinv, sizeof(double) * B * B, hipMemcpyDeviceToDevice, st);
  const int nreg = cr.nsb - (cr.rsep ? 1 : 0);          // nodes 0 .. nreg - 1 take part in the elimination tree
  int s_last = 0;
  for (int s = 1; s < nreg; s *= 2) s_last = s;
  for (int s = s_last; s >= 1; s /= 2) {
    const int n_elim = (nreg - s + 2 * s - 1) / (2 * s);
    hipLaunchKernelGGL(k_cr_selinv_level<B>, dim3(n_elim), dim3(WAVE), 0, st, cr, s, Sd, So);
  }
}

void launch_cr_selinv(hipStream_t st, const CrView& cr, double* Sd, double* So) {
  switch (cr.B) {
    case 3: cr_selinv_t<3>(st, cr, Sd, So); break;
    case 6: cr_selinv_t<6>(st, cr, Sd, So); break;
    case 9: cr_selinv_t<9>(st, cr, Sd, So); break;
    default: cr_selinv_t<12>(st, cr, Sd, So); break;
  }
}

// pose i: its 3x3 block of S Sigma S, rotated into the robot frame as vector_mapping.cpp:2792-2801 does
// (transform = blockdiag(Rotation2Df(-angle).cast<double>(), 1); one_cov = transform * one_cov * transform^T)
__global__ void k_cov_extract(int n_poses, int Wb, int B, const double* __restrict__ Sd, const double* __restrict__ scale,
                              const unsigned char* __restrict__ fixed, const float2* __restrict__ rot, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_poses) return;
  double* o = out + 9 * (size_t)i;
  if (fixed[i]) { for (int e = 0; e < 9; ++e) o[e] = 0.0; return; }
  const int k = i / Wb, a0 = 3 * (i % Wb);
  const double* blk = Sd + (size_t)k * B * B;
  double S[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) S[r][c] = blk[(a0 + r) * B + a0 + c] * scale[3 * i + r] * scale[3 * i + c];
  const double c = (double)rot[i].x, sn = (double)rot[i].y;     // cos, sin of -angle (fp32 values)
  const double T[3][3] = {{c, -sn, 0.0}, {sn, c, 0.0}, {0.0, 0.0, 1.0}};
  double M[3][3];
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) M[r][q] = T[r][0] * S[0][q] + T[r][1] * S[1][q] + T[r][2] * S[2][q];
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) o[3 * r + q] = M[r][0] * T[q][0] + M[r][1] * T[q][1] + M[r][2] * T[q][2];
}

void launch_cov_extract(hipStream_t st, int n_poses, int Wb, int B, const double* Sd, const double* scale, const unsigned char* fixed,
                        const float2* rot, double* out) {
  hipLaunchKernelGGL(k_cov_extract, dim3((n_poses + 127) / 128), dim3(128), 0, st, n_poses, Wb, B, Sd, scale, fixed, rot, out);
}

}  // namespace hitl
