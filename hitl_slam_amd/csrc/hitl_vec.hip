// hitl_vec.hip -- band products and vector kernels around the cyclic-reduction solve: k_spmv (PCG product, residual
// check, trust-region bookkeeping), the submap interface helpers (unit right-hand sides, interface gather, combine,
// halo scatter), dots / axpby / copy for the PCG.  fp64.

#include <algorithm>

#include "hitl_device.h"

namespace hitl {

// ===========================================================================================
// y = (S H S [+ D^2]) v : band blocks (both triangles) + off-band blocks, one thread per pose.
// MODE 0: plain product (PCG).
// MODE 1: residual check after the cyclic-reduction solve: y -> r = rhs - A v, per-workgroup
//         partials (r.r, rhs.rhs) in slots 0,1 of the 8-wide record.
// MODE 2: trust-region step bookkeeping, fused: q = (S H S) v (no damping), partials
//         (v.rhs, v.q, sum (s v)^2 over free poses, sum x^2 over free poses, #non-finite) in
//         slots 2..6, and the candidate state xc = x + s*v.
// The host adds the few hundred per-workgroup partials in index order (bit-reproducible).
// ===========================================================================================
struct SpmvAux {
  const double* rhs;   // MODE 1, 2
  double* r_out;       // MODE 1
  const double* x;     // MODE 2
  double* xc;          // MODE 2
  double* partials;    // MODE 1, 2 : [gridDim.x][8]
  double2* cs;         // MODE 2, 3 (may be null): (cos, sin) of the candidate's theta, saves the k_pose_trig launch in front of k_linearize_stf
  int all_cols;        // MODE 0: columns of constant poses take part too (distributed PCG: the halo entries of v carry the
                       // neighbouring submaps' values; global pose 0 carries an exact 0)
  // device-driven LM loop (MODE 3, lm != null): lin / x are buffer 0, lin_alt / x_alt buffer 1; lm->cur picks the current
  // one, the candidate state goes to the other x buffer; mu = lm->radius
  const LmDev* lm;
  LinView lin_alt;
  double* x_alt;
};

// WB > 0: the band is WB poses wide (known at compile time): the two block loops are unrolled with clamped addresses, so that
// all of a row's block, scale and vector loads are in flight together instead of one loop iteration's at a time (a thread's
// chain of dependent memory round trips is what this kernel waits for: there is less than one wave per SIMD).
template <int MODE, int WB = 0>
__global__ __launch_bounds__(128) void k_spmv(int n_poses, int npad, FactorView fv, LinView lin,
                                               const double* __restrict__ scale, double mu, double min_diag,
                                               double max_diag, const double* __restrict__ v, double* __restrict__ y,
                                               SpmvAux aux) {
  if (MODE == 3 && aux.lm != nullptr) {          // uniform branch, scalar loads
    if (aux.lm->done) return;
    double* xb1 = aux.x_alt;
    double* xb0 = const_cast<double*>(aux.x);
    if (aux.lm->cur) { lin = aux.lin_alt; aux.x = xb1; aux.xc = xb0; } else { aux.xc = xb1; }
    mu = aux.lm->radius;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // MODE 3 = MODE 1 and MODE 2 in one pass (banded single-GPU solve: v is exactly 0 on constant poses, so including
  // their columns changes nothing): residual partials in slots 0,1 and the step bookkeeping in slots 2..6
  constexpr bool kRes = (MODE == 1 || MODE == 3), kStep = (MODE == 2 || MODE == 3);
  double part[7] = {0, 0, 0, 0, 0, 0, 0};
  if (3 * i < npad) {
    if (i >= n_poses || fv.fixed[i]) {  // padding rows and constant poses: identity rows
      for (int c = 0; c < 3; ++c) {
        const int t = 3 * i + c;
        if (t < npad) {
          if (MODE == 0) y[t] = v[t];
          if (kRes) { const double r = aux.rhs[t] - v[t]; if (MODE == 1) aux.r_out[t] = r; part[0] += r * r; part[1] += aux.rhs[t] * aux.rhs[t]; }
          if (kStep) {
            y[t] = v[t];
            if (i < n_poses) {
              aux.xc[t] = aux.x[t];
              if (kStep && c == 2 && aux.cs != nullptr) { double sn, cn; sincos(aux.x[t], &sn, &cn); aux.cs[i] = make_double2(cn, sn); }
            }
          }
        }
      }
    } else {
      // Columns of constant poses are not part of the linear system (MODE 0/1).  In MODE 2 they
      // ARE included: v then carries the neighbouring submap's step on the halo poses (0 for pose 0),
      // so owned rows see the cross-cut coupling in the model cost change.
      const int Wb = (WB > 0) ? WB : fv.band_poses, W1 = Wb + 1;
      double acc[3] = {0, 0, 0};
      const double* Hi = lin.Hband + (size_t)i * W1 * 9;
      if (WB > 0 && kStep) {
        // same blocks, same order, same sums as the loops below (kStep: no column is skipped)
        double bl[2 * WB + 1][9], vv[2 * WB + 1][3];
        bool ok[2 * WB + 1];
#pragma unroll
        for (int k = 0; k <= 2 * WB; ++k) {
          const int d = (k <= WB) ? k : k - WB;               // k <= WB: block (i, i + d);  k > WB: block (i - d, i), used transposed
          const int j = (k <= WB) ? i + d : i - d;
          ok[k] = (k <= WB) ? (j < n_poses) : (j >= 0);
          const int jc = ok[k] ? j : i;
          const double* blk = (k <= WB) ? Hi + d * 9 : lin.Hband + ((size_t)jc * W1 + d) * 9;
#pragma unroll
          for (int e = 0; e < 9; ++e) bl[k][e] = blk[e];
#pragma unroll
          for (int e = 0; e < 3; ++e) vv[k][e] = scale[3 * jc + e] * v[3 * jc + e];
        }
#pragma unroll
        for (int k = 0; k <= 2 * WB; ++k) {
          if (!ok[k]) continue;
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            if (k <= WB) acc[a] += bl[k][a * 3] * vv[k][0] + bl[k][a * 3 + 1] * vv[k][1] + bl[k][a * 3 + 2] * vv[k][2];
            else acc[a] += bl[k][a] * vv[k][0] + bl[k][3 + a] * vv[k][1] + bl[k][6 + a] * vv[k][2];
          }
        }
      } else {
      for (int d = 0; d <= Wb && i + d < n_poses; ++d) {
        const int j = i + d;
        if (!kStep && !aux.all_cols && fv.fixed[j]) continue;
        const double* blk = Hi + d * 9;
        const double v0 = scale[3 * j] * v[3 * j], v1 = scale[3 * j + 1] * v[3 * j + 1], v2 = scale[3 * j + 2] * v[3 * j + 2];
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[a] += blk[a * 3] * v0 + blk[a * 3 + 1] * v1 + blk[a * 3 + 2] * v2;
      }
      for (int d = 1; d <= Wb && i - d >= 0; ++d) {
        const int j = i - d;
        if (!kStep && !aux.all_cols && fv.fixed[j]) continue;
        const double* blk = lin.Hband + ((size_t)j * W1 + d) * 9;  // block (j, i), used transposed
        const double v0 = scale[3 * j] * v[3 * j], v1 = scale[3 * j + 1] * v[3 * j + 1], v2 = scale[3 * j + 2] * v[3 * j + 2];
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[a] += blk[a] * v0 + blk[3 + a] * v1 + blk[6 + a] * v2;
      }
      }
      if (fv.with_stf && fv.n_offband > 0 && fv.ob_acc != nullptr) {
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[a] += fv.ob_acc[3 * i + a];        // summed by k_spmv_offband in front of this launch
      } else if (fv.with_stf && fv.n_offband > 0) {
        for (int e = fv.obadj_off[i]; e < fv.obadj_off[i + 1]; ++e) {
          const int ent = fv.obadj_ent[e];
          const int en = ent >> 1, tr = ent & 1;
          const double* blk = lin.obblk + (size_t)en * 9;
          const int j = tr ? fv.ob_lo[en] : fv.ob_hi[en];
          if (!kStep && !aux.all_cols && fv.fixed[j]) continue;
          const double v0 = scale[3 * j] * v[3 * j], v1 = scale[3 * j + 1] * v[3 * j + 1], v2 = scale[3 * j + 2] * v[3 * j + 2];
          if (!tr) {
            for (int a = 0; a < 3; ++a) acc[a] += blk[a * 3] * v0 + blk[a * 3 + 1] * v1 + blk[a * 3 + 2] * v2;
          } else {
            for (int a = 0; a < 3; ++a) acc[a] += blk[a] * v0 + blk[3 + a] * v1 + blk[6 + a] * v2;
          }
        }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int t = 3 * i + a;
        const double sa = scale[t];
        const double rq = sa * acc[a];              // (S H S v)[t]
        double r = rq;
        if (MODE != 2 && mu > 0.0) {
          const double dg = sa * sa * Hi[a * 3 + a];
          r += fmin(fmax(dg, min_diag), max_diag) / mu * v[t];
        }
        if (MODE == 0) y[t] = r;
        if (kRes) {
          const double b = aux.rhs[t], res = b - r;
          if (MODE == 1) aux.r_out[t] = res;
          part[0] += res * res; part[1] += b * b;
        }
        if (kStep) {
          constexpr int o = (MODE == 3) ? 2 : 0;
          const double vt = v[t];
          const double dl = sa * vt, xt = aux.x[t];
          y[t] = rq;
          part[o + 0] += vt * aux.rhs[t];
          part[o + 1] += vt * rq;
          part[o + 2] += dl * dl; part[o + 3] += xt * xt;
          aux.xc[t] = xt + dl;
          if (kStep && a == 2 && aux.cs != nullptr) { double sn, cn; sincos(xt + dl, &sn, &cn); aux.cs[i] = make_double2(cn, sn); }
          if (!isfinite(vt)) part[o + 4] += 1.0;
        }
      }
    }
  }
  if (MODE != 0) {
    __shared__ double sm[2][7];
    const int wv = threadIdx.x >> 6;
    constexpr int NP = (MODE == 1) ? 2 : (MODE == 2 ? 5 : 7);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const double t = wave_sum(part[k]);
      if ((threadIdx.x & 63) == 0) sm[wv][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < NP) aux.partials[(size_t)blockIdx.x * 8 + (MODE == 2 ? 2 : 0) + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x];
  }
}

// Off-band part of the product for maps whose poses see many far-away poses (all-pairs search on a corridor: ~90 blocks
// per pose): one wave per pose, lanes stride over the pose's adjacency entries, fixed-order butterfly sum.  ALLCOLS: columns
// of constant poses take part (the step bookkeeping of MODE 2/3).
template <bool ALLCOLS>
__global__ __launch_bounds__(256) void k_spmv_offband(int n_poses, FactorView fv, LinView lin, const double* __restrict__ scale,
                                                       const double* __restrict__ v, double* __restrict__ ob) {
  const int i = (blockIdx.x * 256 + (int)threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= n_poses || fv.fixed[i]) return;        // wave-uniform
  double acc[3] = {0, 0, 0};
  for (int e = fv.obadj_off[i] + lane; e < fv.obadj_off[i + 1]; e += WAVE) {
    const int ent = fv.obadj_ent[e], j = fv.obadj_col[e];
    if (!ALLCOLS && fv.fixed[j]) continue;
    const double* blk = lin.obblk + (size_t)(ent >> 1) * 9;
    const double v0 = scale[3 * j] * v[3 * j], v1 = scale[3 * j + 1] * v[3 * j + 1], v2 = scale[3 * j + 2] * v[3 * j + 2];
    if (!(ent & 1)) {
#pragma unroll
      for (int a = 0; a < 3; ++a) acc[a] += blk[a * 3] * v0 + blk[a * 3 + 1] * v1 + blk[a * 3 + 2] * v2;
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a) acc[a] += blk[a] * v0 + blk[3 + a] * v1 + blk[6 + a] * v2;
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double t = wave_sum(acc[a]);
    if (lane == 0) ob[3 * i + a] = t;
  }
}

static void spmv_offband(hipStream_t st, int n_poses, const FactorView& fv, const LinView& lin, const double* scale, const double* v,
                         bool allcols) {
  if (!(fv.with_stf && fv.n_offband > 0 && fv.ob_acc != nullptr)) return;
  const dim3 grid((n_poses + 3) / 4), block(256);
  if (allcols) hipLaunchKernelGGL(k_spmv_offband<true>, grid, block, 0, st, n_poses, fv, lin, scale, v, fv.ob_acc);
  else hipLaunchKernelGGL(k_spmv_offband<false>, grid, block, 0, st, n_poses, fv, lin, scale, v, fv.ob_acc);
}

int spmv_blocks(int npad) { return ((npad + 2) / 3 + 127) / 128; }

void launch_spmv(hipStream_t st, int n_poses, int npad, const FactorView& fv, const LinView& lin, const double* scale,
                 double mu, double min_diag, double max_diag, const double* v, double* y, bool all_cols) {
  SpmvAux aux = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, all_cols ? 1 : 0};
  spmv_offband(st, n_poses, fv, lin, scale, v, all_cols);
  hipLaunchKernelGGL(k_spmv<0>, dim3(spmv_blocks(npad)), dim3(128), 0, st, n_poses, npad, fv, lin, scale, mu, min_diag,
                     max_diag, v, y, aux);
}

void launch_spmv_residual(hipStream_t st, int n_poses, int npad, const FactorView& fv, const LinView& lin,
                          const double* scale, double mu, double min_diag, double max_diag, const double* v,
                          const double* rhs, double* r_out, double* partials) {
  SpmvAux aux = {rhs, r_out, nullptr, nullptr, partials, nullptr, 0};
  spmv_offband(st, n_poses, fv, lin, scale, v, false);
  hipLaunchKernelGGL(k_spmv<1>, dim3(spmv_blocks(npad)), dim3(128), 0, st, n_poses, npad, fv, lin, scale, mu, min_diag,
                     max_diag, v, (double*)nullptr, aux);
}

void launch_spmv_residual_step(hipStream_t st, int n_poses, int npad, const FactorView& fv, const LinView& lin, const double* scale, double mu,
                               double min_diag, double max_diag, const double* v, const double* rhs, const double* x, double* q, double* xc,
                               double* partials, double2* cs) {
  SpmvAux aux = {rhs, nullptr, x, xc, partials, cs, 0};
  spmv_offband(st, n_poses, fv, lin, scale, v, true);
  if (fv.band_poses == 3) hipLaunchKernelGGL((k_spmv<3, 3>), dim3(spmv_blocks(npad)), dim3(128), 0, st, n_poses, npad, fv, lin, scale, mu, min_diag, max_diag, v, q, aux);
  else hipLaunchKernelGGL((k_spmv<3, 0>), dim3(spmv_blocks(npad)), dim3(128), 0, st, n_poses, npad, fv, lin, scale, mu, min_diag, max_diag, v, q, aux);
}

void launch_spmv_residual_step_lm(hipStream_t st, int n_poses, int npad, const FactorView& fv, const LinView& lin0, const LinView& lin1,
                                  const double* scale, double min_diag, double max_diag, const double* v, const double* rhs, double* x0,
                                  double* x1, double* q, double* partials, double2* cs, const LmDev* lm) {
  SpmvAux aux = {rhs, nullptr, x0, nullptr, partials, cs, 0, lm, lin1, x1};
  if (fv.band_poses == 3) hipLaunchKernelGGL((k_spmv<3, 3>), dim3(spmv_blocks(npad)), dim3(128), 0, st, n_poses, npad, fv, lin0, scale, 1.0, min_diag, max_diag, v, q, aux);
  else hipLaunchKernelGGL((k_spmv<3, 0>), dim3(spmv_blocks(npad)), dim3(128), 0, st, n_poses, npad, fv, lin0, scale, 1.0, min_diag, max_diag, v, q, aux);
}

// The scalars a submap rank contributes to the exchange that ends an LM iteration -- (dt.rhs, dt.q, |step|^2, |x|^2, bad,
// cost, max |g| of the candidate) -- summed from the per-workgroup partials ON THE DEVICE, in the order of ordered_sum32 (the
// host's), so that with the built-in RCCL transport the record can travel device -> device and the host waits once per
// iteration instead of twice (round 4).  lp == null: no linearisation was queued (cost, max |g| = 0).
__global__ __launch_bounds__(512) void k_sum_partials(const double* __restrict__ sp, int nblk_step, const double* __restrict__ lp, int nblk_lin,
                                                      double* __restrict__ rec7) {
  __shared__ double part[7][32];
  const int tid = threadIdx.x, grp = tid >> 5, j = tid & 31;
  if (grp < 7) {
    double acc = 0.0;
    if (grp < 5) { for (int b = j; b < nblk_step; b += 32) acc += sp[8 * b + 2 + grp]; }
    else if (lp != nullptr && grp == 5) { for (int b = j; b < nblk_lin; b += 32) acc += lp[2 * b]; }
    else if (lp != nullptr) { for (int b = j; b < nblk_lin; b += 32) { const double g = lp[2 * b + 1]; if (g > acc || g != g) acc = g; } }
    part[grp][j] = acc;
  }
  __syncthreads();
  if (tid < 7) {
    double t = 0.0;
    if (tid != 6) { for (int q = 0; q < 32; ++q) t += part[tid][q]; }
    else { for (int q = 0; q < 32; ++q) { const double g = part[6][q]; if (g > t || g != g) t = g; } }
    rec7[tid] = t;
  }
}

void launch_sum_partials(hipStream_t st, const double* step_partials, int nblk_step, const double* lin_partials, int nblk_lin, double* rec7) {
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(512), 0, st, step_partials, nblk_step, lin_partials, nblk_lin, rec7);
}

void launch_spmv_step(hipStream_t st, int n_poses, int npad, const FactorView& fv, const LinView& lin,
                      const double* scale, const double* dt, const double* rhs, const double* x, double* q, double* xc,
                      double* partials, double2* cs) {
  SpmvAux aux = {rhs, nullptr, x, xc, partials, cs, 0};
  spmv_offband(st, n_poses, fv, lin, scale, dt, true);
  hipLaunchKernelGGL(k_spmv<2>, dim3(spmv_blocks(npad)), dim3(128), 0, st, n_poses, npad, fv, lin, scale, 0.0, 0.0, 0.0, dt,
                     q, aux);
}

// ===========================================================================================
// Pose covariances of a map with pose pairs OUTSIDE the band (round 6; SURVEY.md 8(f) row 4 on the maps the reference's all-pairs
// search produces): the marginal 3 x 3 blocks of (J^T J)^-1 do not fall out of the band's cyclic reduction there.  The scaled
// matrix S J^T J S is written out densely (3 N x 3 N doubles: 72 MB for 1 000 poses), factorised and inverted by the dense Cholesky
// of rocSOLVER (a plain library operation, bound at run time: hitl_solver.cpp), and the diagonal blocks are unscaled and rotated
// like the banded path's (k_cov_extract).  Constant poses: identity rows and columns, zero covariance.
// ===========================================================================================
__global__ void k_dense_fill_band(int n, int Wb, int64_t d, const double* __restrict__ Hband, const double* __restrict__ scale,
                                  const unsigned char* __restrict__ fixed, double* __restrict__ A) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (pose i, offset dd, entry q)
  if (t >= (int64_t)n * (Wb + 1) * 9) return;
  const int q = (int)(t % 9), dd = (int)((t / 9) % (Wb + 1)), i = (int)(t / (9 * (Wb + 1)));
  const int j = i + dd;
  if (j >= n) return;
  const int a = q / 3, c = q % 3;
  const int64_t r = 3 * (int64_t)i + a, col = 3 * (int64_t)j + c;
  double v = Hband[((size_t)i * (Wb + 1) + dd) * 9 + q] * scale[r] * scale[col];
  if (fixed[i] || fixed[j]) v = (dd == 0 && a == c) ? 1.0 : 0.0;
  A[r + col * d] = v;
  if (dd > 0) A[col + r * d] = v;                                          // the mirrored block (dd == 0: the diagonal block is stored whole)
}
__global__ void k_dense_fill_offband(int E, int64_t d, const int* __restrict__ ob_lo, const int* __restrict__ ob_hi, const double* __restrict__ obblk,
                                     const double* __restrict__ scale, const unsigned char* __restrict__ fixed, double* __restrict__ A) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)E * 9) return;
  const int q = (int)(t % 9), e = (int)(t / 9);
  const int i = ob_lo[e], j = ob_hi[e];
  if (fixed[i] || fixed[j]) return;                                        // (already zero)
  const int a = q / 3, c = q % 3;
  const int64_t r = 3 * (int64_t)i + a, col = 3 * (int64_t)j + c;
  const double v = obblk[(size_t)e * 9 + q] * scale[r] * scale[col];
  A[r + col * d] = v;
  A[col + r * d] = v;
}
// the inverse's lower triangle (column-major, what potri leaves) -> one rotated 3 x 3 block per pose
__global__ void k_dense_cov_extract(int n, int64_t d, const double* __restrict__ Ainv, const double* __restrict__ scale,
                                    const unsigned char* __restrict__ fixed, const float2* __restrict__ rot, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double* o = out + 9 * (size_t)i;
  if (fixed[i]) { for (int e = 0; e < 9; ++e) o[e] = 0.0; return; }
  double S[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      const int64_t rr = 3 * (int64_t)i + (r >= c ? r : c), cc = 3 * (int64_t)i + (r >= c ? c : r);      // lower triangle: row >= column
      S[r][c] = Ainv[rr + cc * d] * scale[3 * i + r] * scale[3 * i + c];
    }
  const double c = (double)rot[i].x, sn = (double)rot[i].y;     // cos, sin of -angle (fp32 values)
  const double T[3][3] = {{c, -sn, 0.0}, {sn, c, 0.0}, {0.0, 0.0, 1.0}};
  double M[3][3];
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) M[r][q] = T[r][0] * S[0][q] + T[r][1] * S[1][q] + T[r][2] * S[2][q];
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) o[3 * r + q] = M[r][0] * T[q][0] + M[r][1] * T[q][1] + M[r][2] * T[q][2];
}
void launch_dense_fill(hipStream_t st, int n, int Wb, const double* Hband, int n_offband, const int* ob_lo, const int* ob_hi, const double* obblk,
                       const double* scale, const unsigned char* fixed, double* A) {
  const int64_t d = 3 * (int64_t)n;
  const int64_t tb = (int64_t)n * (Wb + 1) * 9;
  hipLaunchKernelGGL(k_dense_fill_band, dim3((unsigned)((tb + 255) / 256)), dim3(256), 0, st, n, Wb, d, Hband, scale, fixed, A);
  if (n_offband > 0)
    hipLaunchKernelGGL(k_dense_fill_offband, dim3((unsigned)(((int64_t)n_offband * 9 + 255) / 256)), dim3(256), 0, st, n_offband, d, ob_lo, ob_hi, obblk, scale, fixed, A);
}
void launch_dense_cov_extract(hipStream_t st, int n, const double* Ainv, const double* scale, const unsigned char* fixed, const float2* rot, double* out) {
  hipLaunchKernelGGL(k_dense_cov_extract, dim3((n + 127) / 128), dim3(128), 0, st, n, 3 * (int64_t)n, Ainv, scale, fixed, rot, out);
}

// ===========================================================================================
// small vector kernels
// ===========================================================================================
// submap exchange helpers --------------------------------------------------------------------
// unit right-hand sides for the interface solve: w[r][*] = 0 except w[r][cols[r]] = 1  (r < ncols)
__global__ void k_unit_rhs(int npad, double* __restrict__ w, size_t stride, const int* __restrict__ cols) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npad) return;
  const int r = blockIdx.y;
  w[stride * r + t] = (t == cols[r]) ? 1.0 : 0.0;
}

void launch_unit_rhs(hipStream_t st, int npad, double* w, size_t stride, const int* cols_dev, int ncols) {
  hipLaunchKernelGGL(k_unit_rhs, dim3((npad + 255) / 256, ncols), dim3(256), 0, st, npad, w, stride, cols_dev);
}

// interface rows of z0, of the ncols unit-rhs solutions and of the Jacobi scale:
// out[0..nc) = z0[cols], out[nc + nc*c + r] = Z[c][cols[r]], out[nc + nc*nc + r] = scale[cols[r]]
__global__ void k_gather_iface(const double* __restrict__ z0, const double* __restrict__ Z, size_t stride,
                               const double* __restrict__ scale, const int* __restrict__ cols, int nc,
                               double* __restrict__ out) {
  for (int t = threadIdx.x; t < nc * (nc + 2); t += blockDim.x) {
    if (t < nc) out[t] = z0[cols[t]];
    else if (t < nc + nc * nc) { const int c = (t - nc) / nc, r = (t - nc) % nc; out[t] = Z[stride * c + cols[r]]; }
    else out[t] = scale[cols[t - nc - nc * nc]];
  }
}

void launch_gather_iface(hipStream_t st, const double* z0, const double* Z, size_t stride, const double* scale,
                         const int* cols_dev, int ncols, double* out) {
  hipLaunchKernelGGL(k_gather_iface, dim3(1), dim3(256), 0, st, z0, Z, stride, scale, cols_dev, ncols, out);
}

// cross-cut Hessian blocks of a submap rank, cut from the band where they were assembled (block (i, i+d) lives at
// Hband[i][d]), into the exchange record:  Clo[3ap+r][3bp+c] = block(own_lo + ap, own_lo - Wi + bp)  (transposed band
// block),  Cup[3ap+r][3bp+c] = block(own_hi - Wi + ap, own_hi + bp);  zero beyond the band or at the ends of the map
__global__ void k_iface_blocks(const double* __restrict__ Hband, int Wb, int Wi, int own_lo, int own_hi, int has_lo, int has_up,
                               double* __restrict__ Clo, double* __restrict__ Cup) {
  const int m = 3 * Wi, W1 = Wb + 1;
  for (int t = threadIdx.x; t < 2 * m * m; t += blockDim.x) {
    const int up = t / (m * m), e = t % (m * m);
    const int row = e / m, col = e % m, ap = row / 3, r = row % 3, bp = col / 3, c = col % 3;
    double v = 0.0;
    if (!up && has_lo) {
      const int d = (Wi + ap) - bp;                       // a = own_lo + ap, b = own_lo - Wi + bp
      if (d <= Wb) v = Hband[((size_t)(own_lo - Wi + bp) * W1 + d) * 9 + c * 3 + r];
    } else if (up && has_up) {
      const int d = (Wi + bp) - ap;                       // a = own_hi - Wi + ap, b = own_hi + bp
      if (d <= Wb) v = Hband[((size_t)(own_hi - Wi + ap) * W1 + d) * 9 + r * 3 + c];
    }
    (up ? Cup : Clo)[e] = v;
  }
}

void launch_iface_blocks(hipStream_t st, const double* Hband, int Wb, int Wi, int own_lo, int own_hi, bool has_lo, bool has_up,
                         double* Clo, double* Cup) {
  hipLaunchKernelGGL(k_iface_blocks, dim3(1), dim3(256), 0, st, Hband, Wb, Wi, own_lo, own_hi, has_lo ? 1 : 0, has_up ? 1 : 0, Clo, Cup);
}

// Pose covariances on a submap rank: the diagonal blocks of the GLOBAL inverse are those of the rank's own inverse minus
// a rank-2m correction through its interface columns,  Sigma_ii = (M^-1)_ii - Z_i T Z_i^T  (Z_i: the three rows of pose i of
// Z = M^-1 [E_F, E_L], T: nc x nc from the interface system, hitl_lm.cpp).  One thread per owned pose; Sd holds the B x B
// tiles of the selected inversion, pose i sits at rows / columns 3 (i mod Wb) of tile i / Wb.
__global__ void k_cov_iface_correction(int own_lo, int own_hi, int Wb, int B, const double* __restrict__ Z, size_t stride,
                                       const double* __restrict__ T, int nc, double* __restrict__ Sd) {
  const int i = own_lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= own_hi) return;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = 0; c < nc; ++c) {
    double t[3] = {0, 0, 0};                                  // (Z_i T)[:, c]
    for (int a = 0; a < nc; ++a) {
      const double tac = T[(size_t)a * nc + c];
      for (int r = 0; r < 3; ++r) t[r] += Z[stride * a + 3 * i + r] * tac;
    }
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) acc[r * 3 + q] += t[r] * Z[stride * c + 3 * i + q];
  }
  double* tile = Sd + (size_t)(i / Wb) * B * B;
  const int o = 3 * (i % Wb);
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) tile[(o + r) * B + o + q] -= acc[r * 3 + q];
}

void launch_cov_iface_correction(hipStream_t st, int own_lo, int own_hi, int Wb, int B, const double* Z, size_t stride, const double* T_dev,
                                 int nc, double* Sd) {
  const int cnt = own_hi - own_lo;
  if (cnt <= 0) return;
  hipLaunchKernelGGL(k_cov_iface_correction, dim3((cnt + 127) / 128), dim3(128), 0, st, own_lo, own_hi, Wb, B, Z, stride, T_dev, nc, Sd);
}

// dt = z0 - sum_k coef[k] * Z[k]   (Z: ncols columns, `stride` apart; coef in device memory)
__global__ void k_combine(int npad, const double* __restrict__ z0, const double* __restrict__ Z, size_t stride,
                          const double* __restrict__ coef, int ncols, double* __restrict__ dt) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npad) return;
  double v = z0[t];
  for (int k = 0; k < ncols; ++k) v -= coef[k] * Z[stride * k + t];
  dt[t] = v;
}

void launch_combine(hipStream_t st, int npad, const double* z0, const double* Z, size_t stride, const double* coef_dev,
                    int ncols, double* dt) {
  hipLaunchKernelGGL(k_combine, dim3((npad + 255) / 256), dim3(256), 0, st, npad, z0, Z, stride, coef_dev, ncols, dt);
}

// halo poses (constant in the local system) take the neighbouring submap's step:
// dt[idx[k]] = val[k], scale[idx[k]] = sc[k]  (k < cnt)
__global__ void k_scatter_halo(int cnt, const int* __restrict__ idx, const double* __restrict__ val,
                               const double* __restrict__ sc, double* __restrict__ dt, double* __restrict__ scale) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  dt[idx[k]] = val[k];
  scale[idx[k]] = sc[k];
}

void launch_scatter_halo(hipStream_t st, int cnt, const int* idx, const double* val, const double* sc, double* dt,
                         double* scale) {
  if (cnt <= 0) return;
  hipLaunchKernelGGL(k_scatter_halo, dim3((cnt + 63) / 64), dim3(64), 0, st, cnt, idx, val, sc, dt, scale);
}

// distributed PCG: one exchange record = `nhead` scalars followed by the owned part of a vector, zero-padded to the widest rank
__global__ void k_pack_owned(int nhead, const double* __restrict__ head, const double* __restrict__ vec, int off, int cnt, int maxcnt,
                             double* __restrict__ rec) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nhead) rec[t] = head[t];
  if (t < maxcnt) rec[nhead + t] = (t < cnt) ? vec[off + t] : 0.0;
}

// the same record with only the two ends of the owned range: entries [0, edge) and [cnt - edge, cnt) of the owned part (all of
// it when cnt <= maxcnt = 2 * edge), zero-padded to maxcnt
__global__ void k_pack_edges(int nhead, const double* __restrict__ head, const double* __restrict__ vec, int off, int cnt, int edge, int maxcnt,
                             double* __restrict__ rec) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nhead) rec[t] = head[t];
  if (t < maxcnt) {
    double v = 0.0;
    if (cnt <= maxcnt) { if (t < cnt) v = vec[off + t]; }
    else v = (t < edge) ? vec[off + t] : vec[off + cnt - edge + (t - edge)];
    rec[nhead + t] = v;
  }
}

void launch_pack_edges(hipStream_t st, int nhead, const double* head, const double* vec, int off, int cnt, int edge, int maxcnt, double* rec) {
  const int tot = std::max(maxcnt, nhead);
  hipLaunchKernelGGL(k_pack_edges, dim3((tot + 255) / 256), dim3(256), 0, st, nhead, head, vec, off, cnt, edge, maxcnt, rec);
}

void launch_pack_owned(hipStream_t st, int nhead, const double* head, const double* vec, int off, int cnt, int maxcnt, double* rec) {
  const int tot = std::max(maxcnt, nhead);
  hipLaunchKernelGGL(k_pack_owned, dim3((tot + 255) / 256), dim3(256), 0, st, nhead, head, vec, off, cnt, maxcnt, rec);
}

// vec[idx[k]] = val[k]
__global__ void k_scatter_idx(int cnt, const int* __restrict__ idx, const double* __restrict__ val, double* __restrict__ vec) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < cnt) vec[idx[k]] = val[k];
}

void launch_scatter_idx(hipStream_t st, int cnt, const int* idx, const double* val, double* vec) {
  if (cnt <= 0) return;
  hipLaunchKernelGGL(k_scatter_idx, dim3((cnt + 255) / 256), dim3(256), 0, st, cnt, idx, val, vec);
}

// candidate state of the halo poses: xc = x + scale * dt  (k_spmv<2> leaves xc = x on constant poses)
__global__ void k_halo_candidate(int cnt, const int* __restrict__ idx, const double* __restrict__ x,
                                 const double* __restrict__ scale, const double* __restrict__ dt, double* __restrict__ xc, double2* __restrict__ cs) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cnt) return;
  const int t = idx[k];
  const double v = x[t] + scale[t] * dt[t];
  xc[t] = v;
  if (cs != nullptr && t % 3 == 2) { double sn, cn; sincos(v, &sn, &cn); cs[t / 3] = make_double2(cn, sn); }   // the pose's theta: trig table of the candidate
}

void launch_halo_candidate(hipStream_t st, int cnt, const int* idx, const double* x, const double* scale, const double* dt,
                           double* xc, double2* cs) {
  if (cnt <= 0) return;
  hipLaunchKernelGGL(k_halo_candidate, dim3((cnt + 63) / 64), dim3(64), 0, st, cnt, idx, x, scale, dt, xc, cs);
}

struct DotArgs { const double* a[4]; const double* b[4]; };

__global__ __launch_bounds__(kRedBlock) void k_dots(int n, int ndots, DotArgs args, double* out) {
  __shared__ double sm[kRedBlock / WAVE];
  for (int k = 0; k < ndots; ++k) {
    double s = 0.0;
    const double* a = args.a[k];
    const double* b = args.b[k];
    for (int t = threadIdx.x; t < n; t += kRedBlock) s += a[t] * b[t];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) out[k] = s;
  }
}

void launch_dots(hipStream_t st, int n, int ndots, const double* const* a, const double* const* b, double* out) {
  DotArgs args;
  for (int k = 0; k < 4; ++k) { args.a[k] = a[k < ndots ? k : 0]; args.b[k] = b[k < ndots ? k : 0]; }
  hipLaunchKernelGGL(k_dots, dim3(1), dim3(kRedBlock), 0, st, n, ndots, args, out);
}

// ---- PCG updates whose scalars stay on the device (no host round trip inside a CG iteration) ----
// slots of the scalar block s[]: 0 = r.z, 1 = previous r.z, 2 = p.q, 3 = r.r, 4 = breakdown flag
// p = z + (rz / rz_old) p      (first: p = z)
__global__ void k_pcg_update_p(int n, const double* __restrict__ z, double* __restrict__ p, const double* __restrict__ s, int first) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const double beta = (first || s[1] == 0.0) ? 0.0 : s[0] / s[1];
  p[t] = z[t] + (first ? 0.0 : beta * p[t]);
}

// alpha = rz / pq;  x += alpha p;  r -= alpha q;  rz_old <- rz   (pq <= 0 or NaN: breakdown flag, no update)
__global__ void k_pcg_update_xr(int n, const double* __restrict__ p, const double* __restrict__ q, double* __restrict__ x, double* __restrict__ r,
                                double* __restrict__ s) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const double rz = s[0], pq = s[2];
  const bool ok = pq > 0.0;
  const double alpha = ok ? rz / pq : 0.0;
  if (t < n) { x[t] += alpha * p[t]; r[t] -= alpha * q[t]; }
  // s[1] is read by the NEXT k_pcg_update_p only.  p.q <= 0 with r.z != 0 is a breakdown (the matrix is SPD); with
  // r.z == 0 the residual is exactly zero (an iteration past convergence inside a batch) and nothing is left to do.
  if (t == 0) { s[1] = rz; if (!ok && rz != 0.0) s[4] = 1.0; }
}

void launch_pcg_update_p(hipStream_t st, int n, const double* z, double* p, const double* s, bool first) {
  hipLaunchKernelGGL(k_pcg_update_p, dim3((n + 255) / 256), dim3(256), 0, st, n, z, p, s, first ? 1 : 0);
}

void launch_pcg_update_xr(hipStream_t st, int n, const double* p, const double* q, double* x, double* r, double* s) {
  hipLaunchKernelGGL(k_pcg_update_xr, dim3((n + 255) / 256), dim3(256), 0, st, n, p, q, x, r, s);
}

// one workgroup: s[0] = r.z, then p = z + (r.z / previous r.z) p          (k_dots + k_pcg_update_p in one launch)
__global__ __launch_bounds__(kRedBlock) void k_pcg_dir(int n, const double* __restrict__ r, const double* __restrict__ z, double* __restrict__ p,
                                                        double* __restrict__ s, int first) {
  __shared__ double sm[kRedBlock / WAVE];
  double acc = 0.0;
  for (int t = threadIdx.x; t < n; t += kRedBlock) acc += r[t] * z[t];
  const double rz = block_sum(acc, sm);
  const double rz_old = s[1];
  const double beta = (first || rz_old == 0.0) ? 0.0 : rz / rz_old;
  for (int t = threadIdx.x; t < n; t += kRedBlock) p[t] = z[t] + (first ? 0.0 : beta * p[t]);
  if (threadIdx.x == 0) s[0] = rz;
}

// one workgroup: s[2] = p.q, alpha = r.z / p.q, x += alpha p, r -= alpha q, s[3] = r.r, s[1] <- r.z   (k_dots, k_pcg_update_xr, k_dots)
__global__ __launch_bounds__(kRedBlock) void k_pcg_step(int n, const double* __restrict__ p, const double* __restrict__ q, double* __restrict__ x,
                                                         double* __restrict__ r, double* __restrict__ s) {
  __shared__ double sm[kRedBlock / WAVE];
  double acc = 0.0;
  for (int t = threadIdx.x; t < n; t += kRedBlock) acc += p[t] * q[t];
  const double pq = block_sum(acc, sm);
  const double rz = s[0];
  const bool ok = pq > 0.0;
  const double alpha = ok ? rz / pq : 0.0;
  acc = 0.0;
  for (int t = threadIdx.x; t < n; t += kRedBlock) {
    x[t] += alpha * p[t];
    const double rt = r[t] - alpha * q[t];
    r[t] = rt;
    acc += rt * rt;
  }
  const double rr = block_sum(acc, sm);
  if (threadIdx.x == 0) { s[1] = rz; s[2] = pq; s[3] = rr; if (!ok && rz != 0.0) s[4] = 1.0; }
}

void launch_pcg_dir(hipStream_t st, int n, const double* r, const double* z, double* p, double* s, bool first) {
  hipLaunchKernelGGL(k_pcg_dir, dim3(1), dim3(kRedBlock), 0, st, n, r, z, p, s, first ? 1 : 0);
}

void launch_pcg_step(hipStream_t st, int n, const double* p, const double* q, double* x, double* r, double* s) {
  hipLaunchKernelGGL(k_pcg_step, dim3(1), dim3(kRedBlock), 0, st, n, p, q, x, r, s);
}

__global__ void k_axpby(int n, double alpha, const double* __restrict__ x, double beta, double* __restrict__ y) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  y[t] = alpha * x[t] + ((beta != 0.0) ? beta * y[t] : 0.0);
}

void launch_axpby(hipStream_t st, int n, double alpha, const double* x, double beta, double* y) {
  hipLaunchKernelGGL(k_axpby, dim3((n + 255) / 256), dim3(256), 0, st, n, alpha, x, beta, y);
}

// one store of a sequence number into pinned host memory, behind everything queued before it: the host polls the word
// instead of an event (hipEventRecord + hipEventQuery cost the LM loop several microseconds per turn)
__global__ void k_host_flag(volatile double* flag, double seq) { *flag = seq; }
void launch_host_flag(hipStream_t st, double* flag_pinned, double seq) {
  hipLaunchKernelGGL(k_host_flag, dim3(1), dim3(1), 0, st, flag_pinned, seq);
}

// the copy the measured ceiling of bench.py is taken with: ONE float4 per thread, a workgroup per 4 KB -- measured on MI355X
// (tools/dev/probes/copy_probe.hip, 1 GiB): 6.28 TB/s; a grid-stride loop with four loads in flight per lane (the first form tried
// here) 4.4-4.9, 16 KB per workgroup 5.9 (6.28 with nontemporal accesses), hipMemcpy device-to-device 5.0
__global__ __launch_bounds__(256) void k_copy_f4(size_t n, const float4* __restrict__ src, float4* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
void launch_copy_f4(hipStream_t st, size_t n_f4, const float4* src, float4* dst, int /*blocks: unused, one float4 per thread*/) {
  if (n_f4 == 0) return;
  hipLaunchKernelGGL(k_copy_f4, dim3((unsigned)((n_f4 + 255) / 256)), dim3(256), 0, st, n_f4, src, dst);
}

void launch_copy(hipStream_t st, int n, const double* src, double* dst) {
  (void)hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
}

}  // namespace hitl
