// hitl_solver.cpp -- the solver object and the C ABI of libhitlgpu.so (include/hitl_gpu.h).
//
// Replaces the body of JointOpt::Run (JointOptimization.cpp:1295-1385): factor construction
// (AddOdometryConstraints :736-825, AddHumanConstraints :969-1054, AddSTFConstraints
// :539-559) becomes device-resident SoA tables (this file); ceres::Solve (:1093, :1208) becomes the
// Levenberg-Marquardt loop of hitl_lm.cpp; FindSTFCorrespondences (:561-642) is driven from hitl_corr.cpp.
// There is no CPU fallback anywhere in this library.

#include <dlfcn.h>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>
#include <numeric>
#include <thread>

#include <rccl/rccl.h>

#include "hitl_solver_impl.h"

namespace hitl {

static thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

}  // namespace hitl

using namespace hitl;

namespace hitl {

int solver_ctx(hitl_solver* s, SolverCtx* out) {
  if (!s) return fail(HITL_ERR_INVALID_ARGUMENT, "null solver");
  out->device = s->device; out->stream = s->stream; out->n = s->n; out->n_obs = s->n_obs; out->world = s->world;
  out->map = s->map_view();
  out->own_lo = s->own_lo; out->own_hi = s->own_hi; out->pose_base = s->g_begin - s->hl;
  return HITL_OK;
}

int solver_allgather(hitl_solver* s, const void* send, void* recv, int64_t bytes_per_rank) { return comm_allgather(s, send, recv, bytes_per_rank); }

int solver_pose_rt(hitl_solver* s, const float* poses, const float** T4_dev) {
  const size_t n = (size_t)s->n;
  if (s->pose_rt.p != nullptr && s->pose_rt_of.size() == 3 * n && std::memcmp(s->pose_rt_of.data(), poses, sizeof(float) * 3 * n) == 0) {
    *T4_dev = s->pose_rt.p;
    return HITL_OK;
  }
  std::vector<float> t(4 * n);
  for (size_t i = 0; i < n; ++i) {      // Rotation2Df(angle), Translation2f: fp32 trig on the host
    t[4 * i] = cosf(poses[3 * i + 2]); t[4 * i + 1] = sinf(poses[3 * i + 2]); t[4 * i + 2] = poses[3 * i]; t[4 * i + 3] = poses[3 * i + 1];
  }
  s->pose_rt_of.clear();                // (not valid until the upload has completed)
  TRY(s->pose_rt.upload(t, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->pose_rt_of.assign(poses, poses + 3 * n);
  *T4_dev = s->pose_rt.p;
  return HITL_OK;
}

// ---------------------------------------------------------------------------------------
// derived pair structures (host, O(P)); called whenever the correspondence set changes
// ---------------------------------------------------------------------------------------
int ensure_host_pairs(hitl_solver* s) {
  if (!s->host_pairs_stale) return HITL_OK;
  const size_t P = (size_t)s->n_pairs;
  s->h_pair_src.resize(P); s->h_pair_tgt.resize(P); s->h_match_off.resize(P + 1);
  hipStream_t st = s->stream;
  if (P) {
    HIP_TRY(hipMemcpyAsync(s->h_pair_src.data(), s->pair_src.p, P * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(s->h_pair_tgt.data(), s->pair_tgt.p, P * sizeof(int), hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipMemcpyAsync(s->h_match_off.data(), s->match_off.p, (P + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  s->h_perm.resize(P);
  std::iota(s->h_perm.begin(), s->h_perm.end(), 0);
  s->host_pairs_stale = false;
  return HITL_OK;
}

int build_pair_structures(hitl_solver* s) {
  ++s->corr_generation;
  s->host_pairs_stale = false;      // the caller has just written the host lists
  s->obadj_nonzero = true;
  const int n = s->n, P = s->n_pairs;
  int maxd = 1;
  for (int p = 0; p < P; ++p) maxd = std::max(maxd, std::abs(s->h_pair_src[p] - s->h_pair_tgt[p]));
  const int cap = std::min(std::max(s->opt.max_band_poses, 1), kMaxBandPoses);
  const int Wb = std::min(std::max(maxd, s->min_band_poses), cap);      // min_band_poses: a submap rank's band is at least as wide as the interface (prepare)
  s->band_pairs_Wb = Wb;
  std::vector<int> ppo(n + 1, 0), inc_off(n + 1, 0);
  for (int p = 0; p < P; ++p) { ppo[s->h_pair_src[p] + 1]++; inc_off[s->h_pair_src[p] + 1]++; inc_off[s->h_pair_tgt[p] + 1]++; }
  for (int i = 0; i < n; ++i) { ppo[i + 1] += ppo[i]; inc_off[i + 1] += inc_off[i]; }
  std::vector<int> inc_ent(std::max(2 * P, 1)), fill(inc_off.begin(), inc_off.end() - 1);
  std::vector<int> band_pair((size_t)n * Wb * 2, -1);
  std::unordered_map<int64_t, int> ob_index;                // (lo, hi) -> entry, entries numbered in the order the pairs arrive
  ob_index.reserve((size_t)std::min(P, 1 << 15));
  std::vector<int> ob_lo, ob_hi, ob_fwd, ob_rev;
  for (int p = 0; p < P; ++p) {
    const int a = s->h_pair_src[p], b = s->h_pair_tgt[p];
    inc_ent[fill[a]++] = 2 * p;
    inc_ent[fill[b]++] = 2 * p + 1;
    const int lo = std::min(a, b), hi = std::max(a, b), d = hi - lo;
    if (d <= Wb) {
      int& slot = band_pair[((size_t)lo * Wb + (d - 1)) * 2 + (a < b ? 0 : 1)];
      if (slot >= 0) return fail(HITL_ERR_INVALID_ARGUMENT, "duplicate directed pose pair in the correspondence set");
      slot = p;
    } else {
      const int64_t key = (int64_t)lo * n + hi;
      auto it = ob_index.find(key);
      int e;
      if (it == ob_index.end()) {
        e = (int)ob_lo.size();
        ob_index[key] = e;
        ob_lo.push_back(lo); ob_hi.push_back(hi); ob_fwd.push_back(-1); ob_rev.push_back(-1);
      } else {
        e = it->second;
      }
      int& slot = (a < b) ? ob_fwd[e] : ob_rev[e];
      if (slot >= 0) return fail(HITL_ERR_INVALID_ARGUMENT, "duplicate directed pose pair in the correspondence set");
      slot = p;
    }
  }
  s->n_pairs_owned = 0; s->n_matches_owned = 0;
  for (int p = 0; p < P; ++p)
    if (s->h_pair_src[p] >= s->own_lo && s->h_pair_src[p] < s->own_hi) { s->n_pairs_owned++; s->n_matches_owned += s->h_match_off[p + 1] - s->h_match_off[p]; }
  const int E = (int)ob_lo.size();
  std::vector<int> obadj_off(n + 1, 0);
  for (int e = 0; e < E; ++e) { obadj_off[ob_lo[e] + 1]++; obadj_off[ob_hi[e] + 1]++; }
  for (int i = 0; i < n; ++i) obadj_off[i + 1] += obadj_off[i];
  std::vector<int> obadj_ent(std::max(2 * E, 1)), obadj_col(std::max(2 * E, 1)), fill2(obadj_off.begin(), obadj_off.end() - 1);
  for (int e = 0; e < E; ++e) {
    obadj_col[fill2[ob_lo[e]]] = ob_hi[e]; obadj_ent[fill2[ob_lo[e]]++] = 2 * e;
    obadj_col[fill2[ob_hi[e]]] = ob_lo[e]; obadj_ent[fill2[ob_hi[e]]++] = 2 * e + 1;
  }
  s->n_offband = E;
  s->h_obadj_off = obadj_off;                  // kept on the host: the persistent PCG launch's product items are cut from it
  hipStream_t st = s->stream;
  TRY(s->pose_pair_off.upload(ppo, st));
  TRY(s->inc_off.upload(inc_off, st));
  TRY(s->inc_ent.upload(inc_ent, st));
  TRY(s->band_pair.upload(band_pair, st));
  if (ob_lo.empty()) { ob_lo.push_back(0); ob_hi.push_back(0); ob_fwd.push_back(-1); ob_rev.push_back(-1); }
  TRY(s->ob_lo.upload(ob_lo, st)); TRY(s->ob_hi.upload(ob_hi, st));
  TRY(s->ob_fwd.upload(ob_fwd, st)); TRY(s->ob_rev.upload(ob_rev, st));
  TRY(s->obadj_off.upload(obadj_off, st)); TRY(s->obadj_ent.upload(obadj_ent, st)); TRY(s->obadj_col.upload(obadj_col, st));
  HIP_TRY(hipStreamSynchronize(st));
  return HITL_OK;
}

int ensure_empty_pair_tables(hitl_solver* s) {
  // mode A never reads pair tables, but the views must hold valid pointers
  if (s->pose_pair_off.n == (size_t)s->n + 1) return HITL_OK;
  s->n_pairs = 0; s->n_matches = 0;
  s->h_pair_src.clear(); s->h_pair_tgt.clear(); s->h_perm.clear(); s->h_match_off.assign(1, 0);
  TRY(s->pair_src.resize(0)); TRY(s->pair_tgt.resize(0)); TRY(s->matches.resize(kMatchPad));
  std::vector<int64_t> mo(1, 0);
  TRY(s->match_off.upload(mo, s->stream));
  return build_pair_structures(s);
}

// ---------------------------------------------------------------------------------------
// work buffers
// ---------------------------------------------------------------------------------------
int ensure_work(hitl_solver* s, int Wb) {
  const int n = s->n;
  s->B = 3 * Wb;
  // (a submap rank: one super-block of slack -- the chain of its owned poses starts at own_lo, not at a multiple of Wb, and its
  // last block may reach up to Wb - 1 poses past the end)
  s->nsb = (n + Wb - 1) / Wb + (s->dist_mode() ? 1 : 0);
  s->npad = s->nsb * s->B;
  const size_t npad = (size_t)s->npad;
  for (DevArray<double>* v : {&s->x, &s->xc, &s->scale, &s->rhs, &s->dt, &s->wv, &s->qv, &s->rv, &s->zv, &s->pv}) {
    const bool fresh = v->cap < npad;
    TRY(v->resize(npad));
    if (fresh || s->wb_Wb != Wb) TRY(v->zero(s->stream));
  }
  for (int k = 0; k < 2; ++k) {
    TRY(s->pairRJ[k].resize((size_t)std::max(s->n_pairs, 1) * kPairRJ));
    TRY(s->Hband[k].resize((size_t)n * (Wb + 1) * 9));
    const bool fresh = s->gvec[k].cap < npad;
    TRY(s->gvec[k].resize(npad));
    if (fresh || s->wb_Wb != Wb) TRY(s->gvec[k].zero(s->stream));
    TRY(s->obblk[k].resize((size_t)std::max(s->n_offband, 1) * 9));
    TRY(s->lin_partials[k].resize((size_t)assemble_blocks(n) * 2));
  }
  TRY(s->step_partials.resize((size_t)spmv_blocks(s->npad) * 8));
  TRY(s->ob_acc.resize(npad));
  TRY(s->pose_cs.resize((size_t)n));
  {
    // step partials at offset 0, linearisation partials behind them: one iteration reads both with one sync
    const size_t need = (size_t)assemble_blocks(n) * 2 + (size_t)spmv_blocks(s->npad) * 8;
    if (need > s->h_partials_cap) {
      if (s->h_partials) (void)hipHostFree(s->h_partials);
      s->h_partials = nullptr; s->h_partials_cap = 0;
      HIP_TRY(hipHostMalloc((void**)&s->h_partials, need * sizeof(double)));
      s->h_partials_cap = need;
    }
  }
  const size_t bb = (size_t)s->nsb * s->B * s->B;
  TRY(s->crD.resize(bb)); TRY(s->crL.resize(bb)); TRY(s->crDinv.resize(bb)); TRY(s->crGl.resize(bb)); TRY(s->crGr.resize(bb));
  {
    const size_t np = (size_t)cr_contrib_slots(s->nsb);
    s->cr_pw_rhs = s->dist_mode() ? 1 + 6 * Wb : 1;       // the submap solve pushes its interface columns through the factorisation
    TRY(s->crP.resize(4 * np * s->B * s->B)); TRY(s->crPw.resize((size_t)s->cr_pw_rhs * 4 * np * s->B));
    if (s->cr_ticket.p == nullptr) { TRY(s->cr_ticket.resize(4)); TRY(s->cr_ticket.zero(s->stream)); }
  }
  s->wb_Wb = Wb; s->wb_n = n; s->wb_pairs = s->n_pairs; s->wb_offband = s->n_offband;
  return HITL_OK;
}

// ---------------------------------------------------------------------------------------
// factor tables derived from the fp32 poses handed to a solve
// ---------------------------------------------------------------------------------------
static void odom_setup(const float* pp, const float* pc, float4* A, float2* rr) {
  // JointOptimization.cpp:748-769 in fp32 (Vector2f / Rotation2Df arithmetic)
  const float kEpsilon = 1e-6f;
  const float tx = pc[0] - pp[0], ty = pc[1] - pp[1];
  float rx, ry, rho;
  if (std::fabs(tx) < kEpsilon && std::fabs(ty) < kEpsilon) {
    rx = cosf(pc[2]); ry = sinf(pc[2]); rho = 0.0f;
  } else {
    const float c = cosf(-pp[2]), sn = sinf(-pp[2]);
    const float vx = c * tx - sn * ty, vy = sn * tx + c * ty;
    const float nrm = sqrtf(vx * vx + vy * vy);
    rx = vx / nrm; ry = vy / nrm;
    rho = sqrtf(tx * tx + ty * ty);
  }
  const float c90 = cosf((float)M_PI_2), s90 = sinf((float)M_PI_2);  // Rotation2Df(M_PI_2)
  A->x = rx; A->y = ry; A->z = c90 * rx - s90 * ry; A->w = s90 * rx + c90 * ry;
  rr->x = rho;
  rr->y = (float)angle_mod((double)(pc[2] - pp[2]));
}

static void hc_target(const hitl_human_constraint_t& row, const float* a, float4* t) {
  // JointOptimization.cpp:981-989, :1023-1025, :1037-1039 in fp32
  const float ang = a[2];
  const float pax = cosf(ang), pay = sinf(ang);
  t->x = (a[0] + row.delta_parallel * pax) + row.delta_perpendicular * (-pay);
  t->y = (a[1] + row.delta_parallel * pay) + row.delta_perpendicular * pax;
  const float t_angle = ang + row.delta_angle;
  t->z = atan2f(sinf(t_angle), cosf(t_angle));
  t->w = ang + row.relative_penalty_dir;
}

int setup_factors(hitl_solver* s, const float* poses, const float* poses_global, bool with_odom_hc) {
  // Per solve: the odometry measurements re-derived from the current fp32 poses (JointOptimization.cpp:748-769) and the human
  // rows' targets (they depend on the anchor poses).  Staged in page-locked buffers the solver keeps: no allocation, real DMA,
  // and no synchronisation here -- the stream orders the uploads before the first kernel that reads them, and the buffers are
  // not touched again before the solve has returned.
  const int n = s->n;
  hipStream_t st = s->stream;
  if (s->staging_queued) {
    // uploads of an earlier call may still read the pinned staging buffers refilled below (a solve that returned early on an
    // error never drained the stream): wait for them -- on an idle stream this returns at once
    HIP_TRY(hipStreamSynchronize(st));
    s->staging_queued = false;
  }
  TRY(s->h_odom_A.resize(n)); TRY(s->h_odom_rr.resize(n)); TRY(s->h_hc_off.resize((size_t)n + 1));
  float4* A = s->h_odom_A.p;
  float2* rr = s->h_odom_rr.p;
  A[0] = make_float4(0, 0, 0, 0); rr[0] = make_float2(0, 0);
  const size_t H = with_odom_hc ? s->h_hc.size() : 0;
  TRY(s->h_hc_type.resize(H)); TRY(s->h_hc_tgt.resize(H));          // (at least one element each)
  uint32_t* hc_type = s->h_hc_type.p;
  float4* hc_tgt = s->h_hc_tgt.p;
  hc_type[0] = 0; hc_tgt[0] = make_float4(0, 0, 0, 0);
  s->num_hc_residuals = 0;
  if (with_odom_hc) {
    // the edges are independent: a few threads on large maps (50 000 fp32 set-ups are 0.25 ms on one core, most of what this
    // function costs; spawning a thread ~20 us)
    const int nt = (n >= 20000) ? (int)std::max(1u, std::min(4u, std::thread::hardware_concurrency())) : 1;
    auto edges = [&](int lo, int hi) { for (int i = lo; i < hi; ++i) odom_setup(poses + 3 * (i - 1), poses + 3 * i, &A[i], &rr[i]); };
    if (nt > 1) {
      // std::thread's constructor may throw (std::system_error: no thread to be had); nothing may leave through the C ABI:
      // whatever range got no thread is done here
      std::vector<std::thread> th;
      int started = 1;
      try {
        for (int t = 1; t < nt; ++t) {
          th.emplace_back(edges, 1 + (int)((int64_t)(n - 1) * t / nt), 1 + (int)((int64_t)(n - 1) * (t + 1) / nt));
          started = t + 1;
        }
      } catch (...) {}
      edges(1, 1 + (n - 1) / nt);
      for (int t = started; t < nt; ++t) edges(1 + (int)((int64_t)(n - 1) * t / nt), 1 + (int)((int64_t)(n - 1) * (t + 1) / nt));
      for (auto& x : th) x.join();
    } else edges(1, n);
    int* hc_off = s->h_hc_off.p;
    std::fill(hc_off, hc_off + n + 1, 0);
    for (size_t h = 0; h < H; ++h) hc_off[s->h_hc[h].constrained_pose_id + 1]++;
    for (int i = 0; i < n; ++i) hc_off[i + 1] += hc_off[i];
    std::vector<int> fill(H ? n : 0);
    if (H) std::copy(hc_off, hc_off + n, fill.begin());
    for (size_t h = 0; h < H; ++h) {  // stable: rows of one pose keep their caller order
      const hitl_human_constraint_t& row = s->h_hc[h];
      const int dst = fill[row.constrained_pose_id]++;
      hc_type[dst] = row.type;
      hc_target(row, poses_global + 3 * (size_t)row.anchor_pose_id, &hc_tgt[dst]);
      if (row.type == HITL_HC_LINE_SEGMENT) s->num_hc_residuals += 3;
      else if (row.type == HITL_HC_COLINEAR) s->num_hc_residuals += 2;
      else if (row.type == HITL_HC_PERPENDICULAR || row.type == HITL_HC_PARALLEL) s->num_hc_residuals += 1;
    }
  } else {
    std::fill(A, A + n, make_float4(0, 0, 0, 0));
    std::fill(rr, rr + n, make_float2(0, 0));
    std::fill(s->h_hc_off.p, s->h_hc_off.p + n + 1, 0);
  }
  s->staging_queued = true;       // from the first queued copy on, also when one of the later uploads fails
  TRY(s->odom_A.upload(A, (size_t)n, st)); TRY(s->odom_rr.upload(rr, (size_t)n, st));
  TRY(s->hc_off.upload(s->h_hc_off.p, (size_t)n + 1, st));
  TRY(s->hc_type.upload(hc_type, std::max<size_t>(H, 1), st)); TRY(s->hc_tgt.upload(hc_tgt, std::max<size_t>(H, 1), st));
  s->prepare_pcie_bytes = (double)n * (sizeof(float4) + sizeof(float2)) + ((double)n + 1) * sizeof(int) + (double)std::max<size_t>(H, 1) * (sizeof(uint32_t) + sizeof(float4));
  return HITL_OK;
}

FactorView factor_view(hitl_solver* s, int mode, int Wb) {
  FactorView f;
  f.with_stf = (mode != HITL_MODE_ODOM_HC) && s->n_pairs > 0;
  f.with_odom = (mode != HITL_MODE_STF_ONLY);
  f.with_hc = (mode != HITL_MODE_STF_ONLY) && !s->h_hc.empty();
  f.band_poses = Wb;
  f.inc_off = s->inc_off.p; f.inc_ent = s->inc_ent.p; f.band_pair = s->band_pair.p;
  f.odom_A = s->odom_A.p; f.odom_rr = s->odom_rr.p;
  f.sig_r = s->opt.odom_radial_std_dev; f.sig_t = s->opt.odom_tangential_std_dev; f.sig_a = s->opt.odom_angular_std_dev;
  f.hc_off = s->hc_off.p; f.hc_type = s->hc_type.p; f.hc_tgt = s->hc_tgt.p;
  f.n_offband = f.with_stf ? s->n_offband : 0;
  f.ob_lo = s->ob_lo.p; f.ob_hi = s->ob_hi.p; f.ob_fwd = s->ob_fwd.p; f.ob_rev = s->ob_rev.p;
  f.obadj_off = s->obadj_off.p; f.obadj_ent = s->obadj_ent.p; f.obadj_col = s->obadj_col.p;
  // 8 or more off-band neighbours per pose on average: a serial walk per pose is what the product then waits for
  f.ob_acc = (f.n_offband > 0 && (int64_t)2 * f.n_offband >= (int64_t)8 * s->n) ? s->ob_acc.p : nullptr;
  f.fixed = s->fixed.p; f.own_lo = s->own_lo; f.own_hi = s->own_hi;
  return f;
}

int check_ready(hitl_solver* s, int mode) {
  if (!s) return fail(HITL_ERR_INVALID_ARGUMENT, "null solver");
  if (s->n <= 0) return fail(HITL_ERR_NOT_READY, "hitl_solver_set_map has not been called");
  if (mode < 0 || mode > 2) return fail(HITL_ERR_INVALID_ARGUMENT, "mode must be 0 (ODOM_HC), 1 (STF_ONLY) or 2 (JOINT)");
  return HITL_OK;
}

// the rank-local part of prepare: everything that can fail before the ranks talk to each other
static int prepare_local(hitl_solver* s, int mode, const float* poses_ref, const float* poses_global, int Wb) {
  HIP_TRY(hipSetDevice(s->device));
  if (dev_switches().fail_prepare) return fail(HITL_ERR_OUT_OF_MEMORY, "prepare: failure injected by the test hook fail_prepare");
  TRY(ensure_empty_pair_tables(s));
  TRY(ensure_work(s, Wb));
  if (s->opt.stf_moments && mode != HITL_MODE_ODOM_HC && s->moments_of != s->corr_generation) {
    // once per correspondence set: the point data enter the scan-to-scan factors only through these sums
    TRY(s->pairM.resize((size_t)std::max(s->n_pairs, 1) * kPairMomStride));
    launch_stf_moments(s->stream, s->map_view(), s->pair_view(), s->pairM.p);
    HIP_TRY(hipGetLastError());
    s->moments_of = s->corr_generation;
  }
  if (!s->opt.stf_moments && mode != HITL_MODE_ODOM_HC && s->n_pairs > 0 && s->stf_pose_recs_of != s->corr_generation) {
    // once per correspondence set: the per-pose descriptor records (what the first two dependent load rounds of a workgroup fetch)
    TRY(s->stf_pose_recs.resize((size_t)s->n * stf_pose_rec_bytes()));
    launch_build_stf_pose_recs(s->stream, s->map_view(), s->pair_view(), s->stf_pose_recs.p);
    HIP_TRY(hipGetLastError());
    s->stf_pose_recs_of = s->corr_generation;
  }
  TRY(setup_factors(s, poses_ref, poses_global, mode != HITL_MODE_STF_ONLY));
  return HITL_OK;
}

int prepare(hitl_solver* s, int mode, const float* poses_ref, const float* poses_global, FactorView* fv) {
  // mode A reads no pair table: a 1-pose band; the pair-derived band_pair table is unused
  int Wb = (mode == HITL_MODE_ODOM_HC) ? 1 : s->band_pairs_Wb;
  const int local_rc = prepare_local(s, mode, poses_ref, poses_global, Wb);
  if (!s->dist_mode() && local_rc != HITL_OK) return local_rc;
  if (s->dist_mode()) {
    // interface width = how deep (in poses) any factor reaches across a cut, maximised over all
    // ranks: 1 when only the odometry edges straddle the cuts, up to the pose band otherwise
    int wi_local = 1, min_Wb = Wb, max_Wb = Wb;
    bool sep_cand = !dev_switches().dist_columns;          // the separator form of the linear solve (hitl_internal.h) is a candidate
    if (mode != HITL_MODE_ODOM_HC && local_rc == HITL_OK && ensure_host_pairs(s) == HITL_OK)
      for (int p = 0; p < s->n_pairs; ++p) {
        const int a = std::min(s->h_pair_src[p], s->h_pair_tgt[p]), b = std::max(s->h_pair_src[p], s->h_pair_tgt[p]);
        if (a < s->own_lo && b >= s->own_lo) wi_local = std::max(wi_local, std::max(s->own_lo - a, b - s->own_lo + 1));
        if (a < s->own_hi && b >= s->own_hi) wi_local = std::max(wi_local, std::max(s->own_hi - a, b - s->own_hi + 1));
      }
    {
      // one exchange carries the interface width AND every rank's local verdict, so that all ranks leave at the same
      // point: a rank that failed above (or holds pose pairs outside the band, which submap solves do not support yet)
      // must not let the others walk into the collectives of the LM loop alone
      const bool offband_here = (mode != HITL_MODE_ODOM_HC) && s->n_offband > 0;
      // (the last entry: this rank is willing to take the separator form -- HITL_DIST_COLUMNS is a per-process switch, and a rank
      // that differs from the others must not leave them waiting in the separator form's own exchange below)
      double mine[6] = {(double)wi_local, (double)local_rc, offband_here ? 1.0 : 0.0, (double)Wb, (double)(s->own_hi - s->own_lo), sep_cand ? 1.0 : 0.0};
      std::vector<double> all((size_t)6 * s->world);
      TRY(comm_allgather(s, mine, all.data(), sizeof(mine)));
      int worst = HITL_OK;
      bool any_offband = false;
      for (int r = 0; r < s->world; ++r) {
        wi_local = std::max(wi_local, (int)all[(size_t)6 * r]);
        if ((int)all[(size_t)6 * r + 1] != HITL_OK && worst == HITL_OK) worst = (int)all[(size_t)6 * r + 1];
        any_offband = any_offband || all[(size_t)6 * r + 2] != 0.0;
        min_Wb = std::min(min_Wb, (int)all[(size_t)6 * r + 3]);
        max_Wb = std::max(max_Wb, (int)all[(size_t)6 * r + 3]);
      }
      if (local_rc != HITL_OK) return local_rc;
      if (worst != HITL_OK) return fail(worst, "another rank failed while preparing the solve");
      s->dist_pcg = any_offband;
      // The separator form wants ONE band width on all ranks (the records are B x B tiles), every rank but the last owning a
      // multiple of it (its last super-block is full: nothing couples across it) and at least two super-blocks per rank.
      // Decided from the gathered values: the same verdict everywhere.
      if (any_offband || max_Wb > std::min(std::max((int)s->opt.max_band_poses, 1), kMaxBandPoses)) sep_cand = false;
      for (int r = 0; r < s->world; ++r) {
        const int64_t cnt = (int64_t)all[(size_t)6 * r + 4];
        if (cnt <= max_Wb || (r + 1 < s->world && cnt % max_Wb != 0) || all[(size_t)6 * r + 5] == 0.0) sep_cand = false;
      }
    }
    s->Wi = wi_local;
    s->dist_sep = false;
    const int band_target = sep_cand ? std::max(max_Wb, s->Wi) : s->Wi;      // (mode A: 1 everywhere)
    if (!s->dist_pcg && mode != HITL_MODE_ODOM_HC) {
      // The interface width is a maximum over ALL ranks; the band of this rank comes from its own kept pairs.  A rank that
      // holds only short pairs while a cut elsewhere is crossed by longer ones widens its band to the interface (the exact
      // interface system needs Wi <= Wb, and the multi-right-hand-side work buffers are sized by Wb).  The verdict must be the
      // same on every rank -- Wi and the cap are -- or the others would walk into the collectives of the LM loop alone.
      const int cap = std::min(std::max((int)s->opt.max_band_poses, 1), kMaxBandPoses);
      if (s->Wi > cap)
        return fail(HITL_ERR_UNSUPPORTED, "a factor reaches " + std::to_string(s->Wi) + " poses across a cut, further than the pose band may be wide (max_band_poses)");
      int widen_rc = HITL_OK;
      if (band_target > Wb) {
        s->min_band_poses = band_target; // reset by set_map / set_partition / every new correspondence set
        widen_rc = ensure_host_pairs(s);
        if (widen_rc == HITL_OK) widen_rc = build_pair_structures(s);
        if (widen_rc == HITL_OK) { Wb = s->band_pairs_Wb; widen_rc = prepare_local(s, mode, poses_ref, poses_global, Wb); }
      }
      if (band_target > min_Wb) {
        // somebody rebuilt its tables (every rank knows from the gathered bands): agree on the outcome before anyone enters
        // the LM loop's collectives -- a rank that ran out of memory here must not leave the others waiting in them
        double mine_w = (double)widen_rc;
        std::vector<double> all_w((size_t)s->world);
        TRY(comm_allgather(s, &mine_w, all_w.data(), sizeof(mine_w)));
        if (widen_rc != HITL_OK) return widen_rc;
        for (int r = 0; r < s->world; ++r)
          if ((int)all_w[(size_t)r] != HITL_OK) return fail((int)all_w[(size_t)r], "another rank failed while widening its pose band to the interface");
      }
    }
    if (s->dist_pcg) {
      // Pose pairs outside the band somewhere: the ranks iterate together (distributed PCG, hitl_lm.cpp).  A factor that
      // straddles a cut must be seen by both ranks it touches, i.e. reach no deeper than the halos on either side.
      if (s->Wi > s->halo_req)
        return fail(HITL_ERR_UNSUPPORTED, "a pose pair reaches " + std::to_string(s->Wi) + " poses across a cut but the halos hold " +
                                          std::to_string(s->halo_req) + ": raise halo_poses of hitl_solver_set_partition");
      double mine2[3] = {(double)s->g_begin, (double)(s->own_hi - s->own_lo), (double)s->halo_req};
      std::vector<double> all2((size_t)3 * s->world);
      TRY(comm_allgather(s, mine2, all2.data(), sizeof(mine2)));
      s->part_begin.assign(s->world, 0); s->part_count.assign(s->world, 0);
      s->maxown = 1;
      int hmax = 1;
      for (int r = 0; r < s->world; ++r) {
        s->part_begin[r] = (int64_t)all2[(size_t)3 * r]; s->part_count[r] = (int64_t)all2[(size_t)3 * r + 1];
        s->maxown = std::max(s->maxown, (int)s->part_count[r]);
        hmax = std::max(hmax, (int)all2[(size_t)3 * r + 2]);
      }
      // The exchange record of the distributed PCG carries only what somebody's halo can need (round 3; the whole owned vector
      // until round 2: 1.2 MB per CG iteration at 50 000 poses): a halo pose lies within `halo` poses of one of its owner's
      // two boundaries, so the first and the last pcg_edge owned poses of every rank are enough (a rank no longer than two
      // edges sends everything).  Slot of owned offset o of a rank with c poses: o if c <= 2 edge or o < edge, else
      // edge + (o - (c - edge)).
      s->pcg_edge = std::min(hmax, s->maxown);
      const int rec_poses = std::min(s->maxown, 2 * s->pcg_edge);
      const int rec_len = 3 * rec_poses;         // head excluded
      std::vector<int> hidx;
      s->halo_owner_slot.clear();
      for (int li = 0; li < s->n; ++li) {
        if (li >= s->own_lo && li < s->own_hi) continue;
        const int64_t gi = s->g_begin - s->hl + li;
        int owner = -1;
        for (int r = 0; r < s->world; ++r) if (gi >= s->part_begin[r] && gi < s->part_begin[r] + s->part_count[r]) owner = r;
        if (owner < 0) return fail(HITL_ERR_INVALID_ARGUMENT, "the ranks' owned ranges do not tile the map");
        for (int c = 0; c < 3; ++c) {
          hidx.push_back(3 * li + c);
          const int oo = (int)(gi - s->part_begin[owner]), cc = (int)s->part_count[owner];
          int slot_pose;
          if (cc <= rec_poses || oo < s->pcg_edge) slot_pose = oo;
          else if (oo >= cc - s->pcg_edge) slot_pose = s->pcg_edge + (oo - (cc - s->pcg_edge));
          else return fail(HITL_ERR_INVALID_ARGUMENT, "a halo pose lies deeper inside its owner's range than any rank's halo: inconsistent partition");
          s->halo_owner_slot.push_back(owner * rec_len + 3 * slot_pose + c);
        }
      }
      s->n_halo = (int)hidx.size();
      if (hidx.empty()) hidx.push_back(0);
      TRY(s->halo_idx.upload(hidx, s->stream));
      TRY(s->halo_val.resize((size_t)std::max(s->n_halo, 1)));
      TRY(s->pcg_rec.resize((size_t)rec_len + 4)); TRY(s->pcg_head.resize(4));
      HIP_TRY(hipStreamSynchronize(s->stream));
      *fv = factor_view(s, mode, Wb);
      return HITL_OK;
    }
    const int Wi = s->Wi, nc = 6 * Wi;
    if (Wi > Wb) return fail(HITL_ERR_UNSUPPORTED, "a factor reaches further across a cut than the pose band");
    if (sep_cand) {
      // every rank checks that its own chain runs as fused stages, then all agree (one more tiny exchange, once per prepare)
      s->sep_nsb = (s->own_hi - s->own_lo + Wb - 1) / Wb;
      CrSource probe = {};
      probe.Wb = Wb;
      double mine_s = (Wb == band_target && cr_sep_applies(s->sep_chain_view(), probe, s->wv.p, s->dt.p)) ? 1.0 : 0.0;
      if (mine_s != 0.0) {
        // the separator form's buffers are allocated BEFORE the ranks agree on it and a failure counts as "does not apply here":
        // a rank that cannot allocate must not return alone while the others walk into the LM loop's collectives
        auto alloc = [&]() -> int {
          const size_t rl = (size_t)sep_rec_len(s->B), nn = (size_t)2 * s->world, bb2 = nn * s->B * s->B;
          TRY(s->sep_rec.resize(rl)); TRY(s->sep_all.resize(rl * s->world));
          TRY(s->sep_w2.resize(nn * s->B)); TRY(s->sep_x2.resize(nn * s->B));
          TRY(s->sep_D.resize(bb2)); TRY(s->sep_L.resize(bb2)); TRY(s->sep_Dinv.resize(bb2)); TRY(s->sep_Gl.resize(bb2)); TRY(s->sep_Gr.resize(bb2));
          const size_t np2 = (size_t)cr_contrib_slots((int)nn);
          TRY(s->sep_P.resize(4 * np2 * s->B * s->B)); TRY(s->sep_Pw.resize(4 * np2 * s->B));
          return HITL_OK;
        };
        if (alloc() != HITL_OK) { (void)hipGetLastError(); mine_s = 0.0; }
      }
      std::vector<double> all_s((size_t)s->world);
      TRY(comm_allgather(s, &mine_s, all_s.data(), sizeof(mine_s)));
      s->dist_sep = true;
      for (int r = 0; r < s->world; ++r) if (all_s[(size_t)r] == 0.0) s->dist_sep = false;
    }
    std::vector<int> cols(nc), hidx;
    for (int p = 0; p < Wi; ++p)
      for (int c = 0; c < 3; ++c) { cols[3 * p + c] = 3 * (s->own_lo + p) + c; cols[3 * Wi + 3 * p + c] = 3 * (s->own_hi - Wi + p) + c; }
    if (s->dist_sep) {
      // the neighbours' separators cover up to Wb poses beyond either cut: all of them that are held here take their steps from them
      SepGeom& gm = s->sep_geom;
      gm.own_lo = s->own_lo; gm.own_hi = s->own_hi; gm.r_pose0 = s->own_lo + (s->sep_nsb - 1) * Wb;
      gm.halo_lo = s->rank > 0 ? std::min(s->hl, Wb) : 0;
      gm.halo_hi = s->rank < s->world - 1 ? std::min(s->hh, Wb) : 0;
      gm.rank = s->rank; gm.world = s->world; gm.Wb = Wb;
      for (int p = 0; p < gm.halo_lo; ++p) for (int c = 0; c < 3; ++c) hidx.push_back(3 * (s->own_lo - gm.halo_lo + p) + c);
      for (int p = 0; p < gm.halo_hi; ++p) for (int c = 0; c < 3; ++c) hidx.push_back(3 * (s->own_hi + p) + c);
      s->n_halo = (int)hidx.size();
      TRY(s->dt.zero(s->stream));          // rows outside the chain and the filled halo entries stay zero (the separator buffers exist: allocated above)
    } else {
    if (s->rank > 0) for (int p = 0; p < Wi; ++p) for (int c = 0; c < 3; ++c) hidx.push_back(3 * (s->own_lo - Wi + p) + c);
    if (s->rank < s->world - 1) for (int p = 0; p < Wi; ++p) for (int c = 0; c < 3; ++c) hidx.push_back(3 * (s->own_hi + p) + c);
    }
    TRY(s->if_cols.upload(cols, s->stream));
    if (hidx.empty()) hidx.push_back(0);
    TRY(s->halo_idx.upload(hidx, s->stream));
    if (!s->dist_sep) {                           // the interface-column form's buffers (the separator form needs none of them)
      TRY(s->iface_dev.resize((size_t)nc * (nc + 2) + 2 * (size_t)(nc / 2) * (nc / 2)));   // interface rows + the two cross-cut blocks
      TRY(s->rhsN.resize((size_t)(1 + nc) * s->npad)); TRY(s->zN.resize((size_t)(1 + nc) * s->npad));
      TRY(s->coef_dev.resize((size_t)3 * nc));    // coefficients | halo steps | halo scales: one upload per LM iteration
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  *fv = factor_view(s, mode, Wb);
  return HITL_OK;
}

}  // namespace hitl

namespace hitl {
static DevSwitches g_dev;
static DevSwitches g_test_hooks;          // what hitl_test_set_switch has set; only the test-hook fields are ever taken from it
const DevSwitches& dev_switches() { return g_dev; }
void read_dev_switches() {
  DevSwitches d = g_test_hooks;
  auto on = [](const char* name) { return std::getenv(name) != nullptr; };
  // the five environment switches; everything else a DevSwitches field can be is a test hook
  d.timing = on("HITL_TIMING");
  d.force_dist = on("HITL_FORCE_DIST");
  d.force_rccl = on("HITL_FORCE_RCCL");
  d.dist_columns = on("HITL_DIST_COLUMNS");
  g_dev = d;
  roctx_resolve(on("HITL_ROCTX"));
}
}  // namespace hitl

// Test hook (declared in hitl_host.h, exported, not part of the C ABI of include/hitl_gpu.h): sets one of the DevSwitches
// test fields for every solver created afterwards in this process; name == nullptr clears all of them.  Returns 0, or -1 for
// an unknown name.
extern "C" int hitl_test_set_switch(const char* name, long long value) {
  using namespace hitl;
  if (name == nullptr) { g_test_hooks = DevSwitches(); return 0; }
  const std::string k(name);
  DevSwitches& d = g_test_hooks;
  if (k == "cr_legacy") d.cr_legacy = value != 0;
  else if (k == "cr_unfused_build") d.cr_unfused_build = value != 0;
  else if (k == "cr_seq") d.cr_seq = value != 0;
  else if (k == "cr_seq_m") d.cr_seq_m = (int)std::max(0ll, std::min(64ll, value));
  else if (k == "cr_tail") d.cr_tail = value != 0;
  else if (k == "cr_tail_starve") d.cr_tail_starve = value != 0;
  else if (k == "pcg_fused") d.pcg_fused = value != 0;
  else if (k == "kd_grid") d.kd_grid = value != 0;
  else if (k == "kd_spec") d.kd_spec = (int)value;
  else if (k == "kd_order") d.kd_order = value != 0;
  else if (k == "kd_split") d.kd_split = (int)value;
  else if (k == "stf_lds_targets") d.stf_lds_targets = value != 0;
  else if (k == "pcg_starve") d.pcg_starve = value != 0;
  else if (k == "cu_count") d.cu_count = (int)std::max(0ll, std::min(4096ll, value));
  else if (k == "stf_addr64") d.stf_addr64 = value != 0;
  else if (k == "device_lm") d.device_lm = value != 0;
  else if (k == "lm_ahead") d.lm_ahead = (int)std::max(1ll, std::min(16ll, value));
  else if (k == "kd_cap") d.kd_cap = (int)std::max(0ll, value);
  else if (k == "kd_redo_cap") d.kd_redo_cap = (int)std::max(1ll, std::min((long long)(1 << 18), value));
  else if (k == "cand_tf_check") d.cand_tf_check = value != 0;
  else if (k == "cand_tile_bytes") d.cand_tile_bytes = (size_t)std::max(1ll, value);
  else if (k == "fail_prepare") d.fail_prepare = value != 0;
  else if (k == "poison_step_from") d.poison_step_from = (int)std::max(0ll, value);
  else return -1;
  return 0;
}

namespace hitl {
static RoctxApi g_roctx;
const RoctxApi& roctx_api() { return g_roctx; }
void roctx_resolve(bool enable) {
  if (!enable) { g_roctx = RoctxApi(); return; }
  if (g_roctx.push) return;
  for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
    void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if (!h) continue;
    auto push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    auto pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (push && pop) { g_roctx.push = push; g_roctx.pop = pop; return; }
  }
}
}  // namespace hitl

// =========================================================================================
// C ABI
// =========================================================================================
// ---- pose covariances of a map with pose pairs outside the band: the dense inverse (hitl_vec.hip: k_dense_*) -----------------------
// rocSOLVER's dense Cholesky (dpotrf + dpotri) is bound at run time: the library carries no link-time dependency on it, and a box
// without it gets HITL_ERR_UNSUPPORTED for exactly this call, as before round 6.
namespace {
struct DenseSolverLib {
  void* handle = nullptr;            // rocblas_handle
  int (*create)(void**) = nullptr;
  int (*set_stream)(void*, hipStream_t) = nullptr;
  int (*potrf)(void*, int, int, double*, int, int*) = nullptr;
  int (*potri)(void*, int, int, double*, int, int*) = nullptr;
  bool ok = false;
};
DenseSolverLib& dense_solver_lib() {
  static DenseSolverLib L = [] {
    DenseSolverLib l;
    void* so = nullptr;
    for (const char* name : {"librocsolver.so", "librocsolver.so.0", "/opt/rocm/lib/librocsolver.so"})
      if ((so = dlopen(name, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (so == nullptr) return l;
    l.create = reinterpret_cast<int (*)(void**)>(dlsym(so, "rocblas_create_handle"));
    l.set_stream = reinterpret_cast<int (*)(void*, hipStream_t)>(dlsym(so, "rocblas_set_stream"));
    l.potrf = reinterpret_cast<int (*)(void*, int, int, double*, int, int*)>(dlsym(so, "rocsolver_dpotrf"));
    l.potri = reinterpret_cast<int (*)(void*, int, int, double*, int, int*)>(dlsym(so, "rocsolver_dpotri"));
    if (l.create && l.set_stream && l.potrf && l.potri && l.create(&l.handle) == 0 && l.handle != nullptr) l.ok = true;
    return l;
  }();
  return L;
}
}  // namespace

static int pose_covariances_dense(hitl_solver* s, const FactorView& fv, const float* poses_xyt, double* cov9) {
  if (s->dist_mode()) return fail(HITL_ERR_UNSUPPORTED, "pose_covariances: pose pairs outside the band on a submap rank");
  const int n = s->n;
  const int64_t d = 3 * (int64_t)n;
  if (d > 32768) return fail(HITL_ERR_UNSUPPORTED, "pose_covariances: pose pairs outside the band on a map of more than 10 922 poses (the dense inverse would take " +
                                                   std::to_string((d * d * 8) >> 30) + " GB)");
  DenseSolverLib& L = dense_solver_lib();
  if (!L.ok) return fail(HITL_ERR_UNSUPPORTED, "pose_covariances: pose pairs outside the band need rocSOLVER's dense Cholesky (librocsolver.so could not be bound)");
  hipStream_t st = s->stream;
  std::vector<double> hx((size_t)s->npad, 0.0);
  for (int64_t t = 0; t < d; ++t) hx[(size_t)t] = (double)poses_xyt[t];
  HIP_TRY(hipMemcpyAsync(s->x.p, hx.data(), sizeof(double) * s->npad, hipMemcpyHostToDevice, st));
  LinResult lr;
  TRY(run_linearize(s, fv, s->x.p, 0, &lr));            // J^T J at these poses: band blocks + off-band blocks (no trust-region diagonal)
  launch_jacobi_scale(st, n, s->npad, fv.band_poses, s->Hband[0].p, s->opt.jacobi_scaling, s->fixed.p, s->scale.p);
  DevArray<double> A, d_out;
  DevArray<float2> d_rot;
  DevArray<int> d_info;
  std::vector<float2> rot((size_t)n);
  for (int i = 0; i < n; ++i) { const float a = -poses_xyt[3 * i + 2]; rot[(size_t)i] = make_float2(cosf(a), sinf(a)); }   // Rotation2Df(-angle)
  TRY(A.resize((size_t)d * (size_t)d)); TRY(d_out.resize((size_t)9 * n)); TRY(d_info.resize(2)); TRY(d_rot.upload(rot, st));
  TRY(A.zero(st)); TRY(d_info.zero(st));
  launch_dense_fill(st, n, fv.band_poses, s->Hband[0].p, fv.n_offband, s->ob_lo.p, s->ob_hi.p, s->obblk[0].p, s->scale.p, s->fixed.p, A.p);
  HIP_TRY(hipGetLastError());
  constexpr int kFillLower = 122;                        // rocblas_fill_lower
  if (L.set_stream(L.handle, st) != 0 || L.potrf(L.handle, kFillLower, (int)d, A.p, (int)d, d_info.p) != 0 ||
      L.potri(L.handle, kFillLower, (int)d, A.p, (int)d, d_info.p + 1) != 0)
    return fail(HITL_ERR_HIP, "pose_covariances: the dense Cholesky could not be run");
  launch_dense_cov_extract(st, n, A.p, s->scale.p, s->fixed.p, d_rot.p, d_out.p);
  int info[2] = {0, 0};
  if (hipMemcpyAsync(cov9, d_out.p, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(info, d_info.p, sizeof(info), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return fail(HITL_ERR_HIP, "pose_covariances: kernels failed");
  if (info[0] != 0 || info[1] != 0) return fail(HITL_ERR_SOLVER_FAILURE, "pose_covariances: J^T J is singular at these poses");
  for (size_t t = 0; t < (size_t)9 * n; ++t)
    if (!(cov9[t] == cov9[t])) return fail(HITL_ERR_SOLVER_FAILURE, "pose_covariances: J^T J is singular at these poses");
  return HITL_OK;
}


extern "C" {

void hitl_options_default(hitl_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->abi_version = HITL_ABI_VERSION;
  o->device = 0;
  o->laser_std_dev = 0.05f;
  o->point_correlation_factor = 1.0f / 40.0f;
  o->point_match_threshold = 0.15f;
  o->max_stf_angle_error = (float)(25.0 * M_PI / 180.0);
  o->max_correspondences_per_point = 6;
  o->num_skip_readings = 1;
  o->min_inter_pose_correspondence = 10;
  o->odom_radial_std_dev = 0.03f;
  o->odom_tangential_std_dev = 0.03f;
  o->odom_angular_std_dev = 0.01f;
  o->max_num_iterations = 100;
  o->jacobi_scaling = 1;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->linear_solver_tolerance = 1e-12;
  o->max_linear_iterations = 500;     // Ceres: Solver::Options::max_linear_solver_iterations
  o->max_band_poses = 4;
  o->rank = 0;
  o->world_size = 1;
  o->timing_level = 1;
  o->kd_tree_search = 1;          // the reference's own search (BuildKDTrees + FindNearestPointNormal); 0 = the brute-force definition
  o->rccl_unique_id = nullptr;
}

int hitl_abi_version(void) { return HITL_ABI_VERSION; }

int hitl_device_memory(int32_t device, int64_t* free_bytes, int64_t* total_bytes) {
  if (!free_bytes || !total_bytes) return fail(HITL_ERR_INVALID_ARGUMENT, "device_memory: null argument");
  size_t f = 0, t = 0;
  if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) {
    (void)hipGetLastError();
    return fail(HITL_ERR_NO_DEVICE, "device_memory: no such HIP device (there is no CPU fallback)");
  }
  *free_bytes = (int64_t)f; *total_bytes = (int64_t)t;
  return HITL_OK;
}

int hitl_device_copy_bandwidth(int32_t device, int64_t bytes, int32_t repeats, double* gb_per_s) {
  if (!gb_per_s || bytes < 16 || repeats < 1) return fail(HITL_ERR_INVALID_ARGUMENT, "device_copy_bandwidth: bad argument");
  *gb_per_s = 0.0;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count || hipSetDevice(device) != hipSuccess) {
    (void)hipGetLastError();
    return fail(HITL_ERR_NO_DEVICE, "device_copy_bandwidth: no such HIP device (there is no CPU fallback)");
  }
  const size_t n = (size_t)bytes / 16;
  DevArray<float4> a, b;
  TRY(a.resize(n)); TRY(b.resize(n));
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
  int rc = HITL_OK;
  float ms = 0.f;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = HITL_ERR_HIP;
  if (rc == HITL_OK) {
    (void)hipMemsetAsync(a.p, 0x3f, n * 16, st);
    const int blocks = std::max(1, cus) * 8;                 // 8 workgroups of 256 per compute unit: every CU streams, no tail
    for (int k = 0; k < 2; ++k) launch_copy_f4(st, n, a.p, b.p, blocks);
    (void)hipEventRecord(e0, st);
    for (int k = 0; k < repeats; ++k) launch_copy_f4(st, n, a.p, b.p, blocks);
    (void)hipEventRecord(e1, st);
    if (hipStreamSynchronize(st) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)) rc = HITL_ERR_HIP;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  if (rc != HITL_OK) { (void)hipGetLastError(); return fail(rc, "device_copy_bandwidth: HIP error"); }
  *gb_per_s = 2.0 * (double)n * 16.0 * repeats / ((double)ms * 1e-3) / 1e9;
  return HITL_OK;
}

const char* hitl_last_error(void) { return g_last_error.c_str(); }

const char* hitl_status_string(int st) {
  switch (st) {
    case HITL_OK: return "OK";
    case HITL_ERR_INVALID_ARGUMENT: return "invalid argument";
    case HITL_ERR_NO_DEVICE: return "no usable HIP device";
    case HITL_ERR_HIP: return "HIP runtime error";
    case HITL_ERR_OUT_OF_MEMORY: return "out of device memory";
    case HITL_ERR_NOT_READY: return "solver not ready (map or correspondences missing)";
    case HITL_ERR_UNSUPPORTED: return "unsupported input";
    case HITL_ERR_SOLVER_FAILURE: return "solver FAILURE";
    case HITL_ERR_COMM: return "RCCL error";
    default: return "unknown status";
  }
}

int hitl_solver_create(const hitl_options* opt, hitl_solver** out) {
  if (!out) return fail(HITL_ERR_INVALID_ARGUMENT, "out is null");
  *out = nullptr;
  hitl_options o;
  if (opt) o = *opt; else hitl_options_default(&o);
  if (o.abi_version != HITL_ABI_VERSION) return fail(HITL_ERR_INVALID_ARGUMENT, "hitl_options.abi_version mismatch");
  read_dev_switches();
  (void)hitl_libm_sincos_exact();      // the search's once-per-process self-check of the C library (5-9 ms: a million sinf / cosf / sincosf calls) belongs here, not in the first search a user waits for
  if (o.world_size < 1 || o.rank < 0 || o.rank >= o.world_size) return fail(HITL_ERR_INVALID_ARGUMENT, "bad rank/world_size");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(HITL_ERR_NO_DEVICE, "no HIP device visible (hipGetDeviceCount); this library has no CPU fallback");
  if (o.device < 0 || o.device >= count) return fail(HITL_ERR_NO_DEVICE, "hitl_options.device out of range");
  HIP_TRY(hipSetDevice(o.device));
  hitl_solver* s = new (std::nothrow) hitl_solver();
  if (!s) return fail(HITL_ERR_OUT_OF_MEMORY, "host allocation failed");
  s->opt = o;
  s->device = o.device;
  auto bail = [&](int code, const std::string& m) { hitl_solver_destroy(s); return fail(code, m); };
  if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) return bail(HITL_ERR_HIP, "hipStreamCreate failed");
  for (hipEvent_t* e : {&s->ev0, &s->ev1, &s->ev2, &s->ev3, &s->ev4, &s->ev5})
    if (hipEventCreate(e) != hipSuccess) return bail(HITL_ERR_HIP, "hipEventCreate failed");
  if (hipHostMalloc((void**)&s->h_scalars, sizeof(double) * (kNumScalars + 8)) != hipSuccess) return bail(HITL_ERR_HIP, "hipHostMalloc failed");
  std::memset(s->h_scalars, 0, sizeof(double) * (kNumScalars + 8));
  if (s->lm_dev.resize(1) != HITL_OK) return bail(HITL_ERR_OUT_OF_MEMORY, "hipMalloc failed");     // a recycled pinned block may hold an old sequence word (spin_sync polls it)
  if (s->d_scalars.resize(kNumScalars) != HITL_OK) return bail(HITL_ERR_OUT_OF_MEMORY, "hipMalloc failed");
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, o.device) == hipSuccess && cus > 0) s->cu_count = cus;
    if (dev_switches().cu_count > 0) s->cu_count = dev_switches().cu_count;
  }
  s->rank = o.rank;
  s->world = o.world_size;
  s->force_dist = dev_switches().force_dist;
  if ((o.world_size > 1 || dev_switches().force_rccl) && o.rccl_unique_id != nullptr) {
    ncclUniqueId id;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(&id, o.rccl_unique_id, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&s->nccl, o.world_size, id, o.rank);
    if (r != ncclSuccess) return bail(HITL_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  }
  s->opt.rccl_unique_id = nullptr;   // not retained
  *out = s;
  return HITL_OK;
}

void hitl_solver_destroy(hitl_solver* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (DevArray<double>* v : {&s->d_scalars, &s->x, &s->xc, &s->scale, &s->rhs, &s->dt, &s->wv, &s->qv, &s->rv, &s->zv, &s->pv,
                              &s->crD, &s->crL, &s->crDinv, &s->crGl, &s->crGr, &s->crP, &s->crPw})
    v->release();
  for (int k = 0; k < 2; ++k) { s->pairRJ[k].release(); s->Hband[k].release(); s->gvec[k].release(); s->obblk[k].release(); s->lin_partials[k].release(); }
  s->obs_off.release(); s->match_off.release();
  s->obs.release(); s->cov9.release(); s->step_partials.release(); s->pose_cs.release();
  if (s->h_partials) (void)hipHostFree(s->h_partials);
  for (DevArray<int>* v : {&s->pair_src, &s->pair_tgt, &s->pose_pair_off, &s->inc_off, &s->inc_ent, &s->band_pair, &s->ob_lo,
                           &s->ob_hi, &s->ob_fwd, &s->ob_rev, &s->obadj_off, &s->obadj_ent, &s->hc_off})
    v->release();
  s->d_cand_tgt.release(); s->d_cand_counts.release(); s->d_cand_kept.release(); s->d_cand_tf.release(); s->d_cand_table.release();
  s->d_cand_trig.release(); s->d_cand_poses.release(); s->d_cand_cs.release();
  s->pairM.release(); s->kd_pts.release(); s->kd_meta.release(); s->d_kd_ovf.release(); s->world_xy.release();
  s->matches.release(); s->odom_A.release(); s->hc_tgt.release(); s->odom_rr.release(); s->hc_type.release();
  if (s->nccl) (void)ncclCommDestroy(s->nccl);
  s->comm_send.release(); s->comm_recv.release(); s->fixed.release(); s->if_cols.release(); s->halo_idx.release();
  s->iface_dev.release(); s->rhsN.release(); s->zN.release(); s->coef_dev.release(); s->halo_val.release(); s->halo_sc.release();
  if (s->h_scalars) (void)hipHostFree(s->h_scalars);
  for (hipEvent_t e : {s->ev0, s->ev1, s->ev2, s->ev3, s->ev4, s->ev5, s->ev_stage[0], s->ev_stage[1]}) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : s->lm_events) if (e) (void)hipEventDestroy(e);
  s->lm_dev.release();
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

int hitl_solver_set_comm(hitl_solver* s, const hitl_comm_ops* ops) {
  if (!s) return fail(HITL_ERR_INVALID_ARGUMENT, "null solver");
  if (ops) s->comm = *ops; else { s->comm.allgather = nullptr; s->comm.ctx = nullptr; }
  return HITL_OK;
}

int hitl_solver_comm_selftest(hitl_solver* s) {
  if (!s) return fail(HITL_ERR_INVALID_ARGUMENT, "null solver");
  HIP_TRY(hipSetDevice(s->device));
  uint64_t send[8];
  for (int k = 0; k < 8; ++k) send[k] = 0x4849544c00000000ull + (uint64_t)s->rank * 8 + k;
  std::vector<uint64_t> recv((size_t)8 * s->world, 0);
  TRY(comm_allgather(s, send, recv.data(), sizeof(send)));
  for (int r = 0; r < s->world; ++r)
    for (int k = 0; k < 8; ++k)
      if (recv[(size_t)8 * r + k] != 0x4849544c00000000ull + (uint64_t)r * 8 + k) return fail(HITL_ERR_COMM, "all-gather self-test: wrong payload");
  // the same payload from a device buffer: the path the interface records of a submap solve take
  DevArray<uint64_t> dsend;
  TRY(dsend.upload(send, 8, s->stream));
  std::fill(recv.begin(), recv.end(), 0);
  std::vector<double> stage;
  TRY(comm_allgather_dev(s, dsend.p, recv.data(), sizeof(send), stage));
  for (int r = 0; r < s->world; ++r)
    for (int k = 0; k < 8; ++k)
      if (recv[(size_t)8 * r + k] != 0x4849544c00000000ull + (uint64_t)r * 8 + k) return fail(HITL_ERR_COMM, "all-gather self-test (device record): wrong payload");
  return HITL_OK;
}

int hitl_solver_set_timing_level(hitl_solver* s, int32_t level) {
  if (!s || level < 0 || level > 2) return fail(HITL_ERR_INVALID_ARGUMENT, "set_timing_level: level must be 0, 1 or 2");
  s->opt.timing_level = level;
  return HITL_OK;
}

int hitl_solver_comm_ranks(hitl_solver* s, int32_t* ranks, int32_t* kind) {
  if (!s || !ranks) return fail(HITL_ERR_INVALID_ARGUMENT, "comm_ranks: null argument");
  int n = 1, k = 0;
  if (s->comm.allgather) { n = s->world; k = 2; }
  else if (s->nccl) {
    const ncclResult_t r = ncclCommCount(s->nccl, &n);
    if (r != ncclSuccess) return fail(HITL_ERR_COMM, std::string("ncclCommCount: ") + ncclGetErrorString(r));
    k = 1;
  }
  *ranks = n;
  if (kind) *kind = k;
  return HITL_OK;
}

int hitl_solver_set_partition(hitl_solver* s, int64_t global_begin, int64_t n_global_poses, int32_t halo_poses) {
  if (!s || global_begin < 0 || n_global_poses < 1 || global_begin >= n_global_poses) return fail(HITL_ERR_INVALID_ARGUMENT, "set_partition: bad range");
  if (s->world == 1 && (global_begin != 0)) return fail(HITL_ERR_INVALID_ARGUMENT, "set_partition: a single-GPU solver owns the whole map");
  if (s->world > 1 && ((s->rank == 0) != (global_begin == 0))) return fail(HITL_ERR_INVALID_ARGUMENT, "set_partition: rank 0 (and only rank 0) starts at pose 0");
  if (s->world > 1 && (halo_poses < 1 || (s->rank > 0 && global_begin < halo_poses))) return fail(HITL_ERR_INVALID_ARGUMENT, "set_partition: halo_poses must be >= 1 and not reach below pose 0");
  s->g_begin = global_begin;
  s->g_n = n_global_poses;
  s->halo_req = (s->world > 1) ? halo_poses : 0;
  s->hl = (s->world > 1 && s->rank > 0) ? halo_poses : 0;
  s->hh = (s->world > 1 && s->rank < s->world - 1) ? halo_poses : 0;
  s->n = 0;   // a map must be (re)installed for this range
  s->min_band_poses = 1;
  return HITL_OK;
}

int hitl_solver_set_partition_range(hitl_solver* s, int64_t global_begin, int64_t global_end, int64_t n_global_poses, int32_t halo_poses) {
  if (!s || global_begin < 0 || n_global_poses < 1 || global_end <= global_begin || global_end > n_global_poses)
    return fail(HITL_ERR_INVALID_ARGUMENT, "set_partition_range: bad range");
  if (s->world == 1 && (global_begin != 0 || global_end != n_global_poses)) return fail(HITL_ERR_INVALID_ARGUMENT, "set_partition_range: a single-GPU solver owns the whole map");
  if (s->world > 1 && ((s->rank == 0) != (global_begin == 0))) return fail(HITL_ERR_INVALID_ARGUMENT, "set_partition_range: rank 0 (and only rank 0) starts at pose 0");
  if (s->world > 1 && ((s->rank == s->world - 1) != (global_end == n_global_poses))) return fail(HITL_ERR_INVALID_ARGUMENT, "set_partition_range: the last rank (and only it) ends at the last pose");
  if (s->world > 1 && halo_poses == 0) return fail(HITL_ERR_INVALID_ARGUMENT, "set_partition_range: halo_poses must be >= 1, or < 0 for the whole map");
  // halo_poses < 0: the rank holds the scans of the WHOLE map and owns the rows of [begin, end) -- pose pairs may then reach
  // any distance across the cuts (the reference's all-pairs search, revisits); otherwise the halos are clipped at the map's ends
  const int64_t h = (halo_poses < 0) ? n_global_poses : (int64_t)halo_poses;
  s->g_begin = global_begin;
  s->g_n = n_global_poses;
  s->halo_req = (s->world > 1) ? (int)std::min<int64_t>(h, n_global_poses) : 0;
  s->hl = (s->world > 1) ? (int)std::min<int64_t>(h, global_begin) : 0;
  s->hh = (s->world > 1) ? (int)std::min<int64_t>(h, n_global_poses - global_end) : 0;
  s->n = 0;   // a map must be (re)installed for this range
  s->min_band_poses = 1;
  return HITL_OK;
}

int hitl_solver_set_map(hitl_solver* s, int64_t n_poses, const int64_t* obs_offset, const float* px, const float* py,
                        const float* nx, const float* ny, const float* cov9) {
  if (!s || !obs_offset || n_poses < 1) return fail(HITL_ERR_INVALID_ARGUMENT, "set_map: null solver/offsets or n_poses < 1");
  if (n_poses > (int64_t)1 << 30) return fail(HITL_ERR_UNSUPPORTED, "set_map: more than 2^30 poses");
  if (s->world > 1 && s->g_n == 0) return fail(HITL_ERR_NOT_READY, "set_map: call hitl_solver_set_partition first when world_size > 1");
  if (s->world == 1) { s->g_begin = 0; s->g_n = n_poses; s->hl = s->hh = 0; }
  s->min_band_poses = 1;
  // submap rank: n_poses = hl halo poses + owned poses + hh halo poses, i.e. global [g_begin - hl, ...)
  if (s->world > 1 && (n_poses <= s->hl + s->hh || s->g_begin - s->hl + n_poses > s->g_n)) return fail(HITL_ERR_INVALID_ARGUMENT, "set_map: the local range [begin - halo, end + halo) does not fit the global pose count");
  if (s->world > 1 && s->rank == s->world - 1 && s->g_begin - s->hl + n_poses != s->g_n) return fail(HITL_ERR_INVALID_ARGUMENT, "set_map: the last rank's range must end at the last global pose");
  const int64_t O = obs_offset[n_poses];
  if (obs_offset[0] != 0 || O < 0 || (O > 0 && (!px || !py || !nx || !ny))) return fail(HITL_ERR_INVALID_ARGUMENT, "set_map: bad offsets or null scan arrays");
  int max_scan = 0;
  for (int64_t i = 0; i < n_poses; ++i) {
    const int64_t m = obs_offset[i + 1] - obs_offset[i];
    if (m < 0) return fail(HITL_ERR_INVALID_ARGUMENT, "set_map: obs_offset must be non-decreasing");
    if (m > 65535) return fail(HITL_ERR_UNSUPPORTED, "set_map: a scan has more than 65535 points (match indices are packed u16)");
    max_scan = std::max<int>(max_scan, (int)m);
  }
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t st = s->stream;
  s->n = (int)n_poses; s->n_obs = O; s->max_scan = std::max(max_scan, 1);
  s->own_lo = s->hl; s->own_hi = (int)n_poses - s->hh;
  s->h_obs_off.assign(obs_offset, obs_offset + n_poses + 1);
  TRY(s->obs_off.upload(s->h_obs_off, st));
  {  // constant poses: global pose 0 (JointOptimization.cpp:824, :1197) and the halos
    std::vector<unsigned char> fx((size_t)n_poses, 0);
    for (int i = 0; i < (int)n_poses; ++i) fx[i] = (i < s->own_lo || i >= s->own_hi || (s->g_begin - s->hl + i) == 0) ? 1 : 0;
    TRY(s->fixed.upload(fx, st));
  }
  {  // the caller's SoA arrays are packed into one float4 per observation on the device
    DevArray<float> t0, t1, t2, t3;
    int rc = t0.upload(px, O, st);
    if (rc == HITL_OK) rc = t1.upload(py, O, st);
    if (rc == HITL_OK) rc = t2.upload(nx, O, st);
    if (rc == HITL_OK) rc = t3.upload(ny, O, st);
    if (rc == HITL_OK) rc = s->obs.resize((size_t)O);
    if (rc == HITL_OK) {
      launch_pack_obs(st, O, t0.p, t1.p, t2.p, t3.p, s->obs.p);
      if (hipStreamSynchronize(st) != hipSuccess) rc = fail(HITL_ERR_HIP, "pack_obs failed");
    }
    t0.release(); t1.release(); t2.release(); t3.release();
    if (rc != HITL_OK) return rc;
  }
  if (cov9) TRY(s->cov9.upload(cov9, (size_t)n_poses * 9, st));   // carried like covariances_; no live factor reads it
  HIP_TRY(hipStreamSynchronize(st));
  // a new map invalidates correspondences, human rows and the KD-trees
  s->kd_built = false; s->kd_grid_built = false; s->kd_full_depth = false; s->kd_cost_valid = false;
  s->adj_window = 0; s->adj_slots = 0; s->h_adj.clear();      // info_mat: no search on this map yet
  s->n_pairs = 0; s->n_matches = 0; s->pose_pair_off.n = 0;
  s->h_hc.clear();
  return ensure_empty_pair_tables(s);
}

int hitl_solver_set_human_constraints(hitl_solver* s, int64_t n, const hitl_human_constraint_t* rows) {
  if (!s || n < 0 || (n > 0 && !rows)) return fail(HITL_ERR_INVALID_ARGUMENT, "set_human_constraints: bad arguments");
  if (s->n <= 0) return fail(HITL_ERR_NOT_READY, "set_human_constraints before set_map");
  const int64_t own_lo = s->g_begin, own_hi = s->g_begin + (s->own_hi - s->own_lo);
  for (int64_t h = 0; h < n; ++h)
    if (rows[h].constrained_pose_id < 0 || rows[h].constrained_pose_id >= s->g_n || rows[h].anchor_pose_id < 0 || rows[h].anchor_pose_id >= s->g_n)
      return fail(HITL_ERR_INVALID_ARGUMENT, "set_human_constraints: pose id out of range");
  s->h_hc.clear();
  for (int64_t h = 0; h < n; ++h) {
    if (rows[h].constrained_pose_id < own_lo || rows[h].constrained_pose_id >= own_hi) continue;   // another rank's row
    hitl_human_constraint_t r = rows[h];
    r.constrained_pose_id = (int32_t)(r.constrained_pose_id - s->g_begin + s->hl);                  // local; the anchor stays global
    s->h_hc.push_back(r);
  }
  return HITL_OK;
}

int hitl_solver_find_correspondences(hitl_solver* s, const float* poses_xyt, int64_t window) {
  if (!s || !poses_xyt) return fail(HITL_ERR_INVALID_ARGUMENT, "find_correspondences: null argument");
  if (s->n <= 0) return fail(HITL_ERR_NOT_READY, "find_correspondences before set_map");
  // poses_xyt is the GLOBAL array; the local list starts at the halo pose (rank > 0)
  return find_correspondences(s, poses_xyt + 3 * (size_t)(s->g_begin - s->hl), window);
}

int hitl_solver_set_correspondences(hitl_solver* s, int64_t n_pairs, const int32_t* pair_src, const int32_t* pair_tgt,
                                    const int64_t* match_offset, const int32_t* src_idx, const int32_t* tgt_idx) {
  if (!s || n_pairs < 0) return fail(HITL_ERR_INVALID_ARGUMENT, "set_correspondences: bad arguments");
  if (s->n <= 0) return fail(HITL_ERR_NOT_READY, "set_correspondences before set_map");
  if (n_pairs > 0 && (!pair_src || !pair_tgt || !match_offset || !src_idx || !tgt_idx)) return fail(HITL_ERR_INVALID_ARGUMENT, "set_correspondences: null array");
  HIP_TRY(hipSetDevice(s->device));
  s->min_band_poses = 1;                       // a new correspondence set (see find_correspondences)
  const int P = (int)n_pairs;
  const int shift = s->hl - (int)s->g_begin;   // GLOBAL pose id -> local index
  for (int p = 0; p < P; ++p) {
    const int a = pair_src[p] + shift, b = pair_tgt[p] + shift;
    if (a < 0 || a >= s->n || b < 0 || b >= s->n || a == b) return fail(HITL_ERR_INVALID_ARGUMENT, "set_correspondences: pose index outside this solver's local range, or src == tgt");
    if (match_offset[p + 1] <= match_offset[p]) return fail(HITL_ERR_INVALID_ARGUMENT, "set_correspondences: every pair needs >= 1 match (residual_functors.h:783)");
    const int64_t ma = s->h_obs_off[a + 1] - s->h_obs_off[a], mb = s->h_obs_off[b + 1] - s->h_obs_off[b];
    for (int64_t m = match_offset[p]; m < match_offset[p + 1]; ++m)
      if (src_idx[m] < 0 || src_idx[m] >= ma || tgt_idx[m] < 0 || tgt_idx[m] >= mb) return fail(HITL_ERR_INVALID_ARGUMENT, "set_correspondences: point index out of range");
  }
  std::vector<int> order(P);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int u, int v) { return pair_src[u] < pair_src[v]; });   // shift-invariant
  s->h_perm = order;
  s->h_pair_src.resize(P); s->h_pair_tgt.resize(P); s->h_match_off.assign(P + 1, 0);
  const int64_t C = P ? match_offset[P] - match_offset[0] : 0;
  std::vector<uint32_t> packed((size_t)C + kMatchPad, 0u);   // k_linearize_stf prefetches index words unguarded
  int64_t w = 0;
  for (int q = 0; q < P; ++q) {
    const int p = order[q];
    s->h_pair_src[q] = pair_src[p] + shift; s->h_pair_tgt[q] = pair_tgt[p] + shift;
    for (int64_t m = match_offset[p]; m < match_offset[p + 1]; ++m) packed[w++] = (uint32_t)src_idx[m] | ((uint32_t)tgt_idx[m] << 16);
    s->h_match_off[q + 1] = w;
  }
  s->n_pairs = P; s->n_matches = C;
  s->adj_window = 0; s->adj_slots = 0; s->h_adj.clear();          // info_mat: the installed pairs
  hipStream_t st = s->stream;
  TRY(s->pair_src.upload(s->h_pair_src, st)); TRY(s->pair_tgt.upload(s->h_pair_tgt, st));
  TRY(s->match_off.upload(s->h_match_off, st)); TRY(s->matches.upload(packed, st));
  HIP_TRY(hipStreamSynchronize(st));
  return build_pair_structures(s);
}

int hitl_solver_get_kd_trees(hitl_solver* s, float* pts4, uint32_t* meta) {
  if (!s) return fail(HITL_ERR_INVALID_ARGUMENT, "null solver");
  if (s->n <= 0) return fail(HITL_ERR_INVALID_ARGUMENT, "get_kd_trees: no map set");
  HIP_TRY(hipSetDevice(s->device));
  TRY(ensure_kd_trees(s));
  if (pts4 && s->n_obs > 0) HIP_TRY(hipMemcpyAsync(pts4, s->kd_pts.p, sizeof(float4) * (size_t)s->n_obs, hipMemcpyDeviceToHost, s->stream));
  if (meta && s->n_obs > 0) HIP_TRY(hipMemcpyAsync(meta, s->kd_meta.p, sizeof(uint32_t) * (size_t)s->n_obs, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return HITL_OK;
}

int hitl_solver_get_info_mat(hitl_solver* s, int32_t mode, int64_t* n_entries, int32_t* rows, int32_t* cols) {
  int rc = check_ready(s, mode);
  if (rc != HITL_OK) return rc;
  if (!n_entries || ((rows == nullptr) != (cols == nullptr))) return fail(HITL_ERR_INVALID_ARGUMENT, "get_info_mat: n_entries and both or neither of rows / cols");
  if (s->dist_mode()) return fail(HITL_ERR_UNSUPPORTED, "get_info_mat is a single-GPU entry point");
  HIP_TRY(hipSetDevice(s->device));
  std::vector<std::pair<int, int>> e;
  const int n = s->n;
  if (mode != HITL_MODE_STF_ONLY) {
    for (int i = 1; i < n; ++i) e.emplace_back(i - 1, i);                                   // AddOdometryConstraints :744-745
    for (const hitl_human_constraint_t& r : s->h_hc) {                                       // AddHumanConstraints :978-979 (every row, whatever its type)
      const int a = r.anchor_pose_id, c = r.constrained_pose_id;
      if (a >= 0 && a < n && c >= 0 && c < n) e.emplace_back(std::min(a, c), std::max(a, c));
    }
  }
  if (mode != HITL_MODE_ODOM_HC) {
    if (s->adj_window > 0) {          // windowed search: the candidate counts are still on the device
      std::vector<int> counts((size_t)n * s->adj_slots);
      if (!counts.empty()) {
        HIP_TRY(hipMemcpyAsync(counts.data(), s->d_cand_counts.p, sizeof(int) * counts.size(), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
      }
      for (int i = 0; i < n; ++i) {
        const int j0 = std::max(0, i - s->adj_window), j1 = std::min(n - 1, i + s->adj_window);
        int slot = 0;
        for (int j = j0; j <= j1; ++j) {
          if (j == i) continue;
          if (counts[(size_t)i * s->adj_slots + slot] > 0) e.emplace_back(std::min(i, j), std::max(i, j));
          ++slot;
        }
      }
    } else if (!s->h_adj.empty()) {
      e.insert(e.end(), s->h_adj.begin(), s->h_adj.end());
    } else {                          // an installed set (hitl_solver_set_correspondences): its pairs
      TRY(ensure_host_pairs(s));
      for (int p = 0; p < s->n_pairs; ++p) e.emplace_back(std::min(s->h_pair_src[p], s->h_pair_tgt[p]), std::max(s->h_pair_src[p], s->h_pair_tgt[p]));
    }
  }
  std::sort(e.begin(), e.end());
  e.erase(std::unique(e.begin(), e.end()), e.end());
  if (rows) {
    if (*n_entries < (int64_t)e.size()) return fail(HITL_ERR_INVALID_ARGUMENT, "get_info_mat: arrays too small (call with NULL arrays for the count)");
    for (size_t k = 0; k < e.size(); ++k) { rows[k] = e[k].first; cols[k] = e[k].second; }
  }
  *n_entries = (int64_t)e.size();
  return HITL_OK;
}

int hitl_solver_get_correspondence_counts(hitl_solver* s, int64_t* n_pairs, int64_t* n_matches) {
  if (!s) return fail(HITL_ERR_INVALID_ARGUMENT, "null solver");
  if (n_pairs) *n_pairs = s->n_pairs;
  if (n_matches) *n_matches = s->n_matches;
  return HITL_OK;
}

int hitl_solver_get_correspondences(hitl_solver* s, int32_t* pair_src, int32_t* pair_tgt, int64_t* match_offset,
                                    int32_t* src_idx, int32_t* tgt_idx) {
  if (!s || !match_offset) return fail(HITL_ERR_INVALID_ARGUMENT, "get_correspondences: null argument");
  HIP_TRY(hipSetDevice(s->device));
  const int P = s->n_pairs;
  TRY(ensure_host_pairs(s));
  std::vector<uint32_t> packed((size_t)s->n_matches);
  if (s->n_matches) {
    HIP_TRY(hipMemcpyAsync(packed.data(), s->matches.p, packed.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  // return in the caller's original pair order
  std::vector<int> inv(P);
  for (int q = 0; q < P; ++q) inv[s->h_perm[q]] = q;
  int64_t w = 0;
  match_offset[0] = 0;
  for (int p = 0; p < P; ++p) {
    const int q = inv[p];
    pair_src[p] = s->h_pair_src[q] - (s->hl - (int)s->g_begin); pair_tgt[p] = s->h_pair_tgt[q] - (s->hl - (int)s->g_begin);
    for (int64_t m = s->h_match_off[q]; m < s->h_match_off[q + 1]; ++m) {
      src_idx[w] = (int32_t)(packed[m] & 0xffffu); tgt_idx[w] = (int32_t)(packed[m] >> 16); ++w;
    }
    match_offset[p + 1] = w;
  }
  return HITL_OK;
}

int hitl_solver_solve(hitl_solver* s, int32_t mode, const float* poses_in_xyt, float* poses_out_xyt, hitl_summary* summary) {
  TRY(check_ready(s, mode));
  if (!poses_in_xyt || !poses_out_xyt) return fail(HITL_ERR_INVALID_ARGUMENT, "solve: null pose array");
  if (s->world == 1) {
    if (mode == HITL_MODE_STF_ONLY && s->n_pairs == 0) return fail(HITL_ERR_NOT_READY, "mode STF_ONLY needs correspondences (find_/set_correspondences)");
    return lm_solve(s, mode, poses_in_xyt, poses_in_xyt, poses_out_xyt, summary);
  }
  // submap rank: poses are GLOBAL arrays; this rank solves [halo pose +] its owned range
  // Mode B (the dormant STF-only solve) runs on submap ranks since round 3.  Without the odometry chain the ranks are coupled
  // through the scan-to-scan pairs across the cuts only, the problem is ill-conditioned or singular, and -- as on one GPU -- its
  // accept / reject path depends on the rounding of the linear solver: the ranks reach the cost level of
  // the single-GPU solve, not its iteration sequence.
  const size_t n = (size_t)s->n, first = (size_t)(s->g_begin - s->hl);
  std::vector<float> in(poses_in_xyt + 3 * first, poses_in_xyt + 3 * (first + n)), out(3 * n);
  hitl_summary S = {};
  const int rc = lm_solve(s, mode, in.data(), poses_in_xyt, out.data(), &S);
  if (poses_out_xyt != poses_in_xyt) std::memcpy(poses_out_xyt, poses_in_xyt, sizeof(float) * 3 * (size_t)s->g_n);
  if (rc == HITL_OK || rc == HITL_ERR_SOLVER_FAILURE)
    std::memcpy(poses_out_xyt + 3 * (size_t)s->g_begin, out.data() + 3 * (size_t)s->own_lo, sizeof(float) * 3 * (size_t)(s->own_hi - s->own_lo));
  if (rc != HITL_OK && rc != HITL_ERR_SOLVER_FAILURE) { if (summary) *summary = S; return rc; }   // every rank left at the same point (prepare)
  // problem-size counters are global sums
  double cnt[4] = {(double)S.num_residuals, (double)S.num_hc_residuals, (double)S.num_pairs, (double)S.num_matches};
  std::vector<double> all((size_t)4 * s->world);
  if (comm_allgather(s, cnt, all.data(), sizeof(cnt)) == HITL_OK) {
    double t[4] = {0, 0, 0, 0};
    for (int r = 0; r < s->world; ++r) for (int k = 0; k < 4; ++k) t[k] += all[(size_t)4 * r + k];
    S.num_residuals = (int32_t)t[0]; S.num_hc_residuals = (int32_t)t[1]; S.num_pairs = (int64_t)t[2]; S.num_matches = (int64_t)t[3];
  }
  if (summary) *summary = S;
  return rc;
}

int hitl_solver_linearize(hitl_solver* s, int32_t mode, const float* poses_ref_xyt, const double* x, double* cost,
                          double* gradient, double* pair_residuals, double* pair_jacobians) {
  TRY(check_ready(s, mode));
  if (!poses_ref_xyt || !x) return fail(HITL_ERR_INVALID_ARGUMENT, "linearize: null pose array");
  if (s->world > 1) return fail(HITL_ERR_UNSUPPORTED, "linearize/evaluate are single-GPU entry points");
  FactorView fv;
  TRY(prepare(s, mode, poses_ref_xyt, poses_ref_xyt, &fv));
  hipStream_t st = s->stream;
  const int n3 = 3 * s->n;
  HIP_TRY(hipMemcpyAsync(s->x.p, x, sizeof(double) * n3, hipMemcpyHostToDevice, st));
  LinResult lr;
  TRY(run_linearize(s, fv, s->x.p, 0, &lr));
  if (cost) *cost = lr.cost;
  if (gradient) HIP_TRY(hipMemcpyAsync(gradient, s->gvec[0].p, sizeof(double) * n3, hipMemcpyDeviceToHost, st));
  if (fv.with_stf && (pair_residuals || pair_jacobians)) {
    TRY(ensure_host_pairs(s));
    std::vector<double> rj((size_t)s->n_pairs * kPairRJ);
    HIP_TRY(hipMemcpyAsync(rj.data(), s->pairRJ[0].p, rj.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int q = 0; q < s->n_pairs; ++q) {
      const int p = s->h_perm[q];
      if (pair_residuals) { pair_residuals[2 * p] = rj[(size_t)q * kPairRJ]; pair_residuals[2 * p + 1] = rj[(size_t)q * kPairRJ + 1]; }
      if (pair_jacobians) std::memcpy(pair_jacobians + 12 * (size_t)p, &rj[(size_t)q * kPairRJ + 2], 12 * sizeof(double));
    }
  }
  HIP_TRY(hipStreamSynchronize(st));
  return HITL_OK;
}

int hitl_solver_evaluate(hitl_solver* s, int32_t mode, const float* poses_xyt, double* cost, double* gradient) {
  TRY(check_ready(s, mode));
  if (!poses_xyt) return fail(HITL_ERR_INVALID_ARGUMENT, "evaluate: null pose array");
  if (s->world > 1) {
    // submap rank (collective: every rank calls it): poses_xyt is the GLOBAL array; cost is the global cost (every factor is
    // counted by the rank owning it); gradient (3 * n_global doubles) receives this rank's OWNED rows, zeros elsewhere
    HIP_TRY(hipSetDevice(s->device));
    const size_t n = (size_t)s->n, first = (size_t)(s->g_begin - s->hl);
    std::vector<float> in(poses_xyt + 3 * first, poses_xyt + 3 * (first + n));
    std::vector<double> xl(3 * n);
    for (size_t t = 0; t < xl.size(); ++t) xl[t] = (double)in[t];
    FactorView fv;
    TRY(prepare(s, mode, in.data(), poses_xyt, &fv));
    HIP_TRY(hipMemcpyAsync(s->x.p, xl.data(), sizeof(double) * xl.size(), hipMemcpyHostToDevice, s->stream));
    LinResult lr;
    TRY(run_linearize_dist(s, fv, s->x.p, 0, &lr));
    if (cost) *cost = lr.cost;
    if (gradient) {
      std::memset(gradient, 0, sizeof(double) * 3 * (size_t)s->g_n);
      const size_t own = (size_t)(s->own_hi - s->own_lo);
      HIP_TRY(hipMemcpyAsync(gradient + 3 * (size_t)s->g_begin, s->gvec[0].p + 3 * (size_t)s->own_lo, sizeof(double) * 3 * own, hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(hipStreamSynchronize(s->stream));
    }
    return HITL_OK;
  }
  std::vector<double> x((size_t)3 * s->n);
  for (size_t t = 0; t < x.size(); ++t) x[t] = (double)poses_xyt[t];
  return hitl_solver_linearize(s, mode, poses_xyt, x.data(), cost, gradient, nullptr, nullptr);
}

int hitl_solver_evaluate_jacobian(hitl_solver* s, const float* poses_xyt, double* cost, double* residuals, int32_t* crs_rows,
                                  int32_t* crs_cols, double* crs_values) {
  TRY(check_ready(s, HITL_MODE_STF_ONLY));
  if (!poses_xyt) return fail(HITL_ERR_INVALID_ARGUMENT, "evaluate_jacobian: null pose array");
  if (s->world > 1) return fail(HITL_ERR_UNSUPPORTED, "evaluate_jacobian is a single-GPU entry point");
  const int P = s->n_pairs;
  std::vector<double> x((size_t)3 * s->n), pr((size_t)2 * std::max(P, 1)), pj((size_t)12 * std::max(P, 1));
  for (size_t t = 0; t < x.size(); ++t) x[t] = (double)poses_xyt[t];
  double c = 0.0;
  if (P > 0) TRY(hitl_solver_linearize(s, HITL_MODE_STF_ONLY, poses_xyt, x.data(), &c, nullptr, pr.data(), pj.data()));
  if (cost) *cost = c;
  if (residuals) std::memcpy(residuals, pr.data(), sizeof(double) * 2 * (size_t)P);
  // caller order: pair p of the caller is internal pair q with h_perm[q] == p
  TRY(ensure_host_pairs(s));
  std::vector<int> inv((size_t)P);
  for (int q = 0; q < P; ++q) inv[s->h_perm[q]] = q;
  for (int p = 0; p < P; ++p) {
    const int q = inv[p];
    const int a = s->h_pair_src[q], b = s->h_pair_tgt[q];
    for (int r = 0; r < 2; ++r) {
      const size_t row = (size_t)2 * p + r, at = 6 * row;
      if (crs_rows) crs_rows[row] = (int32_t)at;
      const double* J = &pj[(size_t)12 * p + 6 * r];            // (x0,y0,t0 | x1,y1,t1) = (source | target)
      const int first = (a < b) ? 0 : 3, second = 3 - first;    // ascending column indices within the row
      const int ca = (a < b) ? a : b, cb = (a < b) ? b : a;
      for (int k = 0; k < 3; ++k) {
        if (crs_cols) { crs_cols[at + k] = 3 * ca + k; crs_cols[at + 3 + k] = 3 * cb + k; }
        if (crs_values) { crs_values[at + k] = J[first + k]; crs_values[at + 3 + k] = J[second + k]; }
      }
    }
  }
  if (crs_rows) crs_rows[(size_t)2 * P] = (int32_t)(12 * (size_t)P);
  return HITL_OK;
}

// ---- pose covariances: what ceres::Covariance + the robot-frame rotation produce at
// episodic_non_markov_localization/vector_mapping.cpp:2772-2812, from the cyclic-reduction factors ----

int hitl_solver_pose_covariances(hitl_solver* s, int32_t mode, const float* poses_xyt, double* cov9) {
  TRY(check_ready(s, mode));
  if (!poses_xyt || !cov9) return fail(HITL_ERR_INVALID_ARGUMENT, "pose_covariances: null argument");
  if (s->world > 1) {
    // submap rank (collective; mode A or C): poses_xyt is the GLOBAL array, cov9 (9 * n_global) receives the blocks of the
    // rank's OWNED poses (zeros elsewhere): the blocks of the GLOBAL inverse, hitl_lm.cpp dist_cov_correction
    if (mode == HITL_MODE_STF_ONLY) return fail(HITL_ERR_UNSUPPORTED, "submap ranks need the odometry chain (mode A or C)");
    HIP_TRY(hipSetDevice(s->device));
    const size_t nl = (size_t)s->n, first = (size_t)(s->g_begin - s->hl);
    std::vector<float> in(poses_xyt + 3 * first, poses_xyt + 3 * (first + nl));
    FactorView fvd;
    TRY(prepare(s, mode, in.data(), poses_xyt, &fvd));
    hipStream_t sd = s->stream;
    std::vector<double> hx((size_t)s->npad, 0.0);
    for (size_t t = 0; t < 3 * nl; ++t) hx[t] = (double)in[t];
    HIP_TRY(hipMemcpyAsync(s->x.p, hx.data(), sizeof(double) * s->npad, hipMemcpyHostToDevice, sd));
    LinResult lrd;
    TRY(run_linearize(s, fvd, s->x.p, 0, &lrd));
    launch_jacobi_scale(sd, s->n, s->npad, fvd.band_poses, s->Hband[0].p, s->opt.jacobi_scaling, s->fixed.p, s->scale.p);
    const CrView crd = s->cr_view();
    DevArray<double> Sd, So, d_out;
    DevArray<float2> d_rot;
    std::vector<float2> rot(nl);
    for (size_t i = 0; i < nl; ++i) { const float a = -in[3 * i + 2]; rot[i] = make_float2(cosf(a), sinf(a)); }
    TRY(Sd.resize((size_t)crd.nsb * crd.B * crd.B)); TRY(So.resize((size_t)crd.nsb * crd.B * crd.B)); TRY(d_out.resize(9 * nl));
    TRY(d_rot.upload(rot, sd));
    if (s->dist_sep) {
      // separator form: tiles by chain node (node k = owned poses own_lo + k * Wb ..), extraction over the owned range
      TRY(dist_cov_sep(s, fvd, 0, Sd.p, So.p));
      const size_t lo = (size_t)s->own_lo;
      launch_cov_extract(sd, s->own_hi - s->own_lo, fvd.band_poses, crd.B, Sd.p, s->scale.p + 3 * lo, s->fixed.p + lo, d_rot.p + lo, d_out.p + 9 * lo);
    } else {
    TRY(dist_cov_correction(s, fvd, 0, Sd.p, So.p));
    // halo poses are constant locally: the extraction treats them like pose 0 (zero block); only the owned range goes out
    launch_cov_extract(sd, s->n, fvd.band_poses, crd.B, Sd.p, s->scale.p, s->fixed.p, d_rot.p, d_out.p);
    }
    std::memset(cov9, 0, sizeof(double) * 9 * (size_t)s->g_n);
    const size_t own = (size_t)(s->own_hi - s->own_lo);
    HIP_TRY(hipMemcpyAsync(cov9 + 9 * (size_t)s->g_begin, d_out.p + 9 * (size_t)s->own_lo, sizeof(double) * 9 * own, hipMemcpyDeviceToHost, sd));
    HIP_TRY(hipStreamSynchronize(sd));
    for (size_t t = 9 * (size_t)s->g_begin; t < 9 * ((size_t)s->g_begin + own); ++t)
      if (!(cov9[t] == cov9[t])) return fail(HITL_ERR_SOLVER_FAILURE, "pose_covariances: J^T J is singular at these poses");
    return HITL_OK;
  }
  FactorView fv;
  TRY(prepare(s, mode, poses_xyt, poses_xyt, &fv));
  if (fv.n_offband > 0) return pose_covariances_dense(s, fv, poses_xyt, cov9);
  hipStream_t st = s->stream;
  const int n = s->n, n3 = 3 * n;
  std::vector<double> hx((size_t)s->npad, 0.0);
  for (int t = 0; t < n3; ++t) hx[t] = (double)poses_xyt[t];
  HIP_TRY(hipMemcpyAsync(s->x.p, hx.data(), sizeof(double) * s->npad, hipMemcpyHostToDevice, st));
  LinResult lr;
  TRY(run_linearize(s, fv, s->x.p, 0, &lr));
  launch_jacobi_scale(st, n, s->npad, fv.band_poses, s->Hband[0].p, s->opt.jacobi_scaling, s->fixed.p, s->scale.p);
  const CrView cr = s->cr_view();
  // J^T J itself: no trust-region diagonal (clamp(..) / infinity == 0)
  launch_build_system(st, n, fv.band_poses, s->lin_view(0), s->scale.p, s->fixed.p, INFINITY, s->opt.min_lm_diagonal,
                      s->opt.max_lm_diagonal, cr, s->rhs.p);
  launch_cr_factor(st, cr);
  DevArray<double> Sd, So, d_out;
  DevArray<float2> d_rot;
  std::vector<float2> rot((size_t)n);
  for (int i = 0; i < n; ++i) { const float a = -poses_xyt[3 * i + 2]; rot[i] = make_float2(cosf(a), sinf(a)); }   // Rotation2Df(-angle)
  int rc = Sd.resize((size_t)cr.nsb * cr.B * cr.B);
  if (rc == HITL_OK) rc = So.resize((size_t)cr.nsb * cr.B * cr.B);
  if (rc == HITL_OK) rc = d_out.resize((size_t)9 * n);
  if (rc == HITL_OK) rc = d_rot.upload(rot, st);
  if (rc == HITL_OK) {
    launch_cr_selinv(st, cr, Sd.p, So.p);
    launch_cov_extract(st, n, fv.band_poses, cr.B, Sd.p, s->scale.p, s->fixed.p, d_rot.p, d_out.p);
    if (hipMemcpyAsync(cov9, d_out.p, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) rc = fail(HITL_ERR_HIP, "pose_covariances: kernels failed");
  }
  if (rc == HITL_OK)
    for (size_t t = 0; t < (size_t)9 * n; ++t)
      if (!(cov9[t] == cov9[t])) { rc = fail(HITL_ERR_SOLVER_FAILURE, "pose_covariances: J^T J is singular at these poses"); break; }
  Sd.release(); So.release(); d_out.release(); d_rot.release();
  return rc;
}

// ---- observation scans either side of the solve -------------------------
namespace {
int obs_scan_prepare(hitl_solver* s, const float* poses, const float** T4) {
  if (!s || !poses) return fail(HITL_ERR_INVALID_ARGUMENT, "observation scan: null argument");
  if (s->n <= 0) return fail(HITL_ERR_NOT_READY, "observation scan before set_map");
  HIP_TRY(hipSetDevice(s->device));
  // submap rank: poses is the GLOBAL array; the rank's resident scans (halos included) start at global pose g_begin - hl
  return solver_pose_rt(s, poses + 3 * (size_t)(s->g_begin - s->hl), T4);
}
}  // namespace

// world x of every observation, then world y, in a buffer the solver keeps (no 160 MB allocation per call)
static int transform_scans_resident(hitl_solver* s, const float* poses_xyt) {
  const float* T4 = nullptr;
  int rc = obs_scan_prepare(s, poses_xyt, &T4);
  if (rc == HITL_OK) rc = s->world_xy.resize((size_t)2 * std::max<int64_t>(s->n_obs, 1));
  if (rc == HITL_OK) {
    launch_transform(s->stream, s->map_view(), T4, s->world_xy.p, s->world_xy.p + s->n_obs);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) rc = fail(HITL_ERR_HIP, "transform_scans failed");
  }
  return rc;
}

// Device -> PAGEABLE host memory, large: the runtime's own path (a bounce buffer and one copying thread) delivers ~16 GB/s (9.6 ms for
// the 160 MB of C4's display clouds).  Here the data goes through two page-locked staging buffers the solver keeps, 8 MB pieces: the
// DMA of piece k + 1 runs while a few threads copy piece k out (each its slice of every piece).  Destinations that ARE page-locked
// (hitl_host_alloc) take one direct DMA as before.
static bool is_pinned_host(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}

static int staged_d2h(hitl_solver* s, char* dst, const char* src_dev, size_t bytes) {
  constexpr size_t kPiece = (size_t)8 << 20;
  const int n_pieces = (int)((bytes + kPiece - 1) / kPiece);
  TRY(s->h_stage[0].resize(kPiece)); TRY(s->h_stage[1].resize(kPiece));
  for (int k = 0; k < 2; ++k) if (!s->ev_stage[k]) HIP_TRY(hipEventCreateWithFlags(&s->ev_stage[k], hipEventDisableTiming));
  const int nt = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
  std::atomic<int> ready(0);                 // pieces that have arrived in their staging buffer
  std::vector<std::atomic<int>> done(n_pieces);
  for (auto& d : done) d.store(0);
  auto piece_len = [&](int k) { return std::min(kPiece, bytes - (size_t)k * kPiece); };
  std::vector<std::thread> team;
  for (int t = 0; t < nt; ++t)
    team.emplace_back([&, t]() {
      for (int k = 0; k < n_pieces; ++k) {
        while (ready.load(std::memory_order_acquire) <= k) sched_yield();
        const size_t len = piece_len(k), a = len * t / nt, b = len * (t + 1) / nt;
        std::memcpy(dst + (size_t)k * kPiece + a, s->h_stage[k & 1].p + a, b - a);
        done[k].fetch_add(1, std::memory_order_release);
      }
    });
  int rc = HITL_OK;
  for (int k = 0; k < n_pieces; ++k) {
    if (k >= 2) while (done[k - 2].load(std::memory_order_acquire) < nt) sched_yield();     // its staging buffer has been copied out
    if (rc == HITL_OK && (hipMemcpyAsync(s->h_stage[k & 1].p, src_dev + (size_t)k * kPiece, piece_len(k), hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
                          hipEventRecord(s->ev_stage[k & 1], s->stream) != hipSuccess))
      rc = fail(HITL_ERR_HIP, "staged copy to the host: queueing a piece failed");
    if (k >= 1) {
      if (rc == HITL_OK && hipEventSynchronize(s->ev_stage[(k - 1) & 1]) != hipSuccess) rc = fail(HITL_ERR_HIP, "staged copy to the host: a piece did not arrive");
      ready.store(k, std::memory_order_release);           // pieces 0 .. k-1 are in (after an error the team still runs to its end: harmless bytes)
    }
  }
  if (rc == HITL_OK && hipEventSynchronize(s->ev_stage[(n_pieces - 1) & 1]) != hipSuccess) rc = fail(HITL_ERR_HIP, "staged copy to the host: the last piece did not arrive");
  ready.store(n_pieces, std::memory_order_release);
  for (auto& th : team) th.join();
  return rc;
}

int hitl_solver_transform_scans(hitl_solver* s, const float* poses_xyt, float* world_x, float* world_y) {
  if (!world_x || !world_y) return fail(HITL_ERR_INVALID_ARGUMENT, "transform_scans: null output");
  int rc = transform_scans_resident(s, poses_xyt);
  if (rc != HITL_OK || s->n_obs <= 0) return rc;
  const size_t bytes = sizeof(float) * (size_t)s->n_obs;
  if (bytes >= ((size_t)32 << 20) && !is_pinned_host(world_x) && !is_pinned_host(world_y)) {
    rc = staged_d2h(s, reinterpret_cast<char*>(world_x), reinterpret_cast<const char*>(s->world_xy.p), bytes);
    if (rc == HITL_OK) rc = staged_d2h(s, reinterpret_cast<char*>(world_y), reinterpret_cast<const char*>(s->world_xy.p + s->n_obs), bytes);
    return rc;
  }
  if (hipMemcpyAsync(world_x, s->world_xy.p, bytes, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
      hipMemcpyAsync(world_y, s->world_xy.p + s->n_obs, bytes, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
      hipStreamSynchronize(s->stream) != hipSuccess) rc = fail(HITL_ERR_HIP, "transform_scans: the copy to the host failed");
  return rc;
}

int hitl_solver_transform_scans_device(hitl_solver* s, const float* poses_xyt, const float** d_world_x, const float** d_world_y) {
  if (!d_world_x || !d_world_y) return fail(HITL_ERR_INVALID_ARGUMENT, "transform_scans_device: null output");
  *d_world_x = *d_world_y = nullptr;
  const int rc = transform_scans_resident(s, poses_xyt);
  if (rc == HITL_OK) { *d_world_x = s->world_xy.p; *d_world_y = s->world_xy.p + s->n_obs; }
  return rc;
}

void* hitl_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes) != hipSuccess) return nullptr;
  return p;
}

void hitl_host_free(void* p) { if (p) (void)hipHostFree(p); }

int hitl_solver_verify_points(hitl_solver* s, const float* poses_xyt, int32_t n_points, const float* points_xy,
                              float threshold, int32_t* seen) {
  if (n_points < 0 || n_points > 16 || (n_points > 0 && (!points_xy || !seen))) return fail(HITL_ERR_INVALID_ARGUMENT, "verify_points: 0..16 points");
  const float* T4 = nullptr;
  DevArray<int> d_seen;
  int rc = obs_scan_prepare(s, poses_xyt, &T4);
  if (rc == HITL_OK) rc = d_seen.resize(16);
  if (rc == HITL_OK) rc = d_seen.zero(s->stream);
  if (rc == HITL_OK) {
    launch_verify(s->stream, s->map_view(), T4, n_points, points_xy, threshold, d_seen.p);
    int h[16];
    if (hipMemcpyAsync(h, d_seen.p, sizeof(h), hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
        hipStreamSynchronize(s->stream) != hipSuccess) rc = fail(HITL_ERR_HIP, "verify_points failed");
    else for (int q = 0; q < n_points; ++q) seen[q] = h[q];
  }
  d_seen.release();
  return rc;
}

int hitl_solver_observation_sets(hitl_solver* s, const float* poses_xyt, const float* seg_xy8, double threshold,
                                 int32_t* counts, uint8_t* mask) {
  if (!seg_xy8 || !counts) return fail(HITL_ERR_INVALID_ARGUMENT, "observation_sets: null argument");
  const float* T4 = nullptr;
  DevArray<int> d_cnt;
  DevArray<unsigned char> d_mask;
  int rc = obs_scan_prepare(s, poses_xyt, &T4);
  if (rc == HITL_OK) rc = d_cnt.resize((size_t)2 * s->n);
  if (rc == HITL_OK) rc = d_cnt.zero(s->stream);
  if (rc == HITL_OK && mask) rc = d_mask.resize((size_t)s->n_obs);
  if (rc == HITL_OK) {
    launch_obs_sets(s->stream, s->map_view(), T4, seg_xy8, threshold, d_cnt.p, mask ? d_mask.p : nullptr);
    hipError_t e = hipMemcpyAsync(counts, d_cnt.p, sizeof(int) * 2 * (size_t)s->n, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess && mask) e = hipMemcpyAsync(mask, d_mask.p, (size_t)s->n_obs, hipMemcpyDeviceToHost, s->stream);
    if (e != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) rc = fail(HITL_ERR_HIP, "observation_sets failed");
  }
  d_cnt.release(); d_mask.release();
  return rc;
}

int hitl_comm_unique_id(void* out128) {
  if (!out128) return fail(HITL_ERR_INVALID_ARGUMENT, "null buffer");
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return fail(HITL_ERR_COMM, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
  std::memcpy(out128, &id, 128);
  return HITL_OK;
}

int hitl_partition_range(int64_t n_poses, const int64_t* obs_offset, int32_t rank, int32_t world_size, int64_t* begin,
                         int64_t* end) {
  return hitl_partition_range_aligned(n_poses, obs_offset, rank, world_size, 1, begin, end);
}

int hitl_partition_range_aligned(int64_t n_poses, const int64_t* obs_offset, int32_t rank, int32_t world_size, int32_t align,
                                 int64_t* begin, int64_t* end) {
  // contiguous pose ranges balanced by observation count, cuts on multiples of `align`
  if (n_poses < 1 || !obs_offset || world_size < 1 || rank < 0 || rank >= world_size || !begin || !end || align < 1)
    return fail(HITL_ERR_INVALID_ARGUMENT, "partition_range: bad arguments");
  const int64_t total = obs_offset[n_poses] - obs_offset[0];
  // every rank computes the whole cut list, so neighbouring ranges always abut
  std::vector<int64_t> cuts(world_size + 1, 0);
  cuts[world_size] = n_poses;
  for (int r = 1; r < world_size; ++r) {
    int64_t c;
    if (total <= 0) {
      c = n_poses * r / world_size;
    } else {
      const double goal = (double)total * (double)r / (double)world_size;
      c = std::lower_bound(obs_offset, obs_offset + n_poses + 1, (int64_t)std::llround(goal) + obs_offset[0]) - obs_offset;
    }
    if (align > 1 && n_poses >= (int64_t)2 * align * world_size) {
      // nearest multiple, at least two blocks of `align` behind the previous cut and room for two per rank still to come
      c = (c + align / 2) / align * align;
      c = std::max<int64_t>(c, cuts[r - 1] + 2 * (int64_t)align);
      c = std::min<int64_t>(c, (n_poses - 2 * (int64_t)align * (world_size - r)) / align * align);
    }
    if (n_poses >= world_size) c = std::min<int64_t>(std::max<int64_t>(c, cuts[r - 1] + 1), n_poses - (world_size - r));
    else c = std::min<int64_t>(std::max<int64_t>(c, cuts[r - 1]), n_poses);
    cuts[r] = c;
  }
  const int64_t b = cuts[rank], e = cuts[rank + 1];
  *begin = b; *end = e;
  return HITL_OK;
}

}  // extern "C"
