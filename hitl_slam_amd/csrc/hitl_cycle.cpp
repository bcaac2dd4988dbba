// hitl_cycle.cpp -- host side and C ABI of the correction cycle around the batch re-solve:
//   hitl_solver_fit_segments              == EMInput::AutomaticEndpointAdjustment + SegFitEM (EMinput.cpp:153-250)
//   hitl_order_user_input                 == EMInput::OrderAndFilterUserInput               (EMinput.cpp:325-455)
//   hitl_solver_apply_explicit_correction == AppExpCorrect::Run                 (ApplyExplicitCorrection.cpp:150-494)
//   hitl_solver_backprop                  == Backprop::Run                                   (Backprop.cpp:98-213)
//   hitl_normalize_angles                 == the atan2(sin, cos) loop of HitLSLAM::Run       (HitLSLAM.cpp:442-447)
// Every O(N*M), O(N), O(span^2) and O(anchors x corrected) step is a HIP kernel (hitl_obs.hip, hitl_cycle.hip); the
// host keeps the index logic and the O(1)/O(N) scalar set-up whose float transcendental values must be the C
// library's (acosf, atan2f, cosf, sinf -- what the reference's unqualified calls on float resolve to), so that the
// fp32 results stay bit-identical to the CPU restatement.  No CPU fallback: without a device these return an error.

#include <cmath>
#include <cstring>

#include "hitl_host.h"

using namespace hitl;

namespace {

struct FitEval { double cost, g, h; };

// one evaluation of the 1-parameter problem of SegFitEM at theta: residuals and derivatives on the device,
// per-slice partial sums added here in slice order
int fit_eval(const SolverCtx& cx, const float2* pts, int64_t m, double icx, double icy, double len, double theta,
             DevArray<double>* partial, FitEval* out) {
  out->cost = out->g = out->h = 0.0;
  if (m == 0) return HITL_OK;
  double ax = cos(theta), ay = sin(theta);
  const double inv = 1.0 / sqrt(ax * ax + ay * ay);   // alpha.normalize() on Jets: f * (1 / g)
  ax *= inv; ay *= inv;
  const int g = fit_eval_blocks(m);
  TRY(partial->resize((size_t)3 * g));
  launch_fit_eval(cx.stream, pts, m, icx, icy, len, ax, ay, partial->p);
  std::vector<double> h((size_t)3 * g);
  HIP_TRY(hipMemcpyAsync(h.data(), partial->p, h.size() * sizeof(double), hipMemcpyDeviceToHost, cx.stream));
  HIP_TRY(hipStreamSynchronize(cx.stream));
  double c = 0, gg = 0, hh = 0;
  for (int b = 0; b < g; ++b) { c += h[3 * b]; gg += h[3 * b + 1]; hh += h[3 * b + 2]; }
  out->cost = 0.5 * c; out->g = gg; out->h = hh;
  return HITL_OK;
}

// ceres::Solve at EMinput.cpp:178 (DENSE_QR, max_num_iterations 25, everything else default): the same trust-region
// rules as the pose solve on a 1x1 system.
int seg_fit_em(const SolverCtx& cx, const float2* pts, int64_t m, const float* sp, DevArray<double>* partial, float ep[4], int* iterations) {
  const double p1x = sp[0], p1y = sp[1], p2x = sp[2], p2y = sp[3];
  const double icx = (p1x + p2x) / 2.0, icy = (p1y + p2y) / 2.0;
  const double hy = sqrt((p1x - p2x) * (p1x - p2x) + (p1y - p2y) * (p1y - p2y));
  const double ad = fabs(p1x - p2x);
  const double theta0 = acos(ad / hy);   // sign of the slope is dropped here, as in the reference
  double theta = theta0;
  const double len = hy / 2.0;
  FitEval cur, cand;
  TRY(fit_eval(cx, pts, m, icx, icy, len, theta, partial, &cur));
  // A residual that is exactly 0 (a clicked point that is a scan point and an endpoint) has derivative 0 * inf = NaN,
  // under Jet sqrt as much as in the closed form.  Ceres then terminates with FAILURE and leaves the user's parameter
  // untouched (Summary::IsSolutionUsable); SegFitEM never reads the summary, so the fit silently keeps its initial
  // angle.  Same after 5 invalid steps in a row.  A candidate is judged on its cost alone (Ceres evaluates cost-only).
  bool failed = !(std::isfinite(cur.cost) && std::isfinite(cur.g) && std::isfinite(cur.h));
  const double scale = 1.0 / (1.0 + sqrt(cur.h));
  double radius = 1e4, decrease_factor = 2.0;
  int iter = 0, invalid_run = 0;
  bool done = failed || std::fabs(cur.g) <= 1e-10;
  while (!done) {
    if (iter >= 25) break;
    ++iter;
    const double hs = cur.h * scale * scale;
    double dg = hs;
    if (dg < 1e-6) dg = 1e-6;
    if (dg > 1e32) dg = 1e32;
    const double dt = -scale * cur.g / (hs + dg / radius);
    const double mcc = -dt * scale * cur.g - 0.5 * dt * dt * hs;
    if (!std::isfinite(dt) || mcc < 0.0) {
      if (++invalid_run >= 5) { failed = true; break; }
      radius /= decrease_factor; decrease_factor *= 2.0;
      continue;
    }
    invalid_run = 0;
    const double step = scale * dt, cth = theta + step;
    TRY(fit_eval(cx, pts, m, icx, icy, len, cth, partial, &cand));
    if (std::fabs(step) <= 1e-8 * (std::fabs(theta) + 1e-8)) break;
    if (!std::isfinite(cand.cost)) cand.cost = 1.7976931348623157e308;   // "a step with infinite cost"
    const double cost_change = cur.cost - cand.cost;
    if (std::fabs(cost_change) <= 1e-6 * cur.cost) break;
    const double rho = cost_change / mcc;
    if (rho > 1e-3) {
      theta = cth; cur = cand;
      if (!(std::isfinite(cur.g) && std::isfinite(cur.h))) { failed = true; break; }
      if (std::fabs(cur.g) <= 1e-10) break;
      const double q = 2.0 * rho - 1.0;
      double f = 1.0 - q * q * q;
      if (f < 1.0 / 3.0) f = 1.0 / 3.0;
      radius = radius / f;
      if (radius > 1e16) radius = 1e16;
      decrease_factor = 2.0;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0;
      if (radius < 1e-32) break;
    }
  }
  if (iterations) *iterations = iter;
  if (failed) theta = theta0;
  double ax = cos(theta), ay = sin(theta);
  const double nrm = sqrt(ax * ax + ay * ay);
  ax /= nrm; ay /= nrm;
  ep[0] = (float)(icx + len * ax); ep[1] = (float)(icy + len * ay);
  ep[2] = (float)(icx - len * ax); ep[3] = (float)(icy - len * ay);
  return HITL_OK;
}

void normalized2(float x, float y, float* ox, float* oy) {
  const float n = sqrtf(x * x + y * y);
  *ox = x / n; *oy = y / n;
}

}  // namespace

extern "C" {

int hitl_solver_fit_segments(hitl_solver* s, const float* poses_xyt, int32_t n_segments, float* seg_xy4, int32_t* n_inliers,
                             int32_t* passes, int32_t* lm_iterations) {
  SolverCtx cx;
  TRY(solver_ctx(s, &cx));
  if (!poses_xyt || !seg_xy4 || n_segments < 0) return fail(HITL_ERR_INVALID_ARGUMENT, "fit_segments: null argument");
  if (cx.n <= 0) return fail(HITL_ERR_NOT_READY, "fit_segments before set_map");
  // Submap ranks: every rank scans the observations of its OWNED poses, the inlier points travel in one all-gather (a few
  // hundred points; rank order = pose order = the reading order of the single-GPU list) and every rank then runs the same
  // M step on the same list: identical endpoints everywhere, equal to the single-GPU ones.
  HIP_TRY(hipSetDevice(cx.device));
  const int n_own = cx.own_hi - cx.own_lo;
  MapView own = cx.map;
  own.n_poses = n_own; own.obs_off = cx.map.obs_off + cx.own_lo;
  std::vector<float2> h_pts, h_all;
  std::vector<int64_t> h_m;
  const float* T4 = nullptr;
  DevArray<int> d_cnt;
  DevArray<int64_t> d_off;
  DevArray<float2> d_pts;
  DevArray<double> d_partial;
  int rc = solver_pose_rt(s, poses_xyt + 3 * (size_t)cx.pose_base, &T4);     // submap rank: the caller's array is the GLOBAL one
  if (rc == HITL_OK) rc = d_cnt.resize((size_t)n_own);
  std::vector<int> h_cnt((size_t)n_own);
  std::vector<int64_t> h_off((size_t)n_own + 1);
  const float* T4own = (rc == HITL_OK) ? T4 + (size_t)4 * cx.own_lo : nullptr;
  for (int k = 0; k < n_segments && rc == HITL_OK; ++k) {
    float* sp = seg_xy4 + 4 * k;
    const double thresh = 0.05;
    double adj1 = 2 * thresh, adj2 = 2 * thresh;
    int pass = 0, it_total = 0;
    int64_t m = 0;
    while (rc == HITL_OK && (adj1 > thresh || adj2 > thresh)) {
      if (sp[0] == sp[2] && sp[1] == sp[3]) { rc = fail(HITL_ERR_INVALID_ARGUMENT, "fit_segments: zero-length segment"); break; }
      if (++pass > 64) { rc = fail(HITL_ERR_SOLVER_FAILURE, "fit_segments: endpoints did not settle in 64 passes"); break; }
      // E step: observations within 3 cm of the current segment (pass 1 counts, pass 2 gathers in reading order)
      launch_fit_inliers(cx.stream, own, T4own, sp, d_cnt.p, nullptr, nullptr);
      if (hipMemcpyAsync(h_cnt.data(), d_cnt.p, sizeof(int) * h_cnt.size(), hipMemcpyDeviceToHost, cx.stream) != hipSuccess ||
          hipStreamSynchronize(cx.stream) != hipSuccess) { rc = fail(HITL_ERR_HIP, "fit_segments: inlier scan failed"); break; }
      h_off[0] = 0;
      for (int i = 0; i < n_own; ++i) h_off[i + 1] = h_off[i] + h_cnt[i];
      m = h_off[n_own];
      if (m > 0) {
        if ((rc = d_off.upload(h_off, cx.stream)) != HITL_OK) break;
        if ((rc = d_pts.resize((size_t)m)) != HITL_OK) break;
        launch_fit_inliers(cx.stream, own, T4own, sp, nullptr, d_off.p, d_pts.p);
      }
      if (cx.world > 1) {
        h_m.assign((size_t)cx.world, 0);
        const int64_t mine = m;
        if ((rc = solver_allgather(s, &mine, h_m.data(), sizeof(int64_t))) != HITL_OK) break;
        int64_t mmax = 1, mtot = 0;
        for (int r = 0; r < cx.world; ++r) { mmax = std::max(mmax, h_m[r]); mtot += h_m[r]; }
        h_pts.assign((size_t)mmax, make_float2(0.f, 0.f));
        if (m > 0 && (hipMemcpyAsync(h_pts.data(), d_pts.p, sizeof(float2) * (size_t)m, hipMemcpyDeviceToHost, cx.stream) != hipSuccess ||
                      hipStreamSynchronize(cx.stream) != hipSuccess)) { rc = fail(HITL_ERR_HIP, "fit_segments: inlier download failed"); break; }
        h_all.resize((size_t)mmax * cx.world);
        if ((rc = solver_allgather(s, h_pts.data(), h_all.data(), (int64_t)sizeof(float2) * mmax)) != HITL_OK) break;
        std::vector<float2> full((size_t)std::max<int64_t>(mtot, 1));
        int64_t w = 0;
        for (int r = 0; r < cx.world; ++r) for (int64_t q = 0; q < h_m[r]; ++q) full[(size_t)w++] = h_all[(size_t)mmax * r + q];
        m = mtot;
        if ((rc = d_pts.upload(full.data(), (size_t)std::max<int64_t>(mtot, 1), cx.stream)) != HITL_OK) break;
        if (hipStreamSynchronize(cx.stream) != hipSuccess) { rc = fail(HITL_ERR_HIP, "fit_segments: inlier upload failed"); break; }
      }
      // M step
      float ep[4];
      int it = 0;
      if ((rc = seg_fit_em(cx, d_pts.p, m, sp, &d_partial, ep, &it)) != HITL_OK) break;
      it_total += it;
      const float a0 = sp[0] - ep[0], a1 = sp[1] - ep[1], b0 = sp[2] - ep[2], b1 = sp[3] - ep[3];
      adj1 = sqrtf(a0 * a0 + a1 * a1); adj2 = sqrtf(b0 * b0 + b1 * b1);
      std::memcpy(sp, ep, sizeof(ep));
    }
    if (n_inliers) n_inliers[k] = (int32_t)m;
    if (passes) passes[k] = pass;
    if (lm_iterations) lm_iterations[k] = it_total;
  }
  (void)hipStreamSynchronize(cx.stream);
  d_cnt.release(); d_off.release(); d_pts.release(); d_partial.release();
  return rc;
}

int hitl_order_user_input(int64_t n_poses, const int32_t* counts, float* seg_xy8, int32_t* corrected, int32_t* n_corrected,
                          int32_t* anchor, int32_t* n_anchor, int32_t bounds[2]) {
  if (n_poses < 0 || !counts || !seg_xy8 || !corrected || !n_corrected || !anchor || !n_anchor || !bounds)
    return fail(HITL_ERR_INVALID_ARGUMENT, "order_user_input: null argument");
  std::vector<int32_t> first, second, overlaps;
  for (int64_t i = 0; i < n_poses; ++i) {          // EstablishObservationSets keeps poses with more than 5 inliers
    if (counts[2 * i] > 5) first.push_back((int32_t)i);
    if (counts[2 * i + 1] > 5) second.push_back((int32_t)i);
  }
  {   // both lists ascend, so the reference's nested loop yields the ascending intersection
    size_t a = 0, b = 0;
    while (a < first.size() && b < second.size()) {
      if (first[a] == second[b]) { overlaps.push_back(first[a]); ++a; ++b; }
      else if (first[a] < second[b]) ++a;
      else ++b;
    }
  }
  auto erase_all = [&](std::vector<int32_t>& v) {
    std::vector<int32_t> keep;
    size_t o = 0;
    for (int32_t x : v) {
      while (o < overlaps.size() && overlaps[o] < x) ++o;
      if (o < overlaps.size() && overlaps[o] == x) continue;
      keep.push_back(x);
    }
    v.swap(keep);
  };
  *n_corrected = 0; *n_anchor = 0;
  bounds[0] = bounds[1] = -1;
  const size_t no = overlaps.size();
  if (no == first.size() && no == second.size()) { /* "complete selection overlap": nothing is erased */ }
  else if (no == first.size()) erase_all(second);
  else if (no == second.size()) erase_all(first);
  else if (no > 0) { erase_all(first); erase_all(second); }
  if (first.empty() || second.empty())
    return fail(HITL_ERR_INVALID_ARGUMENT, "order_user_input: a selection is seen by no pose (the reference indexes an empty vector here)");
  const int fmin = first.front(), fmax = first.back(), smin = second.front(), smax = second.back();
  if (fmin > smax) {            // "user was good": the first selection moves onto the second
    std::memcpy(corrected, first.data(), first.size() * 4); *n_corrected = (int32_t)first.size();
    std::memcpy(anchor, second.data(), second.size() * 4); *n_anchor = (int32_t)second.size();
    bounds[0] = smax + 1; bounds[1] = fmin - 1;
  } else if (fmax < smin) {     // "bad user": swap the selections
    float t[4];
    std::memcpy(t, seg_xy8, 16); std::memcpy(seg_xy8, seg_xy8 + 4, 16); std::memcpy(seg_xy8 + 4, t, 16);
    std::memcpy(corrected, second.data(), second.size() * 4); *n_corrected = (int32_t)second.size();
    std::memcpy(anchor, first.data(), first.size() * 4); *n_anchor = (int32_t)first.size();
    bounds[0] = fmax + 1; bounds[1] = smin - 1;
  }                             // else "selection overlap": (-1, -1)
  return HITL_OK;
}

int hitl_solver_apply_explicit_correction(hitl_solver* s, uint32_t type, const float* sp, int64_t n_poses, float* poses,
                                          int32_t n_corr, const int32_t* corr, int32_t n_anchor, const int32_t* anchor,
                                          float correction[3], hitl_human_constraint_t* rows) {
  SolverCtx cx;
  TRY(solver_ctx(s, &cx));
  if (!sp || !poses || !corr || !correction || n_poses < 1 || n_corr < 1 || n_anchor < 0 || (n_anchor > 0 && (!anchor || !rows)))
    return fail(HITL_ERR_INVALID_ARGUMENT, "apply_explicit_correction: bad arguments");
  if (type != HITL_HC_LINE_SEGMENT && type != HITL_HC_COLINEAR && type != HITL_HC_PERPENDICULAR && type != HITL_HC_PARALLEL)
    return fail(HITL_ERR_UNSUPPORTED, "apply_explicit_correction: point and corner corrections are not supported (nor in the reference)");
  if (n_poses > 0x7fffffff / 3) return fail(HITL_ERR_UNSUPPORTED, "apply_explicit_correction: too many poses");
  HIP_TRY(hipSetDevice(cx.device));
  // ---- the rigid transform of the correction (ApplyExplicitCorrection.cpp:150-180, 229-316), fp32 + libm ----
  float Ax, Ay, Bx, By;
  normalized2(sp[2] - sp[0], sp[3] - sp[1], &Ax, &Ay);
  normalized2(sp[6] - sp[4], sp[7] - sp[5], &Bx, &By);
  const float cmAx = (sp[2] + sp[0]) * 0.5f, cmAy = (sp[3] + sp[1]) * 0.5f;
  const float cmBx = (sp[6] + sp[4]) * 0.5f, cmBy = (sp[7] + sp[5]) * 0.5f;
  const float dot = Ax * Bx + Ay * By, cross = Ax * By - Ay * Bx;
  float theta, dstx, dsty;
  if (type == HITL_HC_LINE_SEGMENT) {
    theta = acosf(dot);
    if (cross < 0.0f) theta = -theta;
    dstx = cmBx; dsty = cmBy;
  } else if (type == HITL_HC_COLINEAR) {
    theta = (cross >= 0.0f) ? acosf(dot) : -acosf(dot);
    const float alpha = (cmAx - cmBx) * Bx + (cmAy - cmBy) * By;
    dstx = cmBx + alpha * Bx; dsty = cmBy + alpha * By;
  } else if (type == HITL_HC_PERPENDICULAR) {
    double th = (cross < 0.0f) ? -(double)acosf(dot) : (double)acosf(dot);
    if (th == M_PI / 2.0 || th == -M_PI / 2.0) th = 0.0;
    else if (th > 0.0) th = -(-th + M_PI / 2.0);
    else th = -(-th - M_PI / 2.0);
    theta = (float)th; dstx = cmAx; dsty = cmAy;
  } else {
    theta = (cross >= 0.0f) ? acosf(dot) : -acosf(dot);
    dstx = cmAx; dsty = cmAy;
  }
  const float c = cosf(theta), sn = sinf(theta);
  // ---- FindContiguousGroups (:358-385): group 0 = the first run of consecutive corrected ids ----
  std::vector<unsigned char> member((size_t)n_poses, 0);
  for (int32_t j = 0; j < n_corr; ++j) {
    if (corr[j] < 0 || corr[j] >= n_poses) return fail(HITL_ERR_INVALID_ARGUMENT, "apply_explicit_correction: corrected pose id out of range");
    member[corr[j]] = 1;
  }
  for (int32_t a = 0; a < n_anchor; ++a)
    if (anchor[a] < 0 || anchor[a] >= n_poses) return fail(HITL_ERR_INVALID_ARGUMENT, "apply_explicit_correction: anchor pose id out of range");
  int64_t g0 = 0;
  while (!member[g0]) ++g0;
  int64_t g1 = g0;
  while (g1 + 1 < n_poses && member[g1 + 1]) ++g1;
  auto corr_of = [&](int64_t i, float T[3]) {
    const float p0x = poses[3 * i], p0y = poses[3 * i + 1];
    const float qx = p0x - cmAx, qy = p0y - cmAy;
    const float p1x = dstx + (c * qx - sn * qy), p1y = dsty + (sn * qx + c * qy);
    T[0] = p1x - p0x; T[1] = p1y - p0y; T[2] = theta;
  };
  float c0[3], cl[3];
  corr_of(g0, c0);
  corr_of(g1, cl);
  ExplicitArgs ea;
  ea.n = (int)n_poses; ea.g0 = (int)g0; ea.g1 = (int)g1;
  ea.c = c; ea.s = sn; ea.theta = theta; ea.pivx = cmAx; ea.pivy = cmAy; ea.dstx = dstx; ea.dsty = dsty;
  ea.lx = poses[3 * g1] + cl[0]; ea.ly = poses[3 * g1 + 1] + cl[1];      // poses_[last_pose] after its own correction
  ea.cl = cosf(cl[2]); ea.sl = sinf(cl[2]); ea.ltx = cl[0]; ea.lty = cl[1]; ea.lth = cl[2];
  DevArray<float> d_poses;
  DevArray<AnchorRow> d_anchor;
  DevArray<hitl_human_constraint_t> d_rows;
  DevArray<int> d_corr;
  int rc = d_poses.upload(poses, (size_t)3 * n_poses, cx.stream);
  if (rc == HITL_OK) {
    launch_explicit_apply(cx.stream, d_poses.p, ea);
    if (hipMemcpyAsync(poses, d_poses.p, sizeof(float) * 3 * (size_t)n_poses, hipMemcpyDeviceToHost, cx.stream) != hipSuccess ||
        hipStreamSynchronize(cx.stream) != hipSuccess) rc = fail(HITL_ERR_HIP, "apply_explicit_correction: kernel failed");
  }
  std::memcpy(correction, c0, sizeof(c0));
  // ---- calculateConstraintTargets (:447-487): per-anchor scalars here, the anchors x corrected table on the device ----
  if (rc == HITL_OK && n_anchor > 0) {
    std::vector<AnchorRow> at((size_t)n_anchor);
    const float corr_angle = atan2f(sp[7] - sp[5], sp[6] - sp[4]);
    for (int32_t a = 0; a < n_anchor; ++a) {
      const float aa = poses[3 * (size_t)anchor[a] + 2];
      AnchorRow& r = at[a];
      r.x = poses[3 * (size_t)anchor[a]]; r.y = poses[3 * (size_t)anchor[a] + 1]; r.angle = aa;
      r.c = cosf(aa); r.s = sinf(aa);
      r.rel = (float)((double)atan2f(sinf(corr_angle - aa), cosf(corr_angle - aa)) + M_PI / 2.0);
      r.id = anchor[a]; r.pad = 0;
    }
    const size_t n_rows = (size_t)n_anchor * n_corr;
    rc = d_anchor.upload(at, cx.stream);
    if (rc == HITL_OK) rc = d_corr.upload(corr, (size_t)n_corr, cx.stream);
    if (rc == HITL_OK) rc = d_rows.resize(n_rows);
    if (rc == HITL_OK) {
      launch_constraint_rows(cx.stream, d_poses.p, d_anchor.p, d_corr.p, n_anchor, n_corr, type, d_rows.p);
      if (hipMemcpyAsync(rows, d_rows.p, n_rows * sizeof(hitl_human_constraint_t), hipMemcpyDeviceToHost, cx.stream) != hipSuccess ||
          hipStreamSynchronize(cx.stream) != hipSuccess) rc = fail(HITL_ERR_HIP, "apply_explicit_correction: row kernel failed");
    }
  }
  d_poses.release(); d_anchor.release(); d_rows.release(); d_corr.release();
  return rc;
}

int hitl_solver_backprop(hitl_solver* s, int64_t n_poses, float* poses, float* cov9, int32_t lo, int32_t hi, const float correction[3]) {
  SolverCtx cx;
  TRY(solver_ctx(s, &cx));
  if (!poses || !cov9 || !correction || n_poses < 1) return fail(HITL_ERR_INVALID_ARGUMENT, "backprop: null argument");
  if (!(lo < hi)) return HITL_OK;     // "No poses between bounds. Cannot run backprop." (Backprop.cpp:204-210)
  if (lo < 0 || hi >= n_poses || n_poses > 0x7fffffff / 9) return fail(HITL_ERR_INVALID_ARGUMENT, "backprop: bounds outside the pose chain");
  HIP_TRY(hipSetDevice(cx.device));
  const int L = hi - lo;
  // ---- weights (Backprop.cpp:103-150): sequential fp32 sums, exactly in the reference's order ----
  const float destx = poses[3 * (size_t)hi] + correction[0], desty = poses[3 * (size_t)hi + 1] + correction[1];
  const float drv = 0.0001f, dtv = 0.001f;
  auto rs = [&](int64_t i) { return cov9[9 * i + 8]; };
  auto ts = [&](int64_t i) { return (float)((double)(cov9[9 * i] + cov9[9 * i + 4]) / 2.0); };
  float sum_r = 0.0f, sum_t = 0.0f;
  for (int64_t i = lo; i <= hi; ++i) { sum_r += rs(i); sum_t += ts(i); }
  sum_r += drv; sum_t += dtv;
  const float rot_beta = 1 / (1 + (rs(hi - 1) / drv)), trans_beta = 1 / (1 + (ts(hi - 1) / dtv));
  const float theta = correction[2];
  std::vector<float> piv((size_t)4 * L), tw((size_t)L);
  for (int q = 0; q < L; ++q) {
    const float dth = (rs(lo + q) / sum_r) * theta;
    piv[4 * q] = dth; piv[4 * q + 1] = cosf(dth); piv[4 * q + 2] = sinf(dth); piv[4 * q + 3] = 0.f;
    tw[q] = ts(lo + q) / sum_t;
  }
  DevArray<float> d_poses, d_cov, d_piv, d_tw, d_xf, d_trans;
  DevArray<int> d_sync;
  // the kernels touch poses lo .. hi and covariances lo .. hi - 1 only: that span travels, not the whole chain (a correction's span
  // is a few dozen poses of 50 000)
  int rc = d_poses.upload(poses + 3 * (size_t)lo, (size_t)3 * (L + 1), cx.stream);
  if (rc == HITL_OK) rc = d_cov.upload(cov9 + 9 * (size_t)lo, (size_t)9 * L, cx.stream);
  if (rc == HITL_OK) rc = d_piv.upload(piv, cx.stream);
  if (rc == HITL_OK) rc = d_tw.upload(tw, cx.stream);
  if (rc == HITL_OK) rc = d_xf.resize((size_t)4 * L);
  if (rc == HITL_OK) rc = d_trans.resize(2);
  if (rc == HITL_OK) rc = d_sync.resize((size_t)backprop_blocks(L) + 1);
  if (rc == HITL_OK) {
    launch_backprop(cx.stream, d_poses.p, d_cov.p, 0, L, d_piv.p, d_tw.p, d_xf.p, d_sync.p, destx, desty, trans_beta, rot_beta, d_trans.p);
    hipError_t e = hipMemcpyAsync(poses + 3 * (size_t)lo, d_poses.p, sizeof(float) * 3 * (size_t)(L + 1), hipMemcpyDeviceToHost, cx.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(cov9 + 9 * (size_t)lo, d_cov.p, sizeof(float) * 9 * (size_t)L, hipMemcpyDeviceToHost, cx.stream);
    if (e != hipSuccess || hipStreamSynchronize(cx.stream) != hipSuccess) rc = fail(HITL_ERR_HIP, "backprop: kernels failed");
  }
  d_poses.release(); d_cov.release(); d_piv.release(); d_tw.release(); d_xf.release(); d_trans.release(); d_sync.release();
  return rc;
}

int hitl_normalize_angles(int64_t n_poses, float* poses) {
  if (n_poses < 0 || (n_poses > 0 && !poses)) return fail(HITL_ERR_INVALID_ARGUMENT, "normalize_angles: null argument");
  for (int64_t i = 0; i < n_poses; ++i) {
    const float a = poses[3 * i + 2];
    poses[3 * i + 2] = atan2f(sinf(a), cosf(a));
  }
  return HITL_OK;
}

}  // extern "C"
