// CPU harness around hitl_slam_amd/csrc/hitl_kdwalk.h (test infrastructure, g++ -ffp-contract=off):
//  * kdw_build  -- per-scan tree in the in-order layout, built by the rules of KDTree::BuildKDTree (kdtree.cpp:37-139):
//                  split plane of the larger sum of squares, stable sort, median node;
//  * kdw_check  -- every query through `kd_step` (the state machine the GPU runs) and through the literal explicit-stack
//                  unrolling of the recursion; reports both results, the number of steps, the deepest stored-best stack and
//                  whether a stack of `cap` entries overflowed (such results are void and not compared);
//  * kdw_sim    -- 64 lanes in lock step over a queue of queries with the kernel's refill rule: how many wave iterations
//                  a target scan costs (design aid for the refill threshold; no effect on results).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../hitl_slam_amd/csrc/hitl_kdwalk.h"

using namespace hitl;

namespace {
struct P { float x, y, nx, ny; int id; };

void build_rec(std::vector<P> v, int lo, float4* pts, uint32_t* meta) {
  const int n = (int)v.size();
  if (n == 0) return;
  float mx = 0.0f, my = 0.0f;
  for (int q = 0; q < n; ++q) { mx = mx + v[q].x; my = my + v[q].y; }
  mx = mx / (float)n; my = my / (float)n;
  float sx = 0.0f, sy = 0.0f;
  for (int q = 0; q < n; ++q) {
    sx = sx + (v[q].x - mx) * (v[q].x - mx);
    sy = sy + (v[q].y - my) * (v[q].y - my);
  }
  int plane = 0;
  float best = 0.0f;
  if (sx > best) { plane = 0; best = sx; }
  if (sy > best) { plane = 1; best = sy; }
  std::stable_sort(v.begin(), v.end(), [plane](const P& a, const P& b) { return plane ? a.y < b.y : a.x < b.x; });
  const int ind = n / 2;
  pts[lo + ind] = float4{v[ind].x, v[ind].y, v[ind].nx, v[ind].ny};
  meta[lo + ind] = (uint32_t)v[ind].id | ((uint32_t)plane << 16);
  build_rec(std::vector<P>(v.begin(), v.begin() + ind), lo, pts, meta);
  build_rec(std::vector<P>(v.begin() + ind + 1, v.end()), lo + ind + 1, pts, meta);
}
}  // namespace

extern "C" {

void kdw_build(int m, const float* x, const float* y, const float* nx, const float* ny, float* pts4, uint32_t* meta) {
  std::vector<P> v(m);
  for (int k = 0; k < m; ++k) v[k] = P{x[k], y[k], nx[k], ny[k], k};
  build_rec(v, 0, reinterpret_cast<float4*>(pts4), meta);
}

// returns the number of queries on which the two walks disagree (distance bits or winner)
// fast: kd_step_fast calls behind every kd_step (the kernels' schedule); `evals` then counts the pairs
int64_t kdw_check2(int m, const float* pts4, const uint32_t* meta, int64_t nq, const float* qx, const float* qy, float thr,
                   int32_t* pos_new, float* d_new, int32_t* pos_old, float* d_old, int32_t* evals, int cap, int fast);
int64_t kdw_check(int m, const float* pts4, const uint32_t* meta, int64_t nq, const float* qx, const float* qy, float thr,
                  int32_t* pos_new, float* d_new, int32_t* pos_old, float* d_old, int32_t* evals, int cap) {
  return kdw_check2(m, pts4, meta, nq, qx, qy, thr, pos_new, d_new, pos_old, d_old, evals, cap, 0);
}
int64_t kdw_check2(int m, const float* pts4, const uint32_t* meta, int64_t nq, const float* qx, const float* qy, float thr,
                   int32_t* pos_new, float* d_new, int32_t* pos_old, float* d_old, int32_t* evals, int cap, int fast) {
  const float4* pts = reinterpret_cast<const float4*>(pts4);
  std::vector<float> f_inc(64), st_best(64), st_pd(64);
  std::vector<uint32_t> f_seg(64), f_oc(64);
  std::vector<uint32_t> st_seg(64), st_out(64);
  int64_t bad = 0;
  const float thr2 = thr * thr;
  for (int64_t q = 0; q < nq; ++q) {
    KdWalk w;
    kd_begin(w, qx[q], qy[q], m);
    int it = 0;
    int deep = 0;
    w.ovf = 0;
    w.t_inc = 0.0f; w.t_seg = 0; w.t_oc = 0;
    while (w.mode != 2) {
      kd_step(w, pts, meta, thr, thr2, f_inc.data(), f_seg.data(), f_oc.data(), 1, cap); ++it; deep = std::max(deep, w.sp);
      for (int f = 0; f < fast; ++f) kd_step_fast(w, pts, meta, thr, thr2);
    }
    pos_new[q] = w.ret_c; d_new[q] = w.ret_d; evals[q] = it | (deep << 16) | (w.ovf << 30);
    int po = -1;
    float dd = 3.402823466e+38f;
    if (m > 0) dd = kd_find_nearest_point_normal(pts, meta, m, qx[q], qy[q], thr, st_seg.data(), st_best.data(), st_out.data(), st_pd.data(), 1, &po);
    pos_old[q] = po; d_old[q] = dd;
    if (!w.ovf && (po != w.ret_c || std::memcmp(&dd, &w.ret_d, 4) != 0)) ++bad;
  }
  return bad;
}

// lock-step simulation: `lanes` lanes share a queue of nq queries (handed out in order); idle lanes are refilled when at
// least `refill_min` of them are idle (or nothing else is running).  Returns wave iterations; *busy = lane-iterations that
// evaluated or popped something.
int64_t kdw_sim2(int m, const float* pts4, const uint32_t* meta, int64_t nq, const float* qx, const float* qy, float thr, int lanes,
                 int refill_min, int64_t* busy, int64_t* refills, int fast);
int64_t kdw_sim(int m, const float* pts4, const uint32_t* meta, int64_t nq, const float* qx, const float* qy, float thr, int lanes,
                int refill_min, int64_t* busy, int64_t* refills) {
  return kdw_sim2(m, pts4, meta, nq, qx, qy, thr, lanes, refill_min, busy, refills, 0);
}
int64_t kdw_sim2(int m, const float* pts4, const uint32_t* meta, int64_t nq, const float* qx, const float* qy, float thr, int lanes,
                 int refill_min, int64_t* busy, int64_t* refills, int fast) {
  const float4* pts = reinterpret_cast<const float4*>(pts4);
  const float thr2 = thr * thr;
  std::vector<KdWalk> w(lanes);
  std::vector<std::vector<float>> sb(lanes, std::vector<float>(64));
  std::vector<std::vector<uint32_t>> so(lanes, std::vector<uint32_t>(64)), sq(lanes, std::vector<uint32_t>(64));
  for (auto& x : w) { x.mode = 2; x.lo = x.n = 0; x.sp = 0; x.ovf = 0; x.ret_d = 0.0f; x.ret_c = -1; x.inc_d = 0.0f; x.inc_c = -1; x.qx = x.qy = 0.0f; x.t_inc = 0.0f; x.t_seg = 0; x.t_oc = 0; }
  int64_t next = 0, iters = 0, b = 0, rf = 0;
  for (;;) {
    int idle = 0;
    for (auto& x : w) idle += x.mode == 2;
    if (idle == lanes && next >= nq) break;
    if (next < nq && (idle >= refill_min || idle == lanes)) {
      ++rf;
      for (auto& x : w) if (x.mode == 2 && next < nq) { kd_begin(x, qx[next], qy[next], m); ++next; }
    }
    for (int l = 0; l < lanes; ++l) if (w[l].mode != 2) { kd_step(w[l], pts, meta, thr, thr2, sb[l].data(), so[l].data(), sq[l].data(), 1, 62); ++b; }
    for (int f = 0; f < fast; ++f)
      for (int l = 0; l < lanes; ++l) if (w[l].mode != 2) kd_step_fast(w[l], pts, meta, thr, thr2);
    ++iters;
  }
  *busy = b; *refills = rf;
  return iters;
}

}  // extern "C"
