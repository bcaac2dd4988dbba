// hitl_kdwalk.h -- KDTree::FindNearestPointNormal (perception_tools/kdtree.cpp:141-197) as a per-lane state machine.
//
// The reference walk is a recursion whose pruning uses each call's LOCAL best (the point-to-plane distance found so far
// inside that sub-tree), so its result depends on the tree and cannot be replaced by a closed form.  This header restates
// it, decision for decision, as `kd_step`: one call = one node evaluation of one query.  A wave calls it for its 64
// queries in lock step; lanes are at different nodes and different depths of their own trees walk.
//
// What a recursion frame keeps, and where it went:
//   * the frame's segment [lo, lo + n) -- recovered from the child's segment, the side the child hangs on and one parity
//     bit (the in-order layout makes the split deterministic: node at lo + n/2, children [lo, n/2) and [lo+n/2+1, n-n/2-1));
//   * the stage the frame is in -- 3 bits; with the parity bit 4 bits per level in a 64-bit register (16 levels = scans
//     of up to 65535 points);
//   * the node's own score and plane distance -- recomputed from the node when its near child returns (one LDS read);
//   * the running best of a frame that went down a SECOND child (the far side of the split plane, or the right child
//     after the left one when the query lies on the plane) -- the only values that really have to be stored: a small
//     (best, winner) stack, touched by the few frames that take the far side.
// A frame whose plane distance is at least the gate (|pd| >= thr) has its own point outside the gate and can never take
// the far side: it hands its child's result through.  Such frames are not pushed at all while the stack is empty (the
// walk starts at the first node the query comes close to), and pop without a node read otherwise.
//
// No intrinsics: compiled by hipcc for the device and by g++ for the CPU check in tests/ (-ffp-contract=off on both sides:
// every fp32 operation is a single IEEE mul/add in the order of the reference).
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define HITL_HD __host__ __device__ __forceinline__
#else
#define HITL_HD inline
struct float4 { float x, y, z, w; };
#endif

namespace hitl {

enum KdCode : uint32_t {      // what the parent does when this child returns
  KD_LEFT_NEAR = 0,           // child = left, taken first (pd <= 0): merge, then maybe the right child
  KD_RIGHT_NEAR = 1,          // child = right, taken first (pd > 0): merge, then maybe the left child
  KD_FINAL_R = 2,             // child = right, taken second: merge with the stored best, return
  KD_FINAL_L = 3,             // child = left, taken second
  KD_PASS_L = 4,              // child = left, parent out of reach (|pd| >= thr): hand the result through
  KD_PASS_R = 5,
};

struct KdWalk {
  float qx, qy;               // query in the target's frame
  int lo, n;                  // segment of the node to evaluate next (mode 0) / of the child that just returned (mode 1)
  int mode;                   // 0 = enter (lo, n); 1 = returning (ret_d, ret_c) from (lo, n); 2 = finished
  float ret_d; int ret_c;     // FindNearestPointNormal's return value and the tree position of *neighbor_node (-1: never set)
  uint64_t stk;               // 4 bits per pushed frame under a sentinel bit
  int ssp;                    // depth of the stored-best stack
  int ovf;                    // the stored-best stack was too small for this query: the result is void (the caller repeats the search with a full-depth stack)
};

HITL_HD void kd_begin(KdWalk& w, float qx, float qy, int m) {
  w.qx = qx; w.qy = qy; w.lo = 0; w.n = m; w.mode = (m > 0) ? 0 : 2;
  w.ret_d = 3.402823466e+38f; w.ret_c = -1; w.stk = 1ull; w.ssp = 0;
}

// One step of one query: at most one pop and at most one node evaluation.  Written as straight-line selects -- the 64 lanes
// of a wave are in all states at once, so every branch would be taken by somebody; masked-off work is cheaper than the
// exec-mask bookkeeping of ~40 nested branches (first version: 150 instruction slots per step).
// pts[pos] = (x, y, nx, ny) of the tree node at in-order position pos (position 0 must be readable even for an empty
// tree), meta[pos] = original index | split plane << 16.  sv_best / sv_out: this lane's stored-best stack, entry e at
// [e * stride], entries 0 .. cap (entry cap is a write-only dump for lanes that store nothing).
HITL_HD void kd_step(KdWalk& w, const float4* pts, const uint32_t* meta, float thr, float thr2, float* sv_best, int* sv_out, int stride, int cap) {
  constexpr float kFltMax = 3.402823466e+38f, kFltMin = 1.175494351e-38f;
  // ---- a returning lane pops one frame ----
  const bool m1 = (w.mode == 1);
  const bool empty = (w.stk == 1ull);
  const bool popping = m1 && !empty;
  const uint32_t e = (uint32_t)w.stk & 15u;
  const uint32_t code = e & 7u;
  const int par = (int)(e >> 3);
  const bool left = ((0x19u >> code) & 1u) != 0;                 // KD_LEFT_NEAR, KD_FINAL_L, KD_PASS_L: the child hangs on the left
  const int pn = left ? 2 * w.n + par : 2 * w.n + 2 - par;         // n -> (n/2, n - n/2 - 1) inverted with the parity bit
  const int plo = left ? w.lo : w.lo - (pn >> 1) - 1;
  w.stk = popping ? (w.stk >> 4) : w.stk;
  w.lo = popping ? plo : w.lo;
  w.n = popping ? pn : w.n;
  const bool ret_eval = popping && code < KD_FINAL_R;              // near child returned: the node is read again
  const bool is_final = popping && (code == KD_FINAL_R || code == KD_FINAL_L);
  const int si = (w.ssp > 0) ? ((w.ssp <= cap) ? w.ssp - 1 : cap) : 0;     // entries past cap were never stored (ovf is set)
  const float sb = sv_best[si * stride];
  const int so = sv_out[si * stride];
  const bool take_saved = is_final && !(w.ret_d < sb);             // kdtree.cpp:189-192 (and :174-177 for pd == 0)
  w.ret_d = take_saved ? sb : w.ret_d;
  w.ret_c = take_saved ? so : w.ret_c;
  w.ssp -= is_final ? 1 : 0;
  w.mode = (m1 && empty) ? 2 : w.mode;
  // ---- node of (lo, n): kdtree.cpp:145-156 ----
  const bool enter = (w.mode == 0);
  const int n = w.n, ind = n >> 1, pos = w.lo + ind;
  const int rp = (enter || ret_eval) ? pos : 0;
  const float4 g = pts[rp];
  const int split = (int)(meta[rp] >> 16);
  const float dx = g.x - w.qx, dy = g.y - w.qy;
  const bool ingate = dx * dx + dy * dy < thr2;
  const float sc = __builtin_fabsf(g.z * (w.qx - g.x) + g.w * (w.qy - g.y));
  const float best = ingate ? sc : kFltMax;
  const int out = ingate ? pos : -1;
  const float pd = (split ? w.qy : w.qx) - (split ? g.y : g.x);
  const float apd = __builtin_fabsf(pd);
  // ---- the near child returned (ret_d, ret_c): kdtree.cpp:165-168 / :174-177, then :183-186 ----
  const bool child_wins = w.ret_d < best;
  const float mb = child_wins ? w.ret_d : best;
  const int mo = child_wins ? w.ret_c : out;
  const float lim = (mb < thr) ? mb : thr;
  // KD_LEFT_NEAR: the right child, as the second near one (pd == 0) or as the far side; KD_RIGHT_NEAR: pd > 0, the left child exists
  const bool go = ret_eval && ((code == KD_LEFT_NEAR) ? ((ind < n - 1) && (pd == 0.0f || apd < lim)) : (apd < lim));
  // ---- first visit: kdtree.cpp:149-151, :161, :170 ----
  const bool early = enter && ingate && sc < kFltMin;
  const bool go_l = enter && !early && pd <= 0.0f && ind > 0;
  const bool go_r = enter && !early && !go_l && pd >= 0.0f && ind < n - 1;
  const bool no_child = enter && !early && !go_l && !go_r;         // other_tree stays NULL (kdtree.cpp:159, :183)
  // own point outside the gate, no second near child (pd != 0), far side never taken (|pd| < min(best, thr) needs |pd| < thr):
  // the frame hands its child's result through.  (|pd| >= thr > 0 implies the first condition; spelled out so that any thr is exact.)
  const bool pass = !ingate && (pd != 0.0f) && !(apd < thr);
  const bool push_enter = (go_l || go_r) && !(pass && w.stk == 1ull);
  const uint32_t code_enter = (go_l ? 0u : 1u) + (pass ? 4u : 0u);
  const uint32_t code_go = (code == KD_LEFT_NEAR) ? (uint32_t)KD_FINAL_R : (uint32_t)KD_FINAL_L;
  // ---- stored best of a frame that takes its second child ----
  const bool fits = w.ssp < cap;
  const int wi = (go && fits) ? w.ssp : cap;
  sv_best[wi * stride] = mb;
  sv_out[wi * stride] = mo;
  w.ovf |= (go && !fits) ? 1 : 0;
  w.ssp += go ? 1 : 0;
  const bool push = go || push_enter;
  const uint64_t pushed = (w.stk << 4) | (uint64_t)((go ? code_go : code_enter) | ((uint32_t)(n & 1) << 3));
  w.stk = push ? pushed : w.stk;
  const bool to_right = go_r || (go && code == KD_LEFT_NEAR);
  const bool to_left = go_l || (go && code != KD_LEFT_NEAR);
  w.lo = to_right ? w.lo + ind + 1 : w.lo;
  w.n = to_right ? n - ind - 1 : (to_left ? ind : n);
  const bool ret_here = ret_eval && !go;
  w.ret_d = early ? 0.0f : (no_child ? best : (ret_here ? mb : w.ret_d));
  w.ret_c = early ? pos : (no_child ? out : (ret_here ? mo : w.ret_c));
  w.mode = (early || no_child) ? 1 : (go ? 0 : w.mode);
}

// The same walk as a literal unrolling of the recursion onto an explicit frame stack (round 2's kernel; bit for bit the
// CPU restatement's method 1 up to 105.9 M matches): kept as the yardstick `kd_step` is held to.
HITL_HD float kd_find_nearest_point_normal(const float4* __restrict__ pts, const uint32_t* __restrict__ meta, int m,
                                           float qx, float qy, float thr, uint32_t* st_seg, float* st_best,
                                           uint32_t* st_out, float* st_pd, int stride, int* out_pos) {
  constexpr float kFltMax = 3.402823466e+38f, kFltMin = 1.175494351e-38f;
  int sp = 1;
  st_seg[0] = (uint32_t)m << 16;            // lo = 0, n = m
  st_out[0] = 0;                            // stage 0
  float ret_d = kFltMax;
  int ret_c = -1;
  while (sp > 0) {
    const int f = (sp - 1) * stride;
    const uint32_t seg = st_seg[f];
    const int lo = (int)(seg & 0xffffu), n = (int)(seg >> 16), ind = n / 2, pos = lo + ind;
    const uint32_t os = st_out[f];
    int stage = (int)(os >> 28);
    int out = (int)(os & 0x0fffffffu) - 1;  // -1 = none
    float best = st_best[f], pd = st_pd[f];
    bool went_left = false, other_left = false, have_other = false;
    if (stage == 0) {
      const float4 g = pts[pos];
      const float dx = g.x - qx, dy = g.y - qy;
      best = kFltMax; out = -1;
      if (dx * dx + dy * dy < thr * thr) {
        out = pos;
        best = __builtin_fabsf(g.z * (qx - g.x) + g.w * (qy - g.y));
        if (best < kFltMin) { ret_d = 0.0f; ret_c = pos; --sp; continue; }
      }
      const int split = (int)(meta[pos] >> 16);
      pd = (split ? qy : qx) - (split ? g.y : g.x);
      if (pd <= 0.0f && ind > 0) {            // search the left sub-tree first
        st_best[f] = best; st_pd[f] = pd; st_out[f] = (1u << 28) | (uint32_t)(out + 1);
        const int c = sp * stride;
        st_seg[c] = (uint32_t)lo | ((uint32_t)ind << 16); st_out[c] = 0;
        ++sp;
        continue;
      }
      stage = 2;
    } else if (stage == 1) {                  // back from the left sub-tree
      if (ret_d < best) { best = ret_d; out = ret_c; }
      went_left = true;
      stage = 2;
    }
    if (stage == 2) {
      if (pd >= 0.0f && ind < n - 1) {        // the right sub-tree
        st_best[f] = best; st_pd[f] = pd; st_out[f] = (3u << 28) | (uint32_t)(out + 1);
        const int c = sp * stride;
        st_seg[c] = (uint32_t)(lo + ind + 1) | ((uint32_t)(n - ind - 1) << 16); st_out[c] = 0;
        ++sp;
        continue;
      }
      have_other = went_left && (ind < n - 1);   // `other = nd->right` was set in the left branch (NULL without a right child)
      other_left = false;
      stage = 4;
    } else if (stage == 3) {                  // back from the right sub-tree
      if (ret_d < best) { best = ret_d; out = ret_c; }
      have_other = ind > 0;                   // `other = nd->left`
      other_left = true;
      stage = 4;
    }
    if (stage == 4) {
      if (have_other && pd != 0.0f && __builtin_fabsf(pd) < ((best < thr) ? best : thr)) {   // the other side of the split plane may hold a closer point
        st_best[f] = best; st_pd[f] = pd; st_out[f] = (5u << 28) | (uint32_t)(out + 1);
        const int c = sp * stride;
        st_seg[c] = other_left ? ((uint32_t)lo | ((uint32_t)ind << 16)) : ((uint32_t)(lo + ind + 1) | ((uint32_t)(n - ind - 1) << 16));
        st_out[c] = 0;
        ++sp;
        continue;
      }
    } else if (stage == 5) {                  // back from the other side
      if (ret_d < best) { best = ret_d; out = ret_c; }
    }
    ret_d = best; ret_c = out;
    --sp;
  }
  *out_pos = ret_c;
  return ret_d;
}

}  // namespace hitl
