// hitl_kdwalk.h -- KDTree::FindNearestPointNormal (perception_tools/kdtree.cpp:141-197) as a per-lane state machine.
//
// The reference walk is a recursion whose pruning uses each call's LOCAL best (the point-to-plane distance found so far
// inside that sub-tree), so its result depends on the tree and cannot be replaced by a closed form.  This header restates
// it, decision for decision, as `kd_step`: one call = at most one node evaluation of one query.  A wave calls it for its 64
// queries in lock step; lanes are at different nodes and different depths of their own walks.
//
// Which recursion frames have to exist.  A frame (a call on node X) does three things after its near child returns:
// merge the child's result with X's own point (the child wins only if strictly better), decide whether the far side of
// the split plane is searched too -- `|pd| < min(best, thr)` with the MERGED best, which is never larger than X's own
// score -- and hand the result up.  If at the first visit already |pd| >= min(own score, thr) (or the query lies off the
// plane and there is no far child), the far side can never be taken whatever the child returns: the frame only merges.
// Merging is a minimum with ties to the outer frame, so a run of merge-only frames folds into ONE running value, the
// INCUMBENT (inc): visiting such a node costs an evaluation and no frame.  Only frames that may still take the far side
// ("real" frames, 1.1 on average for a 400-point tree, 6 at most seen) are pushed: they keep the incumbent of the run above
// them, their segment and which child was taken first; the run below them starts with an empty incumbent, because a real
// frame's far-side test must see the result of its own sub-tree only.  When a real frame takes its second child, the
// merged value becomes the incumbent of that child's run (the final merge of kdtree.cpp:189-192 is again a minimum with
// ties to the earlier value).  Steps per query: nodes visited + real frames + second descents = 12.6 on scan queries; the
// first state machine (round 3, first half: every frame within reach of the gate pushed, its segment recovered through a
// parity bit, its score recomputed on return) needed 17.4.
// The TOP frame lives in registers (t_inc, t_seg, t_oc): a pop feeds the node read of the same step, and through LDS that
// would be two dependent round trips per step (measured: slower than the 17.4-step machine).  The frame below the top is
// read from LDS at the start of every step and only consumed at its end, when a pop refills the registers.
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

enum KdCode : uint32_t {      // what a pushed frame does when the run below it returns
  KD_LEFT_NEAR = 0,           // the left child was taken first (pd <= 0): merge, then maybe the right child
  KD_RIGHT_NEAR = 1,          // the right child was taken first (pd > 0): merge, then maybe the left child
  KD_FINAL = 2,               // the second child was taken: its run carried the merged value as incumbent, hand the result up
};

struct KdWalk {
  float qx, qy;               // query in the target's frame
  int lo, n;                  // segment of the node to evaluate next (mode 0)
  int mode;                   // 0 = enter (lo, n); 1 = the run below returned (ret_d, ret_c); 2 = finished
  float ret_d; int ret_c;     // FindNearestPointNormal's return value and the tree position of *neighbor_node (-1: never set)
  float inc_d; int inc_c;     // incumbent of the current run of merge-only frames
  int sp;                     // frames on the stack (the top one in t_*, the others in LDS entries 0 .. sp-2)
  float t_inc; uint32_t t_seg, t_oc;   // top frame: saved incumbent's distance, segment lo | n << 16, saved incumbent's position + 1 | code << 20
  int ovf;                    // the frame stack was too small for this query: the result is void
};

// Lane-varying conditions are plain bools combined with & and | (no && / ||: every operand is side-effect free, and the
// short-circuit forms come out of the compiler as exec-masked regions -- s_and_saveexec / s_or_b64 around one or two
// instructions -- until a step of the walk spends as many scalar instructions on mask bookkeeping as vector instructions on
// the walk: -7 % on the whole search, same-box A/B).  kd_sel is the plain ?: (forcing both operands to be evaluated, i.e. one
// v_cndmask per choice, was measured 3 % slower than what the compiler makes of the nested ?: chains).
#define kd_sel(c, a, b) ((c) ? (a) : (b))

HITL_HD void kd_begin(KdWalk& w, float qx, float qy, int m) {
  w.qx = qx; w.qy = qy; w.lo = 0; w.n = m; w.mode = (m > 0) ? 0 : 2;
  w.ret_d = 3.402823466e+38f; w.ret_c = -1; w.inc_d = 3.402823466e+38f; w.inc_c = -1; w.sp = 0; w.ovf = 0;
}

// One step of one query: at most one frame popped and at most one node evaluated, as straight-line selects (the 64 lanes of
// a wave are in all states at once: masked-off work is cheaper than the exec-mask bookkeeping of nested branches).
// pts[pos] = (x, y, nx, ny) of the tree node at in-order position pos (position 0 must be readable even for an empty
// tree), meta[pos] = original index | split plane << 16.  st_inc / st_seg / st_oc: this lane's frame stack below the top,
// entry e at [e * stride], entries 0 .. cap (entry cap is a write-only dump for lanes that store nothing).
HITL_HD void kd_step(KdWalk& w, const float4* pts, const uint32_t* meta, float thr, float thr2, float* st_inc, uint32_t* st_seg, uint32_t* st_oc,
                     int stride, int cap) {
  constexpr float kFltMax = 3.402823466e+38f, kFltMin = 1.175494351e-38f;
  // the frame below the top, in case this step pops (consumed at the end of the step only)
  const int bi = (w.sp >= 2) ? ((w.sp - 2 < cap) ? w.sp - 2 : cap) : 0;
  const float b_inc = st_inc[bi * stride];
  const uint32_t b_seg = st_seg[bi * stride], b_oc = st_oc[bi * stride];
  // ---- the run below returned: its result against the run's incumbent (ties to the incumbent: the outer frame) ----
  const bool m1 = (w.mode == 1);
  const bool ret_wins = w.ret_d < w.inc_d;
  const float rd = ret_wins ? w.ret_d : w.inc_d;
  const int rc = ret_wins ? w.ret_c : w.inc_c;
  const bool empty = (w.sp == 0);
  const bool popping = m1 & !empty;
  const uint32_t code = w.t_oc >> 20;
  const float f_inc = w.t_inc;
  const int f_inc_c = (int)(w.t_oc & 0xfffffu) - 1;
  const bool is_final = popping & (code == KD_FINAL);
  const bool ret_eval = popping & (code != KD_FINAL);               // a near child's run returned: the frame's node is read again
  w.ret_d = (m1 ? rd : w.ret_d); w.ret_c = (m1 ? rc : w.ret_c);
  w.mode = (m1 & empty) ? 2 : w.mode;
  w.lo = ret_eval ? (int)(w.t_seg & 0xffffu) : w.lo;
  w.n = ret_eval ? (int)(w.t_seg >> 16) : w.n;
  // ---- node of (lo, n): kdtree.cpp:145-156 ----
  const bool enter = (w.mode == 0);
  const int n = w.n, ind = n >> 1, pos = w.lo + ind;
  const int rp = (enter | ret_eval) ? pos : 0;
  const float4 g = pts[rp];
  const int split = (int)(meta[rp] >> 16);
  const float dx = g.x - w.qx, dy = g.y - w.qy;
  const bool ingate = dx * dx + dy * dy < thr2;
  // kdtree.cpp:150 scores with (q - p) and :157 splits on q - p as well; q - p is the exact negative of dx / dy, products and sums of
  // exact negatives are exact negatives, and only |.|, the sign and the zero test of the results are used: same bits from dx, dy
  const float sc = __builtin_fabsf(g.z * dx + g.w * dy);
  const float best = ingate ? sc : kFltMax;
  const int out = ingate ? pos : -1;
  const float npd = split ? dy : dx;                    // -(q - p) along the split plane: pd <= 0 <=> npd >= 0, pd >= 0 <=> npd <= 0
  const float apd = __builtin_fabsf(npd);
  const bool has_right = ind < n - 1, has_left = ind > 0;
  // ---- a near child's run returned (rd, rc): kdtree.cpp:165-168 / :174-177, then :183-186 ----
  const bool child_wins = rd < best;
  const float mb = child_wins ? rd : best;
  const int mo = child_wins ? rc : out;
  const float lim = (mb < thr) ? mb : thr;
  // KD_LEFT_NEAR: the right child, as the second near one (pd == 0) or as the far side; KD_RIGHT_NEAR: pd > 0, the left child exists
  const bool near_lim = apd < lim;
  const bool go_if_left = has_right & ((npd == 0.0f) | near_lim);
  const bool go = ret_eval & ((code == KD_LEFT_NEAR) ? go_if_left : near_lim);
  const bool ret_here = ret_eval & !go;                           // the frame returns the merged value into the outer run
  // ---- first visit: kdtree.cpp:149-151, :161, :170 ----
  const bool early = enter & ingate & (sc < kFltMin);
  const bool go_l = enter & !early & (npd >= 0.0f) & has_left;
  const bool go_r = enter & !early & !go_l & (npd <= 0.0f) & has_right;
  const bool no_child = enter & !early & !go_l & !go_r;         // other_tree stays NULL (kdtree.cpp:159, :183)
  // the far side can never be taken, whatever the near child returns: |pd| >= min(own score, thr) >= min(merged, thr) and the
  // query is off the plane -- or there is no far child (a left child without a right sibling)
  const float lim0 = (best < thr) ? best : thr;
  const bool merge_only = ((npd != 0.0f) & !(apd < lim0)) | (go_l & !has_right);
  const bool fold = (go_l | go_r) & merge_only;                  // own point into the incumbent, no frame
  const bool push = (go_l | go_r) & !merge_only;
  const bool own_beats_inc = best < w.inc_d;
  // ---- the frame stack.  push: the old top goes to LDS entry sp - 1, the new frame (outer incumbent, segment, first child)
  //      becomes the top.  go: the top's code becomes KD_FINAL.  is_final / ret_here: the top is dropped, the frame below
  //      comes up from LDS ----
  const bool spill = push & (w.sp > 0);
  const bool fits = w.sp - 1 < cap;
  const int wi = (spill & fits) ? w.sp - 1 : cap;
  st_inc[wi * stride] = w.t_inc; st_seg[wi * stride] = w.t_seg; st_oc[wi * stride] = w.t_oc;
  w.ovf |= (spill & !fits) ? 1 : 0;
  const bool pop = is_final | ret_here;
  const uint32_t n_oc_push = (uint32_t)(w.inc_c + 1) | ((go_l ? (uint32_t)KD_LEFT_NEAR : (uint32_t)KD_RIGHT_NEAR) << 20);
  const uint32_t n_oc_go = (w.t_oc & 0xfffffu) | ((uint32_t)KD_FINAL << 20);
  w.t_oc = kd_sel(push, n_oc_push, kd_sel(go, n_oc_go, kd_sel(pop, b_oc, w.t_oc)));
  w.t_seg = kd_sel(push, (uint32_t)w.lo | ((uint32_t)n << 16), kd_sel(pop, b_seg, w.t_seg));
  w.t_inc = kd_sel(push, w.inc_d, kd_sel(pop, b_inc, w.t_inc));
  w.sp += (push ? 1 : 0) - (pop ? 1 : 0);
  // ---- the incumbent of the run that continues ----
  //   fold: min(inc, own), ties to inc;  push: a fresh run;  go: the merged value;  is_final / ret_here: the outer incumbent comes back
  const float inc_d = kd_sel(fold, kd_sel(own_beats_inc, best, w.inc_d), kd_sel(push, kFltMax, kd_sel(go, mb, kd_sel(pop, f_inc, w.inc_d))));
  const int inc_c = kd_sel(fold, kd_sel(own_beats_inc, out, w.inc_c), kd_sel(push, -1, kd_sel(go, mo, kd_sel(pop, f_inc_c, w.inc_c))));
  w.inc_d = inc_d; w.inc_c = inc_c;
  // ---- where the walk goes ----
  const bool to_right = go_r | (go & (code == KD_LEFT_NEAR));
  const bool to_left = go_l | (go & (code != KD_LEFT_NEAR));
  w.lo = kd_sel(to_right, w.lo + ind + 1, w.lo);
  w.n = kd_sel(to_right, n - ind - 1, kd_sel(to_left, ind, n));
  w.ret_d = kd_sel(early, 0.0f, kd_sel(no_child, best, kd_sel(ret_here, mb, w.ret_d)));
  w.ret_c = kd_sel(early, pos, kd_sel(no_child, out, kd_sel(ret_here, mo, w.ret_c)));
  w.mode = kd_sel(early | no_child, 1, kd_sel(go, 0, w.mode));
}

// The cheap half of kd_step (round 4): a query that is about to ENTER a node and will not push a frame there -- the node ends the
// run (early return, no child on the query's side) or folds into the incumbent and the walk descends -- takes exactly the
// transition kd_step would give it; every other lane (returning runs, finished queries, nodes whose far side may still be taken:
// those need the frame stack) is left untouched for the next kd_step.  Four of five steps of a scan query are of this kind, and
// without the stack, the pops and the return logic the step costs about a third of kd_step's instructions: the kernels alternate
// the two.  Same expressions as kd_step, so the walk -- winner and distance bits -- cannot change.
HITL_HD void kd_step_fast(KdWalk& w, const float4* pts, const uint32_t* meta, float thr, float thr2) {
  constexpr float kFltMax = 3.402823466e+38f, kFltMin = 1.175494351e-38f;
  const bool enter = (w.mode == 0);
  const int n = w.n, ind = n >> 1, pos = w.lo + ind;
  const int rp = enter ? pos : 0;
  const float4 g = pts[rp];
  const int split = (int)(meta[rp] >> 16);
  const float dx = g.x - w.qx, dy = g.y - w.qy;
  const bool ingate = dx * dx + dy * dy < thr2;
  // kdtree.cpp:150 scores with (q - p) and :157 splits on q - p as well; q - p is the exact negative of dx / dy, products and sums of
  // exact negatives are exact negatives, and only |.|, the sign and the zero test of the results are used: same bits from dx, dy
  const float sc = __builtin_fabsf(g.z * dx + g.w * dy);
  const float best = ingate ? sc : kFltMax;
  const int out = ingate ? pos : -1;
  const float npd = split ? dy : dx;                    // -(q - p) along the split plane: pd <= 0 <=> npd >= 0, pd >= 0 <=> npd <= 0
  const float apd = __builtin_fabsf(npd);
  const bool has_right = ind < n - 1, has_left = ind > 0;
  const bool early = enter & ingate & (sc < kFltMin);
  const bool go_l = enter & !early & (npd >= 0.0f) & has_left;
  const bool go_r = enter & !early & !go_l & (npd <= 0.0f) & has_right;
  const bool no_child = enter & !early & !go_l & !go_r;
  const float lim0 = (best < thr) ? best : thr;
  const bool merge_only = ((npd != 0.0f) & !(apd < lim0)) | (go_l & !has_right);
  const bool fold = (go_l | go_r) & merge_only;          // (go_l || go_r) && !merge_only would push: not here
  const bool own_beats_inc = best < w.inc_d;
  w.inc_d = (fold & own_beats_inc) ? best : w.inc_d;
  w.inc_c = (fold & own_beats_inc) ? out : w.inc_c;
  const bool to_right = fold & go_r, to_left = fold & go_l;
  w.lo = kd_sel(to_right, w.lo + ind + 1, w.lo);
  w.n = kd_sel(to_right, n - ind - 1, kd_sel(to_left, ind, n));
  w.ret_d = kd_sel(early, 0.0f, kd_sel(no_child, best, w.ret_d));
  w.ret_c = kd_sel(early, pos, kd_sel(no_child, out, w.ret_c));
  w.mode = kd_sel(early | no_child, 1, w.mode);
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
