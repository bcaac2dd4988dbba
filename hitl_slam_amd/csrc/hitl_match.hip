// hitl_match.hip -- scan-to-scan correspondence scoring (K2 of SURVEY.md section 2.3):
// JointOpt::FindSTFCorrespondences (JointOptimization.cpp:561-642) with the KD-tree query
// KDTree::FindNearestPointNormal (perception_tools/kdtree.cpp:141-197) replaced by its
// brute-force definition evaluated out of LDS: among the target points inside the Euclidean
// gate, the one with the smallest point-to-plane distance.
//
// Compiled with -ffp-contract=off: every fp32 operation below is a single IEEE mul/add in a
// fixed order, so the result (an index set) is bit-identical to the CPU restatement.
//
// One workgroup per SOURCE pose.  Target poses are visited in ascending order because the
// per-source-point cap (kMaxCorrespondencesPerPoint) is counted across targets
// (JointOptimization.cpp:596-599) -- that dependence is sequential by definition.  Each
// target scan is staged once in LDS as float4 (x, y, nx, ny); every thread owns source
// points k = tid, tid+256, ... and walks the staged scan with broadcast ds_read_b128.

#include "hitl_internal.h"
#include "hitl_kdwalk.h"

namespace hitl {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kMatchBlock = 256;
constexpr int kMatchGroup = 16;    // target points per bounding box

__global__ __launch_bounds__(kMatchBlock) void k_match(MapView map, int chunk, int max_scan, int slots,
                                                        const int* __restrict__ cand_tgt,
                                                        const float* __restrict__ cand_tf, MatchParams mp,
                                                        uint16_t* __restrict__ table, int* __restrict__ counts, int i0, int n_tile) {
  extern __shared__ float4 s_tgt[];                 // [max_scan]
  int* s_cnt = (int*)(s_tgt + ((max_scan + 1) & ~1));   // [max_scan] accepted matches per source point
  float4* s_box = (float4*)(s_cnt + ((max_scan + 3) & ~3));   // [ceil(max_scan / kMatchGroup)] (min x, min y, max x, max y), 16-byte aligned
  __shared__ int s_pair_count;
  const int b = blockIdx.x;
  const int idx = b >> 3;
  const int il = (b & 7) * chunk + idx;             // consecutive poses share an XCD (L2 reuse of neighbours)
  if (idx >= chunk || il >= n_tile) return;
  const int i = i0 + il;                            // candidate arrays, table and counts hold the tile [i0, i0 + n_tile) only
  const int64_t oi = map.obs_off[i];
  const int mi = (int)(map.obs_off[i + 1] - oi);
  for (int k = threadIdx.x; k < mi; k += kMatchBlock) s_cnt[k] = 0;
  const float thr = mp.thr, thr2 = mp.thr * mp.thr;
  for (int slot = 0; slot < slots; ++slot) {
    const int j = cand_tgt[(size_t)il * slots + slot];
    if (j < 0) continue;                            // block-uniform
    const int64_t oj = map.obs_off[j];
    const int mj = (int)(map.obs_off[j + 1] - oj);
    __syncthreads();
    if (threadIdx.x == 0) s_pair_count = 0;
    // Two consecutive target points per LDS record pair -- s_tgt[2h] = (x, x', y, y'), s_tgt[2h + 1] = (nx, nx', ny, ny') of
    // points 2h and 2h + 1 -- so that the gate and the point-to-plane score of both come out of packed fp32 instructions
    // (v_pk_add_f32 / v_pk_mul_f32: two IEEE results per instruction, each identical to the scalar one; no contraction).
    // An odd scan ends with a copy of its last point (never selected: its index is past the end).
    for (int h = threadIdx.x; 2 * h < mj; h += kMatchBlock) {
      const float4 a = map.obs[oj + 2 * h];
      const float4 b = (2 * h + 1 < mj) ? map.obs[oj + 2 * h + 1] : a;
      s_tgt[2 * h] = make_float4(a.x, b.x, a.y, b.y);
      s_tgt[2 * h + 1] = make_float4(a.z, b.z, a.w, b.w);
    }
    __syncthreads();
    // Bounding box of every run of kMatchGroup consecutive target points (a scan is ordered by bearing, so a run is a
    // short piece of wall).  A source point farther than the gate from a box cannot pass the gate for any point in
    // it -- exactly so in fp32 too: rounding is monotone, so every dsq of the run is >= the box's -- and the run is
    // skipped.  The points that are tested are still visited in ascending order: same winner, same ties.
    const int ngrp = (mj + kMatchGroup - 1) / kMatchGroup;
    for (int g = threadIdx.x; g < ngrp; g += kMatchBlock) {
      float4 bx = make_float4(3.402823466e+38f, 3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f);
      const int t1 = min(mj, (g + 1) * kMatchGroup);
      for (int t = g * kMatchGroup; t < t1; ++t) {
        const float4 v = s_tgt[t & ~1];
        const float vx = (t & 1) ? v.y : v.x, vy = (t & 1) ? v.w : v.z;
        bx.x = fminf(bx.x, vx); bx.y = fminf(bx.y, vy); bx.z = fmaxf(bx.z, vx); bx.w = fmaxf(bx.w, vy);
      }
      s_box[g] = bx;
    }
    __syncthreads();
    const float* tf = cand_tf + ((size_t)il * slots + slot) * 8;
    const float m00 = tf[0], m01 = tf[1], m10 = tf[2], m11 = tf[3], tx = tf[4], ty = tf[5], cr = tf[6], sr = tf[7];
    uint16_t* row = table + ((size_t)il * slots + slot) * max_scan;
    int local = 0;
    for (int k = threadIdx.x; k < mi; k += kMatchBlock) {
      uint16_t res = 0xFFFFu;
      if ((k % mp.skip) == 0 && s_cnt[k] < mp.cap) {
        const float4 a = map.obs[oi + k];
        const float ax = a.x, ay = a.y;
        const float qx = (m00 * ax + m01 * ay) + tx;
        const float qy = (m10 * ax + m11 * ay) + ty;
        float best = 3.402823466e+38f;
        int bi = -1;
        const f32x2 q2x = {qx, qx}, q2y = {qy, qy};
        for (int gi = 0; gi < ngrp; ++gi) {
          const float4 bx = s_box[gi];
          // distance from q to the box, with the same subtractions the point test uses (q - coordinate)
          const float ex = fmaxf(fmaxf(qx - bx.z, -(qx - bx.x)), 0.0f);
          const float ey = fmaxf(fmaxf(qy - bx.w, -(qy - bx.y)), 0.0f);
          if (!(ex * ex + ey * ey < thr2)) continue;
          const int t1 = min(mj, (gi + 1) * kMatchGroup);
          for (int t = gi * kMatchGroup; t < t1; t += 2) {            // kMatchGroup is even: t is even
            const float4 P = s_tgt[t], N = s_tgt[t + 1];
            const f32x2 dx = q2x - f32x2{P.x, P.y}, dy = q2y - f32x2{P.z, P.w};
            const f32x2 dsq = dx * dx + dy * dy;
            const f32x2 sc = f32x2{N.x, N.y} * dx + f32x2{N.z, N.w} * dy;
            const float s0 = fabsf(sc.x), s1 = fabsf(sc.y);
            if (dsq.x < thr2 && s0 < best) { best = s0; bi = t; }
            if (t + 1 < t1 && dsq.y < thr2 && s1 < best) { best = s1; bi = t + 1; }
          }
        }
        if (bi >= 0 && best < thr) {
          const float anx = a.z, any = a.w;
          const float nqx = cr * anx - sr * any, nqy = sr * anx + cr * any;
          const float4 gn = s_tgt[(bi & ~1) + 1];
          const float gnx = (bi & 1) ? gn.y : gn.x, gny = (bi & 1) ? gn.w : gn.z;
          const float cosang = gnx * nqx + gny * nqy;
          if (cosang > mp.min_cos) { res = (uint16_t)bi; s_cnt[k] += 1; ++local; }
        }
      }
      row[k] = res;
    }
    if (local) atomicAdd(&s_pair_count, local);
    __syncthreads();
    if (threadIdx.x == 0) counts[(size_t)il * slots + slot] = s_pair_count;
  }
}

// ===========================================================================================
// kd_tree_search (hitl_options): the reference's OWN search.  BuildKDTrees (JointOptimization.cpp:514-537) builds one
// median-split tree per scan (KDTree::BuildKDTree, kdtree.cpp:106-139: split plane of larger variance, kdtree.cpp:37-69;
// std::sort by that coordinate -- ties taken in their incoming order, i.e. the stable outcome; node = element n/2) and
// FindNearestPointNormal (kdtree.cpp:141-197) walks it, pruning sub-trees with the PLANE distance.  On flat walls many
// target points score almost equally, so which one the pruned walk returns depends on the tree: its index sets differ
// from the brute-force definition on a large share of the matches.  Both pieces are reproduced here bit for
// bit (fp32, no FMA contraction, the same order of every sum), so that with this option the index sets are the
// reference's own.
//
// Tree layout: the scan's points in the order the recursive sorts leave them ("in-order"): the node of segment [lo, lo+n)
// sits at lo + n/2, its children are the nodes of [lo, lo + n/2) and [lo + n/2 + 1, lo + n).  No pointers.
// ===========================================================================================
constexpr int kKdBlock = 256;

// one workgroup per scan; level by level: split plane per live segment (one thread walks the segment: the fp32 sums of
// kdtree.cpp:40-58 are order-dependent), stable rank sort inside every segment, the median becomes a node.
// LDS: the scan's coordinates by original index (ox, oy), the 16-bit permutation that is being sorted (cur -> nxt) and the
// coordinates in the CURRENT order (cx, cy: gathered by all threads once per level, so that the sequential sums and the
// rank counts read contiguous memory instead of chasing the permutation): 26 bytes per point, scans of ~6000 points.
__global__ __launch_bounds__(kKdBlock) void k_kd_build(MapView map, int max_scan, float4* __restrict__ kd_pts, uint32_t* __restrict__ kd_meta) {
  extern __shared__ float s_raw[];
  float* ox = s_raw;                                    // coordinates by original index
  float* oy = ox + max_scan;
  float* cx = oy + max_scan;                            // coordinates in the current order
  float* cy = cx + max_scan;
  uint32_t* seg = (uint32_t*)(cy + max_scan);           // lo | n << 16 of the segment the element at this position is in (n == 0: already a node)
  uint16_t* cur = (uint16_t*)(seg + max_scan);          // original index of the element at this position
  uint16_t* nxt = cur + max_scan;
  unsigned char* s_plane = (unsigned char*)(nxt + max_scan);   // split plane of the live segment that starts at this position
  unsigned char* s_split = s_plane + max_scan;                 // split plane of the node sitting at this position
  __shared__ int s_live;
  const int i = blockIdx.x;
  const int64_t o = map.obs_off[i];
  const int m = (int)(map.obs_off[i + 1] - o);
  if (m <= 0) return;
  for (int p = threadIdx.x; p < m; p += kKdBlock) {
    const float4 v = map.obs[o + p];
    ox[p] = v.x; oy[p] = v.y; cx[p] = v.x; cy[p] = v.y; cur[p] = (uint16_t)p; seg[p] = (uint32_t)m << 16; s_plane[p] = 0; s_split[p] = 0;
  }
  __syncthreads();
  for (;;) {
    if (threadIdx.x == 0) s_live = 0;
    __syncthreads();
    // 1. split plane of every live segment (its first position does the sequential sums, in the order of the segment)
    for (int p = threadIdx.x; p < m; p += kKdBlock) {
      const uint32_t sg = seg[p];
      const int n = (int)(sg >> 16);
      if (n > 0 && (int)(sg & 0xffffu) == p) {
        // the sums run in segment order (fp32, order-dependent); eight elements are loaded at a time so that the adds wait for
        // one LDS round trip per eight elements instead of one each
        float mx = 0.0f, my = 0.0f;
        int q = p;
        for (; q + 8 <= p + n; q += 8) {
          float vx[8], vy[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) { vx[k] = cx[q + k]; vy[k] = cy[q + k]; }
#pragma unroll
          for (int k = 0; k < 8; ++k) { mx = mx + vx[k]; my = my + vy[k]; }
        }
        for (; q < p + n; ++q) { mx = mx + cx[q]; my = my + cy[q]; }
        mx = mx / (float)n; my = my / (float)n;
        float sx = 0.0f, sy = 0.0f;
        q = p;
        for (; q + 8 <= p + n; q += 8) {
          float vx[8], vy[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) { vx[k] = cx[q + k]; vy[k] = cy[q + k]; }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            sx = sx + (vx[k] - mx) * (vx[k] - mx);
            sy = sy + (vy[k] - my) * (vy[k] - my);
          }
        }
        for (; q < p + n; ++q) {
          sx = sx + (cx[q] - mx) * (cx[q] - mx);
          sy = sy + (cy[q] - my) * (cy[q] - my);
        }
        int plane = 0;
        float best = 0.0f;
        if (sx > best) { plane = 0; best = sx; }
        if (sy > best) { plane = 1; best = sy; }
        s_plane[p] = (unsigned char)plane;
        s_live = 1;
      }
    }
    __syncthreads();
    if (!s_live) break;
    // 2. stable sort of every live segment by its split coordinate: new position = lo + #{smaller, or equal and earlier}
    for (int p = threadIdx.x; p < m; p += kKdBlock) {
      const uint32_t sg = seg[p];
      const int n = (int)(sg >> 16);
      int dst = p;
      if (n > 0) {
        const int lo = (int)(sg & 0xffffu);
        const float* co = s_plane[lo] ? cy : cx;
        const float key = co[p];
        int rank = 0;
        int q = lo;
        for (; q + 8 <= lo + n; q += 8) {                  // eight keys per LDS round trip
          float k8[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) k8[k] = co[q + k];
#pragma unroll
          for (int k = 0; k < 8; ++k) rank += (k8[k] < key || (k8[k] == key && q + k < p)) ? 1 : 0;
        }
        for (; q < lo + n; ++q) {
          const float kq = co[q];
          rank += (kq < key || (kq == key && q < p)) ? 1 : 0;
        }
        dst = lo + rank;
      }
      nxt[dst] = cur[p];
    }
    __syncthreads();
    // 3. the median of a segment becomes its node; the two halves are the next level's segments; coordinates follow the new order
    for (int p = threadIdx.x; p < m; p += kKdBlock) {
      const uint32_t sg = seg[p];
      const int n = (int)(sg >> 16);
      const int e = nxt[p];
      cur[p] = (uint16_t)e; cx[p] = ox[e]; cy[p] = oy[e];
      if (n > 0) {
        const int lo = (int)(sg & 0xffffu), ind = n / 2;
        if (p == lo + ind) { s_split[p] = s_plane[lo]; seg[p] = (uint32_t)lo; }
        else if (p < lo + ind) { seg[p] = (uint32_t)lo | ((uint32_t)ind << 16); }
        else { seg[p] = (uint32_t)(lo + ind + 1) | ((uint32_t)(n - ind - 1) << 16); }
      }
    }
    __syncthreads();
  }
  for (int p = threadIdx.x; p < m; p += kKdBlock) {
    const int e = cur[p];
    kd_pts[o + p] = map.obs[o + e];
    kd_meta[o + p] = (uint32_t)e | ((uint32_t)s_split[p] << 16);
  }
}

size_t kd_build_lds_bytes(int max_scan) { return (size_t)((max_scan + 3) & ~3) * 26; }

void launch_kd_build(hipStream_t st, const MapView& map, int max_scan, float4* kd_pts, uint32_t* kd_meta) {
  max_scan = (max_scan + 3) & ~3;            // keeps the 4-, 2- and 1-byte arrays aligned behind one another
  const size_t lds = kd_build_lds_bytes(max_scan);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kd_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_kd_build, dim3(map.n_poses), dim3(kKdBlock), lds, st, map, max_scan, kd_pts, kd_meta);
}

// ---------------------------------------------------------------------------------------------------------------------
// The walk itself: hitl_kdwalk.h (`kd_step`, one node evaluation of one query per call; the same source is held to the
// literal unrolling of the recursion on the CPU in tests/test_kdwalk_cpu.py).
//
// k_match_kd_tgt -- one workgroup per TARGET pose.  Its tree is staged in LDS once and serves the queries of every source
// scan that has this pose in its window: one queue of (source, point) items per workgroup (2W x M ~ 2400 queries), handed
// out in order; a lane that finishes its query takes the next item, so the 64 walks of a wave -- each at its own node, at
// its own depth -- keep the lanes busy (lock-step simulation in tests/test_kdwalk_cpu.py: 95 % with refills when >= 8 lanes
// are idle) instead of waiting for the longest walk of a static round.  Legal only when the per-point cap cannot bind:
// FindSTFCorrespondences counts a source point's accepted matches across its targets in ascending order
// (JointOptimization.cpp:596-599), which couples the targets of one source pose -- but a point that has seen s targets has
// at most s matches, so with (number of slots) <= cap the test is vacuous and every (source, target) pair is independent.
// k_match_kd (source-major, targets in order, counters in LDS) serves every other case with the same `kd_step`.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kKdMaxSrc = 2 * kMaxBandPoses;      // slots of a windowed search inside the band
static int kd_depth(int max_scan) { int d = 1; while ((1 << d) <= max_scan) ++d; return d + 1; }
// Frame-stack entries per lane below the register-resident top.  The stack is provisioned per lane in LDS and decides how
// many workgroups a CU holds (C4: full depth 10 -> 3 workgroups of 256, search 9.1 ms; 5 entries -> 6 workgroups, 6.8 ms).
// Scan queries need 1.1 frames on average and 6 or fewer in all but ~1 of 10^5 walks, so the fast attempt gives every lane
// kKdFastCap entries; a walk that wants more is put on a list and repeated by k_match_kd_redo right behind (full depth, private
// stack).  If even the list overflows (never seen; the test hook kd_cap = 1 forces it) the host repeats the whole search at full depth.
constexpr int kKdFastCap = 5;
static int kd_tgt_cap(int max_scan, bool full) {
  const int d = kd_depth(max_scan);
  if (full) return d;
  const int c = dev_switches().kd_cap > 0 ? dev_switches().kd_cap : kKdFastCap;
  return c < d ? c : d;
}

// FAST: kd_step_fast calls behind every kd_step (hitl_kdwalk.h: the cheap transitions of a walk need neither the frame stack nor
// the return logic)
template <int BLOCK, int FAST = 1>
__global__ __launch_bounds__(BLOCK) void k_match_kd_tgt(MapView map, int chunk, int max_scan, int slots, int window, int depth, int refill_min,
                                                         const float* __restrict__ cand_tf, MatchParams mp,
                                                         const float4* __restrict__ kd_pts, const uint32_t* __restrict__ kd_meta,
                                                         uint16_t* __restrict__ table, int* __restrict__ counts, int* __restrict__ redo_count,
                                                         int* __restrict__ redo_list, int redo_cap) {
  extern __shared__ float4 s_tgt[];                 // [max_scan] target tree points (in-order layout)
  uint32_t* s_meta = (uint32_t*)(s_tgt + max_scan); // [max_scan] original index | split << 16
  float* st_inc = (float*)(s_meta + max_scan);      // [depth + 1][BLOCK] frame stack of kd_step below its register-resident top (+ a dump entry):
  uint32_t* st_seg = (uint32_t*)(st_inc + (size_t)(depth + 1) * BLOCK);      // saved incumbent, segment,
  uint32_t* st_oc = st_seg + (size_t)(depth + 1) * BLOCK;                    // incumbent position | code
  __shared__ int s_next, s_total, s_nsrc;
  __shared__ int s_count[kKdMaxSrc], s_cand[kKdMaxSrc], s_start[kKdMaxSrc + 1];
  __shared__ long long s_obs[kKdMaxSrc];
  __shared__ float s_tf[kKdMaxSrc][8];
  const int b = blockIdx.x;
  const int idx = b >> 3;
  const int j = (b & 7) * chunk + idx;              // consecutive poses share an XCD (L2 reuse of the neighbouring scans)
  if (idx >= chunk || j >= map.n_poses) return;
  const int64_t oj = map.obs_off[j];
  const int mj = (int)(map.obs_off[j + 1] - oj);
  if (threadIdx.x == 0) {
    // the sources that have j in their window, ascending; slot of j in source i's candidate list as fill_candidates lays it out
    int ns = 0, start = 0;
    const int i0 = max(0, j - window), i1 = min(map.n_poses - 1, j + window);
    for (int i = i0; i <= i1; ++i) {
      if (i == j) continue;
      const int sl = (j - max(0, i - window)) - (j > i ? 1 : 0);
      const int64_t oi = map.obs_off[i];
      const int mi = (int)(map.obs_off[i + 1] - oi);
      s_cand[ns] = i * slots + sl; s_obs[ns] = oi; s_start[ns] = start; s_count[ns] = 0;
      start += (mi + mp.skip - 1) / mp.skip;        // items = source points k with k % skip == 0
      ++ns;
    }
    s_start[ns] = start; s_nsrc = ns; s_total = start; s_next = 0;
  }
  for (int t = threadIdx.x; t < mj; t += BLOCK) { s_tgt[t] = kd_pts[oj + t]; s_meta[t] = kd_meta[oj + t]; }
  __syncthreads();
  const int ns = s_nsrc, total = s_total;
  for (int t = threadIdx.x; t < ns * 8; t += BLOCK) s_tf[t >> 3][t & 7] = cand_tf[(size_t)s_cand[t >> 3] * 8 + (t & 7)];
  __syncthreads();
  const float thr = mp.thr, thr2 = mp.thr * mp.thr;
  float* my_inc = st_inc + threadIdx.x;
  uint32_t* my_seg = st_seg + threadIdx.x;
  uint32_t* my_oc = st_oc + threadIdx.x;
  KdWalk w;
  w.mode = 2; w.qx = w.qy = 0.0f; w.lo = w.n = 0; w.ret_d = 0.0f; w.ret_c = -1; w.inc_d = 0.0f; w.inc_c = -1; w.sp = 0; w.ovf = 0;
  w.t_inc = 0.0f; w.t_seg = 0u; w.t_oc = 0u;
  bool have = false;                                // this lane holds a query (running or finished, not yet written)
  int src = 0, k = 0;
  float nqx = 0.0f, nqy = 0.0f;
  bool more = total > 0;                            // wave-uniform: the queue may still hold items
  for (;;) {
    const bool idle = (w.mode == 2);
    const int nidle = __popcll(__ballot(idle));
    if (nidle >= (more ? refill_min : 64) || nidle == 64) {
      bool got = false;
      if (idle) {
        if (have) {                                 // finish: JointOptimization.cpp:606-621 on the walk's winner
          uint16_t res = 0xFFFFu;
          if (w.ovf) {                              // the walk needed more frames than this launch gives a lane: k_match_kd_redo repeats it
            const int at = atomicAdd(redo_count, 1);
            if (at < redo_cap) { redo_list[2 * at] = s_cand[src]; redo_list[2 * at + 1] = k; }
          } else if (w.ret_c >= 0 && w.ret_d < thr) {
            const float4 g = s_tgt[w.ret_c];
            const float cosang = g.z * nqx + g.w * nqy;
            if (cosang > mp.min_cos) { res = (uint16_t)(s_meta[w.ret_c] & 0xffffu); atomicAdd(&s_count[src], 1); }
          }
          table[(size_t)s_cand[src] * max_scan + k] = res;
          have = false;
        }
        if (more) {
          const int t = atomicAdd(&s_next, 1);
          if (t < total) {
            int sidx = 0;
#pragma unroll
            for (int q = 1; q < kKdMaxSrc; ++q) sidx += (q < ns && t >= s_start[q]) ? 1 : 0;
            src = sidx;
            k = (t - s_start[sidx]) * mp.skip;
            const float4 a = map.obs[s_obs[sidx] + k];
            const float* tf = s_tf[sidx];
            const float qx = (tf[0] * a.x + tf[1] * a.y) + tf[4];
            const float qy = (tf[2] * a.x + tf[3] * a.y) + tf[5];
            nqx = tf[6] * a.z - tf[7] * a.w; nqy = tf[7] * a.z + tf[6] * a.w;
            kd_begin(w, qx, qy, mj);
            have = true; got = true;
          }
        }
      }
      if (more && !__any(got)) more = false;        // the queue ran dry (t >= total on every asking lane)
      if (!more && __all(w.mode == 2)) {
        if (__any(have)) continue;                  // walks of empty trees finish at once: one more round writes them
        break;
      }
    }
    kd_step(w, s_tgt, s_meta, thr, thr2, my_inc, my_seg, my_oc, BLOCK, depth);
#pragma unroll
    for (int f = 0; f < FAST; ++f) kd_step_fast(w, s_tgt, s_meta, thr, thr2);
  }
  __syncthreads();
  if ((int)threadIdx.x < ns) counts[s_cand[threadIdx.x]] = s_count[threadIdx.x];
}

// The few walks of k_match_kd_tgt's fast attempt that needed a deeper frame stack (about 1 query in 10^5 on corridor maps),
// once more: one thread per listed (candidate, source point), the tree read where it lies in global memory, a private
// full-depth stack.  Results go where the first pass would have put them; the candidate's count is completed atomically
// (integer adds: order-free, bit-reproducible).
__global__ __launch_bounds__(64) void k_match_kd_redo(MapView map, int max_scan, int slots, int window, const float* __restrict__ cand_tf, MatchParams mp,
                                                      const float4* __restrict__ kd_pts, const uint32_t* __restrict__ kd_meta,
                                                      uint16_t* __restrict__ table, int* __restrict__ counts, const int* __restrict__ redo_count,
                                                      const int* __restrict__ redo_list, int redo_cap) {
  constexpr int kDeep = 18;                                    // scans of up to 65535 points: 17 levels
  const int total = min(*redo_count, redo_cap);
  const float thr = mp.thr, thr2 = mp.thr * mp.thr;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int cand = redo_list[2 * t], k = redo_list[2 * t + 1];
    const int i = cand / slots, sl = cand % slots;
    const int jj = max(0, i - window) + sl;
    const int j = (jj >= i) ? jj + 1 : jj;                     // the slot layout of fill_candidates
    const int64_t oj = map.obs_off[j];
    const int mj = (int)(map.obs_off[j + 1] - oj);
    const float4 a = map.obs[map.obs_off[i] + k];
    const float* tf = cand_tf + (size_t)cand * 8;
    const float qx = (tf[0] * a.x + tf[1] * a.y) + tf[4];
    const float qy = (tf[2] * a.x + tf[3] * a.y) + tf[5];
    float st_inc[kDeep];
    uint32_t st_seg[kDeep], st_oc[kDeep];
    KdWalk w;
    w.t_inc = 0.0f; w.t_seg = 0u; w.t_oc = 0u;
    kd_begin(w, qx, qy, mj);
    while (w.mode != 2) kd_step(w, kd_pts + oj, kd_meta + oj, thr, thr2, st_inc, st_seg, st_oc, 1, kDeep - 1);
    uint16_t res = 0xFFFFu;
    if (w.ret_c >= 0 && w.ret_d < thr) {
      const float nqx = tf[6] * a.z - tf[7] * a.w, nqy = tf[7] * a.z + tf[6] * a.w;
      const float4 g = kd_pts[oj + w.ret_c];
      const float cosang = g.z * nqx + g.w * nqy;
      if (cosang > mp.min_cos) { res = (uint16_t)(kd_meta[oj + w.ret_c] & 0xffffu); atomicAdd(&counts[cand], 1); }
    }
    table[(size_t)cand * max_scan + k] = res;
  }
}

// ===========================================================================================
// Occupancy grid of a scan (round 6): the all-pairs search asks a target's tree about EVERY point of EVERY other scan -- 360 M walks on a
// 1 000-pose map, of which 0.6 % end in a match.  FindNearestPointNormal (kdtree.cpp:141-197) can only return a node that lies within
// `threshold` of the query in the plane (kdtree.cpp:147: the gate in front of every score), so a query with NO scan point within the
// threshold has no match whatever the walk does.  The grid answers that conservatively in one bit test: cells of at least 1.25 x the
// threshold over the scan's bounding box (two cells of margin), a cell's bit set when the cell or one of its eight neighbours holds
// a point.  Two points closer than the threshold lie in the same or in neighbouring cells (their cell coordinates differ by less than
// 0.8 + rounding), so a query whose cell bit is clear -- or that falls outside the grid -- skips its walk: the result (no match, the
// per-point counter untouched) is exactly the walk's.
// Layout per pose: kKdGridWords words: [0] x0, [1] y0, [2] 1 / cell (floats), [3] nx, [4] ny, [8 ...] the bits, row-major.
// nx = 0: the scan has no point that could match (empty, or NaN coordinates only); nx = -1: points, but no grid (all queries walk).
// ===========================================================================================
constexpr int kKdGridBits = 32768;
constexpr int kKdGridWords = 8 + kKdGridBits / 32;
size_t kd_grid_words() { return kKdGridWords; }

__global__ __launch_bounds__(256) void k_kd_grid(MapView map, float thr, const float4* __restrict__ kd_pts, uint32_t* __restrict__ grid) {
  __shared__ uint32_t bits[kKdGridBits / 32];
  __shared__ float s_min[2][4], s_max[2][4];
  __shared__ float s_hdr[3];
  __shared__ int s_dim[2];
  const int j = blockIdx.x, tid = threadIdx.x;
  const int64_t oj = map.obs_off[j];
  const int mj = (int)(map.obs_off[j + 1] - oj);
  uint32_t* out = grid + (size_t)j * kKdGridWords;
  for (int t = tid; t < kKdGridBits / 32; t += 256) bits[t] = 0u;
  float mnx = 3.0e38f, mny = 3.0e38f, mxx = -3.0e38f, mxy = -3.0e38f;
  for (int t = tid; t < mj; t += 256) {
    const float4 g = kd_pts[oj + t];
    if (g.x == g.x && g.y == g.y) { mnx = fminf(mnx, g.x); mny = fminf(mny, g.y); mxx = fmaxf(mxx, g.x); mxy = fmaxf(mxy, g.y); }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    mnx = fminf(mnx, __shfl_xor(mnx, m, 64)); mny = fminf(mny, __shfl_xor(mny, m, 64));
    mxx = fmaxf(mxx, __shfl_xor(mxx, m, 64)); mxy = fmaxf(mxy, __shfl_xor(mxy, m, 64));
  }
  if ((tid & 63) == 0) { s_min[0][tid >> 6] = mnx; s_min[1][tid >> 6] = mny; s_max[0][tid >> 6] = mxx; s_max[1][tid >> 6] = mxy; }
  __syncthreads();
  if (tid == 0) {
    float ax = s_min[0][0], ay = s_min[1][0], bx = s_max[0][0], by = s_max[1][0];
    for (int k = 1; k < 4; ++k) { ax = fminf(ax, s_min[0][k]); ay = fminf(ay, s_min[1][k]); bx = fmaxf(bx, s_max[0][k]); by = fmaxf(by, s_max[1][k]); }
    int nx = 0, ny = 0;
    float cell = 1.25f * thr, x0 = 0.0f, y0 = 0.0f, inv = 0.0f;
    const bool any = mj > 0 && ax <= bx && ay <= by;              // a point without NaN coordinates exists
    if (any && cell > 0.0f && bx - ax < 1.0e30f && by - ay < 1.0e30f) {
      for (int it = 0; it < 64; ++it) {
        inv = 1.0f / cell;
        x0 = ax - 2.0f * cell; y0 = ay - 2.0f * cell;
        const float fx = floorf((bx - x0) * inv), fy = floorf((by - y0) * inv);
        if (fx < 30000.0f && fy < 30000.0f) {
          nx = (int)fx + 3; ny = (int)fy + 3;
          if ((long long)nx * ny <= kKdGridBits) break;
        }
        nx = ny = 0;
        cell *= 1.5f;                                  // a long-range scan: coarser cells (still >= 1.25 thr: still conservative)
      }
    }
    if (any && nx == 0) nx = ny = -1;                  // points, but no grid over them (infinite coordinates, a zero threshold): every query of this target walks
    s_hdr[0] = x0; s_hdr[1] = y0; s_hdr[2] = inv; s_dim[0] = nx; s_dim[1] = ny;
  }
  __syncthreads();
  const float x0 = s_hdr[0], y0 = s_hdr[1], inv = s_hdr[2];
  const int nx = s_dim[0], ny = s_dim[1];
  if (nx > 0) {
    for (int t = tid; t < mj; t += 256) {
      const float4 g = kd_pts[oj + t];
      if (!(g.x == g.x && g.y == g.y)) continue;
      const int ix = (int)floorf((g.x - x0) * inv), iy = (int)floorf((g.y - y0) * inv);
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int cx = ix + dx, cy = iy + dy;
          if (cx >= 0 && cx < nx && cy >= 0 && cy < ny) { const int b = cy * nx + cx; atomicOr(&bits[b >> 5], 1u << (b & 31)); }
        }
    }
  }
  __syncthreads();
  if (tid < 8) {
    uint32_t v = 0u;
    if (tid == 0) v = __float_as_uint(x0); else if (tid == 1) v = __float_as_uint(y0); else if (tid == 2) v = __float_as_uint(inv);
    else if (tid == 3) v = (uint32_t)nx; else if (tid == 4) v = (uint32_t)ny;
    out[tid] = v;
  }
  for (int t = tid; t < kKdGridBits / 32; t += 256) out[8 + t] = bits[t];
}

void launch_kd_grid(hipStream_t st, const MapView& map, float thr, const float4* kd_pts, uint32_t* grid) {
  if (map.n_poses <= 0) return;
  hipLaunchKernelGGL(k_kd_grid, dim3(map.n_poses), dim3(256), 0, st, map, thr, kd_pts, grid);
}

// source-major form (any number of slots, tiles of source poses): targets in ascending order, the per-point counters of
// JointOptimization.cpp:596-599 in LDS.  Round 6: three stages per (source, target) pair, each cheaper than the next and each
// deciding whether the next one runs at all --
//   0. (uniform) the source scan's bounding box, carried into the target's frame, against the extent of the target's grid: poses
//      that look at different places are done after ~30 scalar-ish instructions;
//   1. every query against the grid's bits, read through the caches (a grid is 0.1-4 KB, the 1 000 grids of a map stay in L2):
//      no LDS staging, ONE barrier (the workgroup-wide OR).  No bit set anywhere: the pair has no match, its count stays at the
//      zero the driver cleared it to, and its row of the candidate table is never written (k_compact only reads rows of kept pairs);
//   2. only now the target's tree comes into LDS (8.6 KB for 360 points: staging it for each of the 10^6 pairs of a 1 000-pose map
//      WAS the search's time: 8.6 GB through L2), the surviving queries are compacted into a list so that the walks fill whole
//      waves, and walk.
// Without grids (test hook kd_grid = 0) stages 0 and 1 pass everything: rounds 3-5's search.
__global__ __launch_bounds__(kMatchBlock) void k_match_kd(MapView map, int chunk, int max_scan, int slots, int depth,
                                                           const int* __restrict__ cand_tgt, const float* __restrict__ cand_tf, MatchParams mp,
                                                           const float4* __restrict__ kd_pts, const uint32_t* __restrict__ kd_meta,
                                                           const uint32_t* __restrict__ kd_grid /* k_kd_grid, or null: every query walks */,
                                                           uint16_t* __restrict__ table, int* __restrict__ counts, int i0, int n_tile,
                                                           unsigned long long* __restrict__ stats /* HITL_TIMING: pairs past stage 0, past stage 1, walks; or null */) {
  extern __shared__ float4 s_tgt[];                 // [max_scan] target tree points (in-order layout)
  uint32_t* s_meta = (uint32_t*)(s_tgt + max_scan); // [max_scan] original index | split << 16
  int* s_cnt = (int*)(s_meta + max_scan);           // [max_scan] accepted matches per source point
  float* st_inc = (float*)(s_cnt + max_scan);       // [depth + 1][kMatchBlock] frame stack of kd_step below its register-resident top (+ a dump entry)
  uint32_t* st_seg = (uint32_t*)(st_inc + (size_t)(depth + 1) * kMatchBlock);
  uint32_t* st_oc = st_seg + (size_t)(depth + 1) * kMatchBlock;
  unsigned long long n_st0 = 0, n_st1 = 0, n_walk = 0, t_st1 = 0, t_st2 = 0;       // (t_*: ticks of the 100 MHz clock)
  uint16_t* s_list = (uint16_t*)(st_oc + (size_t)(depth + 1) * kMatchBlock);      // [max_scan] the queries that walk
  __shared__ int s_pair_count, s_npass;
  __shared__ float s_box[4][kMatchBlock / 64];
  const int b = blockIdx.x;
  const int idx = b >> 3;
  const int il = (b & 7) * chunk + idx;
  if (idx >= chunk || il >= n_tile) return;
  const int i = i0 + il;                            // tile [i0, i0 + n_tile): see k_match
  const int64_t oi = map.obs_off[i];
  const int mi = (int)(map.obs_off[i + 1] - oi);
  for (int k = threadIdx.x; k < mi; k += kMatchBlock) s_cnt[k] = 0;
  if (threadIdx.x == 0) { s_pair_count = 0; s_npass = 0; }
  // stage 0's box of the source scan (its own frame; NaN points stay out: they match nothing)
  float bx0 = 3.0e38f, by0 = 3.0e38f, bx1 = -3.0e38f, by1 = -3.0e38f;
  for (int k = threadIdx.x; k < mi; k += kMatchBlock) {
    const float4 a = map.obs[oi + k];
    if (a.x == a.x && a.y == a.y) { bx0 = fminf(bx0, a.x); by0 = fminf(by0, a.y); bx1 = fmaxf(bx1, a.x); by1 = fmaxf(by1, a.y); }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    bx0 = fminf(bx0, __shfl_xor(bx0, m, 64)); by0 = fminf(by0, __shfl_xor(by0, m, 64));
    bx1 = fmaxf(bx1, __shfl_xor(bx1, m, 64)); by1 = fmaxf(by1, __shfl_xor(by1, m, 64));
  }
  if ((threadIdx.x & 63) == 0) { const int wv = threadIdx.x >> 6; s_box[0][wv] = bx0; s_box[1][wv] = by0; s_box[2][wv] = bx1; s_box[3][wv] = by1; }
  __syncthreads();
  bx0 = s_box[0][0]; by0 = s_box[1][0]; bx1 = s_box[2][0]; by1 = s_box[3][0];
#pragma unroll
  for (int wv = 1; wv < kMatchBlock / 64; ++wv) { bx0 = fminf(bx0, s_box[0][wv]); by0 = fminf(by0, s_box[1][wv]); bx1 = fmaxf(bx1, s_box[2][wv]); by1 = fmaxf(by1, s_box[3][wv]); }
  const bool box_ok = bx0 <= bx1 && by0 <= by1 && bx1 - bx0 < 1.0e30f && by1 - by0 < 1.0e30f;      // (false: no usable box -- stage 0 passes everything)
  const float thr = mp.thr, thr2 = mp.thr * mp.thr;
  for (int slot = 0; slot < slots; ++slot) {
    const int j = cand_tgt[(size_t)il * slots + slot];
    if (j < 0) continue;                            // block-uniform
    const int64_t oj = map.obs_off[j];
    const int mj = (int)(map.obs_off[j + 1] - oj);
    if (mj <= 0) continue;                          // an empty scan: nothing to find (count stays 0)
    const float* tf = cand_tf + ((size_t)il * slots + slot) * 8;
    const float m00 = tf[0], m01 = tf[1], m10 = tf[2], m11 = tf[3], tx = tf[4], ty = tf[5], cr = tf[6], sr = tf[7];
    const uint32_t* gj = nullptr;
    float gx0 = 0.0f, gy0 = 0.0f, ginv = 0.0f;
    int gnx = 0, gny = 0;
    if (kd_grid != nullptr) {
      gj = kd_grid + (size_t)j * kKdGridWords;
      gx0 = __uint_as_float(gj[0]); gy0 = __uint_as_float(gj[1]); ginv = __uint_as_float(gj[2]); gnx = (int)gj[3]; gny = (int)gj[4];
      if (gnx == 0) continue;                       // the target has no point without NaN coordinates: no query can pass the gate (kdtree.cpp:147)
      if (gnx < 0) gj = nullptr;                    // no grid could be laid over this target (k_kd_grid): all its queries walk
      else if (box_ok) {
        // stage 0: the four corners of the source's box in the target's frame; one cell of slack on every side (the grid itself keeps
        // two cells = 2.5 thresholds around the target's points, a match needs less than one threshold: 1.5 thresholds minus this cell
        // of slack remain for fp32 rounding of the corners against the queries -- metres against micrometres)
        const float c0x = (m00 * bx0 + m01 * by0) + tx, c0y = (m10 * bx0 + m11 * by0) + ty;
        const float c1x = (m00 * bx1 + m01 * by0) + tx, c1y = (m10 * bx1 + m11 * by0) + ty;
        const float c2x = (m00 * bx0 + m01 * by1) + tx, c2y = (m10 * bx0 + m11 * by1) + ty;
        const float c3x = (m00 * bx1 + m01 * by1) + tx, c3y = (m10 * bx1 + m11 * by1) + ty;
        const float lo_x = (fminf(fminf(c0x, c1x), fminf(c2x, c3x)) - gx0) * ginv, hi_x = (fmaxf(fmaxf(c0x, c1x), fmaxf(c2x, c3x)) - gx0) * ginv;
        const float lo_y = (fminf(fminf(c0y, c1y), fminf(c2y, c3y)) - gy0) * ginv, hi_y = (fmaxf(fmaxf(c0y, c1y), fmaxf(c2y, c3y)) - gy0) * ginv;
        if (hi_x < -1.0f || lo_x > (float)gnx + 1.0f || hi_y < -1.0f || lo_y > (float)gny + 1.0f) continue;
      }
    }
    // stage 1: which of this thread's queries may find anything (bit p: the query k = threadIdx.x + p * kMatchBlock; scans of up to 32 x 256 points -- the
    // tree kernels' LDS budget ends at ~5 900)
    uint32_t mask = 0u;
    ++n_st0;
    const unsigned long long t_a = stats != nullptr ? wall_clock64() : 0ull;
    {
      int p = 0;
      for (int k = threadIdx.x; k < mi; k += kMatchBlock, ++p) {
        bool ok = (k % mp.skip) == 0 && s_cnt[k] < mp.cap;
        if (ok && gj != nullptr) {
          const float4 a = map.obs[oi + k];
          const float qx = (m00 * a.x + m01 * a.y) + tx, qy = (m10 * a.x + m11 * a.y) + ty;
          const float fx = floorf((qx - gx0) * ginv), fy = floorf((qy - gy0) * ginv);
          ok = fx >= 0.0f && fx < (float)gnx && fy >= 0.0f && fy < (float)gny;        // (NaN coordinates: false, as every gate of the walk would be)
          if (ok) { const int bit = (int)fy * gnx + (int)fx; ok = (gj[8 + (bit >> 5)] >> (bit & 31)) & 1u; }
        }
        mask |= ok ? (1u << p) : 0u;
      }
    }
    const bool any_q = __syncthreads_or(mask != 0u);
    const unsigned long long t_b = stats != nullptr ? wall_clock64() : 0ull;
    t_st1 += t_b - t_a;
    if (!any_q) continue;                           // no query has a point of the target within the threshold: FindNearestPointNormal would return "none" 360 times
    // stage 2: the tree into LDS, the surviving queries into a list, rows of the others closed
    for (int t = threadIdx.x; t < mj; t += kMatchBlock) { s_tgt[t] = kd_pts[oj + t]; s_meta[t] = kd_meta[oj + t]; }
    uint16_t* row = table + ((size_t)il * slots + slot) * max_scan;
    {
      int p = 0;
      for (int k = threadIdx.x; k < mi; k += kMatchBlock, ++p) {
        if ((mask >> p) & 1u) s_list[atomicAdd(&s_npass, 1)] = (uint16_t)k;      // (any order: every query's result depends on k alone)
        else row[k] = 0xFFFFu;
      }
    }
    __syncthreads();
    const int np = s_npass;
    ++n_st1; n_walk += (unsigned)np;
    int local = 0;
    for (int e = threadIdx.x; e < np; e += kMatchBlock) {
      const int k = s_list[e];
      uint16_t res = 0xFFFFu;
      const float4 a = map.obs[oi + k];
      const float ax = a.x, ay = a.y;
      const float qx = (m00 * ax + m01 * ay) + tx;
      const float qy = (m10 * ax + m11 * ay) + ty;
      KdWalk w;
      w.ovf = 0; w.t_inc = 0.0f; w.t_seg = 0u; w.t_oc = 0u;
      kd_begin(w, qx, qy, mj);
      while (w.mode != 2) kd_step(w, s_tgt, s_meta, thr, thr2, st_inc + threadIdx.x, st_seg + threadIdx.x, st_oc + threadIdx.x, kMatchBlock, depth);
      if (w.ret_c >= 0 && w.ret_d < thr) {
        const float anx = a.z, any = a.w;
        const float nqx = cr * anx - sr * any, nqy = sr * anx + cr * any;
        const float4 g = s_tgt[w.ret_c];
        const float cosang = g.z * nqx + g.w * nqy;
        if (cosang > mp.min_cos) { res = (uint16_t)(s_meta[w.ret_c] & 0xffffu); s_cnt[k] += 1; ++local; }
      }
      row[k] = res;
    }
    if (local) atomicAdd(&s_pair_count, local);
    __syncthreads();
    if (threadIdx.x == 0) { counts[(size_t)il * slots + slot] = s_pair_count; s_pair_count = 0; s_npass = 0; }      // (both untouched by the others until the next pair's OR barrier)
    if (stats != nullptr) t_st2 += wall_clock64() - t_b;
  }
  if (stats != nullptr && threadIdx.x == 0) {
    atomicAdd(stats, n_st0); atomicAdd(stats + 1, n_st1); atomicAdd(stats + 2, n_walk);
    atomicAdd(stats + 3, t_st1); atomicAdd(stats + 4, t_st2); atomicMax(stats + 5, t_st1 + t_st2);
  }
}


// The same search, G targets at a time (round 6).  k_match_kd's time on a map that revisits places is the time of its SLOWEST
// workgroup: a source pose whose points keep finding neighbours but rarely a match (the per-point cap of JointOptimization.cpp:596-599
// never closes them) stages a tree and walks for every one of its 999 targets, ~13 us of dependent latency each, one wave busy.
// The cap is the only thing that orders a source point's targets -- whether point k is ASKED about target j depends on its accepted
// matches with earlier targets; what the walk RETURNS does not.  So: the next G targets are examined together against the counters
// as they stand (a superset of the queries the sequential order would make: counters only grow), wave g stages target g's tree and
// walks its list, and then every point's results are committed in ascending target order by the thread that owns the point --
// accepted while the counter is below the cap, dropped (0xFFFF, not counted) after.  Identical rows, counts and counters; G trees
// and G walks in flight per workgroup, and the G bit tests of stage 1 overlap their memory latency.
template <int G>
__global__ __launch_bounds__(64 * G) void k_match_kd_spec(MapView map, int chunk, int max_scan, int slots, int depth,
                                                           const int* __restrict__ cand_tgt, const float* __restrict__ cand_tf, MatchParams mp,
                                                           const float4* __restrict__ kd_pts, const uint32_t* __restrict__ kd_meta,
                                                           const uint32_t* __restrict__ kd_grid, uint16_t* __restrict__ table, int* __restrict__ counts,
                                                           int i0, int n_tile, unsigned long long* __restrict__ stats,
                                                           const int* __restrict__ order /* workgroup -> pose of the tile | part << 20 | parts << 24, or null */,
                                                           int n_wg /* entries of order */,
                                                           unsigned* __restrict__ cost /* [n_tile] ticks this pose's workgroup took (zeroed by the driver), or null */) {
  constexpr int BLOCK = 64 * G;
  extern __shared__ float4 s_tgt[];                               // [G][max_scan] target trees (in-order layout)
  uint32_t* s_meta = (uint32_t*)(s_tgt + (size_t)G * max_scan);   // [G][max_scan]
  int* s_cnt = (int*)(s_meta + (size_t)G * max_scan);             // [max_scan] accepted matches per source point
  float* st_inc = (float*)(s_cnt + max_scan);                     // [depth + 1][BLOCK] frame stacks (see k_match_kd)
  uint32_t* st_seg = (uint32_t*)(st_inc + (size_t)(depth + 1) * BLOCK);
  uint32_t* st_oc = st_seg + (size_t)(depth + 1) * BLOCK;
  uint16_t* s_list = (uint16_t*)(st_oc + (size_t)(depth + 1) * BLOCK);      // [G][max_scan] the queries that walk
  uint16_t* s_res = s_list + (size_t)G * max_scan;                          // [G][max_scan] what the walks returned, by source point
  __shared__ int s_np[2][G], s_pc[2][G];
  __shared__ unsigned s_any[2];
  __shared__ float s_box[4][G];
  unsigned long long n_st0 = 0, n_st1 = 0, n_walk = 0, t_st1 = 0, t_st2 = 0;
  const int b = blockIdx.x;
  const int idx = b >> 3;
  // The kernel ends with its slowest workgroup, and two workgroups per CU mean two rounds on a map of 1 000 poses: with an order
  // (the poses sorted by what their workgroups took in the previous search of this map) the long ones start first
  // An order entry is pose | part << 20 | parts << 24: a pose whose workgroup took much longer than the others' last time is served by
  // `parts` workgroups, each with the points k = part (mod parts) -- exact, because the cap counts per POINT: a point's targets are
  // ordered, the points of a scan are independent of each other.  The parts add their counts of a pair (atomics on the zeroed
  // counts) and write their own points' entries of its row (the driver pre-fills the rows of split poses with 0xFFFF: a part that
  // stages no tree for a pair leaves its entries alone).
  const int code = order != nullptr ? (b < n_wg ? order[b] : -1) : 0;
  const int il = order != nullptr ? (code < 0 ? n_tile : (code & 0xFFFFF)) : (b & 7) * chunk + idx;
  const int part = order != nullptr && code >= 0 ? ((code >> 20) & 15) : 0;
  const int parts = order != nullptr && code >= 0 && ((code >> 24) & 15) > 1 ? ((code >> 24) & 15) : 1;
  if ((order == nullptr && idx >= chunk) || il >= n_tile) return;
  const int k0 = part + parts * (int)threadIdx.x, kstep = parts * BLOCK;          // this thread's points: k0, k0 + kstep, ...
  const unsigned long long t_begin = cost != nullptr ? wall_clock64() : 0ull;
  const int i = i0 + il;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t oi = map.obs_off[i];
  const int mi = (int)(map.obs_off[i + 1] - oi);
  for (int k = k0; k < mi; k += kstep) s_cnt[k] = 0;
  if (tid < G) { s_np[0][tid] = 0; s_np[1][tid] = 0; s_pc[0][tid] = 0; s_pc[1][tid] = 0; }
  if (tid < 2) s_any[tid] = 0u;
  float bx0 = 3.0e38f, by0 = 3.0e38f, bx1 = -3.0e38f, by1 = -3.0e38f;
  for (int k = k0; k < mi; k += kstep) {
    const float4 a = map.obs[oi + k];
    if (a.x == a.x && a.y == a.y) { bx0 = fminf(bx0, a.x); by0 = fminf(by0, a.y); bx1 = fmaxf(bx1, a.x); by1 = fmaxf(by1, a.y); }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    bx0 = fminf(bx0, __shfl_xor(bx0, m, 64)); by0 = fminf(by0, __shfl_xor(by0, m, 64));
    bx1 = fmaxf(bx1, __shfl_xor(bx1, m, 64)); by1 = fmaxf(by1, __shfl_xor(by1, m, 64));
  }
  if (lane == 0) { s_box[0][wave] = bx0; s_box[1][wave] = by0; s_box[2][wave] = bx1; s_box[3][wave] = by1; }
  __syncthreads();
  bx0 = s_box[0][0]; by0 = s_box[1][0]; bx1 = s_box[2][0]; by1 = s_box[3][0];
#pragma unroll
  for (int wv = 1; wv < G; ++wv) { bx0 = fminf(bx0, s_box[0][wv]); by0 = fminf(by0, s_box[1][wv]); bx1 = fmaxf(bx1, s_box[2][wv]); by1 = fmaxf(by1, s_box[3][wv]); }
  const bool box_ok = bx0 <= bx1 && by0 <= by1 && bx1 - bx0 < 1.0e30f && by1 - by0 < 1.0e30f;
  const float thr = mp.thr, thr2 = mp.thr * mp.thr;
  int par = 0;
  for (int slot0 = 0; slot0 < slots; slot0 += G) {
    const unsigned long long t_a = stats != nullptr ? wall_clock64() : 0ull;
    // ---- per target of the group (all uniform): candidate, transform, grid header, stage 0 ----
    int tj[G], tmj[G], gnx[G], gny[G];
    int64_t toj[G];
    float tf[G][8], gx0[G], gy0[G], ginv[G];
    const uint32_t* gptr[G];
    bool live[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int slot = slot0 + g;
      tj[g] = slot < slots ? cand_tgt[(size_t)il * slots + slot] : -1;
      live[g] = tj[g] >= 0;
      const int jj = live[g] ? tj[g] : 0;
      toj[g] = map.obs_off[jj];
      tmj[g] = (int)(map.obs_off[jj + 1] - toj[g]);
      live[g] = live[g] && tmj[g] > 0;
      const float* t = cand_tf + ((size_t)il * slots + (live[g] ? slot : 0)) * 8;
#pragma unroll
      for (int q = 0; q < 8; ++q) tf[g][q] = t[q];
      gptr[g] = nullptr; gx0[g] = 0.0f; gy0[g] = 0.0f; ginv[g] = 0.0f; gnx[g] = 0; gny[g] = 0;
      if (kd_grid != nullptr) {
        const uint32_t* gj = kd_grid + (size_t)jj * kKdGridWords;
        gx0[g] = __uint_as_float(gj[0]); gy0[g] = __uint_as_float(gj[1]); ginv[g] = __uint_as_float(gj[2]); gnx[g] = (int)gj[3]; gny[g] = (int)gj[4];
        if (gnx[g] == 0) live[g] = false;               // no point that could match
        else if (gnx[g] > 0) {
          gptr[g] = gj;
          if (box_ok) {                                 // stage 0 (see k_match_kd)
            const float m00 = tf[g][0], m01 = tf[g][1], m10 = tf[g][2], m11 = tf[g][3], tx = tf[g][4], ty = tf[g][5];
            const float c0x = (m00 * bx0 + m01 * by0) + tx, c0y = (m10 * bx0 + m11 * by0) + ty;
            const float c1x = (m00 * bx1 + m01 * by0) + tx, c1y = (m10 * bx1 + m11 * by0) + ty;
            const float c2x = (m00 * bx0 + m01 * by1) + tx, c2y = (m10 * bx0 + m11 * by1) + ty;
            const float c3x = (m00 * bx1 + m01 * by1) + tx, c3y = (m10 * bx1 + m11 * by1) + ty;
            const float lo_x = (fminf(fminf(c0x, c1x), fminf(c2x, c3x)) - gx0[g]) * ginv[g], hi_x = (fmaxf(fmaxf(c0x, c1x), fmaxf(c2x, c3x)) - gx0[g]) * ginv[g];
            const float lo_y = (fminf(fminf(c0y, c1y), fminf(c2y, c3y)) - gy0[g]) * ginv[g], hi_y = (fmaxf(fmaxf(c0y, c1y), fmaxf(c2y, c3y)) - gy0[g]) * ginv[g];
            if (hi_x < -1.0f || lo_x > (float)gnx[g] + 1.0f || hi_y < -1.0f || lo_y > (float)gny[g] + 1.0f) live[g] = false;
          }
        }
      }
      n_st0 += live[g] ? 1u : 0u;
    }
    // ---- stage 1: this thread's queries against the G grids (bit p of mask[g]: the query k = tid + p * BLOCK) ----
    uint32_t mask[G];
#pragma unroll
    for (int g = 0; g < G; ++g) mask[g] = 0u;
    {
      int p = 0;
      for (int k = k0; k < mi; k += kstep, ++p) {
        if (!((k % mp.skip) == 0 && s_cnt[k] < mp.cap)) continue;
        const float4 a = map.obs[oi + k];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          bool ok = live[g];
          if (ok && gptr[g] != nullptr) {
            const float qx = (tf[g][0] * a.x + tf[g][1] * a.y) + tf[g][4], qy = (tf[g][2] * a.x + tf[g][3] * a.y) + tf[g][5];
            const float fx = floorf((qx - gx0[g]) * ginv[g]), fy = floorf((qy - gy0[g]) * ginv[g]);
            ok = fx >= 0.0f && fx < (float)gnx[g] && fy >= 0.0f && fy < (float)gny[g];
            if (ok) { const int bit = (int)fy * gnx[g] + (int)fx; ok = (gptr[g][8 + (bit >> 5)] >> (bit & 31)) & 1u; }
          }
          mask[g] |= ok ? (1u << p) : 0u;
        }
      }
    }
    unsigned wbits = 0u;
#pragma unroll
    for (int g = 0; g < G; ++g) wbits |= (__ballot(mask[g] != 0u) != 0ull) ? (1u << g) : 0u;
    if (lane == 0 && wbits != 0u) atomicOr(&s_any[par], wbits);
    __syncthreads();
    const unsigned anyb = s_any[par];
    const unsigned long long t_b = stats != nullptr ? wall_clock64() : 0ull;
    t_st1 += t_b - t_a;
    if (anyb == 0u) continue;                       // none of the G targets has a point near any query still open (nothing of [par] was touched)
    if (tid == 0) s_any[par ^ 1] = 0u;              // the other set: last read before this group's barrier, next written after the barrier below
    if (tid < G) { s_np[par ^ 1][tid] = 0; s_pc[par ^ 1][tid] = 0; }
    // ---- stage 2a: lists, result rows opened, wave g stages tree g ----
    {
      int p = 0;
      for (int k = k0; k < mi; k += kstep, ++p) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (!((anyb >> g) & 1u)) continue;
          s_res[(size_t)g * max_scan + k] = 0xFFFFu;
          if ((mask[g] >> p) & 1u) s_list[(size_t)g * max_scan + atomicAdd(&s_np[par][g], 1)] = (uint16_t)k;
        }
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (wave == g && ((anyb >> g) & 1u))
        for (int t = lane; t < tmj[g]; t += 64) { s_tgt[(size_t)g * max_scan + t] = kd_pts[toj[g] + t]; s_meta[(size_t)g * max_scan + t] = kd_meta[toj[g] + t]; }
    }
    __syncthreads();
    // ---- stage 2b: wave g walks list g in tree g (no counter is read or written here) ----
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (wave != g || !((anyb >> g) & 1u)) continue;
      const int np = s_np[par][g];
      const float4* tg = s_tgt + (size_t)g * max_scan;
      const uint32_t* tm = s_meta + (size_t)g * max_scan;
      ++n_st1; n_walk += (unsigned)np;
      for (int e = lane; e < np; e += 64) {
        const int k = s_list[(size_t)g * max_scan + e];
        const float4 a = map.obs[oi + k];
        const float qx = (tf[g][0] * a.x + tf[g][1] * a.y) + tf[g][4];
        const float qy = (tf[g][2] * a.x + tf[g][3] * a.y) + tf[g][5];
        KdWalk w;
        w.ovf = 0; w.t_inc = 0.0f; w.t_seg = 0u; w.t_oc = 0u;
        kd_begin(w, qx, qy, tmj[g]);
        while (w.mode != 2) kd_step(w, tg, tm, thr, thr2, st_inc + tid, st_seg + tid, st_oc + tid, BLOCK, depth);
        if (w.ret_c >= 0 && w.ret_d < thr) {
          const float cr = tf[g][6], sr = tf[g][7];
          const float nqx = cr * a.z - sr * a.w, nqy = sr * a.z + cr * a.w;
          const float4 gp = tg[w.ret_c];
          const float cosang = gp.z * nqx + gp.w * nqy;
          if (cosang > mp.min_cos) s_res[(size_t)g * max_scan + k] = (uint16_t)(tm[w.ret_c] & 0xffffu);
        }
      }
    }
    __syncthreads();
    // ---- stage 2c: commit in ascending target order, by the owner of each point: JointOptimization.cpp:596-599's counter ----
    int local[G];
#pragma unroll
    for (int g = 0; g < G; ++g) local[g] = 0;
    for (int k = k0; k < mi; k += kstep) {
      int c = s_cnt[k];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (!((anyb >> g) & 1u)) continue;
        uint16_t r = s_res[(size_t)g * max_scan + k];
        if (r != 0xFFFFu) {
          if (c < mp.cap) { ++c; ++local[g]; }
          else r = 0xFFFFu;                         // the sequential order would not have asked
        }
        table[((size_t)il * slots + slot0 + g) * max_scan + k] = r;
      }
      s_cnt[k] = c;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) if (local[g]) atomicAdd(&s_pc[par][g], local[g]);
    __syncthreads();
    if (tid < G && ((anyb >> tid) & 1u)) {
      if (parts > 1) { if (s_pc[par][tid]) atomicAdd(&counts[(size_t)il * slots + slot0 + tid], s_pc[par][tid]); }
      else counts[(size_t)il * slots + slot0 + tid] = s_pc[par][tid];
    }
    par ^= 1;
    if (stats != nullptr) t_st2 += wall_clock64() - t_b;
  }
  if (cost != nullptr && tid == 0) {              // what the WHOLE pose would take, roughly: a part's time does not fall in proportion (per-target latency stays)
    const unsigned t = (unsigned)(wall_clock64() - t_begin);
    atomicMax(&cost[il], parts == 1 ? t : (parts == 2 ? t + (t >> 1) + (t >> 2) : (parts == 3 ? 2 * t + (t >> 1) : 3 * t)));
  }
  if (stats != nullptr && lane == 0) {
    atomicAdd(stats + 1, n_st1); atomicAdd(stats + 2, n_walk);
    if (tid == 0) { atomicAdd(stats, n_st0); atomicAdd(stats + 3, t_st1); atomicAdd(stats + 4, t_st2); atomicMax(stats + 5, t_st1 + t_st2); atomicMax(stats + 6 + il, t_st1 + t_st2); }
  }
}

template <int G>
static size_t match_kd_spec_lds_bytes(int max_scan) {
  const size_t ms = (size_t)std::max(max_scan, 1);
  return (size_t)G * ms * (sizeof(float4) + sizeof(uint32_t)) + ms * sizeof(int) + (size_t)(kd_depth(max_scan) + 1) * 64 * G * 12 +
         (((size_t)2 * G * ms * sizeof(uint16_t) + 15) & ~(size_t)15);
}

size_t match_kd_lds_bytes(int max_scan) {
  return (size_t)std::max(max_scan, 1) * (sizeof(float4) + 2 * sizeof(int)) + (size_t)(kd_depth(max_scan) + 1) * kMatchBlock * 12 + (((size_t)std::max(max_scan, 1) * 2 + 15) & ~(size_t)15);
}

bool match_kd_takes_order(int max_scan) { return dev_switches().kd_spec >= 2 && match_kd_spec_lds_bytes<2>(max_scan) <= (size_t)80 * 1024; }

void launch_match_kd(hipStream_t st, const MapView& map, int max_scan, int slots, const int* cand_tgt, const float* cand_tf, MatchParams mp,
                     const float4* kd_pts, const uint32_t* kd_meta, const uint32_t* kd_grid, uint16_t* table, int* counts, int i0, int n_tile,
                     unsigned long long* stats, const int* order, int n_wg, unsigned* cost) {
  if (n_tile < 0) n_tile = map.n_poses - i0;
  const int chunk = (n_tile + 7) / 8;
  const unsigned grid = order != nullptr ? (unsigned)n_wg : (unsigned)(8 * chunk);
  // G targets in flight per workgroup while two workgroups still fit a CU's LDS (scans of up to ~470 points: four; ~900: two); longer scans: one
  const int want = dev_switches().kd_spec;
  const size_t budget = 80 * 1024;
  if (want >= 4 && match_kd_spec_lds_bytes<4>(max_scan) <= budget) {
    const size_t l4 = match_kd_spec_lds_bytes<4>(max_scan);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_match_kd_spec<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4);
    hipLaunchKernelGGL(k_match_kd_spec<4>, dim3(grid), dim3(256), l4, st, map, chunk, max_scan, slots, kd_depth(max_scan), cand_tgt, cand_tf,
                       mp, kd_pts, kd_meta, kd_grid, table, counts, i0, n_tile, stats, order, n_wg, cost);
    return;
  }
  if (want >= 2 && match_kd_spec_lds_bytes<2>(max_scan) <= budget) {
    const size_t l2 = match_kd_spec_lds_bytes<2>(max_scan);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_match_kd_spec<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
    hipLaunchKernelGGL(k_match_kd_spec<2>, dim3(grid), dim3(128), l2, st, map, chunk, max_scan, slots, kd_depth(max_scan), cand_tgt, cand_tf,
                       mp, kd_pts, kd_meta, kd_grid, table, counts, i0, n_tile, stats, order, n_wg, cost);
    return;
  }
  const size_t lds = match_kd_lds_bytes(max_scan);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_match_kd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_match_kd, dim3(8 * chunk), dim3(kMatchBlock), lds, st, map, chunk, max_scan, slots, kd_depth(max_scan), cand_tgt, cand_tf,
                     mp, kd_pts, kd_meta, kd_grid, table, counts, i0, n_tile, stats);
}

// Candidate transforms of a windowed search on the device (round 4): RelativePoseTransform (JointOptimization.cpp:296-305) in
// fp32 with exactly the operations, in exactly the order, of make_rel_tf (hitl_corr.cpp) -- multiplications, additions and one
// correctly rounded division, no contraction in this file -- from the poses, the per-pose (cos, sin) and the per-candidate
// (cos, sin) of the angle difference, which stay with the host's libm (the only part whose bits a GPU cannot be trusted to
// reproduce).  One thread per candidate (i, slot): j = slot-th pose of [i - window, i + window] without i; slots past the end
// of the list get zeros.  The test hook cand_tf_check holds every launch to the host routine, bit for bit.
__global__ void k_cand_tf(int n, int slots, int window, int i0, int cnt, const float* __restrict__ poses, const float* __restrict__ cs,
                          const float* __restrict__ trig, float* __restrict__ cand_tf, int* __restrict__ cand_tgt) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= (int64_t)cnt * slots) return;
  const int i = i0 + (int)(c / slots), slot = (int)(c % slots);
  const int j0 = max(0, i - window), j1 = min(n - 1, i + window);
  int j = j0 + slot;
  if (j >= i) ++j;
  float* t = cand_tf + (size_t)c * 8;
  if (j > j1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = 0.0f;
    cand_tgt[c] = -1;
    return;
  }
  cand_tgt[c] = j;
  const float* pi = poses + 3 * (size_t)i;
  const float* pj = poses + 3 * (size_t)j;
  const float ci = cs[2 * i], si = cs[2 * i + 1], cj = cs[2 * j], sj = cs[2 * j + 1];
  const float nsj = -sj, nsi = -si;
  const float det = cj * cj - sj * nsj;
  const float invdet = 1.0f / det;
  const float a00 = cj * invdet, a01 = (-nsj) * invdet, a10 = (-sj) * invdet, a11 = cj * invdet;
  t[0] = a00 * ci + a01 * si; t[1] = a00 * nsi + a01 * ci;
  t[2] = a10 * ci + a11 * si; t[3] = a10 * nsi + a11 * ci;
  const float ix = (-a00) * pj[0] + (-a01) * pj[1], iy = (-a10) * pj[0] + (-a11) * pj[1];
  t[4] = (a00 * pi[0] + a01 * pi[1]) + ix;
  t[5] = (a10 * pi[0] + a11 * pi[1]) + iy;
  t[6] = trig[2 * (size_t)c]; t[7] = trig[2 * (size_t)c + 1];
}

void launch_cand_tf(hipStream_t st, int n, int slots, int window, int i0, int cnt, const float* poses, const float* cs, const float* trig, float* cand_tf,
                    int* cand_tgt) {
  const int64_t tot = (int64_t)cnt * slots;
  if (tot <= 0) return;
  hipLaunchKernelGGL(k_cand_tf, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, n, slots, window, i0, cnt, poses, cs, trig, cand_tf, cand_tgt);
}

// target-major search over the whole map (a windowed search inside the band whose slots cannot reach the per-point cap)
bool match_kd_tgt_applies(int slots, int window, const MatchParams& mp) {
  return window >= 1 && 2 * window <= kKdMaxSrc && slots <= 2 * window && slots <= mp.cap && mp.skip >= 1;
}

constexpr int kKdRefillMin = 16;      // idle lanes per wave that trigger a refill from the query queue (sweeps of rounds 3-4: 4 .. 32 within 2 %)
template <int BLOCK>
static void launch_match_kd_tgt_t(hipStream_t st, const MapView& map, int max_scan, int slots, int window, const float* cand_tf,
                                  MatchParams mp, const float4* kd_pts, const uint32_t* kd_meta, uint16_t* table, int* counts, int* redo_count,
                                  int* redo_list, int redo_cap, bool full_depth) {
  const int chunk = (map.n_poses + 7) / 8;
  const int depth = kd_tgt_cap(max_scan, full_depth);
  const size_t lds = match_kd_tgt_lds_bytes(max_scan, BLOCK, full_depth);
  // one kd_step_fast behind every kd_step (0 and 2 were measured in round 4: 15 % slower / no gain)
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_match_kd_tgt<BLOCK, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_match_kd_tgt<BLOCK, 1>), dim3(8 * chunk), dim3(BLOCK), lds, st, map, chunk, max_scan, slots, window, depth, kKdRefillMin, cand_tf, mp,
                     kd_pts, kd_meta, table, counts, redo_count, redo_list, redo_cap);
  if (!full_depth)
    hipLaunchKernelGGL(k_match_kd_redo, dim3(256), dim3(64), 0, st, map, max_scan, slots, window, cand_tf, mp, kd_pts, kd_meta, table, counts, redo_count,
                       redo_list, redo_cap);
}

size_t match_kd_tgt_lds_bytes(int max_scan, int block, bool full_depth) {
  return (size_t)max_scan * (sizeof(float4) + sizeof(uint32_t)) + (size_t)(kd_tgt_cap(max_scan, full_depth) + 1) * block * 12;
}

void launch_match_kd_tgt(hipStream_t st, const MapView& map, int max_scan, int slots, int window, const float* cand_tf, MatchParams mp,
                         const float4* kd_pts, const uint32_t* kd_meta, uint16_t* table, int* counts, int* redo_count,
                         int* redo_list, int redo_cap, bool full_depth) {
  int block = 256;                                     // 64 / 128 threads: 13.5 / 9.4 ms against 8.6 at C4 (round 3)
  while (block > 64 && match_kd_tgt_lds_bytes(max_scan, block, full_depth) > (size_t)150 * 1024) block /= 2;     // long scans: fewer lanes share the tree
  if (block == 64) launch_match_kd_tgt_t<64>(st, map, max_scan, slots, window, cand_tf, mp, kd_pts, kd_meta, table, counts, redo_count, redo_list, redo_cap, full_depth);
  else if (block == 256) launch_match_kd_tgt_t<256>(st, map, max_scan, slots, window, cand_tf, mp, kd_pts, kd_meta, table, counts, redo_count, redo_list, redo_cap, full_depth);
  else launch_match_kd_tgt_t<128>(st, map, max_scan, slots, window, cand_tf, mp, kd_pts, kd_meta, table, counts, redo_count, redo_list, redo_cap, full_depth);
}

size_t match_lds_bytes(int max_scan) {
  return (size_t)((max_scan + 1) & ~1) * sizeof(float4) + (size_t)((max_scan + 3) & ~3) * sizeof(int) +
         (size_t)((max_scan + kMatchGroup - 1) / kMatchGroup) * sizeof(float4);
}

void launch_match(hipStream_t st, const MapView& map, int max_scan, int slots, const int* cand_tgt, const float* cand_tf,
                  MatchParams mp, uint16_t* table, int* counts, int i0, int n_tile) {
  if (n_tile < 0) n_tile = map.n_poses - i0;
  const int chunk = (n_tile + 7) / 8;
  const size_t lds = match_lds_bytes(max_scan);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_match), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_match, dim3(8 * chunk), dim3(kMatchBlock), lds, st, map, chunk, max_scan, slots, cand_tgt, cand_tf,
                     mp, table, counts, i0, n_tile);
}

// ordered stream compaction of one kept pair's table row into packed (src | tgt << 16) matches
__global__ __launch_bounds__(64) void k_compact(MapView map, int max_scan, int slots, const int* __restrict__ kept_cand,
                                                 const int64_t* __restrict__ match_off,
                                                 const uint16_t* __restrict__ table, uint32_t* __restrict__ matches, int i0) {
  const int p = blockIdx.x;
  const int cand = kept_cand[p];                    // index into the tile's candidate arrays; its source pose is i0 + cand / slots
  const int i = i0 + cand / slots;
  const int mi = (int)(map.obs_off[i + 1] - map.obs_off[i]);
  const uint16_t* row = table + (size_t)cand * max_scan;
  int64_t w = match_off[p];
  const int lane = threadIdx.x;
  // eight 64-entry pieces of the row are requested together (one memory round trip per 512 entries instead of eight: a 400-point
  // row is one round), then compacted in order
  constexpr int U = 8;
  for (int base0 = 0; base0 < mi; base0 += 64 * U) {
    uint16_t v[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int k = base0 + 64 * q + lane;
      v[q] = (k < mi) ? row[k] : (uint16_t)0xFFFFu;
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int k = base0 + 64 * q + lane;
      const bool hit = (v[q] != 0xFFFFu);
      const unsigned long long mask = __ballot(hit);
      if (hit) {
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));
        matches[w + rank] = (uint32_t)k | ((uint32_t)v[q] << 16);
      }
      w += __popcll(mask);
    }
  }
}

void launch_compact(hipStream_t st, const MapView& map, int max_scan, int slots, int n_kept, const int* kept_cand,
                    const int64_t* match_off, const uint16_t* table, uint32_t* matches, int i0) {
  if (n_kept == 0) return;
  hipLaunchKernelGGL(k_compact, dim3(n_kept), dim3(64), 0, st, map, max_scan, slots, kept_cand, match_off, table, matches, i0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Pair tables of a windowed search, built where the counts are (JointOptimization.cpp:632-641: keep a candidate with more
// than kMinInterPoseCorrespondence matches, pose order).  The candidate list of a windowed search is regular -- pose i's
// slots are j = max(0, i - w) .. min(n - 1, i + w) without i -- so a pose can find the candidates that point AT it without
// any list.  Three launches: per-pose counts, one-workgroup scan, per-pose fill.  The layout written is the one
// build_pair_structures (hitl_solver.cpp) derives on the host from arbitrary pair lists, entry for entry.
// ---------------------------------------------------------------------------------------------------------------------
struct WinCand {
  int n, slots, window, min_count, own_lo, own_hi;
  const int* __restrict__ counts;
  __device__ int j0(int i) const { return max(0, i - window); }
  __device__ int j1(int i) const { return min(n - 1, i + window); }
  __device__ int slot_of(int i, int j) const { return (j - j0(i)) - (j > i ? 1 : 0); }       // j != i, inside i's window
  __device__ int target_of(int i, int sl) const { const int j = j0(i) + sl; return j >= i ? j + 1 : j; }
  __device__ int n_slots(int i) const { return j1(i) - j0(i); }
  __device__ bool kept(int i, int sl, int j) const {        // halo-halo pairs belong to other ranks
    const bool relevant = (i >= own_lo && i < own_hi) || (j >= own_lo && j < own_hi);
    return relevant && counts[(size_t)i * slots + sl] > min_count;
  }
};

__global__ void k_pair_count(WinCand w, int* __restrict__ pose_pairs, int64_t* __restrict__ pose_matches, int* __restrict__ pose_inc,
                             int* __restrict__ pose_maxd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w.n) return;
  int np = 0, ninc = 0, maxd = 0;
  int64_t nm = 0;
  const int ns = w.n_slots(i);
  for (int sl = 0; sl < ns; ++sl) {
    const int j = w.target_of(i, sl);
    if (w.kept(i, sl, j)) { ++np; nm += w.counts[(size_t)i * w.slots + sl]; maxd = max(maxd, abs(i - j)); }
    if (w.kept(j, w.slot_of(j, i), i)) ++ninc;       // the candidate j -> i (i lies in j's window exactly when j lies in i's)
  }
  pose_pairs[i] = np; pose_matches[i] = nm; pose_inc[i] = np + ninc; pose_maxd[i] = maxd;
}

// Exclusive scans over the poses in three steps (a single workgroup streaming 50 000 x 20 bytes took 236 us): per-block
// totals, one small scan over the blocks (+ the five totals of launch_pair_count_scan), then every block scans its own
// poses and adds its offset.  Blocks of kPairBlock poses.
constexpr int kPairBlock = 256;
__device__ __forceinline__ int64_t wave_incl_scan(int64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int64_t o = __shfl_up(v, d, 64); if (lane >= d) v += o; }
  return v;
}

// in-block exclusive prefix of three per-thread values; returns the block totals through tot[3] (valid in every thread)
__device__ __forceinline__ void block_excl_scan3(int64_t a, int64_t b, int64_t c, int64_t* ea, int64_t* eb, int64_t* ec, int64_t tot[3]) {
  constexpr int NW = kPairBlock / 64;
  __shared__ int64_t s_w[3][NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t ia = wave_incl_scan(a, lane), ib = wave_incl_scan(b, lane), ic = wave_incl_scan(c, lane);
  if (lane == 63) { s_w[0][wave] = ia; s_w[1][wave] = ib; s_w[2][wave] = ic; }
  __syncthreads();
  int64_t pa = 0, pb = 0, pc = 0;
  tot[0] = tot[1] = tot[2] = 0;
  for (int w = 0; w < NW; ++w) {
    if (w < wave) { pa += s_w[0][w]; pb += s_w[1][w]; pc += s_w[2][w]; }
    tot[0] += s_w[0][w]; tot[1] += s_w[1][w]; tot[2] += s_w[2][w];
  }
  *ea = pa + ia - a; *eb = pb + ib - b; *ec = pc + ic - c;
}

// block b: totals of (pairs, matches, incidences), max |i - j|, owned (pairs, matches) -> blk[6 * b ..]
__global__ __launch_bounds__(kPairBlock) void k_pair_block_sums(int n, int own_lo, int own_hi, const int* __restrict__ pose_pairs,
                                                                 const int64_t* __restrict__ pose_matches, const int* __restrict__ pose_inc,
                                                                 const int* __restrict__ pose_maxd, int64_t* __restrict__ blk) {
  __shared__ int64_t s_red[3][kPairBlock / 64];
  const int i = blockIdx.x * kPairBlock + threadIdx.x;
  const bool in = i < n;
  const int64_t a = in ? pose_pairs[i] : 0, b = in ? pose_matches[i] : 0, c = in ? pose_inc[i] : 0;
  int64_t ea, eb, ec, tot[3];
  block_excl_scan3(a, b, c, &ea, &eb, &ec, tot);
  const bool own = in && i >= own_lo && i < own_hi;
  int64_t d = in ? pose_maxd[i] : 0, oa = own ? a : 0, ob = own ? b : 0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { d = max(d, __shfl_xor(d, m, 64)); oa += __shfl_xor(oa, m, 64); ob += __shfl_xor(ob, m, 64); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_red[0][wave] = d; s_red[1][wave] = oa; s_red[2][wave] = ob; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t dd = 0, soa = 0, sob = 0;
    for (int w = 0; w < kPairBlock / 64; ++w) { dd = max(dd, s_red[0][w]); soa += s_red[1][w]; sob += s_red[2][w]; }
    int64_t* o = blk + 6 * (size_t)blockIdx.x;
    o[0] = tot[0]; o[1] = tot[1]; o[2] = tot[2]; o[3] = dd; o[4] = soa; o[5] = sob;
  }
}

// one workgroup: exclusive scan of the block totals in place (slots 0..2), totals to out[5]
__global__ __launch_bounds__(kPairBlock) void k_pair_block_scan(int nblk, int64_t* __restrict__ blk, int64_t* __restrict__ out) {
  int64_t ca = 0, cb = 0, cc = 0, dd = 0, oa = 0, ob = 0;
  for (int base = 0; base < nblk; base += kPairBlock) {
    const int k = base + threadIdx.x;
    const bool in = k < nblk;
    int64_t* e = blk + 6 * (size_t)(in ? k : 0);
    const int64_t a = in ? e[0] : 0, b = in ? e[1] : 0, c = in ? e[2] : 0;
    if (in) { dd = max(dd, e[3]); oa += e[4]; ob += e[5]; }
    int64_t ea, eb, ec, tot[3];
    __syncthreads();
    block_excl_scan3(a, b, c, &ea, &eb, &ec, tot);
    if (in) { e[0] = ca + ea; e[1] = cb + eb; e[2] = cc + ec; }
    ca += tot[0]; cb += tot[1]; cc += tot[2];
  }
  __shared__ int64_t s_red[3][kPairBlock / 64];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { dd = max(dd, __shfl_xor(dd, m, 64)); oa += __shfl_xor(oa, m, 64); ob += __shfl_xor(ob, m, 64); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { s_red[0][wave] = dd; s_red[1][wave] = oa; s_red[2][wave] = ob; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t d2 = 0, soa = 0, sob = 0;
    for (int w = 0; w < kPairBlock / 64; ++w) { d2 = max(d2, s_red[0][w]); soa += s_red[1][w]; sob += s_red[2][w]; }
    out[0] = ca; out[1] = cb; out[2] = d2; out[3] = soa; out[4] = sob;
  }
}

// every block: exclusive offsets of its poses = block offset + in-block prefix, in place; the last entries (index n) = totals
__global__ __launch_bounds__(kPairBlock) void k_pair_offsets(int n, const int64_t* __restrict__ blk, int* __restrict__ pose_pairs, int64_t* __restrict__ pose_matches,
                                                              int* __restrict__ pose_inc) {
  const int i = blockIdx.x * kPairBlock + threadIdx.x;
  const bool in = i < n;
  const int64_t a = in ? pose_pairs[i] : 0, b = in ? pose_matches[i] : 0, c = in ? pose_inc[i] : 0;
  int64_t ea, eb, ec, tot[3];
  block_excl_scan3(a, b, c, &ea, &eb, &ec, tot);
  const int64_t* o = blk + 6 * (size_t)blockIdx.x;
  if (in) { pose_pairs[i] = (int)(o[0] + ea); pose_matches[i] = o[1] + eb; pose_inc[i] = (int)(o[2] + ec); }
  if (i == n - 1) { pose_pairs[n] = (int)(o[0] + ea + a); pose_matches[n] = o[1] + eb + b; pose_inc[n] = (int)(o[2] + ec + c); }
}

__global__ void k_pair_fill(WinCand w, int Wb, const int* __restrict__ pose_pair_off, const int64_t* __restrict__ pose_match_off,
                            const int* __restrict__ inc_off, int* __restrict__ kept_cand, int* __restrict__ pair_src,
                            int* __restrict__ pair_tgt, int64_t* __restrict__ match_off, int* __restrict__ inc_ent,
                            int* __restrict__ band_pair) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w.n) return;
  int f = inc_off[i];
  // index of the kept pair j -> i: j's first pair + the kept candidates of j in the slots before i's
  auto pair_into = [&](int j) -> int {
    const int sl = w.slot_of(j, i);
    if (!w.kept(j, sl, i)) return -1;
    int q = pose_pair_off[j];
    for (int k = 0; k < sl; ++k) q += w.kept(j, k, w.target_of(j, k)) ? 1 : 0;
    return q;
  };
  // incidence list in ascending pair index: pairs into i from earlier poses, i's own pairs, pairs into i from later poses
  for (int j = w.j0(i); j < i; ++j) { const int q = pair_into(j); if (q >= 0) inc_ent[f++] = 2 * q + 1; }
  int q = pose_pair_off[i];
  int64_t mo = pose_match_off[i];
  const int ns = w.n_slots(i);
  for (int sl = 0; sl < ns; ++sl) {
    const int j = w.target_of(i, sl);
    if (!w.kept(i, sl, j)) continue;
    kept_cand[q] = i * w.slots + sl; pair_src[q] = i; pair_tgt[q] = j; match_off[q] = mo;
    mo += w.counts[(size_t)i * w.slots + sl];
    const int lo = min(i, j), d = abs(i - j);
    band_pair[((size_t)lo * Wb + (d - 1)) * 2 + (i < j ? 0 : 1)] = q;       // d <= window <= Wb's cap: never off the band
    inc_ent[f++] = 2 * q;
    ++q;
  }
  for (int j = i + 1; j <= w.j1(i); ++j) { const int qq = pair_into(j); if (qq >= 0) inc_ent[f++] = 2 * qq + 1; }
  if (i == w.n - 1) match_off[q] = mo;      // the last pose closes the offsets: q == n_pairs, mo == n_matches
}

static WinCand win_cand(int n, int slots, int window, int min_count, int own_lo, int own_hi, const int* counts) {
  WinCand w;
  w.n = n; w.slots = slots; w.window = window; w.min_count = min_count; w.own_lo = own_lo; w.own_hi = own_hi; w.counts = counts;
  return w;
}

void launch_pair_count_scan(hipStream_t st, int n, int slots, int window, int min_count, int own_lo, int own_hi, const int* counts,
                            int* pose_pair_off, int64_t* pose_match_off, int* inc_off, int* pose_maxd, int64_t* blk, int64_t* totals) {
  const WinCand w = win_cand(n, slots, window, min_count, own_lo, own_hi, counts);
  hipLaunchKernelGGL(k_pair_count, dim3((n + 255) / 256), dim3(256), 0, st, w, pose_pair_off, pose_match_off, inc_off, pose_maxd);
  const int nblk = (n + kPairBlock - 1) / kPairBlock;
  hipLaunchKernelGGL(k_pair_block_sums, dim3(nblk), dim3(kPairBlock), 0, st, n, own_lo, own_hi, pose_pair_off, pose_match_off, inc_off, pose_maxd, blk);
  hipLaunchKernelGGL(k_pair_block_scan, dim3(1), dim3(kPairBlock), 0, st, nblk, blk, totals);
  hipLaunchKernelGGL(k_pair_offsets, dim3(nblk), dim3(kPairBlock), 0, st, n, blk, pose_pair_off, pose_match_off, inc_off);
}

int pair_scan_blocks(int n) { return (n + kPairBlock - 1) / kPairBlock; }

void launch_pair_fill(hipStream_t st, int n, int slots, int window, int min_count, int own_lo, int own_hi, const int* counts, int Wb,
                      const int* pose_pair_off, const int64_t* pose_match_off, const int* inc_off, int* kept_cand, int* pair_src,
                      int* pair_tgt, int64_t* match_off, int* inc_ent, int* band_pair) {
  const WinCand w = win_cand(n, slots, window, min_count, own_lo, own_hi, counts);
  hipLaunchKernelGGL(k_pair_fill, dim3((n + 255) / 256), dim3(256), 0, st, w, Wb, pose_pair_off, pose_match_off, inc_off, kept_cand,
                     pair_src, pair_tgt, match_off, inc_ent, band_pair);
}


}  // namespace hitl
