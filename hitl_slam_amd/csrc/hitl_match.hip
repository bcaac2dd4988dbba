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

namespace hitl {

constexpr int kMatchBlock = 256;
constexpr int kMatchGroup = 16;    // target points per bounding box

__global__ __launch_bounds__(kMatchBlock) void k_match(MapView map, int chunk, int max_scan, int slots,
                                                        const int* __restrict__ cand_tgt,
                                                        const float* __restrict__ cand_tf, MatchParams mp,
                                                        uint16_t* __restrict__ table, int* __restrict__ counts) {
  extern __shared__ float4 s_tgt[];                 // [max_scan]
  int* s_cnt = (int*)(s_tgt + max_scan);            // [max_scan] accepted matches per source point
  float4* s_box = (float4*)(s_cnt + ((max_scan + 3) & ~3));   // [ceil(max_scan / kMatchGroup)] (min x, min y, max x, max y), 16-byte aligned
  __shared__ int s_pair_count;
  const int b = blockIdx.x;
  const int idx = b >> 3;
  const int i = (b & 7) * chunk + idx;              // consecutive poses share an XCD (L2 reuse of neighbours)
  if (idx >= chunk || i >= map.n_poses) return;
  const int64_t oi = map.obs_off[i];
  const int mi = (int)(map.obs_off[i + 1] - oi);
  for (int k = threadIdx.x; k < mi; k += kMatchBlock) s_cnt[k] = 0;
  const float thr = mp.thr, thr2 = mp.thr * mp.thr;
  for (int slot = 0; slot < slots; ++slot) {
    const int j = cand_tgt[(size_t)i * slots + slot];
    if (j < 0) continue;                            // block-uniform
    const int64_t oj = map.obs_off[j];
    const int mj = (int)(map.obs_off[j + 1] - oj);
    __syncthreads();
    if (threadIdx.x == 0) s_pair_count = 0;
    for (int t = threadIdx.x; t < mj; t += kMatchBlock) s_tgt[t] = map.obs[oj + t];
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
        const float4 v = s_tgt[t];
        bx.x = fminf(bx.x, v.x); bx.y = fminf(bx.y, v.y); bx.z = fmaxf(bx.z, v.x); bx.w = fmaxf(bx.w, v.y);
      }
      s_box[g] = bx;
    }
    __syncthreads();
    const float* tf = cand_tf + ((size_t)i * slots + slot) * 8;
    const float m00 = tf[0], m01 = tf[1], m10 = tf[2], m11 = tf[3], tx = tf[4], ty = tf[5], cr = tf[6], sr = tf[7];
    uint16_t* row = table + ((size_t)i * slots + slot) * max_scan;
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
        for (int gi = 0; gi < ngrp; ++gi) {
          const float4 bx = s_box[gi];
          // distance from q to the box, with the same subtractions the point test uses (q - coordinate)
          const float ex = fmaxf(fmaxf(qx - bx.z, -(qx - bx.x)), 0.0f);
          const float ey = fmaxf(fmaxf(qy - bx.w, -(qy - bx.y)), 0.0f);
          if (!(ex * ex + ey * ey < thr2)) continue;
          const int t1 = min(mj, (gi + 1) * kMatchGroup);
          for (int t = gi * kMatchGroup; t < t1; ++t) {
            const float4 g = s_tgt[t];
            const float dx = qx - g.x, dy = qy - g.y;
            const float dsq = dx * dx + dy * dy;
            if (dsq < thr2) {
              const float sc = fabsf(g.z * dx + g.w * dy);
              if (sc < best) { best = sc; bi = t; }
            }
          }
        }
        if (bi >= 0 && best < thr) {
          const float anx = a.z, any = a.w;
          const float nqx = cr * anx - sr * any, nqy = sr * anx + cr * any;
          const float4 g = s_tgt[bi];
          const float cosang = g.z * nqx + g.w * nqy;
          if (cosang > mp.min_cos) { res = (uint16_t)bi; s_cnt[k] += 1; ++local; }
        }
      }
      row[k] = res;
    }
    if (local) atomicAdd(&s_pair_count, local);
    __syncthreads();
    if (threadIdx.x == 0) counts[(size_t)i * slots + slot] = s_pair_count;
  }
}

size_t match_lds_bytes(int max_scan) {
  return (size_t)max_scan * sizeof(float4) + (size_t)((max_scan + 3) & ~3) * sizeof(int) +
         (size_t)((max_scan + kMatchGroup - 1) / kMatchGroup) * sizeof(float4);
}

void launch_match(hipStream_t st, const MapView& map, int max_scan, int slots, const int* cand_tgt, const float* cand_tf,
                  MatchParams mp, uint16_t* table, int* counts) {
  const int chunk = (map.n_poses + 7) / 8;
  const size_t lds = match_lds_bytes(max_scan);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_match), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_match, dim3(8 * chunk), dim3(kMatchBlock), lds, st, map, chunk, max_scan, slots, cand_tgt, cand_tf,
                     mp, table, counts);
}

// ordered stream compaction of one kept pair's table row into packed (src | tgt << 16) matches
__global__ __launch_bounds__(64) void k_compact(MapView map, int max_scan, int slots, const int* __restrict__ kept_cand,
                                                 const int64_t* __restrict__ match_off,
                                                 const uint16_t* __restrict__ table, uint32_t* __restrict__ matches) {
  const int p = blockIdx.x;
  const int cand = kept_cand[p];
  const int i = cand / slots;
  const int mi = (int)(map.obs_off[i + 1] - map.obs_off[i]);
  const uint16_t* row = table + (size_t)cand * max_scan;
  int64_t w = match_off[p];
  const int lane = threadIdx.x;
  for (int base = 0; base < mi; base += 64) {
    const int k = base + lane;
    const uint16_t v = (k < mi) ? row[k] : (uint16_t)0xFFFFu;
    const bool hit = (v != 0xFFFFu);
    const unsigned long long mask = __ballot(hit);
    if (hit) {
      const int rank = __popcll(mask & ((1ull << lane) - 1ull));
      matches[w + rank] = (uint32_t)k | ((uint32_t)v << 16);
    }
    w += __popcll(mask);
  }
}

void launch_compact(hipStream_t st, const MapView& map, int max_scan, int slots, int n_kept, const int* kept_cand,
                    const int64_t* match_off, const uint16_t* table, uint32_t* matches) {
  if (n_kept == 0) return;
  hipLaunchKernelGGL(k_compact, dim3(n_kept), dim3(64), 0, st, map, max_scan, slots, kept_cand, match_off, table, matches);
}

}  // namespace hitl
