// hitl_corr.cpp -- host driver of the correspondence search (FindSTFCorrespondences, JointOptimization.cpp:561-642):
// candidate pose pairs and their fp32 relative transforms (host libm, bit-identical to the CPU restatement), k_match,
// the pair filter and the ordered compaction into the packed match list.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <atomic>
#include <thread>

#include "hitl_solver_impl.h"

using namespace hitl;

namespace hitl {

// ---------------------------------------------------------------------------------------
// correspondence search driver
// ---------------------------------------------------------------------------------------
// sincosf(x) == (sinf(x), cosf(x)) and sinf(-x) == -sinf(x), cosf(-x) == cosf(x), bit for bit: true for every finite float on
// glibc (checked exhaustively over all 2^32 patterns on the image this was developed on); verified again on a sample at
// first use, and the search falls back to one cosf + one sinf per candidate if the C library at hand disagrees.
static bool libm_sincos_exact() {
  static const bool ok = [] {
    uint32_t seed = 12345u;
    for (int k = 0; k < 200000; ++k) {
      seed = seed * 1664525u + 1013904223u;
      float x;
      const uint32_t bits = (k & 1) ? seed : ((seed & 0x007fffffu) | ((118u + (seed >> 28)) << 23) | (seed & 0x80000000u));   // any pattern / |x| in [2^-9, 2^6]
      std::memcpy(&x, &bits, 4);
      if (!(x == x) || std::fabs(x) > 3.0e38f) continue;
      float sn, cn;
      sincosf(x, &sn, &cn);
      const float s1 = sinf(x), c1 = cosf(x), s2 = sinf(-x), c2 = cosf(-x), ns = -s1;
      if (std::memcmp(&sn, &s1, 4) || std::memcmp(&cn, &c1, 4) || std::memcmp(&s2, &ns, 4) || std::memcmp(&c2, &c1, 4)) return false;
    }
    return true;
  }();
  return ok;
}

// with_trig == false: t[6], t[7] are written by the mirrored candidate (j -> i), see find_correspondences
static void make_rel_tf(const float* pi, const float* pj, const float* csi, const float* csj, float* t, bool with_trig = true) {
  // RelativePoseTransform (JointOptimization.cpp:296-305) and the normal rotation at :603-605 in fp32, in the operation
  // order of the Eigen expression `target_tf.inverse(Eigen::Affine) * source_tf`: tf = Translation2Df * Rotation2Df (linear
  // [[c, -s], [s, c]]); inverse(Affine) = 2x2 cofactor inverse with invdet = 1 / (m00 m11 - m10 m01) and translation
  // -linear^-1 t; Transform * Transform: linear = L1 L2, translation = L1 t2 + t1; 2-term dot products a*b + c*d (this file
  // is compiled with -ffp-contract=off).  Rounds 1-2 composed R_j^T (t_i - t_j) instead: the same map, different last bits.
  // cs = (cosf, sinf) of each pose angle, computed once per pose by the caller.
  const float ci = csi[0], si = csi[1], cj = csj[0], sj = csj[1];
  const float nsj = -sj, nsi = -si;
  const float det = cj * cj - sj * nsj;
  const float invdet = 1.0f / det;
  const float a00 = cj * invdet, a01 = (-nsj) * invdet, a10 = (-sj) * invdet, a11 = cj * invdet;
  t[0] = a00 * ci + a01 * si; t[1] = a00 * nsi + a01 * ci;
  t[2] = a10 * ci + a11 * si; t[3] = a10 * nsi + a11 * ci;
  const float ix = (-a00) * pj[0] + (-a01) * pj[1], iy = (-a10) * pj[0] + (-a11) * pj[1];
  t[4] = (a00 * pi[0] + a01 * pi[1]) + ix;
  t[5] = (a10 * pi[0] + a11 * pi[1]) + iy;
  if (!with_trig) return;
  const float dth = (float)((double)pj[2] - (double)pi[2]);
  t[6] = cosf(dth); t[7] = sinf(dth);
}

constexpr int kKdRedoCap = 1 << 18;       // (candidate, point) entries of the redo list of k_match_kd_tgt's fast attempt

int ensure_kd_trees(hitl_solver* s) {      // BuildKDTrees: once per map (JointOptimization.cpp:1307, kdtrees_.size() == 0)
  if (s->kd_built) return HITL_OK;
  if (kd_build_lds_bytes(s->max_scan) > (size_t)150 * 1024)
    return fail(HITL_ERR_UNSUPPORTED, "kd_tree_search: a scan of " + std::to_string(s->max_scan) + " points exceeds the LDS budget of the tree build (limit ~5900 points)");
  int rc = s->kd_pts.resize((size_t)std::max<int64_t>(s->n_obs, 1));
  if (rc != HITL_OK) return rc;
  rc = s->kd_meta.resize((size_t)std::max<int64_t>(s->n_obs, 1));
  if (rc != HITL_OK) return rc;
  launch_kd_build(s->stream, s->map_view(), s->max_scan, s->kd_pts.p, s->kd_meta.p);
  if (hipGetLastError() != hipSuccess) return fail(HITL_ERR_HIP, "the tree-build kernel could not be launched");
  s->kd_built = true;
  return HITL_OK;
}

int find_correspondences(hitl_solver* s, const float* poses, int64_t window) {
  RoctxRange r_all("hitl:find_correspondences");
  HIP_TRY(hipSetDevice(s->device));
  s->min_band_poses = 1;      // a new correspondence set: the band a submap rank once widened to belongs to the old one (prepare re-derives it)
  const bool timing = dev_switches().timing;
  auto t_prev = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "find_correspondences: %-22s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  const int n = s->n;
  if (window <= 0 || window > n) window = n;
  const int slots = (int)std::min<int64_t>(2 * window, std::max(n - 1, 1));
  const bool kd = s->opt.kd_tree_search != 0;
  MatchParams mp;
  mp.thr = s->opt.point_match_threshold;
  mp.min_cos = cosf(s->opt.max_stf_angle_error);   // JointOptimization.cpp:564
  mp.cap = s->opt.max_correspondences_per_point;
  mp.skip = std::max(s->opt.num_skip_readings, 1);
  const int band_cap = std::min(std::max((int)s->opt.max_band_poses, 1), kMaxBandPoses);
  const bool device_tables = window <= band_cap;
  const bool kd_tgt = kd && device_tables && match_kd_tgt_applies(slots, (int)window, mp);
  if (kd) {
    const size_t need = kd_tgt ? match_kd_tgt_lds_bytes(s->max_scan, 64, true) : match_kd_lds_bytes(s->max_scan);
    if (need > (size_t)150 * 1024 || kd_build_lds_bytes(s->max_scan) > (size_t)150 * 1024)
      return fail(HITL_ERR_UNSUPPORTED, "find_correspondences (kd_tree_search): a scan of " + std::to_string(s->max_scan) +
                  " points exceeds the LDS budget of the tree kernels (limit ~5900 points: the tree build; the walk kernels would take ~7300 / ~5000)");
  }
  if (match_lds_bytes(s->max_scan) > (size_t)160 * 1024)
    return fail(HITL_ERR_UNSUPPORTED, "find_correspondences: a scan of " + std::to_string(s->max_scan) +
                " points does not fit the 160 KB of LDS the matcher stages a target scan in (limit ~7900 points)");
  // The candidate table (u16 per candidate and source point) of a window inside the band is small (<= 8 slots per pose);
  // wider windows -- the reference's all-pairs search above all -- are cut into tiles of source poses so that the table
  // stays below a budget whatever the map (all slots of a source pose share a tile: the per-point cap counts across them).
  const size_t tile_budget = dev_switches().cand_tile_bytes;     // tests force several tiles
  const size_t per_pose = (size_t)slots * s->max_scan * sizeof(uint16_t);
  const int tile = device_tables ? n : (int)std::min<size_t>((size_t)n, std::max<size_t>(1, tile_budget / std::max<size_t>(per_pose, 1)));
  // scratch kept in the solver between calls (no 10 MB allocate-and-clear per search)
#define TRYP(expr) do { const int rc_ = (expr); if (rc_ != HITL_OK) return rc_; } while (0)
  std::vector<float>& cs = s->h_cand_cs;
  if (cs.size() != (size_t)2 * n) cs.resize((size_t)2 * n);
  // the j of candidate (i, slot): the slot-th pose of [i - window, i + window] without i (k_cand_tf forms the same list on the device)
  auto cand_target = [&](int64_t i, int slot) -> int {
    int64_t j = std::max<int64_t>(0, i - window) + slot;
    if (j >= i) ++j;
    return j <= std::min<int64_t>(n - 1, i + window) ? (int)j : -1;
  };
  // candidate transforms: fp32 host libm (bit-identical to the CPU restatement), spread over a few threads
  // a handful of threads: spawning 32 cost more than the trigonometry (hardware_concurrency() also counts CPUs a container's
  // quota does not grant)
  const int nt = (int)std::max(1u, std::min(12u, std::thread::hardware_concurrency()));
  // ONE team of threads for consecutive phases over [lo, hi) (spawning a team costs ~0.15 ms: more than either phase of a
  // 50 000-pose map): every member takes its slice of phase 0, waits for the others, takes its slice of phase 1, ...  A thread
  // that cannot be started only makes the team smaller (the members learn its size once spawning is over).
  auto run_phases = [&](int64_t lo, int64_t hi, int n_phases, auto&& body /* (phase, a, b) */) {
    std::atomic<int> team{0};
    std::vector<std::atomic<int>> arrived((size_t)std::max(n_phases, 1));
    for (auto& a : arrived) a.store(0, std::memory_order_relaxed);
    const int64_t len = hi - lo;
    auto member = [&](int t) {
      int sz;
      while ((sz = team.load(std::memory_order_acquire)) == 0) std::this_thread::yield();
      for (int ph = 0; ph < n_phases; ++ph) {
        body(ph, lo + len * t / sz, lo + len * (t + 1) / sz);
        arrived[(size_t)ph].fetch_add(1, std::memory_order_acq_rel);
        if (ph + 1 < n_phases) while (arrived[(size_t)ph].load(std::memory_order_acquire) < sz) std::this_thread::yield();
      }
    };
    std::vector<std::thread> th;
    int spawned = 0;
    try {
      th.reserve((size_t)nt);
      for (int t = 1; t < nt; ++t) { th.emplace_back(member, t); ++spawned; }
    } catch (...) {}        // (std::system_error: no more threads) -- the ones that exist do the work
    team.store(spawned + 1, std::memory_order_release);
    member(0);
    for (auto& x : th) x.join();
  };
  auto run = [&](int64_t lo, int64_t hi, auto&& body) { run_phases(lo, hi, 1, [&](int, int64_t a, int64_t b) { body(a, b); }); };
  const bool libm_ok = libm_sincos_exact();
  auto pose_trig = [&](int64_t a, int64_t b) {
    for (int64_t i = a; i < b; ++i) {
      if (libm_ok) sincosf(poses[3 * i + 2], &cs[2 * i + 1], &cs[2 * i]);
      else { cs[2 * i] = cosf(poses[3 * i + 2]); cs[2 * i + 1] = sinf(poses[3 * i + 2]); }
    }
  };
  if (!device_tables) { run(0, n, pose_trig); mark("pose trigonometry"); }      // (one tile: the trigonometry shares a team with the candidates below)
  // candidates of the source poses [i0, i1): the transforms themselves are formed on the device (k_cand_tf: the host routine's
  // operations in the same order); the host keeps what needs the C library: the (cos, sin) of every pose and of every candidate's
  // angle difference -- 2 floats per candidate (indexed from i0) to compute, store and upload, from page-locked memory (round 6: the
  // all-pairs search of a 1 000-pose map uploaded 36 MB of ready-made transforms and target lists from pageable memory per search --
  // more time than its kernels).
  TRYP(s->hp_cand_trig.resize((size_t)tile * slots * 2));
  float* cand_trig = s->hp_cand_trig.p;
  auto fill_candidates = [&](int64_t i0, int64_t i1, bool with_pose_trig = false) {
    const bool paired = libm_ok && i0 == 0 && i1 == n;        // the mirror of a candidate lies in the same tile only when there is one tile
    run_phases(i0, i1, with_pose_trig ? 2 : 1, [&](int ph, int64_t a, int64_t b) {
      if (with_pose_trig && ph == 0) { pose_trig(a, b); return; }
      for (int64_t i = a; i < b; ++i) {
        int slot = 0;
        const int64_t j0 = std::max<int64_t>(0, i - window), j1 = std::min<int64_t>(n - 1, i + window);
        for (int64_t j = j0; j <= j1; ++j) {
          if (j == i) continue;
          float* tr = &cand_trig[((size_t)(i - i0) * slots + slot) * 2];
          if (!paired) {
            const float dth = (float)((double)poses[3 * j + 2] - (double)poses[3 * i + 2]);
            tr[0] = cosf(dth); tr[1] = sinf(dth);
          } else if (j > i) {
            // one sincosf serves the candidate (i -> j) and its mirror (j -> i): the angle of the mirror is the exact
            // negative (fp64 difference, symmetric rounding to fp32), so its cosine is the same float and its sine the negated one
            const float dth = (float)((double)poses[3 * j + 2] - (double)poses[3 * i + 2]);
            float sn, cn;
            sincosf(dth, &sn, &cn);
            tr[0] = cn; tr[1] = sn;
            const int64_t jj0 = std::max<int64_t>(0, j - window);
            float* tm = &cand_trig[((size_t)j * slots + (size_t)(i - jj0)) * 2];   // slot of i in j's list (i < j)
            tm[0] = cn; tm[1] = -sn;
          }
          ++slot;
        }
      }
    });
  };
  if (device_tables) fill_candidates(0, n, /*with_pose_trig=*/true);
  mark("candidate transforms");
  hipStream_t st = s->stream;
  DevArray<int>&d_tgt = s->d_cand_tgt, &d_counts = s->d_cand_counts, &d_kept = s->d_cand_kept;   // device scratch, kept too
  DevArray<float>& d_tf = s->d_cand_tf;
  DevArray<uint16_t>& d_table = s->d_cand_table;
  int rc = HITL_OK;
#define TRYC(expr) do { rc = (expr); if (rc != HITL_OK) return rc; } while (0)
  const MapView mv = s->map_view();
  if (kd && !s->kd_built) {
    TRYC(ensure_kd_trees(s));
    if (timing) (void)hipStreamSynchronize(s->stream);     // the phase line is the build itself, not its enqueue
    mark("k_kd_build");
  }
  // one tile of source poses [i0, i0 + cnt): upload its candidates, match
  auto match_tile = [&](int i0, int cnt) -> int {
    if (i0 == 0) {
      TRYC(s->d_cand_poses.upload(poses, (size_t)3 * n, st));
      TRYC(s->d_cand_cs.upload(cs.data(), (size_t)2 * n, st));
    }
    TRYC(s->d_cand_trig.upload(cand_trig, (size_t)cnt * slots * 2, st));
    TRYC(d_tf.resize((size_t)cnt * slots * 8));
    TRYC(d_tgt.resize((size_t)cnt * slots));
    launch_cand_tf(st, n, slots, (int)window, i0, cnt, s->d_cand_poses.p, s->d_cand_cs.p, s->d_cand_trig.p, d_tf.p, d_tgt.p);
    if (dev_switches().cand_tf_check) {
      // development / tests: every candidate through the host routine as well, bit for bit
      std::vector<float> dev((size_t)cnt * slots * 8), host(8);
      std::vector<int> dev_tgt((size_t)cnt * slots);
      if (hipMemcpyAsync(dev.data(), d_tf.p, dev.size() * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipMemcpyAsync(dev_tgt.data(), d_tgt.p, dev_tgt.size() * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(HITL_ERR_HIP, "find_correspondences: reading the candidate transforms back failed");
      for (int64_t i = i0; i < i0 + cnt; ++i)
        for (int slot = 0; slot < slots; ++slot) {
          const int j = cand_target(i, slot);
          const size_t c = (size_t)(i - i0) * slots + slot;
          if (dev_tgt[c] != j)
            return fail(HITL_ERR_SOLVER_FAILURE, "cand_tf_check: the device's target of candidate (" + std::to_string(i) + ", slot " + std::to_string(slot) + ") differs from the host's");
          if (j < 0) continue;
          make_rel_tf(poses + 3 * i, poses + 3 * j, &cs[2 * i], &cs[2 * j], host.data(), true);
          if (std::memcmp(host.data(), &dev[c * 8], 8 * sizeof(float)) != 0)
            return fail(HITL_ERR_SOLVER_FAILURE, "cand_tf_check: the device's transform of candidate (" + std::to_string(i) + ", " + std::to_string(j) +
                                                 ") differs from the host's");
        }
    }
    auto t_al = std::chrono::steady_clock::now();
    if (timing) { (void)hipStreamSynchronize(st); t_al = std::chrono::steady_clock::now(); }
    TRYC(d_counts.resize((size_t)cnt * slots));
    TRYC(d_counts.zero(st));
    TRYC(d_table.resize((size_t)cnt * slots * s->max_scan));
    if (timing) {
      (void)hipStreamSynchronize(st);
      std::fprintf(stderr, "find_correspondences:     tile: counts + candidate table (%.1f MB; allocated by a solver's first search of this size: ~20 ms of hipMalloc at 720 MB) ready in %.3f ms\n", 1e-6 * 2.0 * (double)cnt * slots * s->max_scan,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_al).count());
    }
    if (kd_tgt) {
      // the per-point cap cannot bind (slots <= cap): every (source, target) pair is independent, one workgroup per target tree
      if (mp.skip > 1 && hipMemsetAsync(d_table.p, 0xFF, sizeof(uint16_t) * (size_t)cnt * slots * s->max_scan, st) != hipSuccess)
        return fail(HITL_ERR_HIP, "find_correspondences: clearing the candidate table failed");
      TRYC(s->d_kd_ovf.resize(1 + 2 * (size_t)kKdRedoCap));
      if (hipMemsetAsync(s->d_kd_ovf.p, 0, sizeof(int), st) != hipSuccess) return fail(HITL_ERR_HIP, "find_correspondences: clearing the redo counter failed");
      launch_match_kd_tgt(st, mv, s->max_scan, slots, (int)window, d_tf.p, mp, s->kd_pts.p, s->kd_meta.p, d_table.p, d_counts.p,
                          s->d_kd_ovf.p, s->d_kd_ovf.p + 1, dev_switches().kd_redo_cap, s->kd_full_depth);
    } else if (kd) {
      const uint32_t* grid = nullptr;
      if (dev_switches().kd_grid) {                 // the occupancy grids of the scans: once per map, on the first search that takes this kernel
        if (!s->kd_grid_built || s->kd_grid_thr != mp.thr) {       // (the cell size follows the threshold: a changed option rebuilds)
          TRYC(s->kd_grid.resize((size_t)std::max(n, 1) * kd_grid_words()));
          launch_kd_grid(st, mv, mp.thr, s->kd_pts.p, s->kd_grid.p);
          s->kd_grid_built = true; s->kd_grid_thr = mp.thr;
        }
        grid = s->kd_grid.p;
      }
      unsigned long long* stats = nullptr;
      if (timing) {                                 // HITL_TIMING: how many pairs each stage of the kernel let through
        TRYC(s->d_kd_stats.resize(6 + (size_t)cnt)); TRYC(s->d_kd_stats.zero(st));
        stats = s->d_kd_stats.p;
      }
      // one tile = the whole map: the workgroups of the poses that took longest in the previous search of this map start first, and a
      // pose that took much longer than the average is served by several workgroups, each with every `parts`-th point (exact: see the
      // kernel).  Test hooks: kd_order = 0: the plain mapping; kd_split = 0: whole poses only.  What each pose took this time is kept
      // for the next search.
      const int* order = nullptr;
      unsigned* cost = nullptr;
      int n_wg = 0;
      if (i0 == 0 && cnt == n && n < (1 << 20) && dev_switches().kd_order && match_kd_takes_order(s->max_scan)) {
        if (s->kd_cost_valid && s->h_kd_cost.size() == (size_t)n) {
          const unsigned* c = s->h_kd_cost.data();
          double mean = 0.0;
          for (int q = 0; q < n; ++q) mean += (double)c[q];
          mean /= std::max(n, 1);
          std::vector<std::pair<double, int>> ent;            // (estimated time of the workgroup, its order entry)
          ent.reserve((size_t)n + 64);
          int n_split = 0;
          for (int q = 0; q < n; ++q) {
            int parts = 1;
            const int ks = dev_switches().kd_split;
            if (ks == 1 && mean > 0.0) parts = std::min(4, std::max(1, (int)std::ceil((double)c[q] / (1.6 * mean))));
            if (parts > 1 && n_split >= 256) parts = 1;      // (a map where "much longer than the average" is common has no stragglers to split)
            if (ks >= 2) parts = std::min(ks, 4);             // tests: every pose in parts
            static const double share[5] = {1.0, 1.0, 1.0 / 1.75, 1.0 / 2.5, 1.0 / 3.0};      // the inverse of the kernel's estimate
            if (parts > 1) {
              ++n_split;
              // the rows of a split pose start as "no match": a part that stages no tree for a pair leaves its entries alone
              if (hipMemsetAsync(d_table.p + (size_t)q * slots * s->max_scan, 0xFF, sizeof(uint16_t) * (size_t)slots * s->max_scan, st) != hipSuccess)
                return fail(HITL_ERR_HIP, "find_correspondences: clearing a split pose's rows failed");
            }
            for (int pt = 0; pt < parts; ++pt) ent.emplace_back((double)c[q] * share[parts], q | (pt << 20) | (parts << 24));
          }
          std::stable_sort(ent.begin(), ent.end(), [](const std::pair<double, int>& x, const std::pair<double, int>& y) { return x.first > y.first; });
          std::vector<int>& ord = s->h_kd_order;
          ord.resize(ent.size());
          for (size_t q = 0; q < ent.size(); ++q) ord[q] = ent[q].second;
          TRYC(s->d_kd_order.upload(ord, st));
          order = s->d_kd_order.p;
          n_wg = (int)ord.size();
        }
        TRYC(s->d_kd_cost.resize((size_t)n)); TRYC(s->d_kd_cost.zero(st));
        cost = s->d_kd_cost.p;
      }
      launch_match_kd(st, mv, s->max_scan, slots, d_tgt.p, d_tf.p, mp, s->kd_pts.p, s->kd_meta.p, grid, d_table.p, d_counts.p, i0, cnt, stats, order, n_wg, cost);
      if (cost != nullptr) {                      // (arrives with the counts: the tile loop synchronises right after)
        s->h_kd_cost.resize((size_t)n);
        s->kd_cost_valid = hipMemcpyAsync(s->h_kd_cost.data(), cost, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToHost, st) == hipSuccess;
      }
      if (timing) {
        unsigned long long h[6] = {0, 0, 0, 0, 0, 0};
        if (hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
          std::fprintf(stderr, "find_correspondences:   k_match_kd, poses [%d, %d): %lld candidate pairs, %llu past the box test, %llu staged a tree, %llu walks; "
                       "per workgroup: bit tests %.3f ms, staging + walks %.3f ms (the slowest workgroup: %.3f ms)\n",
                       i0, i0 + cnt, (long long)cnt * slots, h[0], h[1], h[2], 1e-5 * (double)h[3] / std::max(cnt, 1), 1e-5 * (double)h[4] / std::max(cnt, 1), 1e-5 * (double)h[5]);
        if (std::getenv("HITL_KD_PER_POSE") != nullptr) {       // development: the time of every source pose's workgroup (k_match_kd_spec only)
          std::vector<unsigned long long> per((size_t)cnt);
          if (hipMemcpyAsync(per.data(), stats + 6, per.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
            for (int q = 0; q < cnt; ++q) std::fprintf(stderr, "kd_per_pose %d %.3f\n", i0 + q, 1e-5 * (double)per[(size_t)q]);
        }
      }
    }
    else launch_match(st, mv, s->max_scan, slots, d_tgt.p, d_tf.p, mp, d_table.p, d_counts.p, i0, cnt);
    if (hipGetLastError() != hipSuccess) return fail(HITL_ERR_HIP, "find_correspondences: the match kernel could not be launched");
    return HITL_OK;
  };
  if (device_tables) TRYC(match_tile(0, n));
  // A window inside the pose band cannot produce an off-band pair: the pair tables are then built on the device, next to
  // the counts, and only five totals come back.  Wider windows (all-pairs search) take the general host path below.
  if (device_tables) {
    DevArray<int>& d_maxd = s->d_pose_maxd;
    DevArray<int64_t>&d_pmo = s->d_pose_match_off, &d_tot = s->d_pair_totals;
    TRYC(s->pose_pair_off.resize((size_t)n + 1)); TRYC(s->inc_off.resize((size_t)n + 1));
    TRYC(d_pmo.resize((size_t)n + 1)); TRYC(d_maxd.resize((size_t)n)); TRYC(d_tot.resize(5 + 6 * (size_t)pair_scan_blocks(n)));
    const int min_count = (int)s->opt.min_inter_pose_correspondence;
    launch_pair_count_scan(st, n, slots, (int)window, min_count, s->own_lo, s->own_hi, d_counts.p, s->pose_pair_off.p, d_pmo.p,
                           s->inc_off.p, d_maxd.p, d_tot.p + 5, d_tot.p);
    int64_t tot[5] = {0, 0, 0, 0, 0};
    int kd_ovf = 0;
    if (kd_tgt && hipMemcpyAsync(&kd_ovf, s->d_kd_ovf.p, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess)
      return fail(HITL_ERR_HIP, "find_correspondences: reading the overflow flag failed");
    if (hipMemcpyAsync(tot, d_tot.p, sizeof(tot), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      return fail(HITL_ERR_HIP, std::string("match kernel: ") + hipGetErrorString(hipGetLastError()));
    if (kd_tgt && kd_ovf > dev_switches().kd_redo_cap && !s->kd_full_depth) {
      // more walks overflowed the fast frame stack than the redo list holds: the search is void.  Once more with stacks as deep
      // as the trees (and from now on for this map)
      s->kd_full_depth = true;
      TRYC(d_counts.zero(st));
      TRYC(match_tile(0, n));
      launch_pair_count_scan(st, n, slots, (int)window, min_count, s->own_lo, s->own_hi, d_counts.p, s->pose_pair_off.p, d_pmo.p,
                             s->inc_off.p, d_maxd.p, d_tot.p + 5, d_tot.p);
      if (hipMemcpyAsync(tot, d_tot.p, sizeof(tot), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(HITL_ERR_HIP, std::string("match kernel (full-depth repeat): ") + hipGetErrorString(hipGetLastError()));
      mark("k_match repeated with full-depth stacks");
    }
    mark("k_match + pair counts");
    const int P = (int)tot[0];
    const int Wb = std::min(std::max((int)tot[2], std::max(s->min_band_poses, 1)), band_cap);
    s->n_pairs = P; s->n_matches = tot[1];
    s->n_pairs_owned = (int)tot[3]; s->n_matches_owned = tot[4];
    s->band_pairs_Wb = Wb;
    s->adj_window = (int)window; s->adj_slots = slots; s->h_adj.clear();       // info_mat: from the candidate counts, on demand
    s->n_offband = 0;
    ++s->corr_generation;
    s->h_pair_src.clear(); s->h_pair_tgt.clear(); s->h_perm.clear(); s->h_match_off.clear();
    s->host_pairs_stale = true;                 // ensure_host_pairs() fetches the lists for the entry points that want them
    TRYC(d_kept.resize((size_t)P)); TRYC(s->pair_src.resize((size_t)P)); TRYC(s->pair_tgt.resize((size_t)P));
    TRYC(s->match_off.resize((size_t)P + 1)); TRYC(s->inc_ent.resize((size_t)std::max(2 * P, 1)));
    TRYC(s->band_pair.resize((size_t)n * Wb * 2));
    TRYC(s->matches.resize((size_t)s->n_matches + kMatchPad));      // k_linearize_stf prefetches index words unguarded
    if (hipMemsetAsync(s->band_pair.p, 0xFF, sizeof(int) * (size_t)n * Wb * 2, st) != hipSuccess ||
        hipMemsetAsync(s->matches.p + s->n_matches, 0, sizeof(uint32_t) * kMatchPad, st) != hipSuccess)
      return fail(HITL_ERR_HIP, "find_correspondences: clearing the pair tables failed");
    launch_pair_fill(st, n, slots, (int)window, min_count, s->own_lo, s->own_hi, d_counts.p, Wb, s->pose_pair_off.p, d_pmo.p,
                     s->inc_off.p, d_kept.p, s->pair_src.p, s->pair_tgt.p, s->match_off.p, s->inc_ent.p, s->band_pair.p);
    launch_compact(st, mv, s->max_scan, slots, P, d_kept.p, s->match_off.p, d_table.p, s->matches.p);
    if (s->ob_lo.n != 1 || s->obadj_off.n != (size_t)n + 1 || s->obadj_nonzero) {     // no off-band pairs: the empty tables
      const std::vector<int> zero1(1, 0), minus1(1, -1);
      TRYC(s->ob_lo.upload(zero1, st)); TRYC(s->ob_hi.upload(zero1, st));
      TRYC(s->ob_fwd.upload(minus1, st)); TRYC(s->ob_rev.upload(minus1, st));
      TRYC(s->obadj_off.resize((size_t)n + 1)); TRYC(s->obadj_off.zero(st));
      TRYC(s->obadj_ent.resize(1)); TRYC(s->obadj_col.resize(1));
      s->obadj_nonzero = false;
    }
    if (hipStreamSynchronize(st) != hipSuccess) return fail(HITL_ERR_HIP, "find_correspondences: the pair-table kernels failed");
    mark("pair tables + k_compact");
    return HITL_OK;
  }
  // general path (windows wider than the band): tile by tile -- match, fetch the counts, keep pairs with more than
  // kMinInterPoseCorrespondence matches (:632-634) in pose order (:636-641), compact the tile's rows behind what is there
  std::vector<int> kept;
  TRYC(s->hp_cand_counts.resize((size_t)tile * slots));
  const int* counts = s->hp_cand_counts.p;                    // page-locked: the tile's counts arrive by DMA
  std::vector<int64_t> tile_off;
  s->adj_window = 0; s->adj_slots = 0; s->h_adj.clear();
  s->h_pair_src.clear(); s->h_pair_tgt.clear(); s->h_match_off.assign(1, 0);
  int64_t written = 0;                                   // matches compacted so far
  double t_sub[4] = {0, 0, 0, 0};                        // HITL_TIMING: candidates on the host / uploads + match kernel / the host's filter / k_compact
  auto sub = [&](int k, std::chrono::steady_clock::time_point& t0) {
    if (!timing) return;
    const auto t = std::chrono::steady_clock::now();
    t_sub[k] += std::chrono::duration<double, std::milli>(t - t0).count();
    t0 = t;
  };
  for (int i0 = 0; i0 < n; i0 += tile) {
    const int cnt = std::min(tile, n - i0);
    auto t_s = std::chrono::steady_clock::now();
    fill_candidates(i0, i0 + cnt);
    sub(0, t_s);
    TRYC(match_tile(i0, cnt));
    if (hipMemcpyAsync(s->hp_cand_counts.p, d_counts.p, (size_t)cnt * slots * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return fail(HITL_ERR_HIP, std::string("match kernel: ") + hipGetErrorString(hipGetLastError()));
    sub(1, t_s);
    kept.clear(); tile_off.clear();
    for (int il = 0; il < cnt; ++il)
      for (int sl = 0; sl < slots; ++sl) {
        const size_t c = (size_t)il * slots + sl;
        if (counts[c] <= 0 && s->opt.min_inter_pose_correspondence >= 0) continue;      // (neither an information-matrix mark nor a pair)
        const int i = i0 + il, jt = cand_target(i, sl);
        if (jt >= 0 && counts[c] > 0) s->h_adj.emplace_back(std::min(i, jt), std::max(i, jt));        // info_mat_ marks per accepted match (:624-625)
        const bool relevant = (i >= s->own_lo && i < s->own_hi) || (jt >= s->own_lo && jt < s->own_hi);   // halo-halo pairs belong to other ranks
        if (jt >= 0 && relevant && counts[c] > s->opt.min_inter_pose_correspondence) {
          kept.push_back((int)c);
          tile_off.push_back(s->h_match_off.back());
          s->h_pair_src.push_back(i); s->h_pair_tgt.push_back(jt);
          s->h_match_off.push_back(s->h_match_off.back() + counts[c]);
        }
      }
    const int64_t need = s->h_match_off.back() + kMatchPad;
    if ((size_t)need > s->matches.cap) {                 // grow, keeping what the earlier tiles wrote
      DevArray<uint32_t> bigger;
      TRYC(bigger.resize((size_t)std::max<int64_t>(need, 2 * (int64_t)s->matches.cap)));
      if (written > 0 && hipMemcpyAsync(bigger.p, s->matches.p, sizeof(uint32_t) * (size_t)written, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(HITL_ERR_HIP, "find_correspondences: growing the match array failed");
      HIP_TRY(hipStreamSynchronize(st));
      std::swap(bigger.p, s->matches.p); std::swap(bigger.cap, s->matches.cap); std::swap(bigger.n, s->matches.n);
    }
    s->matches.n = (size_t)need;
    sub(2, t_s);
    if (!kept.empty()) {
      TRYC(d_kept.upload(kept, st));
      TRYC(s->match_off.upload(tile_off, st));           // this tile's write offsets (the full list goes up after the loop)
      launch_compact(st, mv, s->max_scan, slots, (int)kept.size(), d_kept.p, s->match_off.p, d_table.p, s->matches.p, i0);
      if (hipStreamSynchronize(st) != hipSuccess) return fail(HITL_ERR_HIP, "compact kernel failed");
    }
    written = s->h_match_off.back();
    sub(3, t_s);
  }
  if (timing) std::fprintf(stderr, "find_correspondences:   tiles of %d poses: candidates %.3f, uploads + match kernel %.3f, filter %.3f, compact %.3f ms\n",
                           tile, t_sub[0], t_sub[1], t_sub[2], t_sub[3]);
  mark("match + filter + compact (tiles)");
  s->n_pairs = (int)s->h_pair_src.size();
  s->n_matches = s->h_match_off.back();
  s->h_perm.resize(s->n_pairs);
  std::iota(s->h_perm.begin(), s->h_perm.end(), 0);
  TRYC(s->pair_src.upload(s->h_pair_src, st));
  TRYC(s->pair_tgt.upload(s->h_pair_tgt, st));
  TRYC(s->match_off.upload(s->h_match_off, st));
  TRYC(s->matches.resize((size_t)s->n_matches + kMatchPad));      // k_linearize_stf prefetches index words unguarded (capacity is there already)
  if (hipMemsetAsync(s->matches.p + s->n_matches, 0, sizeof(uint32_t) * kMatchPad, st) != hipSuccess) { return fail(HITL_ERR_HIP, "find_correspondences: padding the match array failed"); }
  if (hipStreamSynchronize(st) != hipSuccess) { return fail(HITL_ERR_HIP, "find_correspondences: uploads failed"); }
  mark("uploads");
#undef TRYC
#undef TRYP
  rc = build_pair_structures(s);
  mark("pair structures");
  return rc;
}

}  // namespace hitl

extern "C" int hitl_libm_sincos_exact(void) { return hitl::libm_sincos_exact() ? 1 : 0; }
