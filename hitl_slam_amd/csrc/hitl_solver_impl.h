// hitl_solver_impl.h -- the solver object behind the opaque hitl_solver handle and the internal entry points shared by
// hitl_solver.cpp (set-up, C ABI), hitl_lm.cpp (Levenberg-Marquardt loop, linear solves, submap exchange) and
// hitl_corr.cpp (correspondence search driver).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "hitl_host.h"

using namespace hitl;   // the handle type itself is global (C ABI); its members are hitl:: types

struct hitl_solver {
  hitl_options opt;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  bool staging_queued = false;  // copies out of the pinned staging buffers (setup_factors, the start state) may still be in flight
  int64_t lin_launches = 0;     // linearisations queued so far (every 4th is bracketed by events at timing_level 1)
  bool lin_timed = false;       // the linearisation in flight is bracketed
  double* h_scalars = nullptr;  // pinned, kNumScalars + 8: [8..11] = result record of the persistent PCG launch (or the scalars of the launch-per-stage loop), [14] = k_cr_tail's failure word, [15] = sequence word of spin_sync, [kNumScalars ..] = progress record of the device-driven LM loop
  DevArray<LmDev> lm_dev;       // device-resident state of the LM loop (lm_decide_step in k_assemble)
  DevArray<unsigned> lm_arrivals;             //   workgroups of the deciding launch that have stored their partials (k_assemble)
  std::vector<hipEvent_t> lm_events;   // pairs bracketing k_linearize_stf launches queued ahead (device-driven loop)
  int64_t sync_seq = 0;
  int lm_iter = 0;                 // LM iteration the host-driven loop is in (test hook poison_step_from)
  double* h_partials = nullptr; // pinned, h_partials_cap doubles
  size_t h_partials_cap = 0;
  DevArray<double> step_partials;   // spmv workgroups x 8
  DevArray<double> d_scalars;

  // ---- map ----
  int n = 0;
  int64_t n_obs = 0;
  int max_scan = 0;
  std::vector<int64_t> h_obs_off;
  DevArray<int64_t> obs_off;
  DevArray<float4> obs;                        // (px, py, nx, ny) per observation
  DevArray<float> cov9;

  // ---- pose-range submap (world_size > 1) ----
  // local pose list = [hl halo poses | owned poses | hh halo poses]; halos carry scans and state
  // of the neighbouring ranks' nearest poses and are constant in the local linear system.
  int rank = 0, world = 1;
  int64_t g_begin = 0, g_n = 0;      // first owned GLOBAL pose, global pose count
  int halo_req = 0;                  // halo width asked for at set_partition
  int hl = 0, hh = 0;                // halo poses actually present below / above the owned range
  int own_lo = 0, own_hi = 0;        // owned local range [own_lo, own_hi)
  hitl_comm_ops comm = {nullptr, nullptr};
  ncclComm_t nccl = nullptr;
  DevArray<char> comm_send, comm_recv;
  DevArray<unsigned char> fixed;     // n : constant poses (global pose 0, halos)
  DevArray<int> if_cols, halo_idx;   // interface column indices (2m) ; scalar indices of the nearest halo poses
  DevArray<double> iface_dev, rhsN, zN, coef_dev, halo_val, halo_sc;
  std::vector<double> if_rec, if_all, if_A, if_y, if_Ct;   // host side of the interface exchange
  // off-band pose pairs on a submap rank: distributed PCG (block-Jacobi = the ranks' own band factorisations) instead of the
  // exact interface system.  Every CG iteration all-gathers the owned part of one vector; halo rows are filled from it.
  bool force_dist = false;            // HITL_FORCE_DIST (development): a single rank takes the submap code path (profiling its kernels)
  bool dist_mode() const { return world > 1 || force_dist; }
  bool dist_pcg = false;
  // Separator form of the submap solve (round 4; hitl_internal.h): applies when every pose pair is inside the band and every rank
  // but the last owns a multiple of the (common) band width; the interface-column form above remains for everything else and for
  // the covariance correction.
  bool dist_sep = false;
  int sep_nsb = 0;                               // super-blocks of the chain of owned poses (first = F, last = R)
  SepGeom sep_geom = {};
  DevArray<double> sep_rec, sep_all, sep_w2, sep_x2;          // this rank's record, all ranks' records, the separator chain's rhs / solution
  DevArray<double> sep_D, sep_L, sep_Dinv, sep_Gl, sep_Gr, sep_P, sep_Pw;   // the separator chain (2 * world nodes)
  std::vector<double> sep_host;                  // host side of the record exchange when a caller callback carries it
  CrView sep_chain_view() {                      // the owned super-blocks as a chain whose last node stays
    CrView c = cr_view();
    c.nsb = sep_nsb; c.rsep = 1; c.pw_rhs = cr_pw_rhs;
    return c;
  }
  CrView sep_red_view() {
    CrView c;
    c.nsb = 2 * world; c.B = B; c.D = sep_D.p; c.L = sep_L.p; c.Dinv = sep_Dinv.p; c.Gl = sep_Gl.p; c.Gr = sep_Gr.p;
    c.P = sep_P.p; c.Pw = sep_Pw.p; c.np = cr_contrib_slots(2 * world); c.pw_rhs = 1;
    return c;
  }
  std::vector<int64_t> part_begin, part_count;   // owned global range of every rank
  int maxown = 0;                                // widest owned range, in poses
  int pcg_edge = 0;                              // owned poses at either end of a rank's range that travel in the PCG exchange record
  int n_halo = 0;                                // halo scalars (3 per halo pose, all of them)
  std::vector<int> halo_owner_slot;              // per halo scalar: index into the gathered records (rank * record + 3 * pose + c), head excluded
  DevArray<double> pcg_rec, pcg_head;
  std::vector<double> pcg_all, pcg_stage, pcg_hval;
  int Wi = 1;                        // interface width in poses: deepest reach of a factor across any cut (global max)
  int n_pairs_owned = 0;             // pairs whose SOURCE pose is owned (each pair is counted by one rank)
  int64_t n_matches_owned = 0;

  // ---- human rows: constrained_pose_id is LOCAL, anchor_pose_id stays GLOBAL ----
  std::vector<hitl_human_constraint_t> h_hc;

  // ---- correspondences ----
  int n_pairs = 0;
  int64_t n_matches = 0;
  std::vector<int> h_pair_src, h_pair_tgt;     // internal order: sorted by (src)
  std::vector<int> h_perm;                     // internal index -> caller index
  std::vector<int64_t> h_match_off;            // internal order
  DevArray<int> pair_src, pair_tgt, pose_pair_off;
  DevArray<int64_t> match_off;
  DevArray<uint32_t> matches;
  // derived
  int band_pairs_Wb = 1;                       // Wb implied by the pair set
  // scratch of the correspondence search, kept between calls
  std::vector<float> h_cand_cs;
  PinnedArray<float> hp_cand_trig;             // (cos, sin) of every candidate's angle difference: the upload of a search tile
  PinnedArray<int> hp_cand_counts;             // matches per candidate of a tile of the general (wider than the band) search
  DevArray<int> d_cand_tgt, d_cand_counts, d_cand_kept;
  DevArray<int> d_pose_maxd;                   // device-built pair tables (windowed search inside the band): per-pose scratch
  DevArray<int64_t> d_pose_match_off, d_pair_totals;
  bool host_pairs_stale = false;               // h_pair_src / h_pair_tgt / h_perm / h_match_off not yet fetched from the device
  bool obadj_nonzero = true;                   // the off-band adjacency on the device is not known to be the empty one
  DevArray<float> d_cand_tf;
  DevArray<float> d_cand_trig, d_cand_poses, d_cand_cs;      // device-side candidate transforms: (cos, sin) per candidate, poses, (cos, sin) per pose
  DevArray<uint16_t> d_cand_table;
  DevArray<float4> kd_pts;                     // opt-in (kd_tree_search): every scan in KD in-order layout
  DevArray<uint32_t> kd_meta;                  //   original index | split plane << 16
  PinnedArray<char> h_stage[2];                // staged_d2h: two page-locked pieces for large copies into pageable memory
  hipEvent_t ev_stage[2] = {nullptr, nullptr};
  DevArray<unsigned> cr_ticket;                // k_cr_tail's four ticket words (zero between launches)
  double prepare_pcie_bytes = 0.0;             // host -> device bytes of the last prepare() (factor tables): hitl_summary.pcie_bytes
  int cu_count = 256;                          // compute units of the device
  bool pcg_fused_off = false;                  // a wait inside the persistent PCG launch once ran out on this device: the launch-per-stage loop from then on
  DevArray<double> pcg_work;                   // k_pcg_fused: hand-off buffers (PcgPlan::work_doubles)
  DevArray<unsigned> pcg_sync;                 //   its abort word
  DevArray<unsigned long long> pcg_probe;      //   HITL_TIMING: phase timestamps of one CG iteration of the first launch
  std::vector<int> h_obadj_off;                //   host copy of the off-band adjacency offsets (n + 1)
  DevArray<int> pcg_items;                     //   product items: (pose, first entry, entries, part) x n_items
  DevArray<unsigned char> pcg_parts;           //   parts per pose row (npad / 3)
  int pcg_n_items = 0;
  int64_t pcg_items_of = -1;                   //   correspondence-set generation ...
  int pcg_items_wb = 0;                        //   ... and pose band the items were cut for
  uint64_t pcg_launches = 0;                   //   launches so far on pcg_work (the epoch base of the granule tags)
  hipEvent_t ev4 = nullptr, ev5 = nullptr;     //   brackets of that launch (hitl_summary.pcg_ms)
  bool cr_tail_off = false;                    // a wait inside k_cr_tail once ran out on this device: separate launches from then on
  DevArray<char> stf_pose_recs;                // per-pose descriptor records of the half-wave linearisation (k_linearize_stf_r)
  int64_t stf_pose_recs_of = -1;               //   correspondence-set generation they belong to
  bool kd_full_depth = false;                  //   a walk on this map once needed more frames than the fast attempt provisions: full-depth stacks from then on
  DevArray<int> d_kd_ovf;                      //   k_match_kd_tgt: [0] = walks that overflowed the fast frame stack, [1 ..] = their (candidate, point) list
  bool kd_built = false;                       //   trees belong to the resident map
  DevArray<uint32_t> kd_grid;                  //   occupancy grid of every scan (source-major search only: built on its first use per map)
  bool kd_grid_built = false;
  DevArray<int> d_kd_order;                    //   source-major search, one tile: workgroup -> pose, longest first by the previous search of this map
  DevArray<unsigned> d_kd_cost;                //   ... and what each pose's workgroup took this time
  std::vector<int> h_kd_order;
  std::vector<unsigned> h_kd_cost;
  bool kd_cost_valid = false;
  DevArray<unsigned long long> d_kd_stats;     //   HITL_TIMING: stage counters of k_match_kd
  float kd_grid_thr = 0.0f;                    //   the point-match threshold the grids were laid out for
  DevArray<float> world_xy;                    // hitl_solver_transform_scans*: world x of every observation, then world y
  DevArray<float> pose_rt;                     // solver_pose_rt: (cos, sin, x, y) per resident pose ...
  std::vector<float> pose_rt_of;               // ... of these poses (n * 3)
  // pose pairs with >= 1 accepted match in the last search (info_mat_, JointOptimization.cpp:624-625): either derivable from
  // the candidate counts still on the device (windowed search: adj_window > 0), or collected on the host (tiled search)
  int adj_window = 0, adj_slots = 0;
  std::vector<std::pair<int, int>> h_adj;
  int min_band_poses = 1;                      // lower bound of the pose band (a submap rank: the interface width of the whole partition)
  DevArray<double> pairM;                      // opt-in (stf_moments): 2 x 15 moments per pair
  int64_t moments_of = -1;                     // correspondence-set generation the moments belong to
  int64_t corr_generation = 0;                 // bumped whenever the pair tables change
  int n_offband = 0;
  DevArray<int> inc_off, inc_ent, band_pair, ob_lo, ob_hi, ob_fwd, ob_rev, obadj_off, obadj_ent, obadj_col;
  DevArray<double> ob_acc;                     // per-pose sums of the off-band products (k_spmv_offband)

  // ---- per-solve factor tables ----
  PinnedArray<float4> h_odom_A;                // host staging of the per-solve tables (setup_factors) and of the start state (lm_solve)
  PinnedArray<float2> h_odom_rr;
  PinnedArray<int> h_hc_off;
  PinnedArray<uint32_t> h_hc_type;
  PinnedArray<float4> h_hc_tgt;
  PinnedArray<double> h_x;
  DevArray<float4> odom_A, hc_tgt;
  DevArray<float2> odom_rr;
  DevArray<int> hc_off;
  DevArray<uint32_t> hc_type;
  int num_hc_residuals = 0;

  // ---- work buffers ----
  int wb_Wb = 0, wb_n = 0, wb_pairs = -1, wb_offband = -1;
  int nsb = 0, B = 0, npad = 0;
  DevArray<double> x, xc, scale, rhs, dt, wv, qv, rv, zv, pv;
  DevArray<double> pairRJ[2], Hband[2], gvec[2], obblk[2], lin_partials[2];
  DevArray<double2> pose_cs;        // (cos, sin) of every pose at the state being linearised
  DevArray<double> crD, crL, crDinv, crGl, crGr, crP, crPw;
  int cr_pw_rhs = 1;                // right-hand sides crPw is sized for (1 + 6 * band on a submap rank)

  MapView map_view() const {
    MapView m;
    m.n_poses = n; m.obs_off = obs_off.p; m.obs = obs.p;
    return m;
  }
  PairView pair_view() const {
    PairView p;
    p.n_pairs = n_pairs; p.pose_pair_off = pose_pair_off.p; p.pair_src = pair_src.p; p.pair_tgt = pair_tgt.p;
    p.match_off = match_off.p; p.matches = matches.p;
    return p;
  }
  LinView lin_view(int k) {
    LinView l;
    l.pairRJ = pairRJ[k].p; l.Hband = Hband[k].p; l.g = gvec[k].p; l.obblk = obblk[k].p; l.partials = lin_partials[k].p;
    return l;
  }
  CrView cr_view() {
    CrView c;
    c.nsb = nsb; c.B = B; c.D = crD.p; c.L = crL.p; c.Dinv = crDinv.p; c.Gl = crGl.p; c.Gr = crGr.p;
    c.P = crP.p; c.Pw = crPw.p; c.np = cr_contrib_slots(nsb); c.pw_rhs = cr_pw_rhs;
    c.ticket = cr_ticket.p;
    c.cus = cu_count; c.no_tail = (cr_tail_off || dist_mode()) ? 1 : 0; c.fail_host = h_scalars ? h_scalars + 14 : nullptr;
    return c;
  }
};

namespace hitl {

struct LinResult { double cost, gmax; float stf_ms; int timed; };

static inline double angle_mod(double a) {  // shared/math/util.h:433-439
  return a - 6.28318530717958647692 * std::rint(a / 6.28318530717958647692);
}

// hitl_solver.cpp
int build_pair_structures(hitl_solver* s);
int ensure_host_pairs(hitl_solver* s);      // host mirrors of the pair lists, fetched on first use after a device-side build
int ensure_empty_pair_tables(hitl_solver* s);
int ensure_work(hitl_solver* s, int Wb);
int setup_factors(hitl_solver* s, const float* poses, const float* poses_global, bool with_odom_hc);
FactorView factor_view(hitl_solver* s, int mode, int Wb);
int check_ready(hitl_solver* s, int mode);
int prepare(hitl_solver* s, int mode, const float* poses_ref, const float* poses_global, FactorView* fv);
// hitl_lm.cpp
int run_linearize(hitl_solver* s, const FactorView& fv, const double* xdev, int buf, LinResult* out);
int dist_cov_correction(hitl_solver* s, const FactorView& fv, int buf, double* Sd, double* So);   // submap ranks: global-inverse correction of the selected inversion
int run_linearize_dist(hitl_solver* s, const FactorView& fv, const double* xdev, int buf, LinResult* out);   // cost / |g|max summed over the ranks
int comm_allgather(hitl_solver* s, const void* send, void* recv, int64_t bytes);
int dist_cov_sep(hitl_solver* s, const FactorView& fv, int buf, double* Sd, double* So);      // separator form: Sd / So by chain node
int comm_allgather_dev(hitl_solver* s, const void* send_dev, void* recv_host, int64_t bytes, std::vector<double>& stage);   // send record in device memory
int lm_solve(hitl_solver* s, int mode, const float* poses_in, const float* poses_global, float* poses_out, hitl_summary* summary);
// hitl_corr.cpp
int find_correspondences(hitl_solver* s, const float* poses, int64_t window);
int ensure_kd_trees(hitl_solver* s);     // BuildKDTrees: once per resident map

}  // namespace hitl
