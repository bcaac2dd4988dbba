// hitl_internal.h -- internal declarations shared by the HIP kernels and the host driver of
// libhitlgpu.so (gfx950 only).  Nothing here crosses the C ABI (include/hitl_gpu.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/hitl_gpu.h"

namespace hitl {

constexpr int kMaxBandPoses = 4;          // super-block size B = 3*Wb <= 12
constexpr int kPairRJ = 14;               // r0, r1, J[2][6] per directed STF pair
constexpr int kNumScalars = 16;           // device scalar scratch (doubles)
constexpr int kMatchPad = 512;            // readable words behind the last match (k_linearize_stf fetches one trip ahead)

// Process-wide switches, read in ONE place (hitl_solver_create -> read_dev_switches; nothing on a launch or solve path calls
// getenv) and kept per process: the values the last hitl_solver_create saw.
//  * Environment variables a maintainer may need (INTEGRATION.md lists exactly these): HITL_TIMING, HITL_ROCTX, HITL_FORCE_DIST,
//    HITL_DIST_COLUMNS, HITL_FORCE_RCCL.
//  * Test hooks: fields only hitl_test_set_switch() (hitl_host.h; exported, NOT part of include/hitl_gpu.h) can set.  They force
//    code paths a small test map would not reach by itself (64-bit addressing, the level-per-launch cyclic reduction that backs
//    the fused stages up, overflow of the search's frame stacks ...).  None is read from the environment.
struct DevSwitches {
  bool timing = false;             // HITL_TIMING: phase times of find_correspondences / prepare / lm_solve on stderr
                                   // (HITL_ROCTX: roctx ranges around the phases, hitl_host.h)
  bool force_dist = false;         // HITL_FORCE_DIST: one rank takes the submap code path
  bool force_rccl = false;         // HITL_FORCE_RCCL: in-library RCCL communicator with one rank
  bool dist_columns = false;       // HITL_DIST_COLUMNS: submap ranks keep the interface-column form of the linear solve although the separator form applies
  // ---- test hooks ----
  bool cr_legacy = false;          // "cr_legacy": the level-per-launch cyclic reduction (the fallback of the fused stages)
  bool cr_unfused_build = false;   // "cr_unfused_build": k_build_system as its own launch in front of the generic first stage
  bool cr_seq = true;              // "cr_seq" = 0: the first stage of the LM path's factor-solve as a cyclic-reduction stage (rounds 2-4)
                                   // instead of the two-way sequential order (k_cr_seq, round 5)
  int cr_seq_m = 0;                // "cr_seq_m": chunk size of that first stage (0 = the default)
  bool cr_tail = true;             // "cr_tail" = 0: the last chunk stage, the top and their back-substitutions as four launches (rounds 2-4)
                                   // instead of one (k_cr_tail, round 5)
  bool stf_addr64 = false;         // "stf_addr64": the 64-bit addressing instantiations of k_linearize_stf (arrays beyond 4 GB)
  bool device_lm = false;          // "device_lm": accept / reject / terminate decided on the device (the last workgroup of k_assemble), iterations queued ahead.
                                   // Measured (round 3): 4-8 us per iteration SLOWER than the host-driven loop on every workload
  int lm_ahead = 3;                // "lm_ahead": LM iterations the host queues ahead of the device-driven loop
  int kd_cap = 0;                  // "kd_cap": frame-stack entries per lane of k_match_kd_tgt's first attempt (0 = 5); 1 forces the redo walks
  int kd_redo_cap = 1 << 18;       // "kd_redo_cap": entries of the redo list behind the fast attempt (shrunk to force the full-depth repeat)
  bool cand_tf_check = false;      // "cand_tf_check": the device's candidate transforms held to the host routine, bit for bit
  size_t cand_tile_bytes = (size_t)4 << 30;   // "cand_tile_bytes": candidate-table budget of the tiled search (shrunk to force several tiles)
  bool stf_lds_targets = false;    // "stf_lds_targets": k_linearize_stf_r reads its targets from LDS (the source scan's fp32 copy) -- WRONG results, a timing experiment
  int kd_split = 1;                // "kd_split" = 0: ... and never serves one pose by several workgroups; 2-4: EVERY pose by that many (tests)
  bool kd_order = true;            // "kd_order" = 0: the source-major search maps workgroups to poses the plain way on every call
  int kd_spec = 4;                 // "kd_spec": targets in flight per workgroup of the source-major tree search (4, 2; 0 or 1: one at a time, rounds 3-5's order)
  bool kd_grid = true;             // "kd_grid" = 0: the source-major tree search walks every query (rounds 3-5) instead of skipping those an occupancy grid rules out
  bool pcg_fused = true;           // "pcg_fused" = 0: one launch group per CG iteration, the host looking every fourth (rounds 2-5) instead of one persistent launch per solve
  bool pcg_starve = false;         // "pcg_starve": the persistent PCG launch is started WITHOUT its top workgroup and with short waits (see cr_tail_starve)
  bool cr_tail_starve = false;     // "cr_tail_starve": k_cr_tail is launched WITHOUT its top workgroup and with a short wait: what a device too busy to hold
                                   // all its workgroups at once looks like (the host must notice and repeat that solve as separate launches)
  int cu_count = 0;                // "cu_count" > 0: plan launches as if the device had this many compute units (a CU-masked or partitioned device)
  // failure injection on ONE rank of a submap solve (tests/test_submap_gpu.py: every rank must come back with an error, none may
  // be left waiting in a collective)
  bool fail_prepare = false;       // "fail_prepare": the rank-local part of prepare() reports HITL_ERR_OUT_OF_MEMORY
  int poison_step_from = 0;        // "poison_step_from" = k > 0: from LM iteration k on this rank reports its step as not finite
};
const DevSwitches& dev_switches();
void read_dev_switches();          // hitl_solver_create only

// ---- view structs passed to kernels by value ------------------------------------------
struct MapView {            // SoA robot-frame scans (JointOptimization.h:62-68)
  int n_poses;
  const int64_t* obs_off;   // n+1
  const float4* obs;        // (px, py, nx, ny) per observation: one 16-byte load per point
};

struct PairView {           // directed STF pose pairs, sorted by source pose
  int n_pairs;
  const int* pose_pair_off;   // n_poses+1 : pairs whose source is pose i
  const int* pair_src;        // P
  const int* pair_tgt;        // P
  const int64_t* match_off;   // P+1
  const uint32_t* matches;    // C, packed src_idx | tgt_idx << 16
};

struct FactorView {         // everything assemble needs besides the per-pair r/J
  int with_stf, with_odom, with_hc;
  int band_poses;             // Wb
  const int* inc_off;         // n+1 : incidence list of pose i (pair*2 + role)
  const int* inc_ent;
  const int* band_pair;       // n * Wb * 2 : pair (i -> i+d), pair (i+d -> i) or -1
  const float4* odom_A;       // n : edge (i-1, i) axis transform, row-major
  const float2* odom_rr;      // n : (radial_translation, rotation)
  float sig_r, sig_t, sig_a;
  const int* hc_off;          // n+1 : rows constraining pose i
  const uint32_t* hc_type;
  const float4* hc_tgt;       // (x*, y*, theta*, penalty_dir)
  int n_offband;
  const int* ob_lo; const int* ob_hi; const int* ob_fwd; const int* ob_rev;   // E each
  const int* obadj_off;       // n+1
  const int* obadj_ent;       // entry*2 + transposed
  const int* obadj_col;       // the other pose of the entry (ob_hi or ob_lo), resolved once on the host
  double* ob_acc;             // npad scratch, or null: dense off-band adjacency -> k_spmv_offband sums a pose's off-band
                              // blocks with one wave per pose in front of k_spmv; sparse -> k_spmv walks the few entries itself
  // constant poses (global pose 0, submap halos) and the owned range of a submap
  const unsigned char* fixed; // n : 1 = constant
  int own_lo, own_hi;         // cost / |g|max / step scalars are taken over local poses [own_lo, own_hi)
};

struct LinView {            // one linearisation (double buffered by the driver)
  double* pairRJ;             // P*14
  double* Hband;              // n*(Wb+1)*9 : block (i, i+d) row-major
  double* g;                  // npad
  double* obblk;              // E*9 : block (lo, hi)
  double* partials;           // assemble workgroups x 2 : (cost, max |g|) per workgroup
};

// Device-resident state of the Levenberg-Marquardt loop (round 3: accept / reject / terminate are decided by lm_decide_step -- since round 5 run by the last workgroup of k_assemble -- on
// the device; the host only queues iterations ahead and polls a pinned progress word).  Kernels of the chain take a pointer
// to it: `done` makes the launches queued behind the terminating iteration return at once, `cur` selects which of the two
// linearisation / state buffers is the current one, `radius` is the trust region the next step is solved with.
struct LmDev {
  int cur;                    // linearisation buffer (and state buffer) of the CURRENT state: 0 / 1; the candidate lives in cur ^ 1
  int done;                   // the loop has terminated
  int term;                   // HITL_CONVERGENCE / HITL_NO_CONVERGENCE / HITL_FAILURE
  int iter, successful, unsuccessful, invalid_run, lin_calls;
  double radius, decrease_factor, cost, gmax;
};
struct LmOpts {               // the constants of the decision, by value
  double function_tolerance, gradient_tolerance, parameter_tolerance, max_radius, min_radius, min_relative_decrease, guard2;
  int max_iter, late;
};

struct CrSource {           // where the first cyclic-reduction stage can take its tiles from instead of cr.D / cr.L (Hband == null: unused)
  const double* Hband; const double* g; const double* scale; const unsigned char* fixed;
  double mu, min_diag, max_diag;
  int n_poses, Wb;
  double* rhs;                // receives the scaled right-hand side -S g (the copy the residual check and the step bookkeeping read)
  const int* unit_cols;       // right-hand sides 1 .. n_unit are unit vectors e_{unit_cols[k]} (the interface columns of a submap rank)
  int n_unit;
  // device-driven LM loop (lm != null): Hband / g are buffer 0, the *_alt pointers buffer 1, lm->cur picks; mu = lm->radius
  const LmDev* lm;
  const double* Hband_alt; const double* g_alt;
};

struct CrView {             // block-tridiagonal system + cyclic-reduction factors
  int nsb;                    // super-blocks
  int B;                      // 3*Wb
  double* D;                  // nsb*B*B  (overwritten by the reduction)
  double* L;                  // nsb*B*B  coupling (k, k-stride)
  double* Dinv;               // nsb*B*B
  double* Gl;                 // nsb*B*B  Dinv_k * L_k
  double* Gr;                 // nsb*B*B  Dinv_k * L_{k+s}^T
  // fused factorisation: what a chunk's interior contributes to its two boundary nodes, handed from stage to stage
  double* P = nullptr;        // [parity 2][left/right 2][np][B*B]
  double* Pw = nullptr;       // [rhs pw_rhs][parity 2][left/right 2][np][B]
  int np = 0;                 // cr_contrib_slots(nsb)
  int pw_rhs = 1;             // right-hand sides Pw is sized for
  const LmDev* lm = nullptr;  // device-driven LM loop: the fused stage kernels return at once when lm->done
  // Submap ranks (round 4): the chain's LAST node is a separator like its first one -- never eliminated; the fused stages leave
  // the Schur complement of everything between the two onto them (D, L, w of both nodes + the last stage's contribution slots 0 / 1).
  // It lives at its true index nsb - 1 and takes part in every stage as the node behind the last regular one (fused stages only).
  int rsep = 0;
  unsigned* ticket = nullptr; // 4 words of device memory, zero between launches: k_cr_tail's arrivals / release / error / departures (null: no fused tail)
  // k_cr_tail's workgroups wait for each other: all of them must be resident at once.  The host takes that launch only for at most
  // `cus` workgroups (one of 1024 threads per compute unit of THIS device: a CU-masked or partitioned device has fewer than 256) and
  // not at all with no_tail; a wait that runs out anyway (a device shared with another busy process) is reported through fail_host
  // (pinned host memory, system-scope store) so that the host can repeat that one solve as separate launches.
  int cus = 256;
  int no_tail = 0;
  double* fail_host = nullptr;
  unsigned tail_spins = 1u << 22;   // polls a waiting workgroup of k_cr_tail makes before it gives up (~1.5 us each)
};
// stage nodes alive at stride S (the right separator counts as one), and where stage node sn lives in D / L / w / x
__host__ __device__ inline int64_t cr_stage_nodes(const CrView& cr, int S) {
  return cr.rsep ? ((int64_t)cr.nsb - 2) / S + 2 : ((int64_t)cr.nsb + S - 1) / S;
}
int cr_contrib_slots(int nsb);

// ---- kernel launchers (hitl_kernels.hip) -------------------------------------------------
void launch_pose_trig(hipStream_t st, int n_poses, const double* x, double2* cs);   // cs[i] = (cos, sin) of pose i
// opt-in: the scan-to-scan factors through their 2 x 15 moments per pair (32 doubles per pair in pairM)
constexpr int kPairMomStride = 32;
void launch_stf_moments(hipStream_t st, const MapView& map, const PairView& pr, double* pairM);
void launch_linearize_stf_moments(hipStream_t st, const PairView& pr, const double* pairM, const double* x, const double2* cs, double corr,
                                  double sd, double* pairRJ, const LmDev* lm = nullptr, const double* x_alt = nullptr, double* pairRJ_alt = nullptr);
// lm != null (device-driven LM loop): x / pairRJ are buffer 0, x_alt / pairRJ_alt buffer 1; the kernel works on the
// CANDIDATE's buffers (lm->cur ^ 1) and returns at once when lm->done
void launch_linearize_stf(hipStream_t st, const MapView& map, const PairView& pr, int max_scan, int64_t map_obs_count, int64_t n_matches,
                          const double* x, const double2* cs, float corr, float std_dev, double* pairRJ,
                          const LmDev* lm = nullptr, const double* x_alt = nullptr, double* pairRJ_alt = nullptr,
                          const void* pose_recs = nullptr /* launch_build_stf_pose_recs: one stf_pose_rec_bytes() record per pose, or null */);
size_t stf_pose_rec_bytes();

void launch_build_stf_pose_recs(hipStream_t st, const MapView& map, const PairView& pr, void* recs);
void launch_pack_obs(hipStream_t st, int64_t n, const float* px, const float* py, const float* nx, const float* ny,
                     float4* obs);
// Device-driven LM loop: the decision of the iteration (lm_decide_step) is taken by the LAST workgroup of k_assemble to finish -- the
// kernel that produces the last numbers the decision needs -- instead of by a launch of its own behind it.
struct LmDecide {
  LmDev* lm;                   // null: no decision (host-driven loop)
  LmOpts o;
  const double* step_partials; int nblk_step;      // band pass: 8 per workgroup (r.r, b.b, dt.rhs, dt.q, |step|^2, |x|^2, bad, -)
  const double* lin_partials0; const double* lin_partials1; int nblk_lin;    // k_assemble's own (cost, max |g|), both buffers
  unsigned* arrivals;          // workgroups of this launch that have stored their partials (reset by the last one)
  double* progress;            // pinned host memory: [0] iterations completed (written last), [1] done, [2] termination, [3] cost, [4] radius
  // host-driven loop (lm == null): the last workgroup stores flag_seq into *host_flag (pinned) once every workgroup's partials are out --
  // what the one-thread kernel k_host_flag did behind k_assemble, without the launch (4 us per LM iteration)
  double* host_flag; double flag_seq;
};
void launch_assemble(hipStream_t st, const MapView& map, const PairView& pr, const FactorView& fv,
                     const double* x, const LinView& lin, const LmDev* lm = nullptr, const double* x_alt = nullptr, const LinView* lin_alt = nullptr,
                     const LmDecide* decide = nullptr);
inline int assemble_blocks(int n_poses) { return (n_poses + 63) / 64; }   // k_assemble: 4 lanes per pose, 256 threads
void launch_jacobi_scale(hipStream_t st, int n_poses, int npad, int Wb, const double* Hband, int enable,
                         const unsigned char* fixed, double* scale);
void launch_build_system(hipStream_t st, int n_poses, int Wb, const LinView& lin, const double* scale,
                         const unsigned char* fixed, double mu, double min_diag, double max_diag, const CrView& cr,
                         double* rhs, double* rhs_work = nullptr);
void launch_cr_factor(hipStream_t st, const CrView& cr);
void launch_cr_solve_multi(hipStream_t st, const CrView& cr, double* w, double* x, int nrhs, size_t stride);
void launch_unit_rhs(hipStream_t st, int npad, double* w, size_t stride, const int* cols_dev, int ncols);
void launch_iface_blocks(hipStream_t st, const double* Hband, int Wb, int Wi, int own_lo, int own_hi, bool has_lo, bool has_up,
                         double* Clo, double* Cup);
void launch_cov_iface_correction(hipStream_t st, int own_lo, int own_hi, int Wb, int B, const double* Z, size_t stride, const double* T_dev,
                                 int nc, double* Sd);
void launch_gather_iface(hipStream_t st, const double* z0, const double* Z, size_t stride, const double* scale,
                         const int* cols_dev, int ncols, double* out /* ncols * (ncols + 2) */);
void launch_combine(hipStream_t st, int npad, const double* z0, const double* Z, size_t stride, const double* coef_dev,
                    int ncols, double* dt);
void launch_scatter_halo(hipStream_t st, int cnt, const int* idx, const double* val, const double* sc, double* dt,
                         double* scale);
void launch_halo_candidate(hipStream_t st, int cnt, const int* idx, const double* x, const double* scale, const double* dt,
                           double* xc, double2* cs = nullptr /* (cos, sin) of the halo poses' candidate theta */);
// factorise and solve M x = w in one pass (w is destroyed)
void launch_cr_factor_solve(hipStream_t st, const CrView& cr, double* w, double* x, int nrhs = 1, size_t stride = 0);
// x = M^{-1} w   (w is destroyed)
// build_system + factor + solve with the first stage reading the band blocks itself (no D / L round trip through memory);
// falls back to launch_build_system + launch_cr_factor_solve where the fused stages do not apply
void launch_cr_build_factor_solve(hipStream_t st, const CrView& cr, const CrSource& src, double* w, double* x);
bool cr_build_factor_solve_is_fused(const CrView& cr, const CrSource& src, double* w, double* x);   // no launch: would it run as fused stages only?
// the same with src.n_unit unit right-hand sides behind the first one (w, x: 1 + n_unit vectors `stride` apart)
void launch_cr_build_factor_solve_multi(hipStream_t st, const CrView& cr, const CrSource& src, double* w, double* x, size_t stride);
void launch_cr_solve(hipStream_t st, const CrView& cr, const double* rhs, double* w, double* x);   // x = M^-1 rhs; w = work vector (may be rhs itself: overwritten)

// ---- submap ranks, separator form (round 4).  `chain` views the rank's OWNED super-blocks only (first node
// F, last node R, chain.rsep = 1; src / w / x shifted to the first owned pose).  Forward: the fused stages eliminate everything
// between F and R and leave the Schur complement onto the two; launch_sep_record writes the rank's exchange record
//   [ D~_F | D~_R | K (R <-> F) | C_lo (F x the previous rank's R, unscaled) | w~_F | w~_R | scales of F | scales of R ]
// (sep_rec_len doubles); launch_sep_solve builds the 2 * world node chain of all ranks' separators from the gathered records,
// solves it (the same fused cyclic reduction on a second, tiny CrView) and scatters: x at this rank's F and R, and the steps +
// Jacobi scales of the neighbours' separator poses into this rank's halo entries.  Backward: the stages' back-substitution.
static inline int sep_rec_len(int B) { return 4 * B * B + 4 * B; }
bool cr_sep_applies(const CrView& chain, const CrSource& src, double* w, double* x);
void launch_cr_sep_forward(hipStream_t st, const CrView& chain, const CrSource& src, double* w, double* x);
void launch_cr_sep_backward(hipStream_t st, const CrView& chain, double* w, double* x);
struct SepGeom {            // local pose indices of a rank's separators and the halo poses the neighbours' separators cover
  int own_lo, own_hi;       // owned poses [own_lo, own_hi)
  int r_pose0;              // first pose of R
  int halo_lo, halo_hi;     // poses [own_lo - halo_lo, own_lo) and [own_hi, own_hi + halo_hi) receive the neighbours' steps (<= Wb each)
  int rank, world, Wb;
};
void launch_sep_record(hipStream_t st, const CrView& chain, const double* w, const double* Hband_local, const double* scale_local,
                       const unsigned char* fixed_local, const SepGeom& gm, double* rec);
void launch_sep_solve(hipStream_t st, const CrView& red, const double* rec_all, double* w2, double* x2, const SepGeom& gm, double* dt_local,
                      double* scale_local);
void launch_sep_selinv(hipStream_t st, const CrView& red, const double* rec_all, double* w2, double* x2, const SepGeom& gm, const CrView& chain,
                       double* Sd2, double* So2, double* Sd, double* So);   // diagonal blocks of the GLOBAL inverse over the chain's nodes
// y = (S H S [+ D^2]) v ; damping added when mu > 0
void launch_spmv(hipStream_t st, int n_poses, int npad, const FactorView& fv, const LinView& lin,
                 const double* scale, double mu, double min_diag, double max_diag, const double* v, double* y, bool all_cols = false);
void launch_host_flag(hipStream_t st, double* flag_pinned, double seq);
// (dt.rhs, dt.q, |step|^2, |x|^2, bad, cost, max |g|) of a submap rank from its per-workgroup partials, in ordered_sum32's order
void launch_sum_partials(hipStream_t st, const double* step_partials, int nblk_step, const double* lin_partials, int nblk_lin, double* rec7);
// device-driven LM loop: the band pass on the CURRENT buffers (lin0 / lin1, x0 / x1 picked by lm->cur; mu = lm->radius),
// candidate state into the other x buffer, partials into device memory
void launch_spmv_residual_step_lm(hipStream_t st, int n_poses, int npad, const FactorView& fv, const LinView& lin0, const LinView& lin1,
                                  const double* scale, double min_diag, double max_diag, const double* v, const double* rhs, double* x0,
                                  double* x1, double* q, double* partials, double2* cs, const LmDev* lm);
// the decision of one LM iteration (Ceres' trust-region rules) from the partial sums of the band pass and of
// the candidate's linearisation; progress[0..7] (pinned host memory) = iterations completed, done, termination, cost, ...
void launch_pack_owned(hipStream_t st, int nhead, const double* head, const double* vec, int off, int cnt, int maxcnt, double* rec);
void launch_pack_edges(hipStream_t st, int nhead, const double* head, const double* vec, int off, int cnt, int edge, int maxcnt, double* rec);
void launch_scatter_idx(hipStream_t st, int cnt, const int* idx, const double* val, double* vec);
int spmv_blocks(int npad);
// r = rhs - (S H S + D^2) v ; partials[blk*8 + 0,1] = (r.r, rhs.rhs) per workgroup
void launch_spmv_residual(hipStream_t st, int n_poses, int npad, const FactorView& fv, const LinView& lin,
                          const double* scale, double mu, double min_diag, double max_diag, const double* v,
                          const double* rhs, double* r_out, double* partials);
// q = (S H S) dt ; xc = x + s*dt ; partials[blk*8 + 2..6] = (dt.rhs, dt.q, |s dt|^2, |x|^2, #non-finite)
void launch_spmv_residual_step(hipStream_t st, int n_poses, int npad, const FactorView& fv, const LinView& lin, const double* scale, double mu,
                               double min_diag, double max_diag, const double* v, const double* rhs, const double* x, double* q, double* xc,
                               double* partials, double2* cs = nullptr);   // cs: also (cos, sin) of the candidate's angles
void launch_spmv_step(hipStream_t st, int n_poses, int npad, const FactorView& fv, const LinView& lin,
                      const double* scale, const double* dt, const double* rhs, const double* x, double* q, double* xc,
                      double* partials, double2* cs = nullptr /* (cos, sin) of the candidate's theta per pose: saves k_pose_trig */);
// out[k] = a_k . b_k for k < ndots (<= 4)
void launch_dots(hipStream_t st, int n, int ndots, const double* const* a, const double* const* b, double* out);
// PCG updates with device-resident scalars s[] = (r.z, previous r.z, p.q, r.r, breakdown flag)
void launch_pcg_update_p(hipStream_t st, int n, const double* z, double* p, const double* s, bool first);
void launch_pcg_update_xr(hipStream_t st, int n, const double* p, const double* q, double* x, double* r, double* s);
// the same steps for systems one workgroup can sweep (n <= kPcgFusedMax): dot + update in one launch each, bit-identical sums
constexpr int kPcgFusedMax = 32768;
void launch_pcg_dir(hipStream_t st, int n, const double* r, const double* z, double* p, double* s, bool first);          // r.z, p = z + beta p
void launch_pcg_step(hipStream_t st, int n, const double* p, const double* q, double* x, double* r, double* s);          // p.q, x, r, r.r
// The whole PCG solve as one persistent launch (hitl_pcg.hip): systems with pose pairs outside the band, one GPU, chains of at most
// 64 chunks whose factor tiles fit LDS.  pcg_fused_plan: the launch geometry, or false (the launch-per-stage loop of hitl_lm.cpp runs).
struct PcgPlan { int M, chunks, Tp, G, capw; size_t lds_bytes, work_doubles; };
constexpr int kPcgMaxParts = 8;            // a pose's product is cut into at most this many items ...
constexpr int kPcgPartEntries = 128;       // ... of this many entries (the last one takes the rest)
bool pcg_fused_plan(int nsb, int B, int n_rows, int cus, PcgPlan* out);
size_t pcg_fused_sync_words();
// x: in x0 = M^-1 b, out the solution; r0 = b - A x0 with its (r.r, b.b) partials (launch_spmv_residual); work: PcgPlan::work_doubles
// doubles of device memory, ZERO when allocated and never touched by anything else (hand-off granules tagged by launch_index, which
// must grow by one per launch on that buffer, starting at 1); sync: pcg_fused_sync_words() words, zero when allocated;
// result_pinned[4] (host memory): CG iterations, r.r, status (0 converged or iteration cap, 1 breakdown, 2 a wait between
// workgroups ran out: the solve is void, 3 not finite), b.b
bool launch_pcg_fused(hipStream_t st, const PcgPlan& pl, int n_poses, int npad, const FactorView& fv, const LinView& lin, const CrView& cr, const double* scale,
                      double mu, double min_diag, double max_diag, double tol, int max_it, double* x, const double* r0, const double* partials0, int nblk0,
                      const int* items /* (pose, first entry, entries, part) x n_items: the products cut into parts of <= 64 entries */, int n_items,
                      const unsigned char* parts /* per pose row */,
                      double* work, unsigned* sync, uint64_t launch_index, double* result_pinned, unsigned spins, bool starve = false,
                      unsigned long long* probe = nullptr /* HITL_TIMING: 16 x G timestamps of CG iteration 2 */);
// y = alpha*x + beta*y
void launch_axpby(hipStream_t st, int n, double alpha, const double* x, double beta, double* y);
void launch_copy(hipStream_t st, int n, const double* src, double* dst);
void launch_copy_f4(hipStream_t st, size_t n_f4, const float4* src, float4* dst, int blocks);      // hitl_device_copy_bandwidth

// correspondence search (JointOptimization.cpp:561-642)
struct MatchParams { float thr, min_cos; int cap, skip; };
size_t match_lds_bytes(int max_scan);     // dynamic LDS of k_match: the target scan, per-point counters, bounding boxes
void launch_match(hipStream_t st, const MapView& map, int max_scan, int slots, const int* cand_tgt /* n*slots, -1 = none */,
                  const float* cand_tf /* n*slots*8 */, MatchParams mp, uint16_t* table /* n*slots*max_scan */,
                  int* counts /* n*slots */, int i0 = 0, int n_tile = -1 /* source poses [i0, i0 + n_tile): the arrays hold the tile only */);
// kd_tree_search: the reference's own trees and pruned walk (kdtree.cpp:37-197), bit for bit
size_t kd_build_lds_bytes(int max_scan);
size_t match_kd_lds_bytes(int max_scan);
size_t match_kd_tgt_lds_bytes(int max_scan, int block, bool full_depth);
void launch_kd_build(hipStream_t st, const MapView& map, int max_scan, float4* kd_pts, uint32_t* kd_meta);
// occupancy grid per scan (kd_grid_words() words each): lets the source-major search skip the walks that cannot find anything (exactly: hitl_match.hip)
// dense form of S J^T J S (3 n x 3 n, column-major, symmetric, A zeroed by the caller) and the extraction of the pose blocks of its inverse's
// lower triangle: pose covariances of maps with pose pairs outside the band (hitl_vec.hip)
void launch_dense_fill(hipStream_t st, int n, int Wb, const double* Hband, int n_offband, const int* ob_lo, const int* ob_hi, const double* obblk,
                       const double* scale, const unsigned char* fixed, double* A);
void launch_dense_cov_extract(hipStream_t st, int n, const double* Ainv, const double* scale, const unsigned char* fixed, const float2* rot, double* out);
size_t kd_grid_words();
void launch_kd_grid(hipStream_t st, const MapView& map, float thr, const float4* kd_pts, uint32_t* grid);
void launch_match_kd(hipStream_t st, const MapView& map, int max_scan, int slots, const int* cand_tgt, const float* cand_tf, MatchParams mp,
                     const float4* kd_pts, const uint32_t* kd_meta, const uint32_t* kd_grid /* or null */, uint16_t* table, int* counts, int i0 = 0, int n_tile = -1,
                     unsigned long long* stats = nullptr /* HITL_TIMING: 3 counters (pairs past the box test, pairs that staged a tree, walks) */,
                     const int* order = nullptr /* n_wg entries: workgroup -> pose of the tile | part << 20 | parts << 24 (k_match_kd_spec only) */,
                     int n_wg = 0,
                     unsigned* cost = nullptr /* n_tile, zeroed: out, what each pose's workgroup(s) took (k_match_kd_spec only) */);
bool match_kd_takes_order(int max_scan);      // the launcher would pick k_match_kd_spec for scans of this length
// target-major form of the same search (one workgroup per target tree, one queue of all its queries): whole map, windowed
// candidates, slots <= cap (the per-point cap cannot bind); workgroups of 256 threads (128 / 64 where a long scan's tree
// would not leave room for the frame stacks), a wave refills from the queue when 16 of its lanes are idle
bool match_kd_tgt_applies(int slots, int window, const MatchParams& mp);
// candidate transforms of a windowed search over the whole map on the device; trig = (cos, sin) of theta_j - theta_i per candidate (host libm)
void launch_cand_tf(hipStream_t st, int n, int slots, int window, int i0, int cnt, const float* poses, const float* cs, const float* trig /* from i0 */,
                    float* cand_tf /* from i0 */, int* cand_tgt /* from i0; -1 = none */);
void launch_match_kd_tgt(hipStream_t st, const MapView& map, int max_scan, int slots, int window, const float* cand_tf, MatchParams mp,
                         const float4* kd_pts, const uint32_t* kd_meta, uint16_t* table, int* counts,
                         int* redo_count /* walks that needed more frames than the launch provisioned ... */, int* redo_list /* ... (candidate, point) each,
                         repeated by k_match_kd_redo behind the main kernel; a count beyond redo_cap voids the search */, int redo_cap, bool full_depth);
void launch_compact(hipStream_t st, const MapView& map, int max_scan, int slots, int n_kept,
                    const int* kept_cand /* index into n*slots */, const int64_t* match_off, const uint16_t* table,
                    uint32_t* matches, int i0 = 0);

// pair tables of a windowed search on the device (the layout of build_pair_structures, hitl_solver.cpp):
// count + scan leave exclusive offsets (n + 1 entries) and totals[5] = pairs, matches, max |i - j|, owned pairs, owned matches
void launch_pair_count_scan(hipStream_t st, int n, int slots, int window, int min_count, int own_lo, int own_hi, const int* counts,
                            int* pose_pair_off, int64_t* pose_match_off, int* inc_off, int* pose_maxd,
                            int64_t* blk /* 6 * pair_scan_blocks(n) scratch */, int64_t* totals);
int pair_scan_blocks(int n);
void launch_pair_fill(hipStream_t st, int n, int slots, int window, int min_count, int own_lo, int own_hi, const int* counts, int Wb,
                      const int* pose_pair_off, const int64_t* pose_match_off, const int* inc_off, int* kept_cand, int* pair_src,
                      int* pair_tgt, int64_t* match_off, int* inc_ent, int* band_pair /* n*Wb*2, preset to -1 */);

// observation scans (hitl_obs.hip); T4 = n x (cos, sin, tx, ty) fp32 computed on the host
void launch_transform(hipStream_t st, const MapView& map, const float* T4, float* world_x, float* world_y);
void launch_verify(hipStream_t st, const MapView& map, const float* T4, int n, const float* pts_xy, float thr, int* seen);
void launch_obs_sets(hipStream_t st, const MapView& map, const float* T4, const float* seg8, double thr, int* counts,
                     unsigned char* mask);

// diagonal super-blocks of the inverse from the cyclic-reduction factors; per-pose 3x3 covariance in the robot frame
void launch_cr_selinv(hipStream_t st, const CrView& cr, double* Sd, double* So);
void launch_cov_extract(hipStream_t st, int n_poses, int Wb, int B, const double* Sd, const double* scale, const unsigned char* fixed,
                        const float2* rot, double* out);

// endpoint fit (hitl_obs.hip): inlier scan of one segment (counts per pose, or -- with offsets -- the inliers' world
// points in reading order) and one evaluation of the 1-parameter line problem over them
void launch_fit_inliers(hipStream_t st, const MapView& map, const float* T4, const float* seg4, int* counts, const int64_t* offsets,
                        float2* points);
int fit_eval_blocks(int64_t m);
void launch_fit_eval(hipStream_t st, const float2* pts, int64_t m, double cmx, double cmy, double len, double c, double s,
                     double* partial);

// pose-chain stages of the correction cycle (hitl_cycle.hip)
struct ExplicitArgs {        // AppExpCorrect::ApplyExplicitCorrections (ApplyExplicitCorrection.cpp:387-415), scalars from the host's libm
  int n, g0, g1;             // poses; first contiguous group of corrected poses [g0, g1]
  float c, s, theta;         // Rotation2Df(theta) of the correction
  float pivx, pivy;          // cmA
  float dstx, dsty;          // cmB / new_cmA / cmA
  float lx, ly;              // poses_[last_pose] AFTER its own correction
  float cl, sl;              // Rotation2Df(last_correction(2))
  float ltx, lty, lth;       // last_correction
};
struct AnchorRow { float x, y, angle, c, s, rel; int id; int pad; };   // per anchor pose: position, angle, cos/sin, relative_penalty_dir
void launch_explicit_apply(hipStream_t st, float* poses, const ExplicitArgs& args);
void launch_constraint_rows(hipStream_t st, const float* poses, const AnchorRow* anchors, const int* corrected, int n_anchor, int n_corr,
                            unsigned type, hitl_human_constraint_t* rows);
int backprop_blocks(int L);
void launch_backprop(hipStream_t st, float* poses, float* cov9, int lo, int hi, const float* piv4, const float* tw, float* xf, int* sync,
                     float destx, float desty, float trans_beta, float rot_beta, float* trans2);

const char* kernels_build_info();

}  // namespace hitl
