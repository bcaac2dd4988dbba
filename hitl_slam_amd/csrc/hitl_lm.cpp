// hitl_lm.cpp -- the Levenberg-Marquardt loop of libhitlgpu.so with Ceres' trust-region semantics:
// linearisation passes, the linear solves (exact banded cyclic reduction, PCG for off-band pairs, the submap interface
// system), the tiny all-gathers of the pose-range submaps.  Every O(N), O(P) or O(C) step is a HIP kernel; only a
// handful of scalars per iteration cross to the host.  There is no CPU fallback anywhere in this file.

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "hitl_solver_impl.h"

using namespace hitl;

namespace hitl {


// one pass over every factor at state xdev: residuals, Jacobians, J^T J blocks, gradient, cost.
// enqueue_linearize only queues the kernels and the copy of the per-workgroup partials (behind the step partials in
// the pinned buffer); finish_linearize adds them up once the stream has been synchronised.
static int enqueue_linearize(hitl_solver* s, const FactorView& fv, const double* xdev, int buf, bool direct = false, bool have_trig = false,
                             bool device_only = false /* the partials stay in device memory (launch_sum_partials adds them up there) */,
                             double flag_seq = 0.0 /* != 0 (with direct): k_assemble's last workgroup raises the host flag itself */) {
  hipStream_t st = s->stream;
  const MapView mv = s->map_view();
  const PairView pv = s->pair_view();
  LinView lv = s->lin_view(buf);
  double* host = s->h_partials + (size_t)spmv_blocks(s->npad) * 8;
  if (direct) lv.partials = host;      // pinned host memory is device-visible: the partials land there without a copy
  if (fv.with_stf && !have_trig) launch_pose_trig(st, s->n, xdev, s->pose_cs.p);   // have_trig: the band pass that produced xdev wrote them
  // ev0..ev1 brackets exactly the k_linearize_stf launch.  An event record drains the queue (~6 us each), so at the
  // default timing level only every 4th linearisation is bracketed; level 2 brackets all, level 0 none.
  const int64_t tl = s->opt.timing_level;
  s->lin_timed = fv.with_stf && (tl >= 2 || (tl == 1 && (s->lin_launches & 3) == 0));
  ++s->lin_launches;
  if (s->lin_timed) HIP_TRY(hipEventRecord(s->ev0, st));
  if (fv.with_stf) {
    if (s->opt.stf_moments)
      launch_linearize_stf_moments(st, pv, s->pairM.p, xdev, s->pose_cs.p, s->opt.point_correlation_factor, s->opt.laser_std_dev, lv.pairRJ);
    else
      launch_linearize_stf(st, mv, pv, s->max_scan, s->n_obs, s->n_matches, xdev, s->pose_cs.p, s->opt.point_correlation_factor, s->opt.laser_std_dev, lv.pairRJ,
                           nullptr, nullptr, nullptr, s->stf_pose_recs_of == s->corr_generation ? s->stf_pose_recs.p : nullptr);
  }
  if (s->lin_timed) HIP_TRY(hipEventRecord(s->ev1, st));
  if (flag_seq != 0.0 && direct && !(fv.with_stf && fv.n_offband > 0)) {
    LmDecide dec = {};
    dec.arrivals = s->lm_arrivals.p; dec.host_flag = s->h_scalars + 15; dec.flag_seq = flag_seq;
    launch_assemble(st, mv, pv, fv, xdev, lv, nullptr, nullptr, nullptr, &dec);
  } else
    launch_assemble(st, mv, pv, fv, xdev, lv);
  HIP_TRY(hipGetLastError());
  const int nblk = assemble_blocks(s->n);
  if (!direct && !device_only) HIP_TRY(hipMemcpyAsync(host, lv.partials, sizeof(double) * 2 * nblk, hipMemcpyDeviceToHost, st));
  return HITL_OK;
}

static int finish_linearize(hitl_solver* s, const FactorView& fv, LinResult* out) {
  const int nblk = assemble_blocks(s->n);
  const double* host = s->h_partials + (size_t)spmv_blocks(s->npad) * 8;
  const double cost = ordered_sum32(host, nblk, 2);   // fixed order: bit-reproducible, and the order the device-side decision uses
  double gmax = 0.0;
  for (int b = 0; b < nblk; ++b) {
    const double g = host[2 * b + 1];
    if (g > gmax || g != g) gmax = g;
  }
  out->cost = cost;
  out->gmax = gmax;
  out->stf_ms = 0.f;
  out->timed = s->lin_timed ? 1 : 0;
  if (s->lin_timed) HIP_TRY(hipEventElapsedTime(&out->stf_ms, s->ev0, s->ev1));
  return HITL_OK;
}

// Wait for the stream without sleeping on an interrupt: the LM loop turns around in tens of microseconds.  A one-thread
// kernel behind the queued work stores a sequence number into pinned host memory and the host polls that word: ~0.25 ms per
// re-solve cheaper than hipEventRecord + hipEventQuery (measured in round 2).  The kernels in front of it
// have completed -- and their own stores to pinned memory (the partial sums) are visible -- before it starts.
static int wait_host_flag(hitl_solver* s, double seq);
static int spin_sync(hitl_solver* s) {
  const double seq = (double)(++s->sync_seq);
  launch_host_flag(s->stream, s->h_scalars + 15, seq);   // (hipStreamWriteValue64 instead of this one-thread kernel: measured, no difference)
  return wait_host_flag(s, seq);
}
// the flag may also be raised by the last workgroup of k_assemble (LmDecide::host_flag): no launch of its own
static int wait_host_flag(hitl_solver* s, double seq) {
  volatile double* flag = s->h_scalars + 15;
  for (uint32_t spins = 1;; ++spins) {
    if (*flag == seq) { std::atomic_thread_fence(std::memory_order_acquire); return HITL_OK; }
    if ((spins & 0xfffffu) == 0) {                       // every million polls: is the stream still alive?
      const hipError_t e = hipStreamQuery(s->stream);
      if (e == hipSuccess) {
        if (*flag == seq) { std::atomic_thread_fence(std::memory_order_acquire); return HITL_OK; }
        return fail(HITL_ERR_HIP, "the stream drained without delivering the host flag");
      }
      if (e != hipErrorNotReady) return fail(HITL_ERR_HIP, std::string("while waiting for the stream: ") + hipGetErrorString(e));
    }
  }
}

// k_cr_tail's workgroups wait for one another; a wait that ran out (its workgroups were not all resident: a device shared with
// another busy process) is reported through a pinned word.  Called after a host synchronisation: true = the linear solve just
// waited for is void; the fused tail is off for this solver from now on and the tickets are clean for whatever runs next.
static bool take_cr_tail_failure(hitl_solver* s) {
  volatile double* word = s->h_scalars + 14;
  if (*word == 0.0) return false;
  *word = 0.0;
  s->cr_tail_off = true;
  if (s->cr_ticket.p != nullptr) (void)hipMemsetAsync(s->cr_ticket.p, 0, sizeof(unsigned) * 4, s->stream);
  return true;
}

int run_linearize(hitl_solver* s, const FactorView& fv, const double* xdev, int buf, LinResult* out) {
  TRY(enqueue_linearize(s, fv, xdev, buf));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return finish_linearize(s, fv, out);
}

// ---------------------------------------------------------------------------------------
// pose-range submaps: tiny all-gathers on host buffers (RCCL built in, or a caller callback)
// ---------------------------------------------------------------------------------------
int comm_allgather(hitl_solver* s, const void* send, void* recv, int64_t bytes) {
  if (s->world == 1 && !s->nccl && !s->comm.allgather) { std::memcpy(recv, send, (size_t)bytes); return HITL_OK; }
  if (s->comm.allgather) {
    const int rc = s->comm.allgather(s->comm.ctx, send, recv, bytes);
    return rc == 0 ? HITL_OK : fail(HITL_ERR_COMM, "all-gather callback failed");
  }
  if (!s->nccl) return fail(HITL_ERR_NOT_READY, "world_size > 1 needs hitl_options.rccl_unique_id or hitl_solver_set_comm");
  hipStream_t st = s->stream;
  TRY(s->comm_send.resize((size_t)bytes));
  TRY(s->comm_recv.resize((size_t)bytes * s->world));
  HIP_TRY(hipMemcpyAsync(s->comm_send.p, send, (size_t)bytes, hipMemcpyHostToDevice, st));
  const ncclResult_t r = ncclAllGather(s->comm_send.p, s->comm_recv.p, (size_t)bytes, ncclChar, s->nccl, st);
  if (r != ncclSuccess) return fail(HITL_ERR_COMM, std::string("ncclAllGather: ") + ncclGetErrorString(r));
  HIP_TRY(hipMemcpyAsync(recv, s->comm_recv.p, (size_t)bytes * s->world, hipMemcpyDeviceToHost, st));
  TRY(spin_sync(s));
  return HITL_OK;
}

// the same exchange with the send record already in device memory: RCCL takes it where it is and one copy brings the
// gathered records to the host; a caller callback (host buffers) gets a host copy of the record first
int comm_allgather_dev(hitl_solver* s, const void* send_dev, void* recv_host, int64_t bytes, std::vector<double>& stage) {
  hipStream_t st = s->stream;
  if (s->nccl && !s->comm.allgather) {
    TRY(s->comm_recv.resize((size_t)bytes * s->world));
    const ncclResult_t r = ncclAllGather(send_dev, s->comm_recv.p, (size_t)bytes, ncclChar, s->nccl, st);
    if (r != ncclSuccess) return fail(HITL_ERR_COMM, std::string("ncclAllGather: ") + ncclGetErrorString(r));
    HIP_TRY(hipMemcpyAsync(recv_host, s->comm_recv.p, (size_t)bytes * s->world, hipMemcpyDeviceToHost, st));
    TRY(spin_sync(s));
    return HITL_OK;
  }
  stage.resize((size_t)(bytes + 7) / 8);
  HIP_TRY(hipMemcpyAsync(stage.data(), send_dev, (size_t)bytes, hipMemcpyDeviceToHost, st));
  TRY(spin_sync(s));
  return comm_allgather(s, stage.data(), recv_host, bytes);
}

// all-gather from device memory to device memory: RCCL takes the record where it lies and delivers where the next kernel reads
// -- no host synchronisation at all; a caller callback (host buffers) is fed through `stage`, which must stay untouched until
// the stream has been synchronised again (the upload of the gathered records reads it)
int comm_allgather_d2d(hitl_solver* s, const void* send_dev, void* recv_dev, int64_t bytes, std::vector<double>& stage) {
  hipStream_t st = s->stream;
  if (s->nccl && !s->comm.allgather) {
    const ncclResult_t r = ncclAllGather(send_dev, recv_dev, (size_t)bytes, ncclChar, s->nccl, st);
    if (r != ncclSuccess) return fail(HITL_ERR_COMM, std::string("ncclAllGather: ") + ncclGetErrorString(r));
    return HITL_OK;
  }
  if (s->world == 1 && !s->comm.allgather) {
    HIP_TRY(hipMemcpyAsync(recv_dev, send_dev, (size_t)bytes, hipMemcpyDeviceToDevice, st));
    return HITL_OK;
  }
  const size_t words = (size_t)(bytes + 7) / 8;
  stage.resize(words * (size_t)(1 + s->world));
  HIP_TRY(hipMemcpyAsync(stage.data(), send_dev, (size_t)bytes, hipMemcpyDeviceToHost, st));
  TRY(spin_sync(s));
  TRY(comm_allgather(s, stage.data(), stage.data() + words, bytes));
  HIP_TRY(hipMemcpyAsync(recv_dev, stage.data() + words, (size_t)bytes * s->world, hipMemcpyHostToDevice, st));
  return HITL_OK;
}

static inline double nan_max(double a, double b) { return (b > a || b != b) ? b : a; }

// local linearisation (every rank evaluates all factors touching its owned poses, so its rows are
// complete) + the global cost and gradient max-norm
int run_linearize_dist(hitl_solver* s, const FactorView& fv, const double* xdev, int buf, LinResult* out) {
  TRY(run_linearize(s, fv, xdev, buf, out));
  if (s->world == 1) return HITL_OK;
  const int W = s->world;
  double rec[2] = {out->cost, out->gmax};
  std::vector<double> all((size_t)2 * W);
  TRY(comm_allgather(s, rec, all.data(), sizeof(rec)));
  double cost = 0.0, gmax = 0.0;
  for (int r = 0; r < W; ++r) {          // rank order on every rank: identical sums everywhere
    cost += all[(size_t)2 * r];
    gmax = nan_max(gmax, all[(size_t)2 * r + 1]);
  }
  out->cost = cost;
  out->gmax = gmax;
  return HITL_OK;
}

// dense solve with partial pivoting (the interface system, 6 * band * world unknowns)
static bool solve_dense(int n, std::vector<double>& A, std::vector<double>& b) {
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r) if (std::fabs(A[(size_t)r * n + c]) > std::fabs(A[(size_t)piv * n + c])) piv = r;
    if (!(std::fabs(A[(size_t)piv * n + c]) > 0.0)) return false;
    if (piv != c) { for (int k = 0; k < n; ++k) std::swap(A[(size_t)c * n + k], A[(size_t)piv * n + k]); std::swap(b[c], b[piv]); }
    for (int r = c + 1; r < n; ++r) {
      const double f = A[(size_t)r * n + c] / A[(size_t)c * n + c];
      if (f == 0.0) continue;
      for (int k = c; k < n; ++k) A[(size_t)r * n + k] -= f * A[(size_t)c * n + k];
      b[r] -= f * b[c];
    }
  }
  for (int r = n - 1; r >= 0; --r) {
    double v = b[r];
    for (int k = r + 1; k < n; ++k) v -= A[(size_t)r * n + k] * b[k];
    b[r] = v / A[(size_t)r * n + r];
  }
  return true;
}

// ---------------------------------------------------------------------------------------
// linear solve: PCG on (S H S + D^2) preconditioned by the exact block-cyclic-reduction
// factorisation of its in-band part.  With no off-band pose pairs the preconditioner IS the
// matrix and the first iterate is the exact LM step (what Ceres' sparse Cholesky returns).
// ---------------------------------------------------------------------------------------
struct StepScalars { double dt_rhs, dt_q, step2, xn2, bad; };

static int enqueue_step_partials(hitl_solver* s) {
  const int nblk = spmv_blocks(s->npad);
  HIP_TRY(hipMemcpyAsync(s->h_partials, s->step_partials.p, sizeof(double) * 8 * nblk, hipMemcpyDeviceToHost, s->stream));
  return HITL_OK;
}

static int finish_step_partials(hitl_solver* s, double* rr, double* bb, StepScalars* ss);

static int read_step_partials(hitl_solver* s, double* rr, double* bb, StepScalars* ss) {
  TRY(enqueue_step_partials(s));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return finish_step_partials(s, rr, bb, ss);
}

static int finish_step_partials(hitl_solver* s, double* rr, double* bb, StepScalars* ss) {
  const int nblk = spmv_blocks(s->npad);
  double acc[7];
  ordered_sum32_records<7>(s->h_partials, nblk, 8, acc);     // the order the device-side decision uses (ordered_sum32 per sum), one pass over the records
  if (rr) *rr = acc[0];
  if (bb) *bb = acc[1];
  if (ss) { ss->dt_rhs = acc[2]; ss->dt_q = acc[3]; ss->step2 = acc[4]; ss->xn2 = acc[5]; ss->bad = acc[6]; }
  return HITL_OK;
}

// Solve (S H S + D^2) dt = rhs, then form the trust-region bookkeeping (model cost change
// pieces, step norm, candidate state) -- one host synchronisation in the banded case.
static int linear_solve_and_step(hitl_solver* s, const FactorView& fv, int buf, double mu, const double* xcur,
                                 double* xcand, int* iters, bool* ok, StepScalars* ss) {
  hipStream_t st = s->stream;
  const LinView lv = s->lin_view(buf);
  const CrView cr = s->cr_view();
  const double mind = s->opt.min_lm_diagonal, maxd = s->opt.max_lm_diagonal;
  const int npad = s->npad;
  const double tol = std::max(s->opt.linear_solver_tolerance, 1e-15);
  launch_build_system(st, s->n, fv.band_poses, lv, s->scale.p, s->fixed.p, mu, mind, maxd, cr, s->rhs.p);
  launch_copy(st, npad, s->rhs.p, s->wv.p);
  launch_cr_factor_solve(st, cr, s->wv.p, s->dt.p);   // x0 = M^{-1} b
  *iters = 1;
  *ok = true;
  // r = b - A x0 (partials of r.r and b.b); banded systems: x0 is the exact step up to rounding
  launch_spmv_residual(st, s->n, npad, fv, lv, s->scale.p, mu, mind, maxd, s->dt.p, s->rhs.p, s->rv.p, s->step_partials.p);
  if (fv.n_offband == 0) {
    launch_spmv_step(st, s->n, npad, fv, lv, s->scale.p, s->dt.p, s->rhs.p, xcur, s->qv.p, xcand, s->step_partials.p);
    HIP_TRY(hipGetLastError());
    double rr, bb;
    TRY(read_step_partials(s, &rr, &bb, ss));
    const double guard = std::max(tol, 1e-9);
    if (!(rr == rr) || rr > guard * guard * bb) *ok = false;   // the factorisation broke down
    return HITL_OK;
  }
  // off-band pose pairs present: PCG on the full matrix from x0, preconditioned by the band factorisation.  The
  // iteration runs on the device: its scalars (r.z, p.q, r.r) never leave d_scalars, a CG iteration is the fused
  // preconditioner stages (forward + backward), the product (off-band sums + band) and two single-workgroup kernels that
  // pair each dot with the update that needs it (larger systems: 3 dots + 2 updates), and the host looks at r.r
  // only once per kPcgBatch iterations (up to kPcgBatch - 1 iterations past the tolerance: harmless).
  double rr, bb;
  TRY(read_step_partials(s, &rr, &bb, nullptr));
  const double target = tol * tol * bb;
  if (!(rr == rr)) { *ok = false; return HITL_OK; }
  constexpr int kPcgBatch = 4;
  double* sc = s->d_scalars.p + 8;           // (r.z, previous r.z, p.q, r.r, breakdown)
  HIP_TRY(hipMemsetAsync(sc, 0, sizeof(double) * 5, st));
  for (int it = 0; it < s->opt.max_linear_iterations && rr > target;) {
    const int nb = std::min(kPcgBatch, s->opt.max_linear_iterations - it);
    for (int k = 0; k < nb; ++k, ++it) {
      launch_cr_solve(st, cr, s->rv.p, s->wv.p, s->zv.p);                          // z = M^-1 r
      if (npad <= kPcgFusedMax) {          // one workgroup sweeps the vectors: each dot shares a launch with the update that needs it
        launch_pcg_dir(st, npad, s->rv.p, s->zv.p, s->pv.p, sc, it == 0);          // r.z, p = z + beta p
        launch_spmv(st, s->n, npad, fv, lv, s->scale.p, mu, mind, maxd, s->pv.p, s->qv.p);   // q = A p
        launch_pcg_step(st, npad, s->pv.p, s->qv.p, s->dt.p, s->rv.p, sc);         // p.q, x += alpha p, r -= alpha q, r.r
        continue;
      }
      const double* a1[4] = {s->rv.p, s->rv.p, s->rv.p, s->rv.p};
      const double* b1[4] = {s->zv.p, s->zv.p, s->zv.p, s->zv.p};
      launch_dots(st, npad, 1, a1, b1, sc + 0);                                    // r.z
      launch_pcg_update_p(st, npad, s->zv.p, s->pv.p, sc, it == 0);                // p = z + beta p
      launch_spmv(st, s->n, npad, fv, lv, s->scale.p, mu, mind, maxd, s->pv.p, s->qv.p);   // q = A p
      const double* a2[4] = {s->pv.p, s->pv.p, s->pv.p, s->pv.p};
      const double* b2[4] = {s->qv.p, s->qv.p, s->qv.p, s->qv.p};
      launch_dots(st, npad, 1, a2, b2, sc + 2);                                    // p.q
      launch_pcg_update_xr(st, npad, s->pv.p, s->qv.p, s->dt.p, s->rv.p, sc);      // x += alpha p, r -= alpha q
      const double* a3[4] = {s->rv.p, s->rv.p, s->rv.p, s->rv.p};
      launch_dots(st, npad, 1, a3, a3, sc + 3);                                    // r.r
    }
    *iters += nb;
    HIP_TRY(hipMemcpyAsync(s->h_scalars + 8, sc, sizeof(double) * 5, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    rr = s->h_scalars[8 + 3];
    if (!(rr == rr) || s->h_scalars[8 + 4] != 0.0) { *ok = false; return HITL_OK; }
  }
  launch_spmv_step(st, s->n, npad, fv, lv, s->scale.p, s->dt.p, s->rhs.p, xcur, s->qv.p, xcand, s->step_partials.p);
  HIP_TRY(hipGetLastError());
  TRY(read_step_partials(s, nullptr, nullptr, ss));
  return HITL_OK;
}

// Banded single-GPU case, queue only: the exact step, its residual check and the trust-region bookkeeping.  The caller
// queues the linearisation at the candidate right behind it and synchronises once for both (the candidate does not
// depend on any host decision; if the step turns out invalid the speculative linearisation is simply dropped).
static int enqueue_linear_solve_banded(hitl_solver* s, const FactorView& fv, int buf, double mu, const double* xcur, double* xcand) {
  hipStream_t st = s->stream;
  const LinView lv = s->lin_view(buf);
  const CrView cr = s->cr_view();
  const double mind = s->opt.min_lm_diagonal, maxd = s->opt.max_lm_diagonal;
  const int npad = s->npad;
  const CrSource src = {lv.Hband, lv.g, s->scale.p, s->fixed.p, mu, mind, maxd, s->n, fv.band_poses, s->rhs.p, nullptr, 0};
  launch_cr_build_factor_solve(st, cr, src, s->wv.p, s->dt.p);
  // residual check of the factorisation and the trust-region bookkeeping in one pass over the band
  launch_spmv_residual_step(st, s->n, npad, fv, lv, s->scale.p, mu, mind, maxd, s->dt.p, s->rhs.p, xcur, s->qv.p, xcand, s->h_partials,   // straight into pinned host memory
                            fv.with_stf ? s->pose_cs.p : nullptr);
  HIP_TRY(hipGetLastError());
  return HITL_OK;
}

// Product items of the persistent PCG launch, once per correspondence set and pose band: a pose's entries (2 Wb + 1 band blocks, then
// its off-band partners) in parts of at most kPcgPartEntries = 128 (two gathers per lane) so that a pose with hundreds of partners (a
// map's hub: 270 at c3, against 32 on average) is spread over several workgroups instead of holding one wave, and every chunk that
// waits for it, 2.5 us longer than the others (measured; parts of 64 doubled the items of a map with 60 partners per pose and cost
// more than they saved there).
static int pcg_prepare_items(hitl_solver* s, const FactorView& fv) {
  hipStream_t st = s->stream;
  const int npad = s->npad;
  if (s->pcg_items_of != s->corr_generation || s->pcg_items_wb != fv.band_poses) {
    const int n_rows = npad / 3, nband = 2 * fv.band_poses + 1;
    std::vector<int> items;
    std::vector<unsigned char> parts((size_t)n_rows, 1);
    items.reserve((size_t)4 * (n_rows + 64));
    for (int i = 0; i < n_rows; ++i) {
      const int deg = (i < s->n && (int)s->h_obadj_off.size() == s->n + 1) ? s->h_obadj_off[i + 1] - s->h_obadj_off[i] : 0;
      const int total = nband + deg;
      const int np = std::min(kPcgMaxParts, (total + kPcgPartEntries - 1) / kPcgPartEntries);
      parts[i] = (unsigned char)np;
      for (int q = 0; q < np; ++q) {
        const int first = q * kPcgPartEntries, cnt = (q == np - 1) ? total - first : kPcgPartEntries;
        items.push_back(i); items.push_back(first); items.push_back(cnt); items.push_back(q);
      }
    }
    s->pcg_n_items = (int)(items.size() / 4);
    TRY(s->pcg_items.upload(items, st));
    TRY(s->pcg_parts.upload(parts, st));
    HIP_TRY(hipStreamSynchronize(st));                     // (the host vectors go out of scope; once per correspondence set)
    s->pcg_items_of = s->corr_generation; s->pcg_items_wb = fv.band_poses;
  }
  return HITL_OK;
}

// Pose pairs outside the band, one GPU, queue only (round 6): band factorisation + x0 = M^-1 b, the residual, the WHOLE preconditioned CG
// as one persistent launch (hitl_pcg.hip: its scalars never leave the device, it stops by itself and leaves a four-double record in
// pinned memory), then the trust-region bookkeeping and the candidate.  As in the banded case the caller queues the candidate's
// linearisation right behind and waits ONCE per LM iteration (rounds 2-5: three waits + one per four CG iterations).
static int enqueue_linear_solve_pcg(hitl_solver* s, const FactorView& fv, int buf, double mu, const double* xcur, double* xcand, const PcgPlan& pl, bool timed,
                                    bool refine = false /* continue from the step in s->dt (a launch whose recurrence drifted away from the true residual) */) {
  hipStream_t st = s->stream;
  const LinView lv = s->lin_view(buf);
  const CrView cr = s->cr_view();
  const double mind = s->opt.min_lm_diagonal, maxd = s->opt.max_lm_diagonal;
  const int npad = s->npad;
  const double tol = std::max(s->opt.linear_solver_tolerance, 1e-15);
  if (s->pcg_work.n < pl.work_doubles) {                  // hand-off granules: zero once, tags only ever grow afterwards
    TRY(s->pcg_work.resize(pl.work_doubles + pl.work_doubles / 4));
    TRY(s->pcg_work.zero(st));
  }
  if (s->pcg_sync.p == nullptr) { TRY(s->pcg_sync.resize(pcg_fused_sync_words())); TRY(s->pcg_sync.zero(st)); }
  if (!refine) {
    launch_build_system(st, s->n, fv.band_poses, lv, s->scale.p, s->fixed.p, mu, mind, maxd, cr, s->rhs.p, s->wv.p);
    launch_cr_factor_solve(st, cr, s->wv.p, s->dt.p);   // x0 = M^{-1} b
  }                                                     // (refine: system, factors and the step so far are where the first launch left them)
  launch_spmv_residual(st, s->n, npad, fv, lv, s->scale.p, mu, mind, maxd, s->dt.p, s->rhs.p, s->rv.p, s->step_partials.p);   // the TRUE residual of the start
  double* rec = s->h_scalars + 8;
  rec[0] = 0.0; rec[1] = 0.0; rec[2] = -1.0; rec[3] = 0.0;      // status -1: nobody has reported (the previous launch has long finished: the host waited for it)
  if (timed) HIP_TRY(hipEventRecord(s->ev4, st));
  unsigned long long* probe = nullptr;
  if (dev_switches().timing && s->pcg_launches == 0) {
    TRY(s->pcg_probe.resize((size_t)pl.G * 16));
    TRY(s->pcg_probe.zero(st));
    probe = s->pcg_probe.p;
  }
  const bool starve = dev_switches().pcg_starve;                             // test hook: no top workgroup, short waits -- the chunk workgroups give up
  const unsigned spins = starve ? 1u << 10 : 1u << 20;
  if (!launch_pcg_fused(st, pl, s->n, npad, fv, lv, cr, s->scale.p, mu, mind, maxd, tol, (int)s->opt.max_linear_iterations, s->dt.p, s->rv.p, s->step_partials.p,
                        spmv_blocks(npad), s->pcg_items.p, s->pcg_n_items, s->pcg_parts.p, s->pcg_work.p, s->pcg_sync.p, ++s->pcg_launches, rec, spins, starve, probe))
    return fail(HITL_ERR_HIP, "the persistent PCG kernel could not be launched");
  if (probe != nullptr) {                                  // HITL_TIMING, first solve only: where one CG iteration's time goes, per role
    std::vector<unsigned long long> h((size_t)pl.G * 16);
    HIP_TRY(hipMemcpyAsync(h.data(), probe, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    auto us = [&](int wg, int a_, int b_) { return (h[(size_t)wg * 16 + b_] >= h[(size_t)wg * 16 + a_] && h[(size_t)wg * 16 + a_] != 0) ? (double)(h[(size_t)wg * 16 + b_] - h[(size_t)wg * 16 + a_]) * 0.01 : -1.0; };
    const unsigned long long t00 = h[0];
    auto at = [&](int wg, int slot) { return h[(size_t)wg * 16 + slot] ? ((double)h[(size_t)wg * 16 + slot] - (double)t00) * 0.01 : -1.0; };
    std::fprintf(stderr, "k_pcg_fused probe (CG iteration 2; M %d, chunks %d, G %d), us relative to chunk 0's start of the iteration:\n", pl.M, pl.chunks, pl.G);
    for (int wg : {0, pl.chunks / 2, pl.chunks - 1})
      std::fprintf(stderr, "  chunk %3d: start %.2f | fwd done + ctr out %.2f | top's z taken %.2f | back + z out %.2f | r.r check %.2f | w, delta taken %.2f | end %.2f  (iteration %.2f us)\n",
                   wg, at(wg, 0), at(wg, 1), at(wg, 2), at(wg, 3), at(wg, 4), at(wg, 6), at(wg, 7), us(wg, 0, 7));
    { const int wg = pl.G - 1;
      std::fprintf(stderr, "  top      : start %.2f | ctr taken %.2f | solved %.2f | z out %.2f | r.r check %.2f | end %.2f\n", at(wg, 0), at(wg, 1), at(wg, 2), at(wg, 3), at(wg, 4), at(wg, 7)); }
    for (int wg : {pl.chunks, (pl.chunks + pl.G - 1) / 2, pl.G - 2})
      if (wg >= pl.chunks && wg < pl.G - 1)
        std::fprintf(stderr, "  products %3d: start %.2f | r.r check %.2f | products out %.2f\n", wg, at(wg, 0), at(wg, 4), at(wg, 5));
    std::fprintf(stderr, "  chunk 0 in detail: updates + r.r done %.2f | forward levels done %.2f | ctr out %.2f ... z taken %.2f | back levels done %.2f | z, r.z, r.r out %.2f\n",
                 at(0, 9), at(0, 10), at(0, 1), at(0, 2), at(0, 11), at(0, 3));
    { double worst = -1e9, best = 1e9; int wwg = -1;
      for (int wg = pl.chunks; wg < pl.G - 1; ++wg) { const double t = at(wg, 5); if (t > worst) { worst = t; wwg = wg; } if (t < best) best = t; }
      std::fprintf(stderr, "  products out over all %d product workgroups: first %.2f, last %.2f (workgroup %d)\n", pl.G - 1 - pl.chunks, best, worst, wwg); }
    { const double cyc = (double)(h[8 + 7] - h[8 + 0]), t_us = us(0, 0, 7);
      std::fprintf(stderr, "  shader clock over chunk 0's iteration: %.0f cycles in %.2f us = %.0f MHz\n", cyc, t_us, t_us > 0 ? cyc / t_us : 0.0); }
    s->pcg_probe.release();
  }
  if (timed) HIP_TRY(hipEventRecord(s->ev5, st));
  // the step's bookkeeping AND the true residual b - A dt of what the launch returned (one pass over the matrix: the banded flow's
  // kernel; dt is exactly 0 on constant poses): the recurrence the launch iterates on (r -= alpha s, s = w + beta s) drifts away from
  // it on ill-conditioned systems, and the host holds the solve to the TRUE residual
  launch_spmv_residual_step(st, s->n, npad, fv, lv, s->scale.p, mu, mind, maxd, s->dt.p, s->rhs.p, xcur, s->qv.p, xcand, s->h_partials,
                            fv.with_stf ? s->pose_cs.p : nullptr);
  HIP_TRY(hipGetLastError());
  return HITL_OK;
}

// ---------------------------------------------------------------------------------------
// distributed PCG pieces (submap ranks holding pose pairs outside the band)
// ---------------------------------------------------------------------------------------
// One exchange: every rank contributes `nhead` scalars (device memory) and the owned part of `vec`; afterwards the halo
// rows of `vec` hold their owners' values and heads[r * nhead + k] the scalars of rank r.
static int dist_gather_owned(hitl_solver* s, int nhead, const double* head_dev, double* vec, std::vector<double>* heads) {
  hipStream_t st = s->stream;
  // the record: `nhead` scalars + the owned entries a halo can need (the first and the last pcg_edge owned poses; everything
  // when the range is no longer than two edges)
  const int W = s->world, own = s->own_hi - s->own_lo, rec_poses = std::min(s->maxown, 2 * s->pcg_edge);
  const int maxcnt = 3 * rec_poses, R = nhead + maxcnt;
  launch_pack_edges(st, nhead, head_dev, vec, 3 * s->own_lo, 3 * own, 3 * s->pcg_edge, maxcnt, s->pcg_rec.p);
  s->pcg_all.resize((size_t)R * W);
  TRY(comm_allgather_dev(s, s->pcg_rec.p, s->pcg_all.data(), (int64_t)sizeof(double) * R, s->pcg_stage));
  if (heads) {
    heads->resize((size_t)nhead * W);
    for (int r = 0; r < W; ++r) for (int k = 0; k < nhead; ++k) (*heads)[(size_t)nhead * r + k] = s->pcg_all[(size_t)R * r + k];
  }
  if (s->n_halo > 0) {
    s->pcg_hval.resize((size_t)s->n_halo);
    for (int k = 0; k < s->n_halo; ++k) {
      const int slot = s->halo_owner_slot[k], r = slot / maxcnt, o = slot % maxcnt;
      s->pcg_hval[k] = s->pcg_all[(size_t)R * r + nhead + o];
    }
    HIP_TRY(hipMemcpyAsync(s->halo_val.p, s->pcg_hval.data(), sizeof(double) * s->n_halo, hipMemcpyHostToDevice, st));
    launch_scatter_idx(st, s->n_halo, s->halo_idx.p, s->halo_val.p, vec);
  }
  return HITL_OK;
}

// (S H S + D^2) dt = rhs over ALL ranks by preconditioned CG; the preconditioner is block-Jacobi: every rank's own banded
// block (halo poses constant), factorised by cyclic reduction as in the single-GPU solve.  Products use the halo entries of
// p (filled from the owners by the exchange that also carries r.z, r.r); p.q takes a second, scalar exchange.  Sums run
// over the ranks in rank order on every rank: all ranks take identical decisions.  dt returns with its halo rows filled.
static int dist_pcg_solve(hitl_solver* s, const FactorView& fv, const LinView& lv, const CrView& cr, double mu, int* iters, bool* ok) {
  hipStream_t st = s->stream;
  const double mind = s->opt.min_lm_diagonal, maxd = s->opt.max_lm_diagonal;
  const int npad = s->npad, n = s->n, W = s->world;
  const int off = 3 * s->own_lo, cnt = 3 * (s->own_hi - s->own_lo);
  const double tol = std::max(s->opt.linear_solver_tolerance, 1e-15);
  double* x = s->dt.p;
  double* head = s->pcg_head.p;
  HIP_TRY(hipMemsetAsync(x, 0, sizeof(double) * npad, st));
  HIP_TRY(hipMemsetAsync(s->pv.p, 0, sizeof(double) * npad, st));
  launch_copy(st, npad, s->rhs.p, s->rv.p);
  std::vector<double> heads;
  double rz_old = 0.0, target = 0.0;
  *iters = 0;
  for (int it = 0;; ++it) {
    if (it == 0) { launch_copy(st, npad, s->rv.p, s->wv.p); launch_cr_factor_solve(st, cr, s->wv.p, s->zv.p); }   // factorise + z = M^-1 r
    else launch_cr_solve(st, cr, s->rv.p, s->wv.p, s->zv.p);
    {
      const double* a[4] = {s->rv.p + off, s->rv.p + off, s->rhs.p + off, s->rv.p + off};
      const double* b[4] = {s->zv.p + off, s->rv.p + off, s->rhs.p + off, s->rv.p + off};
      launch_dots(st, cnt, 3, a, b, head);                       // r.z, r.r, b.b over the owned rows
    }
    TRY(dist_gather_owned(s, 3, head, s->zv.p, &heads));
    double rz = 0.0, rr = 0.0, bb = 0.0;
    for (int r = 0; r < W; ++r) { rz += heads[(size_t)3 * r]; rr += heads[(size_t)3 * r + 1]; bb += heads[(size_t)3 * r + 2]; }
    if (it == 0) target = tol * tol * bb;
    if (!(rr == rr) || !(rz == rz)) { *ok = false; break; }
    if (rr <= target || it >= s->opt.max_linear_iterations) break;
    const double beta = (it == 0 || rz_old == 0.0) ? 0.0 : rz / rz_old;
    launch_axpby(st, npad, 1.0, s->zv.p, beta, s->pv.p);         // p = z + beta p  (halo rows: the owners' z, this rank's copy of their p)
    launch_spmv(st, n, npad, fv, lv, s->scale.p, mu, mind, maxd, s->pv.p, s->qv.p, /*all_cols=*/true);
    {
      const double* a[4] = {s->pv.p + off, s->pv.p + off, s->pv.p + off, s->pv.p + off};
      const double* b[4] = {s->qv.p + off, s->qv.p + off, s->qv.p + off, s->qv.p + off};
      launch_dots(st, cnt, 1, a, b, head);
    }
    double mine = 0.0;
    HIP_TRY(hipMemcpyAsync(&mine, head, sizeof(double), hipMemcpyDeviceToHost, st));
    TRY(spin_sync(s));
    std::vector<double> pqs((size_t)W);
    TRY(comm_allgather(s, &mine, pqs.data(), sizeof(double)));
    double pq = 0.0;
    for (int r = 0; r < W; ++r) pq += pqs[r];
    if (!(pq > 0.0)) { if (rz != 0.0) *ok = false; break; }
    const double alpha = rz / pq;
    launch_axpby(st, npad, alpha, s->pv.p, 1.0, x);              // halo rows of x follow their owners' x: same alpha, same p
    launch_axpby(st, npad, -alpha, s->qv.p, 1.0, s->rv.p);
    rz_old = rz;
    ++*iters;
  }
  return HITL_OK;
}

// Submap version.  Rank g's unknowns x_g (owned poses) satisfy
//     M_g x_g + E_F C_lo,g v_{g-1} + E_L C_up,g u_{g+1} = b_g
// with M_g its own banded block (halo poses are constant locally), v_{g-1} / u_{g+1} the steps of
// the previous rank's last Wi poses / the next rank's first Wi poses, and C_lo, C_up the cross-cut
// Hessian blocks (assembled locally: every rank evaluates all factors touching its owned poses).
// Each rank solves M_g Z = [b_g, E_F, E_L] in ONE multi-rhs cyclic-reduction pass, the ranks
// all-gather the 2m interface rows (m = 3 Wi) of Z, their Jacobi scales and C_lo / C_up, and every
// rank solves the same 2m*world dense interface system on the host:
//     y_g + ZF_g (C~_lo,g v_{g-1}) + ZL_g (C~_up,g u_{g+1}) = z0_g ,   y_g = (u_g, v_g).
// The coupling is eliminated exactly; no Schwarz / extra PCG iterations.
static int linear_solve_and_step_dist(hitl_solver* s, const FactorView& fv, int buf, double mu, const double* xcur,
                                      double* xcand, int* iters, bool* ok, StepScalars* ss, LinResult* spec) {
  hipStream_t st = s->stream;
  const LinView lv = s->lin_view(buf);
  const CrView cr = s->cr_view();
  const double mind = s->opt.min_lm_diagonal, maxd = s->opt.max_lm_diagonal;
  const int npad = s->npad, n = s->n, W = s->world, me = s->rank;
  const int Wb = fv.band_poses, Wi = s->Wi, m = 3 * Wi, nc = 2 * m;
  *iters = 1;
  *ok = true;
  if (s->dist_pcg) launch_build_system(st, n, Wb, lv, s->scale.p, s->fixed.p, mu, mind, maxd, cr, s->rhs.p, nullptr);
  int nh = 0;
  std::vector<double> coef, hval, hsc;   // uploaded asynchronously below: alive until the synchronisation at the end
  if (s->dist_pcg) {
    TRY(dist_pcg_solve(s, fv, lv, cr, mu, iters, ok));
    nh = s->n_halo;                      // dt and scale already hold the owners' values on every halo row
  } else if (s->dist_sep) {
    // Separator form (round 4): the rank's first and last owned super-blocks F and R stay, the fused stages eliminate everything
    // between them -- ONE right-hand side, the same kernels and cost as the single-GPU factorisation whatever the interface
    // width -- and leave the Schur complement onto the two.  One all-gather of a 4 B^2 + 4 B record; every rank then solves the
    // same block-tridiagonal chain of the 2 * world separators ON THE DEVICE (the tiny top stage of the same cyclic reduction)
    // and back-substitutes between its own two.  The neighbours' separators cover this rank's halo poses: their steps and
    // scales come out of the same solve.  With RCCL nothing here waits for the host.
    const int lo = s->own_lo;
    const CrView chain = s->sep_chain_view();
    const CrSource src = {lv.Hband + (size_t)lo * (Wb + 1) * 9, lv.g + 3 * (size_t)lo, s->scale.p + 3 * (size_t)lo, s->fixed.p + lo, mu, mind, maxd,
                          s->own_hi - lo, Wb, s->rhs.p + 3 * (size_t)lo, nullptr, 0};
    double* w = s->wv.p + 3 * (size_t)lo;
    double* x = s->dt.p + 3 * (size_t)lo;
    launch_cr_sep_forward(st, chain, src, w, x);
    launch_sep_record(st, chain, w, lv.Hband, s->scale.p, s->fixed.p, s->sep_geom, s->sep_rec.p);
    HIP_TRY(hipGetLastError());
    TRY(comm_allgather_d2d(s, s->sep_rec.p, s->sep_all.p, (int64_t)sizeof(double) * sep_rec_len(s->B), s->sep_host));
    launch_sep_solve(st, s->sep_red_view(), s->sep_all.p, s->sep_w2.p, s->sep_x2.p, s->sep_geom, s->dt.p, s->scale.p);
    launch_cr_sep_backward(st, chain, w, x);
    nh = s->n_halo;
  } else {
  {
    // build + factorise + solve for b and the interface unit vectors in one pass: z0 and Z = M^{-1} [E_F, E_L]; the first
    // stage cuts its tiles and all 1 + nc right-hand sides itself (no k_build_system, no unit-vector fill)
    const CrSource src = {lv.Hband, lv.g, s->scale.p, s->fixed.p, mu, mind, maxd, n, Wb, s->rhs.p, s->if_cols.p, nc};
    launch_cr_build_factor_solve_multi(st, cr, src, s->rhsN.p, s->zN.p, (size_t)npad);
  }
  const double* z0 = s->zN.p;
  const double* Z = s->zN.p + npad;
  const int RI = nc * (nc + 2), RC = m * m, R = RI + 2 * RC;
  // the exchange record (interface rows of z0 / Z / scale, then the two cross-cut blocks) is put together on the device
  launch_gather_iface(st, z0, Z, (size_t)npad, s->scale.p, s->if_cols.p, nc, s->iface_dev.p);
  launch_iface_blocks(st, lv.Hband, Wb, Wi, s->own_lo, s->own_hi, me > 0, me < W - 1, s->iface_dev.p + RI, s->iface_dev.p + RI + RC);
  // host buffers of the exchange live in the solver (sized once per interface width, reused every LM iteration)
  std::vector<double>&rec = s->if_rec, &all = s->if_all, &A = s->if_A, &y = s->if_y, &Ct = s->if_Ct;
  all.resize((size_t)R * W);
  TRY(comm_allgather_dev(s, s->iface_dev.p, all.data(), (int64_t)sizeof(double) * R, rec));
  auto z0_of = [&](int g) { return &all[(size_t)R * g]; };
  auto Z_of = [&](int g) { return &all[(size_t)R * g + nc]; };              // Z[c][r] = Z_of[nc*c + r]
  auto sc_of = [&](int g) { return &all[(size_t)R * g + nc + nc * nc]; };   // [0,m): first Wi poses, [m,2m): last Wi
  auto Clo_of = [&](int g) { return &all[(size_t)R * g + RI]; };
  auto Cup_of = [&](int g) { return &all[(size_t)R * g + RI + RC]; };
  // interface system
  const int NI = nc * W;
  A.assign((size_t)NI * NI, 0.0); y.assign(NI, 0.0); Ct.assign((size_t)m * m, 0.0);
  for (int g = 0; g < W; ++g) {
    for (int r = 0; r < nc; ++r) { A[(size_t)(nc * g + r) * NI + nc * g + r] = 1.0; y[nc * g + r] = z0_of(g)[r]; }
    if (g >= 1) {       // C~_lo,g = diag(s_first,g) C_lo,g diag(s_last,g-1); columns of v_{g-1}
      for (int a = 0; a < m; ++a) for (int k = 0; k < m; ++k) Ct[(size_t)a * m + k] = sc_of(g)[a] * Clo_of(g)[(size_t)a * m + k] * sc_of(g - 1)[m + k];
      for (int r = 0; r < nc; ++r)
        for (int k = 0; k < m; ++k) {
          double v = 0.0;
          for (int c = 0; c < m; ++c) v += Z_of(g)[(size_t)nc * c + r] * Ct[(size_t)c * m + k];
          A[(size_t)(nc * g + r) * NI + nc * (g - 1) + m + k] += v;
        }
    }
    if (g + 1 < W) {    // C~_up,g = diag(s_last,g) C_up,g diag(s_first,g+1); columns of u_{g+1}
      for (int a = 0; a < m; ++a) for (int k = 0; k < m; ++k) Ct[(size_t)a * m + k] = sc_of(g)[m + a] * Cup_of(g)[(size_t)a * m + k] * sc_of(g + 1)[k];
      for (int r = 0; r < nc; ++r)
        for (int k = 0; k < m; ++k) {
          double v = 0.0;
          for (int c = 0; c < m; ++c) v += Z_of(g)[(size_t)nc * (m + c) + r] * Ct[(size_t)c * m + k];
          A[(size_t)(nc * g + r) * NI + nc * (g + 1) + k] += v;
        }
    }
  }
  if (!solve_dense(NI, A, y)) *ok = false;
  // this rank's correction coefficients, and the neighbours' steps / scales for its halo poses
  coef.assign(nc, 0.0);
  if (me >= 1) {
    const double* vprev = &y[nc * (me - 1) + m];
    for (int a = 0; a < m; ++a)
      for (int k = 0; k < m; ++k) coef[a] += sc_of(me)[a] * Clo_of(me)[(size_t)a * m + k] * sc_of(me - 1)[m + k] * vprev[k];
    for (int k = 0; k < m; ++k) { hval.push_back(vprev[k]); hsc.push_back(sc_of(me - 1)[m + k]); }
  }
  if (me + 1 < W) {
    const double* unext = &y[nc * (me + 1)];
    for (int a = 0; a < m; ++a)
      for (int k = 0; k < m; ++k) coef[m + a] += sc_of(me)[m + a] * Cup_of(me)[(size_t)a * m + k] * sc_of(me + 1)[k] * unext[k];
    for (int k = 0; k < m; ++k) { hval.push_back(unext[k]); hsc.push_back(sc_of(me + 1)[k]); }
  }
  nh = (int)hval.size();
  coef.resize((size_t)3 * nc, 0.0);                      // one upload: coefficients | halo steps | halo scales
  std::copy(hval.begin(), hval.end(), coef.begin() + nc);
  std::copy(hsc.begin(), hsc.end(), coef.begin() + 2 * nc);
  HIP_TRY(hipMemcpyAsync(s->coef_dev.p, coef.data(), sizeof(double) * 3 * nc, hipMemcpyHostToDevice, st));
  launch_combine(st, npad, z0, Z, (size_t)npad, s->coef_dev.p, nc, s->dt.p);
  launch_scatter_halo(st, nh, s->halo_idx.p, s->coef_dev.p + nc, s->coef_dev.p + 2 * nc, s->dt.p, s->scale.p);
  }
  // model-cost pieces over the owned rows (halo columns carry the neighbours' steps), candidate state
  double2* cs_out = fv.with_stf ? s->pose_cs.p : nullptr;      // the trig table of the candidate, written by the two kernels that form it
  // With the built-in RCCL transport the whole exchange stays on the device (round 4): the per-workgroup partials are added up
  // by a kernel (the host's order), the 7 scalars go device -> device through ncclAllGather, one copy brings all ranks' scalars
  // to the host and the host waits ONCE.  A caller callback (host buffers) keeps the older flow: partials straight into pinned
  // memory, one wait, host sums, callback.
  const bool on_device = s->nccl != nullptr && !s->comm.allgather;
  double* step_out = on_device ? s->step_partials.p : s->h_partials;
  launch_spmv_step(st, n, npad, fv, lv, s->scale.p, s->dt.p, s->rhs.p, xcur, s->qv.p, xcand, step_out, cs_out);
  launch_halo_candidate(st, nh, s->halo_idx.p, xcur, s->scale.p, s->dt.p, xcand, cs_out);
  HIP_TRY(hipGetLastError());
  // The candidate does not depend on any host decision, so its linearisation is queued right here and its two scalars
  // ride in the same all-gather as the step scalars (exchanges 3 and 1 merged: 7 doubles per rank).
  if (spec) TRY(enqueue_linearize(s, fv, xcand, buf ^ 1, /*direct=*/!on_device, /*have_trig=*/true, /*device_only=*/on_device));
  std::vector<double> all2((size_t)7 * W);
  LinResult lloc = {0.0, 0.0, 0.f, 0};
  if (on_device) {
    TRY(s->comm_send.resize(sizeof(double) * 8)); TRY(s->comm_recv.resize(sizeof(double) * 7 * (size_t)W));
    double* rec7 = reinterpret_cast<double*>(s->comm_send.p);
    launch_sum_partials(st, s->step_partials.p, spmv_blocks(npad), spec ? s->lin_view(buf ^ 1).partials : nullptr, assemble_blocks(n), rec7);
    const ncclResult_t r = ncclAllGather(rec7, s->comm_recv.p, 7 * sizeof(double), ncclChar, s->nccl, st);
    if (r != ncclSuccess) return fail(HITL_ERR_COMM, std::string("ncclAllGather: ") + ncclGetErrorString(r));
    double* host_all = s->h_partials;                      // pinned; nothing else lands there on this path
    HIP_TRY(hipMemcpyAsync(host_all, s->comm_recv.p, sizeof(double) * 7 * (size_t)W, hipMemcpyDeviceToHost, st));
    TRY(spin_sync(s));                   // host vectors above stay alive until here
    std::memcpy(all2.data(), host_all, sizeof(double) * 7 * (size_t)W);
    if (spec) { lloc.timed = s->lin_timed ? 1 : 0; if (s->lin_timed) HIP_TRY(hipEventElapsedTime(&lloc.stf_ms, s->ev0, s->ev1)); }
  } else {
    TRY(spin_sync(s));                     // host vectors above stay alive until here
    StepScalars loc;
    TRY(finish_step_partials(s, nullptr, nullptr, &loc));
    if (dev_switches().poison_step_from > 0 && s->lm_iter >= dev_switches().poison_step_from) loc.bad += 1.0;   // test hook: this rank's step "is not finite"
    if (spec) TRY(finish_linearize(s, fv, &lloc));
    double r2[7] = {loc.dt_rhs, loc.dt_q, loc.step2, loc.xn2, loc.bad, lloc.cost, lloc.gmax};
    TRY(comm_allgather(s, r2, all2.data(), sizeof(r2)));
  }
  ss->dt_rhs = ss->dt_q = ss->step2 = ss->xn2 = ss->bad = 0.0;
  double cost = 0.0, gmax = 0.0;
  for (int g = 0; g < W; ++g) {          // rank order on every rank: identical sums everywhere
    ss->dt_rhs += all2[(size_t)7 * g]; ss->dt_q += all2[(size_t)7 * g + 1]; ss->step2 += all2[(size_t)7 * g + 2];
    ss->xn2 += all2[(size_t)7 * g + 3]; ss->bad += all2[(size_t)7 * g + 4];
    cost += all2[(size_t)7 * g + 5];
    gmax = nan_max(gmax, all2[(size_t)7 * g + 6]);
  }
  if (spec) { spec->cost = cost; spec->gmax = gmax; spec->stf_ms = lloc.stf_ms; spec->timed = lloc.timed; }
  return HITL_OK;
}

// ---------------------------------------------------------------------------------------
// Pose covariances on submap ranks (collective).  With only rank g's right-hand side non-zero, the interface solve of
// linear_solve_and_step_dist gives  x_g = M_g^-1 e - Z_g (B_g A^-1 P_g) Z_g^T e  (A: the interface matrix, P_g: rank g's
// slot, B_g: the map from the interface unknowns to g's correction coefficients), i.e. the diagonal block of the global
// inverse over g's poses is  M_g^-1 - Z_g T_g Z_g^T  with the nc x nc matrix T_g = B_g A^-1 P_g.  The caller has linearised
// into buffer `buf` and Jacobi-scaled; this factorises (with the interface columns riding along), runs the local selected
// inversion into Sd / So and subtracts the correction from the diagonal tiles.
// ---------------------------------------------------------------------------------------
int dist_cov_correction(hitl_solver* s, const FactorView& fv, int buf, double* Sd, double* So) {
  hipStream_t st = s->stream;
  const LinView lv = s->lin_view(buf);
  const CrView cr = s->cr_view();
  const int npad = s->npad, n = s->n, W = s->world, me = s->rank;
  const int Wb = fv.band_poses, Wi = s->Wi, m = 3 * Wi, nc = 2 * m;
  if (s->dist_pcg) return fail(HITL_ERR_UNSUPPORTED, "pose_covariances: pose pairs outside the band (raise max_band_poses)");
  // J^T J itself (no trust-region diagonal); right-hand side 0 in slot 0, the interface unit vectors behind it
  launch_build_system(st, n, Wb, lv, s->scale.p, s->fixed.p, INFINITY, s->opt.min_lm_diagonal, s->opt.max_lm_diagonal, cr, s->rhs.p, nullptr);
  HIP_TRY(hipMemsetAsync(s->rhsN.p, 0, sizeof(double) * npad, st));
  launch_unit_rhs(st, npad, s->rhsN.p + npad, (size_t)npad, s->if_cols.p, nc);
  launch_cr_factor_solve(st, cr, s->rhsN.p, s->zN.p, 1 + nc, (size_t)npad);
  launch_cr_selinv(st, cr, Sd, So);
  const double* z0 = s->zN.p;
  const double* Z = s->zN.p + npad;
  const int RI = nc * (nc + 2), RC = m * m, R = RI + 2 * RC;
  launch_gather_iface(st, z0, Z, (size_t)npad, s->scale.p, s->if_cols.p, nc, s->iface_dev.p);
  launch_iface_blocks(st, lv.Hband, Wb, Wi, s->own_lo, s->own_hi, me > 0, me < W - 1, s->iface_dev.p + RI, s->iface_dev.p + RI + RC);
  std::vector<double>&rec = s->if_rec, &all = s->if_all;
  all.resize((size_t)R * W);
  TRY(comm_allgather_dev(s, s->iface_dev.p, all.data(), (int64_t)sizeof(double) * R, rec));
  auto Z_of = [&](int g) { return &all[(size_t)R * g + nc]; };
  auto sc_of = [&](int g) { return &all[(size_t)R * g + nc + nc * nc]; };
  auto Clo_of = [&](int g) { return &all[(size_t)R * g + RI]; };
  auto Cup_of = [&](int g) { return &all[(size_t)R * g + RI + RC]; };
  const int NI = nc * W;
  std::vector<double> A((size_t)NI * NI, 0.0), Ct((size_t)m * m), X((size_t)NI * nc, 0.0);
  for (int g = 0; g < W; ++g) {                       // the interface matrix of linear_solve_and_step_dist
    for (int r = 0; r < nc; ++r) A[(size_t)(nc * g + r) * NI + nc * g + r] = 1.0;
    if (g >= 1) {
      for (int a = 0; a < m; ++a) for (int k = 0; k < m; ++k) Ct[(size_t)a * m + k] = sc_of(g)[a] * Clo_of(g)[(size_t)a * m + k] * sc_of(g - 1)[m + k];
      for (int r = 0; r < nc; ++r)
        for (int k = 0; k < m; ++k) {
          double v = 0.0;
          for (int c = 0; c < m; ++c) v += Z_of(g)[(size_t)nc * c + r] * Ct[(size_t)c * m + k];
          A[(size_t)(nc * g + r) * NI + nc * (g - 1) + m + k] += v;
        }
    }
    if (g + 1 < W) {
      for (int a = 0; a < m; ++a) for (int k = 0; k < m; ++k) Ct[(size_t)a * m + k] = sc_of(g)[m + a] * Cup_of(g)[(size_t)a * m + k] * sc_of(g + 1)[k];
      for (int r = 0; r < nc; ++r)
        for (int k = 0; k < m; ++k) {
          double v = 0.0;
          for (int c = 0; c < m; ++c) v += Z_of(g)[(size_t)nc * (m + c) + r] * Ct[(size_t)c * m + k];
          A[(size_t)(nc * g + r) * NI + nc * (g + 1) + k] += v;
        }
    }
  }
  // X = A^-1 P_me (NI x nc), one column at a time on a copy of A (NI is a few dozen)
  std::vector<double> T((size_t)nc * nc, 0.0), Acopy, y;
  for (int c = 0; c < nc; ++c) {
    Acopy = A;
    y.assign(NI, 0.0);
    y[nc * me + c] = 1.0;
    if (!solve_dense(NI, Acopy, y)) return fail(HITL_ERR_SOLVER_FAILURE, "pose_covariances: the interface system is singular");
    // this rank's correction coefficients for that right-hand side = column c of T
    if (me >= 1) {
      const double* vprev = &y[nc * (me - 1) + m];
      for (int a = 0; a < m; ++a)
        for (int k = 0; k < m; ++k) T[(size_t)a * nc + c] += sc_of(me)[a] * Clo_of(me)[(size_t)a * m + k] * sc_of(me - 1)[m + k] * vprev[k];
    }
    if (me + 1 < W) {
      const double* unext = &y[nc * (me + 1)];
      for (int a = 0; a < m; ++a)
        for (int k = 0; k < m; ++k) T[(size_t)(m + a) * nc + c] += sc_of(me)[m + a] * Cup_of(me)[(size_t)a * m + k] * sc_of(me + 1)[k] * unext[k];
    }
  }
  DevArray<double> d_T;
  TRY(d_T.upload(T, st));
  launch_cov_iface_correction(st, s->own_lo, s->own_hi, Wb, cr.B, Z, (size_t)npad, d_T.p, nc, Sd);
  HIP_TRY(hipStreamSynchronize(st));
  return HITL_OK;
}

// The same in separator form (ranks whose linear solves take it): no interface columns at all.  The rank's chain is
// factorised between its two separators (J^T J itself: no trust-region diagonal), the separator chain of all ranks is built
// from the gathered records, factorised and selectively inverted -- that yields the blocks of the GLOBAL inverse at this rank's
// F and R and between them -- and the rank's own tree is walked down from those.  Sd / So are indexed by CHAIN node (node k
// = owned poses own_lo + k * Wb ..).
int dist_cov_sep(hitl_solver* s, const FactorView& fv, int buf, double* Sd, double* So) {
  hipStream_t st = s->stream;
  const LinView lv = s->lin_view(buf);
  const int Wb = fv.band_poses, lo = s->own_lo;
  const CrView chain = s->sep_chain_view();
  const CrView red = s->sep_red_view();
  const CrSource src = {lv.Hband + (size_t)lo * (Wb + 1) * 9, lv.g + 3 * (size_t)lo, s->scale.p + 3 * (size_t)lo, s->fixed.p + lo, INFINITY,
                        s->opt.min_lm_diagonal, s->opt.max_lm_diagonal, s->own_hi - lo, Wb, s->rhs.p + 3 * (size_t)lo, nullptr, 0};
  double* w = s->wv.p + 3 * (size_t)lo;
  double* x = s->dt.p + 3 * (size_t)lo;
  launch_cr_sep_forward(st, chain, src, w, x);
  launch_sep_record(st, chain, w, lv.Hband, s->scale.p, s->fixed.p, s->sep_geom, s->sep_rec.p);
  HIP_TRY(hipGetLastError());
  TRY(comm_allgather_d2d(s, s->sep_rec.p, s->sep_all.p, (int64_t)sizeof(double) * sep_rec_len(s->B), s->sep_host));
  DevArray<double> Sd2, So2;
  const size_t bb2 = (size_t)red.nsb * red.B * red.B;
  TRY(Sd2.resize(bb2)); TRY(So2.resize(bb2));
  launch_sep_selinv(st, red, s->sep_all.p, s->sep_w2.p, s->sep_x2.p, s->sep_geom, chain, Sd2.p, So2.p, Sd, So);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));       // the scratch tiles go out of scope here; the step buffers (dt, wv, rhs) were only borrowed
  HIP_TRY(hipMemsetAsync(s->dt.p, 0, sizeof(double) * s->npad, st));
  return HITL_OK;
}

// hitl_summary.hbm_bytes: what one linearisation / one linear solve / one band pass streams through device memory, from the sizes
// of the arrays each launch reads and writes once (this layout's compulsory bytes; DESIGN.md section 4 lists them per kernel)
struct ByteModel { double linearize, solve, step; };
static ByteModel byte_model(const hitl_solver* s, const FactorView& fv) {
  const double n = s->n, P = fv.with_stf ? (double)s->n_pairs : 0.0, C = fv.with_stf ? (double)s->n_matches : 0.0;
  const double O = fv.with_stf ? (double)s->n_obs : 0.0, Wb = fv.band_poses, B = 3.0 * Wb, nsb = std::ceil(n / Wb);
  ByteModel m;
  const double stf = s->opt.stf_moments ? (double)kPairMomStride * 8.0 * P + 112.0 * P + 40.0 * n
                                        : 16.0 * O + 4.0 * C + 24.0 * n + 112.0 * P + 128.0 * n;       // k_linearize_stf_r (P > 0 only)
  const double assemble = 224.0 * P + 8.0 * P + 8.0 * Wb * n       // pair rows read at both ends + band blocks, incidence + band tables
                        + 72.0 * (Wb + 1.0) * n + 24.0 * n + 24.0 * n + 24.0 * n + 16.0 * n;   // Hband + g out, state, odometry tables in
  m.linearize = (P > 0 ? stf : 0.0) + assemble;
  m.solve = 72.0 * (Wb + 1.0) * n + 48.0 * n + 2.0 * 3.0 * 8.0 * B * B * nsb + 4.0 * 24.0 * n;   // band rows in, Dinv / Gl / Gr out and back in, vectors
  m.step = 72.0 * (Wb + 1.0) * n + 10.0 * 24.0 * n;
  return m;
}

static double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------
// Levenberg-Marquardt with Ceres trust-region semantics (DESIGN.md "LM")
// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// The Levenberg-Marquardt loop with the decisions on the device (single GPU, every pose pair inside the band -- the path
// the headline numbers run).  The host queues whole iterations -- banded solve on the CURRENT buffers with the CURRENT
// trust region, band pass + candidate state, the candidate's linearisation with the decision in its last workgroup -- up to DevSwitches::lm_ahead (3) iterations ahead
// of the one the device has finished, and polls one pinned word.  No kernel argument depends on a decision: which buffers
// are current, the radius, and whether the loop is over are read from LmDev on the device.  Launches queued behind the
// terminating iteration return at once.  Same statements, same sums in the same order as the host loop below: identical
// iteration sequences and bit-identical poses (tests/test_gpu_parity.py A/B).
// OPT-IN (test hook device_lm).  Measured on MI355X: rounds 3-4 (a k_lm_decide launch per iteration: 12 us, later 10) 4-8 us per
// iteration SLOWER than the host-driven loop on every workload; round 5 (the decision in k_assemble's last workgroup, no launch of its
// own) 424 against 420 us at C4 and 93.5 against 95.1 us at C2 with the one-launch solve tail: every kernel of this loop starts with a
// dependent scalar load of the loop state, which costs what the host's turn-around costs (profiles/r05_linear_solve.md).
// ---------------------------------------------------------------------------------------
constexpr int kLmEventPairs = 32;

static bool lm_device_loop_applies(hitl_solver* s, const FactorView& fv) {
  if (!dev_switches().device_lm || s->dist_mode() || fv.n_offband != 0 || s->opt.timing_level >= 2) return false;
  const LinView lv0 = s->lin_view(0), lv1 = s->lin_view(1);
  CrView cr = s->cr_view();
  cr.lm = s->lm_dev.p;
  const CrSource src = {lv0.Hband, lv0.g, s->scale.p, s->fixed.p, 1.0, s->opt.min_lm_diagonal, s->opt.max_lm_diagonal, s->n, fv.band_poses, s->rhs.p,
                        nullptr, 0, s->lm_dev.p, lv1.Hband, lv1.g};
  return cr_build_factor_solve_is_fused(cr, src, s->wv.p, s->dt.p);
}

// cur: the buffer holding the initial linearisation; on return *xfinal points at the state the loop ended in
static int lm_device_loop(hitl_solver* s, const FactorView& fv, int cur, double cost, double gmax, hitl_summary* S, double** xfinal,
                          double* radius_out, int* term_out) {
  hipStream_t st = s->stream;
  const hitl_options& o = s->opt;
  const MapView mv = s->map_view();
  const PairView pv = s->pair_view();
  const LinView lv0 = s->lin_view(0), lv1 = s->lin_view(1);
  LmDev init;
  std::memset(&init, 0, sizeof(init));
  init.cur = cur; init.radius = o.initial_trust_region_radius; init.decrease_factor = 2.0; init.cost = cost; init.gmax = gmax;
  init.term = HITL_NO_CONVERGENCE; init.lin_calls = 0;
  HIP_TRY(hipMemcpyAsync(s->lm_dev.p, &init, sizeof(init), hipMemcpyHostToDevice, st));
  volatile double* prog = s->h_scalars + kNumScalars;
  for (int k = 0; k < 8; ++k) prog[k] = 0.0;
  const double guard = std::max(std::max(o.linear_solver_tolerance, 1e-15), 1e-9);
  const LmOpts lo = {o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance, o.max_trust_region_radius, o.min_trust_region_radius,
                     o.min_relative_decrease, guard * guard, (int)o.max_num_iterations, o.tolerances_after_accept != 0 ? 1 : 0};
  CrView cr = s->cr_view();
  cr.lm = s->lm_dev.p;
  const CrSource src = {lv0.Hband, lv0.g, s->scale.p, s->fixed.p, 1.0, o.min_lm_diagonal, o.max_lm_diagonal, s->n, fv.band_poses, s->rhs.p,
                        nullptr, 0, s->lm_dev.p, lv1.Hband, lv1.g};
  const int nblk_step = spmv_blocks(s->npad), nblk_lin = assemble_blocks(s->n);
  const LmDecide decide = {s->lm_dev.p, lo, s->step_partials.p, nblk_step, lv0.partials, lv1.partials, nblk_lin, s->lm_arrivals.p, s->h_scalars + kNumScalars};
  if (s->lm_events.empty() && o.timing_level >= 1) {
    s->lm_events.assign(2 * kLmEventPairs, nullptr);
    for (hipEvent_t& e : s->lm_events) if (hipEventCreate(&e) != hipSuccess) return fail(HITL_ERR_HIP, "hipEventCreate failed");
  }
  const double t_loop0 = now_ms();
  int queued = 0, completed = 0, pairs_used = 0;
  int pair_iter[kLmEventPairs];                          // the (1-based) iteration each event pair brackets
  bool done = false;
  const int max_it = (int)o.max_num_iterations;
  // the state buffers: s->x holds the current state when cur == 0 ... the initial linearisation was taken at s->x with buffer
  // `cur`; buffer index and state index move together, so x of buffer b is (b == cur0 ? s->x : s->xc)
  double* xb[2];
  xb[cur] = s->x.p; xb[cur ^ 1] = s->xc.p;
  while (!done) {
    while (queued < max_it && queued - completed < dev_switches().lm_ahead) {
      launch_cr_build_factor_solve(st, cr, src, s->wv.p, s->dt.p);
      launch_spmv_residual_step_lm(st, s->n, s->npad, fv, lv0, lv1, s->scale.p, o.min_lm_diagonal, o.max_lm_diagonal, s->dt.p, s->rhs.p, xb[0], xb[1],
                                   s->qv.p, s->step_partials.p, fv.with_stf ? s->pose_cs.p : nullptr, s->lm_dev.p);
      const bool timed = fv.with_stf && o.timing_level >= 1 && (s->lin_launches & 3) == 0 && pairs_used < kLmEventPairs;
      ++s->lin_launches;
      if (timed) HIP_TRY(hipEventRecord(s->lm_events[2 * pairs_used], st));
      if (fv.with_stf) {
        if (o.stf_moments)
          launch_linearize_stf_moments(st, pv, s->pairM.p, xb[0], s->pose_cs.p, o.point_correlation_factor, o.laser_std_dev, lv0.pairRJ, s->lm_dev.p, xb[1], lv1.pairRJ);
        else
          launch_linearize_stf(st, mv, pv, s->max_scan, s->n_obs, s->n_matches, xb[0], s->pose_cs.p, o.point_correlation_factor, o.laser_std_dev, lv0.pairRJ,
                               s->lm_dev.p, xb[1], lv1.pairRJ, s->stf_pose_recs_of == s->corr_generation ? s->stf_pose_recs.p : nullptr);
      }
      if (timed) { HIP_TRY(hipEventRecord(s->lm_events[2 * pairs_used + 1], st)); pair_iter[pairs_used] = queued + 1; ++pairs_used; }
      launch_assemble(st, mv, pv, fv, xb[0], lv0, s->lm_dev.p, xb[1], &lv1, &decide);      // its last workgroup takes the iteration's decision
      HIP_TRY(hipGetLastError());
      ++queued;
    }
    // wait until the device has finished another iteration (or the loop)
    for (uint32_t spins = 1;; ++spins) {
      const int c = (int)prog[0];
      if (c > completed) { completed = c; std::atomic_thread_fence(std::memory_order_acquire); break; }
      if ((spins & 0xfffffu) == 0) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) { if ((int)prog[0] > completed) continue; return fail(HITL_ERR_HIP, "the stream drained without progress of the LM loop"); }
        if (e != hipErrorNotReady) return fail(HITL_ERR_HIP, std::string("while waiting for the LM loop: ") + hipGetErrorString(e));
      }
    }
    if (prog[1] != 0.0 || completed >= max_it) done = true;
  }
  const double t_done = now_ms();
  LmDev fin;
  HIP_TRY(hipMemcpyAsync(&fin, s->lm_dev.p, sizeof(fin), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (dev_switches().timing) std::fprintf(stderr, "lm_device_loop: loop %.3f ms, drain %.3f ms, queued %d completed %d\n", t_done - t_loop0, now_ms() - t_done, queued, completed);
  //                    // also drains the launches queued behind the end (they return at once)
  for (int k = 0; k < pairs_used; ++k) {
    if (pair_iter[k] > fin.iter) continue;                 // queued behind the end of the loop: that launch returned at once, not a sample
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->lm_events[2 * k], s->lm_events[2 * k + 1]) == hipSuccess) { S->linearize_ms += ms; S->linearize_timed_calls++; }
    else (void)hipGetLastError();
  }
  S->iterations = fin.iter; S->successful_steps += fin.successful; S->unsuccessful_steps += fin.unsuccessful;
  S->linear_iterations += fin.iter; S->linearize_calls += fin.lin_calls;
  S->final_cost = fin.cost; S->final_gradient_max_norm = fin.gmax;
  *radius_out = fin.radius; *term_out = fin.term;
  *xfinal = xb[fin.cur];
  return HITL_OK;
}

int lm_solve(hitl_solver* s, int mode, const float* poses_in, const float* poses_global, float* poses_out,
                    hitl_summary* summary) {
  const double t_start = now_ms();
  const hitl_options& o = s->opt;
  RoctxRange r_solve(mode == HITL_MODE_ODOM_HC ? "hitl:solve mode A" : mode == HITL_MODE_STF_ONLY ? "hitl:solve mode B" : "hitl:solve mode C");
  FactorView fv;
  { RoctxRange r("hitl:prepare"); TRY(prepare(s, mode, poses_in, poses_global, &fv)); }
  if (dev_switches().timing) std::fprintf(stderr, "lm_solve: prepare (factor tables)   %.3f ms\n", now_ms() - t_start);
  hipStream_t st = s->stream;
  const int n = s->n, n3 = 3 * n;
  TRY(s->lm_arrivals.resize(1));                           // arrival counter of k_assemble's last-workgroup duties: zero at the start of every solve
  HIP_TRY(hipMemsetAsync(s->lm_arrivals.p, 0, sizeof(unsigned), st));
  hitl_summary S;
  std::memset(&S, 0, sizeof(S));
  {  // SetParams (JointOptimization.cpp:392-402): fp32 -> fp64, through a page-locked buffer the solver keeps (queued, not waited for)
    TRY(s->h_x.resize(s->npad));
    double* hx = s->h_x.p;
    for (int t = 0; t < n3; ++t) hx[t] = (double)poses_in[t];
    for (int t = n3; t < s->npad; ++t) hx[t] = 0.0;
    HIP_TRY(hipMemcpyAsync(s->x.p, hx, sizeof(double) * s->npad, hipMemcpyHostToDevice, st));
  }
  double* xcur = s->x.p;
  double* xcand = s->xc.p;
  int cur = 0;
  LinResult lr;
  { RoctxRange r("hitl:initial linearisation"); TRY(run_linearize_dist(s, fv, xcur, cur, &lr)); }
  S.linearize_ms += lr.stf_ms; S.linearize_calls++; S.linearize_timed_calls += lr.timed;
  const ByteModel bm = byte_model(s, fv);
  const double pcie_per_iter = 8.0 * (8.0 * spmv_blocks(s->npad) + 2.0 * assemble_blocks(n) + 1.0);   // partial sums + the flag, written to pinned memory
  S.pcie_bytes = 8.0 * s->npad + 4.0 * 3.0 * n + s->prepare_pcie_bytes + 16.0 * assemble_blocks(n);     // poses in (fp64), out, factor tables, first partials
  double cost = lr.cost, gmax = lr.gmax;
  S.initial_cost = cost;
  S.num_pairs = fv.with_stf ? s->n_pairs_owned : 0;
  S.num_matches = fv.with_stf ? s->n_matches_owned : 0;
  S.num_offband_pairs = fv.n_offband;
  S.band_poses = fv.band_poses;
  S.submap_linear_solver = !s->dist_mode() ? 0.0 : s->dist_pcg ? 3.0 : s->dist_sep ? 2.0 : 1.0;
  S.num_hc_residuals = s->num_hc_residuals;
  // odometry edges are owned by the rank owning their second pose
  const int n_edges = (s->own_hi - s->own_lo) - ((s->g_begin == 0) ? 1 : 0);
  S.num_residuals = (fv.with_stf ? 2 * s->n_pairs_owned : 0) + (fv.with_odom ? 3 * n_edges : 0) + (fv.with_hc ? s->num_hc_residuals : 0);
  launch_jacobi_scale(st, n, s->npad, fv.band_poses, s->Hband[cur].p, o.jacobi_scaling, s->fixed.p, s->scale.p);
  if (s->dist_mode() && s->dist_pcg) TRY(dist_gather_owned(s, 0, nullptr, s->scale.p, nullptr));   // halo rows: the owners' scales (complete diagonal blocks)
  double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
  int iter = 0, invalid_run = 0;
  int pcg_refined_iter = -1;        // the LM iteration whose persistent PCG launch has been continued from its true residual (once per iteration)
  bool pcg_refine = false;
  int pcg_carry_iters = 0;          // CG iterations of the launch(es) in front of a continuation
  int pcg_continued = 0;
  int term = HITL_NO_CONVERGENCE;
  bool done = false;
  if (!(cost == cost) || !(gmax == gmax)) { term = HITL_FAILURE; done = true; }
  else if (gmax <= o.gradient_tolerance) { term = HITL_CONVERGENCE; done = true; }
  bool dev_loop = false;
  const double t_loop = now_ms();
  if (!done && o.max_num_iterations > 0 && lm_device_loop_applies(s, fv)) {
    // decisions on the device, iterations queued ahead (single GPU, banded): see lm_device_loop
    double* xf = nullptr;
    TRY(lm_device_loop(s, fv, cur, cost, gmax, &S, &xf, &radius, &term));
    xcur = xf;
    cost = S.final_cost; gmax = S.final_gradient_max_norm; iter = S.iterations;
    done = true; dev_loop = true;
  }
  while (!done) {
    if (iter >= o.max_num_iterations) { term = HITL_NO_CONVERGENCE; break; }
    ++iter;
    s->lm_iter = iter;
    RoctxRange r_iter("hitl:lm iteration");
    int lin_iters = 0;
    bool ok = true;
    const bool time_solver = o.timing_level >= 2;
    if (time_solver) HIP_TRY(hipEventRecord(s->ev2, st));
    StepScalars ss;
    bool fused = (!s->dist_mode() && fv.n_offband == 0);   // one host synchronisation per LM iteration
    PcgPlan pcg_plan;
    bool fused_pcg = !fused && !s->dist_mode() && fv.n_offband > 0 && dev_switches().pcg_fused && !s->pcg_fused_off;
    if (fused_pcg) {
      TRY(pcg_prepare_items(s, fv));
      fused_pcg = pcg_fused_plan(s->nsb, s->B, std::max(s->npad / 3, s->pcg_n_items), s->cu_count, &pcg_plan);     // one product wave per item where the device has them
    }
    LinResult lr_spec = {0.0, 0.0, 0.f, 0};
    bool have_spec = false;
    if (fused_pcg) {
      const bool pcg_timed = o.timing_level >= 2 || (o.timing_level == 1 && (iter & 3) == 1);
      { RoctxRange r("hitl:queue PCG solve + step"); TRY(enqueue_linear_solve_pcg(s, fv, cur, radius, xcur, xcand, pcg_plan, pcg_timed, pcg_refine)); }
      pcg_refine = false;
      if (time_solver) HIP_TRY(hipEventRecord(s->ev3, st));
      { RoctxRange r("hitl:queue linearisation of the candidate");
        TRY(enqueue_linearize(s, fv, xcand, cur ^ 1, /*direct=*/true, /*have_trig=*/true)); }
      { RoctxRange r("hitl:wait for the device"); TRY(spin_sync(s)); }
      const double* rec = s->h_scalars + 8;
      const double st_pcg = rec[2];
      if (dev_switches().timing && st_pcg != 0.0)
        std::fprintf(stderr, "k_pcg_fused: status %.0f after %.0f iterations, r.r (status 2: workgroup * 1000 + thread + wait site / 100 of the wait that ran out) %.4f, b.b %.3g (M %d, chunks %d, G %d, items %d)\n", st_pcg, rec[0], rec[1], rec[3], pcg_plan.M,
                     pcg_plan.chunks, pcg_plan.G, s->pcg_n_items);
      if (st_pcg == 2.0 || st_pcg < 0.0) {                        // a wait between the workgroups ran out (or nobody reported): this solve is void
        s->pcg_fused_off = true;
        (void)take_cr_tail_failure(s);
        --iter; continue;                                         // the same iteration again, one launch group per CG iteration
      }
      if (take_cr_tail_failure(s)) { --iter; continue; }
      if (st_pcg != 0.0) ok = false;                              // breakdown / not finite: an invalid step
      lin_iters = 1 + (int)rec[0];
      if (pcg_timed) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev4, s->ev5));
        S.pcg_ms += ms; S.pcg_timed_calls += 1.0; S.pcg_timed_iterations += rec[0];
      }
      double rr_true = 0.0, bb_true = 0.0;
      TRY(finish_step_partials(s, &rr_true, &bb_true, &ss));
      if (dev_switches().timing)
        std::fprintf(stderr, "k_pcg_fused: LM iteration %d: %d CG iterations, recurrence |r| / |b| %.3e, true |b - A x| / |b| %.3e (tolerance %.1e)\n", iter, (int)rec[0],
                     std::sqrt(rec[1] / std::max(rec[3], 1e-300)), std::sqrt(rr_true / std::max(bb_true, 1e-300)), std::max(o.linear_solver_tolerance, 1e-15));
      if (st_pcg == 0.0 && !(rr_true == rr_true)) ok = false;     // (not finite: an invalid step)
      {
        // The launch stopped because its RECURRENCE residual met the tolerance; behind a weak preconditioner (hundreds of iterations) the
        // single-reduction recurrence drifts and the true residual of what it returned can be 50-80 x larger (measured: 5e-11 against
        // 6e-13 at 100-470 iterations with a one-pose band; on c3, 10-29 iterations, they agree to 2-3 digits).  Then the solve is
        // CONTINUED once: the same iteration again, starting from this step and its true residual.  A launch that ran into the
        // iteration cap is what it is (Ceres uses such a step too).
        const double tol = std::max(o.linear_solver_tolerance, 1e-15);
        if (st_pcg == 0.0 && ok && rec[0] < (double)o.max_linear_iterations && rr_true > 100.0 * tol * tol * bb_true && pcg_refined_iter != iter) {
          pcg_refined_iter = iter; pcg_refine = true; pcg_carry_iters += (int)rec[0];
          ++pcg_continued;
          --iter; continue;
        }
      }
      lin_iters += pcg_carry_iters; pcg_carry_iters = 0;
      fused = true;                                               // from here on: the banded flow (the candidate's linearisation is in pinned memory)
    } else if (fused) {
      { RoctxRange r("hitl:queue linear solve + step"); TRY(enqueue_linear_solve_banded(s, fv, cur, radius, xcur, xcand)); }
      if (time_solver) HIP_TRY(hipEventRecord(s->ev3, st));
      const double seq = (double)(++s->sync_seq);
      { RoctxRange r("hitl:queue linearisation of the candidate");
        TRY(enqueue_linearize(s, fv, xcand, cur ^ 1, /*direct=*/true, /*have_trig=*/true, false, seq)); }   // partial sums are written to pinned host memory by the kernels
      { RoctxRange r("hitl:wait for the device"); TRY(wait_host_flag(s, seq)); }
      if (take_cr_tail_failure(s)) { --iter; continue; }         // repeat this iteration's solve as separate launches (same radius)
      double rr, bb;
      TRY(finish_step_partials(s, &rr, &bb, &ss));
      const double guard = std::max(std::max(o.linear_solver_tolerance, 1e-15), 1e-9);
      if (!(rr == rr) || rr > guard * guard * bb) ok = false;   // the factorisation broke down
      lin_iters = 1;
    } else {
      RoctxRange r(s->dist_mode() ? "hitl:submap linear solve + exchange" : "hitl:off-band PCG solve");
      if (s->dist_mode()) { TRY(linear_solve_and_step_dist(s, fv, cur, radius, xcur, xcand, &lin_iters, &ok, &ss, &lr_spec)); have_spec = true; }
      else TRY(linear_solve_and_step(s, fv, cur, radius, xcur, xcand, &lin_iters, &ok, &ss));
      if (time_solver) { HIP_TRY(hipEventRecord(s->ev3, st)); HIP_TRY(hipEventSynchronize(s->ev3)); }
      if (!s->dist_mode() && take_cr_tail_failure(s)) { --iter; continue; }
    }
    S.linear_iterations += lin_iters;
    if (time_solver) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, s->ev2, s->ev3));
      S.linear_solver_ms += ms;
    }
    // model_cost_change = -dt.(S g) - 1/2 dt.(S H S) dt
    const double dt_rhs = ss.dt_rhs, dt_q = ss.dt_q, step2 = ss.step2, xn2 = ss.xn2;
    const double mcc = dt_rhs - 0.5 * dt_q;  // rhs = -S g
    if (!ok || ss.bad > 0.0 || !(mcc == mcc) || mcc < 0.0) {  // invalid step
      ++S.unsuccessful_steps;
      if (++invalid_run >= 5) { term = HITL_FAILURE; break; }
      radius *= 0.5;   // LevenbergMarquardtStrategy::StepIsInvalid: halve; decrease_factor is untouched
      continue;
    }
    invalid_run = 0;
    if (fused) TRY(finish_linearize(s, fv, &lr));
    else if (have_spec) lr = lr_spec;
    else TRY(run_linearize_dist(s, fv, xcand, cur ^ 1, &lr));
    S.linearize_ms += lr.stf_ms; S.linearize_calls++; S.linearize_timed_calls += lr.timed;
    const bool late = o.tolerances_after_accept != 0;      // hitl_options: where the two step tolerances are tested
    const bool par_tol = std::sqrt(step2) <= o.parameter_tolerance * (std::sqrt(xn2) + o.parameter_tolerance);
    if (!late && par_tol) { term = HITL_CONVERGENCE; break; }
    const double cost_change = cost - lr.cost;
    const bool fun_tol = std::fabs(cost_change) <= o.function_tolerance * cost;
    if (!late && fun_tol) { term = HITL_CONVERGENCE; break; }
    const double rho = cost_change / mcc;
    if (rho > o.min_relative_decrease) {
      ++S.successful_steps;
      std::swap(xcur, xcand);
      cur ^= 1;
      cost = lr.cost; gmax = lr.gmax;
      if (gmax <= o.gradient_tolerance) { term = HITL_CONVERGENCE; break; }
      if (late && (par_tol || fun_tol)) { term = HITL_CONVERGENCE; break; }
      const double q = 2.0 * rho - 1.0;
      radius = std::min(o.max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - q * q * q));
      decrease_factor = 2.0;
    } else {
      ++S.unsuccessful_steps;
      radius /= decrease_factor; decrease_factor *= 2.0;
      if (radius < o.min_trust_region_radius) { term = HITL_CONVERGENCE; break; }
    }
  }
  if (dev_switches().timing)
    std::fprintf(stderr, "lm_solve: before the loop %.3f ms, %s loop %.3f ms (%d iterations)\n", t_loop - t_start, dev_loop ? "device-driven" : "host-driven", now_ms() - t_loop, iter);
  if (dev_switches().timing && pcg_continued > 0) std::fprintf(stderr, "k_pcg_fused: %d of %d LM iterations continued their linear solve from the true residual\n", pcg_continued, iter);
  S.termination = term; S.iterations = iter;
  S.hbm_bytes = bm.linearize * S.linearize_calls + (bm.solve + bm.step) * S.linear_iterations;
  S.pcie_bytes += pcie_per_iter * iter + 8.0 * 3.0 * n;
  S.final_cost = cost; S.final_gradient_max_norm = gmax; S.final_trust_region_radius = radius;
  {  // CopyParams (JointOptimization.cpp:380-390): fp64 -> fp32, theta through angle_mod
    TRY(s->h_x.resize(s->npad));
    double* hx = s->h_x.p;                                 // page-locked: a direct DMA instead of a copy through a bounce buffer
    HIP_TRY(hipMemcpyAsync(hx, xcur, sizeof(double) * n3, hipMemcpyDeviceToHost, st));
    unsigned tick[4] = {0, 0, 0, 0};
    if (s->cr_ticket.p != nullptr) HIP_TRY(hipMemcpyAsync(tick, s->cr_ticket.p, sizeof(tick), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (tick[2] != 0) {                                    // a wait inside k_cr_tail ran out (never seen): leave clean tickets behind and say so
      (void)hipMemsetAsync(s->cr_ticket.p, 0, sizeof(tick), st);
      return fail(HITL_ERR_HIP, "the fused tail of the cyclic reduction (k_cr_tail) gave up waiting for its workgroups");
    }
    for (int i = 0; i < n; ++i) {
      poses_out[3 * i] = (float)hx[3 * i];
      poses_out[3 * i + 1] = (float)hx[3 * i + 1];
      poses_out[3 * i + 2] = (float)angle_mod(hx[3 * i + 2]);
    }
  }
  S.total_ms = now_ms() - t_start;
  if (summary) *summary = S;
  if (term == HITL_FAILURE) return fail(HITL_ERR_SOLVER_FAILURE, "Levenberg-Marquardt terminated with FAILURE");
  return HITL_OK;
}

}  // namespace hitl
