#!/usr/bin/env python3
"""bench.py -- full batch re-solve throughput of the MI355X pose-graph backend.

A "step" is one complete batch re-solve (hitl_solver_solve, mode C = odometry + human rows +
scan-to-scan factors: every Levenberg-Marquardt iteration re-evaluates all matched
observation rows) of one synthetic map whose scans and correspondences are already resident
in HBM; poses and human-constraint rows are handed over inside the timed region, as at the
reference call site (HitLSLAM.cpp:450-458 -> JointOpt::Run).

Default workload (N=1): BASELINE.json's target configuration "50k poses x 400 obs/pose, 50
human constraints" on one GPU (configs[3]'s map at 1 GPU, the size the north-star target
">= 10x lower wall-clock per full batch re-solve ... at 1 GPU" is quoted on).
--workload c2 selects configs[1] (1k x 360, 2 colinear constraints).

With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) the run is the one
BASELINE.json names for that GPU count: --gpus 2 | 4 split the SAME 50k x 400 map into N
pose-range submaps (configs[3], "scaling": "strong"); --gpus 8 splits the 200k x 512
campus map into 8 (configs[4], --workload c5).  `--scaling weak` keeps the rounds 2-4 run
(every rank gets a whole copy of the workload's trajectory length: N x the poses).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: poses of the map BASELINE.json names, obs per pose, constraint groups of that map, constraint kind, window,
    #       weak_poses = poses PER GPU of a `--scaling weak` run (c5: the named map is 8 such pieces)
    "c4": dict(poses=50000, obs=400, groups=50, kind=0, window=3, weak_poses=50000, baseline="configs[3]",
               desc="synthetic multi-floor: 50k poses x 400 obs/pose, 50 human constraints (mixed), mode C"),
    "c2": dict(poses=1000, obs=360, groups=2, kind=4, window=3, weak_poses=1000, baseline="configs[1]",
               desc="synthetic corridor: 1k poses x 360 obs/pose, 2 colinearity constraints, mode C"),
    "c5": dict(poses=200000, obs=512, groups=200, kind=0, window=3, weak_poses=25000, baseline="configs[4]",
               desc="synthetic campus-scale: 200k poses x 512 obs/pose, 200 mixed constraints, mode C"),
}
WORKLOAD_SEED = {"c2": 2, "c4": 4, "c5": 5}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def kernel_source_sha(path=None):
    """Short hash of the source of the dominant kernel (hitl_slam_amd/csrc/hitl_kernels.hip): profiles/traffic_*.json stores it, and a
    profile taken on another source text is not used for `roofline.traffic`."""
    import hashlib
    path = path or os.path.join(ROOT, "hitl_slam_amd", "csrc", "hitl_kernels.hip")
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def algorithmic_bytes_stf(n_obs, n_matches, n_poses, n_pairs):
    """Algorithmic bytes of ONE k_linearize_stf launch, SURVEY.md section 8(d) `linearize` row
    restricted to what this kernel touches: 16 B per observation (point + normal, read once as
    a source), 24 B per match (u32 src idx + u32 tgt idx + 16 B gathered target point/normal),
    24 B per pose (fp64 state), and 112 B per directed pair (r[2] + J[2x6] written; the 3x3
    blocks of the survey's 72(N+P) term are produced from these by k_assemble)."""
    return 16 * n_obs + 24 * n_matches + 24 * n_poses + 112 * n_pairs


def plan_partition(workload, world, scaling="strong", cut_align=12, poses=0, obs=0):
    """The map and the pose-range submaps of a run (no GPU, no torch): what `--workload W --gpus N` solves.  Returns
    dict(wl, named_poses, obs, ppf, n_global, cuts, seed); rank r owns global poses [cuts[r], cuts[r + 1]).
    tests/test_gpu_splits_fullsize.py runs exactly this plan for BASELINE configs[3] (c4 on 4 ranks) and configs[4] (c5 on 8)."""
    wl = dict(WORKLOADS[workload])
    named_poses = wl["poses"]
    if scaling == "weak" and world > 1:
        wl["groups"] = max(1, int(round(wl["groups"] * wl["weak_poses"] / wl["poses"])))
        wl["poses"] = wl["weak_poses"]                                              # per GPU
    if poses:
        wl["groups"] = max(1, int(round(wl["groups"] * poses / wl["poses"])))   # constraint density of the named workload
        wl["poses"] = poses
    if obs:
        wl["obs"] = obs
    ppf = 200 if wl["poses"] >= 400 else max(wl["poses"] // 2, 8)
    if scaling == "strong" and world > 1:
        # the SAME map on N GPUs (BASELINE configs[3]: 50k poses on 4; configs[4]: --workload c5 --poses 200000 on 8);
        # cuts fall between floors where possible, so only an odometry edge straddles them
        n_global = wl["poses"]
        cuts = [0] + [int(round(r * n_global / world / ppf)) * ppf for r in range(1, world)] + [n_global]
        if any(cuts[r + 1] - cuts[r] < 4 * wl["window"] + 2 for r in range(world)):
            cuts = [r * n_global // world for r in range(world)] + [n_global]
        # (the human constraints stay those of the whole map: every rank is handed all rows -- global pose ids -- and keeps the
        # ones that constrain a pose it owns; anchors may lie on any rank, their targets come from the global start poses)
    else:
        n_global = wl["poses"] * world
        cuts = [r * wl["poses"] for r in range(world + 1)]
    if world > 1 and cut_align > 1 and min(cuts[r + 1] - cuts[r] for r in range(world)) >= 4 * cut_align:
        cuts = [0] + [int(round(c / cut_align)) * cut_align for c in cuts[1:-1]] + [n_global]
    seed = 20260102 + 1000 * WORKLOAD_SEED[workload]
    return dict(wl=wl, named_poses=named_poses, obs=wl["obs"], ppf=ppf, n_global=n_global, cuts=cuts, seed=seed)


# BASELINE configs[2]: "Figure8 dataset + 5 loop-closure + 3 colinear human constraints, 1xMI355X".  The real file is a Google-Drive
# link (README.md:99-103 of the reference; no network here): tools/figure8.py drives the same figure of eight through a room with two
# pillars, at the scale of a real drive (poses x obs below), written and re-read through the reference's .stfs.covars text format.
C3 = dict(poses=1000, obs=360, laps=4, n_loop=5, n_colinear=3, baseline="configs[2]",
          desc="Figure-8 stand-in (tools/figure8.py, through the .stfs.covars loader): 1000 poses x 360 obs/pose, 4 laps, 5 loop-closure (type 2) + "
               "3 colinear (type 4) groups, the reference's all-pairs correspondence search (window 0), mode C")


def pcg_source_sha():
    """Short hash of the persistent PCG kernel's source (hitl_pcg.hip + hitl_pcg_sweep.h): profiles/traffic_c3.json stores it."""
    import hashlib
    h = hashlib.sha256()
    for f in ("hitl_pcg.hip", "hitl_pcg_sweep.h"):
        h.update(open(os.path.join(ROOT, "hitl_slam_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes_cg_iteration(n_poses, n_pairs_undirected):
    """SURVEY.md section 8(d), per CG iteration: 72 (N + P) bytes of 3x3 fp64 blocks (P = block pairs of the reduced Hessian, each
    read once per product) + 192 N bytes of vectors (x, r, p, q, z read and written)."""
    return 72 * (n_poses + n_pairs_undirected) + 192 * n_poses


def run_c3(args):
    import tempfile
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    import hitl_slam_amd as H
    from hitl_slam_amd import io as hio
    from hitl_slam_amd.api import hc_rows
    from tools import figure8

    n_poses, M = (args.poses or C3["poses"]), (args.obs or C3["obs"])
    t0 = time.time()
    d = figure8.make(n_poses, M, seed=7, laps=C3["laps"])
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "figure8_standin.bag.stfs.covars")
        hio.save_stfs_covars(path, "AMRL", 1455638506.25, d["init"], d["off"], d["px"], d["py"], d["nx"], d["ny"], d["cov"])
        g = hio.load_stfs_covars(path, fix_normals=True)
    rows = hc_rows(figure8.constraints(d["gt"], C3["n_loop"], C3["n_colinear"]))
    gen_s = time.time() - t0
    init = np.ascontiguousarray(g["poses"], np.float32)
    opt = H.default_options()
    opt.timing_level = args.timing_level
    solver = H.Solver(opt)
    solver.set_map(g["obs_offset"], g["px"], g["py"], g["nx"], g["ny"], g["cov9"])
    solver.set_human_constraints(rows)
    torch.cuda.synchronize()
    t0 = time.time()
    solver.find_correspondences(init, 0)                      # all pairs, like the reference (JointOptimization.cpp:561-642)
    match_first_ms = (time.time() - t0) * 1e3
    t0 = time.time()
    solver.find_correspondences(init, 0)
    match_ms = (time.time() - t0) * 1e3
    n_obs = int(g["obs_offset"][-1])
    n_pairs, n_matches = solver.correspondence_counts()
    for _ in range(args.warmup):
        solver.set_human_constraints(rows)
        out, summ = solver.solve(H.MODE_JOINT, init)
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    iters = cg = 0
    for _ in range(args.steps):
        solver.set_human_constraints(rows)
        out, summ = solver.solve(H.MODE_JOINT, init)
        iters += summ.iterations
        cg += summ.linear_iterations
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    ms_per_step = elapsed * 1e3 / args.steps
    # side leg (untimed): the linear solves of one more re-solve bracketed by HIP events on the library's stream (timing_level 2)
    solver.set_timing_level(2)
    _, sm2 = solver.solve(H.MODE_JOINT, init)
    solver.set_timing_level(args.timing_level)
    cg2 = max(int(sm2.linear_iterations) - int(sm2.iterations), 1)          # CG iterations proper (every LM iteration counts one for x0 = M^-1 b)
    n_blockpairs = int(sm2.num_offband_pairs) + n_poses * int(sm2.band_poses)        # undirected block pairs of the reduced Hessian (off-band + band)
    alg = algorithmic_bytes_cg_iteration(n_poses, n_blockpairs)
    pcg_launches = int(getattr(sm2, "pcg_timed_calls", 0) or 0)
    pcg_ms = float(getattr(sm2, "pcg_ms", 0.0) or 0.0)
    if pcg_launches > 0 and pcg_ms > 0:        # the fused PCG: one launch per linear solve, bracketed by events; per CG iteration = launch time / its iterations
        us_per_cg = pcg_ms * 1e3 / cg2
        kernel, avg_launch_ms, units = "k_pcg_fused", pcg_ms / pcg_launches, cg2 / pcg_launches
    else:                                      # one launch group per CG iteration: the linear-solver bracket divided by its CG iterations
        us_per_cg = sm2.linear_solver_ms * 1e3 / max(int(sm2.linear_iterations), 1)
        kernel, avg_launch_ms, units = "PCG launch group (preconditioner stages + k_pcg_dir + k_spmv_offband + k_spmv + k_pcg_step)", us_per_cg * 1e-3, 1.0
    achieved = alg / (us_per_cg * 1e-6) / 1e9
    # HBM-side bytes of the persistent launch from the committed counter passes (tools/summarize_pcg_profile.py), while hitl_pcg.hip still is the profiled text
    traffic = traffic_source = traffic_stale = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_c3.json")))
        sha = pcg_source_sha()
        if kernel == "k_pcg_fused" and tj.get("n_matches") == int(n_matches) and tj.get("kernel_source_sha") == sha:
            traffic = tj["hbm_bytes_per_launch"] / max(units, 1e-9)                  # per CG iteration, like `achieved`
            traffic_source = "profiles/traffic_c3.json (%s): %.1f MB per launch of %.1f CG iterations" % (tj.get("source", "?"), tj["hbm_bytes_per_launch"] / 1e6, units)
        elif kernel == "k_pcg_fused":
            traffic_stale = "profiles/traffic_c3.json was taken on another hitl_pcg.hip / workload (sha %s, now %s): not used" % (tj.get("kernel_source_sha"), sha)
    except (OSError, ValueError, KeyError):
        pass
    d_gt = out.astype(np.float64) - d["gt"]
    # pose covariances at the solved poses (pose pairs outside the band: the dense inverse through rocSOLVER, bound at first use)
    t0 = time.time()
    solver.pose_covariances(H.MODE_JOINT, out)
    cov_first_ms = (time.time() - t0) * 1e3
    t0 = time.time()
    cov = solver.pose_covariances(H.MODE_JOINT, out)
    cov_ms = (time.time() - t0) * 1e3
    result = {
        "metric": "full batch re-solve throughput (observation rows x Gauss-Newton/LM iterations per second); "
                  "resolve_ms = full batch re-solve wall-clock, gn_iters_per_sec alongside",
        "value": float(n_obs) * iters / elapsed, "unit": "obs-iters/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": C3["desc"] + (" -- BASELINE.json %s" % C3["baseline"] if not (args.poses or args.obs) else " -- NOT the named size"),
                   "poses_per_gpu": n_poses, "obs_per_pose": M, "n_poses_total": n_poses, "n_obs_per_gpu": n_obs, "matches_per_gpu": int(n_matches),
                   "pairs_per_gpu": int(n_pairs), "offband_block_pairs": int(sm2.num_offband_pairs), "band_poses": int(sm2.band_poses), "window": 0,
                   "human_constraint_rows": rows[1], "mode": "C (joint)", "parallelism": "single GPU"},
        "resolve_ms": ms_per_step, "gn_iters_per_sec": iters / elapsed, "lm_iterations_per_resolve": int(summ.iterations),
        "cg_iterations_per_resolve": int(summ.linear_iterations) - int(summ.iterations), "ms_per_lm_iteration": ms_per_step / max(int(summ.iterations), 1),
        "termination": int(summ.termination), "initial_cost": summ.initial_cost, "final_cost": summ.final_cost,
        "correspondence_search_ms": match_ms, "correspondence_search_first_call_ms": match_first_ms,
        "max_abs_xy_vs_ground_truth_m": float(np.abs(d_gt[:, :2]).max()),
        "pose_covariances_ms": cov_ms, "pose_covariances_first_call_ms": cov_first_ms,
        "pose_covariances_note": "3x3 blocks of (J^T J)^-1 at the solved poses: dense %d x %d inverse (rocSOLVER dpotrf + dpotri, loaded by the first call); "
                                 "largest positional standard deviation %.4f m" % (3 * n_poses, 3 * n_poses, float(np.sqrt(max(cov[:, 0, 0].max(), cov[:, 1, 1].max())))),
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_source, "traffic_stale": traffic_stale,
                     "algorithmic_bytes_per_cg_iteration": alg, "us_per_cg_iteration": us_per_cg, "cg_iterations_per_launch": units,
                     "avg_launch_ms": avg_launch_ms, "launches_bracketed_by_hip_events": pcg_launches or int(sm2.iterations),
                     "linear_solver_ms_per_lm_iteration": sm2.linear_solver_ms / max(int(sm2.iterations), 1),
                     "note": "achieved = SURVEY 8(d)'s algorithmic bytes of one CG iteration (72 (N + P) + 192 N: every 3x3 block of the reduced Hessian and "
                             "the CG vectors once) / the measured time of one CG iteration; the system (a few MB) lives in L2 / LDS across the iterations "
                             "of a solve, and a CG iteration is a chain of dependent phases across workgroups: latency-bound, far below the HBM roof by nature"},
        "synth_gen_s": gen_s,
    }
    if not args.no_cpu_baseline:
        try:
            os.environ["HITL_ORACLE_NATIVE"] = "1"
            from oracle import hitl_oracle as O      # CPU baseline: the oracle (a port, not the reference binary) on the WHOLE workload
            from tools import probe_host
            cores = probe_host.usable_cores()
            corr = solver.get_correspondences()
            O.set_num_threads(cores)

            def oracle_run(max_it):
                oo = H.default_options()
                oo.max_num_iterations = max_it
                orc = O.OracleSolver(oo)
                orc.set_map(g["obs_offset"], g["px"], g["py"], g["nx"], g["ny"], g["cov9"])
                orc.set_human_constraints(rows)
                orc.set_correspondences(*corr)
                t0 = time.perf_counter()
                try:
                    oout, so = orc.solve(H.MODE_JOINT, init)
                finally:
                    dt = time.perf_counter() - t0
                    orc.close()
                return oout, so, dt
            # The oracle's linear solve is an exact banded Cholesky over the whole pair bandwidth -- with the all-pairs search a dense
            # 3N x 3N factorisation per LM iteration -- so the leg is BOUNDED: the first K LM iterations of the same re-solve, K sized
            # from a 2-iteration probe to about --cpu-budget-s seconds (all K = the whole re-solve when it fits).
            _, sp, tp = oracle_run(2)
            per_it = tp / max(int(sp.iterations), 1)
            k_it = int(max(2, min(int(summ.iterations), args.cpu_budget_s // max(per_it, 1e-3))))
            oout, so, best = oracle_run(k_it)
            whole = int(so.iterations) >= int(summ.iterations) or int(so.termination) != 1
            result["cpu_baseline"] = {
                "value": n_obs * so.iterations / best, "unit": "obs-iters/s", "cores": cores, "kind": "port",
                "sample": "%s of the same re-solve (%d poses x %d obs, %d matches, %d directed pairs, mode C): oracle = C/OpenMP restatement, "
                          "linear solve = exact banded Cholesky over the whole pair bandwidth (serial), %d threads for the residual evaluation: "
                          "%.1f s for %d LM iterations (GPU re-solve: %d)"
                          % ("the WHOLE re-solve" if whole else "the first %d LM iterations" % int(so.iterations), n_poses, M, int(n_matches), int(n_pairs),
                             cores, best, so.iterations, summ.iterations),
                "ms_per_lm_iteration": best * 1e3 / max(int(so.iterations), 1), "lm_iterations": int(so.iterations), "lm_iterations_gpu": int(summ.iterations),
                "whole_resolve": bool(whole)}
            if whole:
                dd = oout.astype(np.float64) - out
                dd[:, 2] = np.angle(np.exp(1j * dd[:, 2]))
                result["cpu_baseline"].update(resolve_ms=best * 1e3, final_cost=so.final_cost, final_cost_gpu=summ.final_cost,
                                              max_pose_difference_vs_gpu_m_rad=[float(np.abs(dd[:, :2]).max()), float(np.abs(dd[:, 2]).max())])
            result["ms_per_lm_iteration_cpu_port_over_gpu"] = (best / max(int(so.iterations), 1)) / (ms_per_step * 1e-3 / max(int(summ.iterations), 1))
        except Exception as e:
            result["cpu_baseline"] = {"value": None, "unit": "obs-iters/s", "cores": None, "kind": "port", "sample": "failed: %r" % (e,)}
    print(json.dumps(result))
    sys.stdout.flush()
    solver.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="auto", choices=["auto", "c3"] + sorted(WORKLOADS),
                    help="auto = what BASELINE.json names for the GPU count: c4 (50k x 400) at 1, 2 and 4 GPUs, c5 (200k x 512) at 8; "
                         "c3 = BASELINE configs[2] (a Figure-8 drive + 5 loop-closure + 3 colinear groups, the reference's all-pairs search: "
                         "pose pairs outside the band, the linear solves run the PCG), one GPU")
    ap.add_argument("--poses", type=int, default=0, help="override the map's poses (strong: of the whole map; weak: per GPU)")
    ap.add_argument("--obs", type=int, default=0, help="override observations per pose")
    ap.add_argument("--cpu-sample-poses", type=int, default=6000, help="poses in the 1-thread sample of the cpu_baseline leg")
    ap.add_argument("--cpu-budget-s", type=float, default=30.0, help="CPU seconds the cpu_baseline leg may spend on repeats of the full workload")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="N>1: strong (auto) = the SAME named map split into N pose-range submaps (BASELINE configs[3] / [4]); "
                         "weak = N x the workload's per-GPU poses.  N=1 is the named map on one GPU either way (labelled weak)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-copy-ceiling", action="store_true", help="skip the measured device-copy ceiling (SURVEY.md 8(d)) beside the 8 TB/s peak")
    ap.add_argument("--timing-level", type=int, default=1, help="hitl_options.timing_level: 1 = every 4th stf launch bracketed by "
                    "HIP events (default), 2 = all launches + linear-solver timing")
    ap.add_argument("--cut-align", type=int, default=12,
                    help="N > 1: the cuts between the ranks' pose ranges are moved to multiples of this many poses, so that every rank but the "
                         "last owns a multiple of the pose band and the linear solves take the separator form (12 covers bands of 1, 2, 3, 4 "
                         "and 6 poses); 0 = cuts between floors, only an odometry edge across a cut (rounds 2-3: the interface-column form)")
    ap.add_argument("--comm", default="auto", choices=["auto", "rccl", "torch"],
                    help="N>1 exchange: rccl = in-library ncclAllGather (exit non-zero if it cannot be used), torch = "
                         "torch.distributed all_gather callback, auto = rccl, else torch -- the transport used is in the JSON")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo lets several ranks share one GPU (debugging on a 1-GPU box)")
    args = ap.parse_args()
    if args.workload == "auto":
        args.workload = "c5" if args.gpus == 8 else "c4"
    if args.scaling == "auto":
        args.scaling = "strong" if args.gpus > 1 else "weak"

    if args.workload == "c3":
        if args.gpus != 1:
            sys.stderr.write("bench.py: --workload c3 (BASELINE configs[2]) is a one-GPU configuration\n")
            sys.exit(4)
        return run_c3(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` started directly: N ranks are needed and this process is not one of them.  Start them
        # as CHILD processes (torch.distributed.run, one rank per GPU) before this process has imported torch or touched
        # the GPU -- never a re-exec -- relay rank 0's JSON line and exit with the launcher's code.  A run labelled
        # n_gpus = N is therefore always N ranks; it can never be one rank with a label.
        import subprocess
        # (--standalone: the launcher picks a free rendezvous port itself -- no bind/close/reuse race)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stderr.write("bench.py: --gpus %d without WORLD_SIZE: launching %d ranks: %s\n" % (args.gpus, args.gpus, " ".join(cmd)))
        sys.exit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: the number of ranks IS the number of GPUs of a run; "
                         "start one rank per GPU (torch.distributed.run --nproc-per-node %d)\n" % (args.gpus, world, args.gpus))
        sys.exit(4)

    import torch
    import torch.distributed as dist
    use_dist = world > 1
    if args.backend == "nccl" and use_dist and torch.cuda.device_count() < world:      # device_count() does not initialise the GPU
        sys.stderr.write("bench.py: %d ranks over RCCL need %d GPUs, this node shows %d (use --backend gloo to let ranks share a "
                         "GPU for debugging)\n" % (world, world, torch.cuda.device_count()))
        sys.exit(5)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    tdev = "cuda" if args.backend == "nccl" else "cpu"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend="gloo")
            args.comm = "torch"

    import hitl_slam_amd as H
    from tools import hitl_synth as S

    plan = plan_partition(args.workload, world, args.scaling, args.cut_align, args.poses, args.obs)
    wl, named_poses, M, ppf, n_global, cuts, seed = (plan[k] for k in ("wl", "named_poses", "obs", "ppf", "n_global", "cuts", "seed"))
    n_local = cuts[rank + 1] - cuts[rank]

    # ---- synthetic map: this rank's pose-range submap of the global trajectory ------------------
    t0 = time.time()
    p = S.params(n_global, M, seed=seed, poses_per_floor=ppf)
    gt, init, cov = S.poses(p)
    begin, end = cuts[rank], cuts[rank + 1]                  # this rank's owned pose range
    halo = 2 * wl["window"] if use_dist else 0               # scans + state of the neighbours' nearest poses
    lo, hi = begin - (halo if rank > 0 else 0), end + (halo if rank < world - 1 else 0)
    off, px, py, nx, ny = S.scans(p, lo, hi)
    strong = args.scaling == "strong" and world > 1
    rows = S.constraints(p, gt, wl["groups"], kind=wl["kind"], begin=0 if strong else begin, end=n_global if strong else end)
    init_l = np.ascontiguousarray(init[begin:end])
    gen_s = time.time() - t0

    opt = H.default_options()
    opt.device = dev_index
    opt.timing_level = args.timing_level
    comm_kind = "none"
    comm_ranks = 1
    keep = []
    if use_dist:
        # pose-range submaps: rank r owns global poses [begin, end); the cut falls between two floors,
        # so only the odometry edge straddles it.  Exchange = 2 tiny all-gathers per LM iteration.
        opt.rank, opt.world_size = rank, world
        solver = None
        rccl_error = None
        if args.comm in ("rccl", "auto"):
            ok = 1
            try:
                from hitl_slam_amd.api import comm_unique_id
                uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    uid.copy_(torch.frombuffer(bytearray(comm_unique_id()), dtype=torch.uint8))
                dist.broadcast(uid, 0)
                buf = (C.c_char * 128).from_buffer_copy(bytes(uid.cpu().numpy().tobytes()))
                keep.append(buf)
                opt.rccl_unique_id = C.cast(buf, C.c_void_p)
                solver = H.Solver(opt)
            except Exception as e:
                rccl_error = repr(e)
                sys.stderr.write("rank %d: in-library RCCL unavailable (%r)\n" % (rank, e))
                ok, solver = 0, None
            # the choice of transport must be the same on every rank
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                try:
                    solver.comm_selftest()
                    comm_ranks = solver.comm_ranks()
                    if comm_ranks != world:
                        raise RuntimeError("RCCL communicator holds %d ranks, expected %d" % (comm_ranks, world))
                except Exception as e:
                    rccl_error = repr(e)
                    sys.stderr.write("rank %d: RCCL self-test failed (%r)\n" % (rank, e))
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                comm_kind = "rccl (in-library ncclAllGather, %d ranks in the communicator)" % comm_ranks
            else:
                if solver is not None:
                    solver.close()
                solver = None
                if args.comm == "rccl":      # asked for explicitly: no silent change of transport
                    sys.stderr.write("rank %d: --comm rccl requested but in-library RCCL is not usable (%s)\n" % (rank, rccl_error))
                    dist.destroy_process_group()
                    sys.exit(3)
        if solver is None:
            from hitl_slam_amd.api import torch_allgather_ops
            opt.rccl_unique_id = None
            solver = H.Solver(opt)
            ops, fn = torch_allgather_ops(device=torch.device("cuda", dev_index) if args.backend == "nccl" else None)
            solver.set_comm(ops, fn)
            comm_ranks = solver.comm_ranks()
            comm_kind = "torch.distributed all_gather callback (%s backend)%s" % (
                "nccl = RCCL" if args.backend == "nccl" else "gloo",
                "; in-library RCCL was tried first and failed" if args.comm == "auto" and args.backend == "nccl" else "")
        solver.set_partition(begin, n_global, halo)
        solver.set_map(off, px, py, nx, ny, cov[lo:hi])
        solver.set_human_constraints(rows)          # global pose ids
        pose_arg = init                             # global pose array
    else:
        solver = H.Solver(opt)
        solver.set_map(off, px, py, nx, ny, cov[begin:end])
        solver.set_human_constraints(rows)
        pose_arg = init_l
    torch.cuda.synchronize()
    t0 = time.time()
    solver.find_correspondences(pose_arg, wl["window"])
    match_first_ms = (time.time() - t0) * 1e3                   # first call: allocates the candidate table and the pair tables
    t0 = time.time()
    solver.find_correspondences(pose_arg, wl["window"])        # what every later correction pays (JointOptimization.cpp:1311)
    match_ms = (time.time() - t0) * 1e3
    n_obs = int(off[end - lo] - off[begin - lo])               # observations of the OWNED poses
    n_pairs_l, n_matches_l = solver.correspondence_counts()     # this rank's submap (incl. the few halo-sourced pairs)

    # ---- warm-up, then exactly K timed re-solves -----------------------------------------------------
    for _ in range(args.warmup):
        solver.set_human_constraints(rows)
        out, summ = solver.solve(H.MODE_JOINT, pose_arg)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    iters = 0
    stf_ms = 0.0
    stf_calls = 0
    stf_launches = 0
    lin_solver_ms = 0.0
    for _ in range(args.steps):
        solver.set_human_constraints(rows)                       # constraints + poses are handed over inside the timed region
        out, summ = solver.solve(H.MODE_JOINT, pose_arg)
        iters += summ.iterations
        stf_ms += summ.linearize_ms
        stf_calls += summ.linearize_timed_calls
        stf_launches += summ.linearize_calls
        lin_solver_ms += summ.linear_solver_ms
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(n_obs)], dtype=torch.float64, device=tdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        # one global Levenberg-Marquardt loop: every rank reports the same iteration count
        iters_all, obs_iters_all = float(iters), float(tot[0].item()) * iters
    else:
        iters_all, obs_iters_all = float(iters), float(n_obs) * iters

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    if world > 1 and comm_ranks != args.gpus:
        raise SystemExit("bench.py: the communicator holds %d ranks but the line would say n_gpus = %d" % (comm_ranks, args.gpus))
    ms_per_step = elapsed * 1e3 / args.steps
    avg_stf_ms = stf_ms / max(stf_calls, 1)
    alg_bytes = algorithmic_bytes_stf(n_obs, n_matches_l, n_local, n_pairs_l)
    alg_achieved = alg_bytes / (avg_stf_ms * 1e-3) / 1e9 if avg_stf_ms > 0 else 0.0
    compulsory = 16 * n_obs + 4 * int(n_matches_l) + 24 * n_local + 112 * int(n_pairs_l)
    # HBM bytes per launch are NOT measured in this run: they come from the committed PMC passes of the same command
    # (tools/profile_bench.sh: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, separate rocprofv3 --pmc runs)
    # and are used only when that profile was taken on exactly this workload; otherwise the layout's compulsory bytes.
    traffic, traffic_source, traffic_stale = None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            # the counters describe the kernel that was profiled: the profile carries the hash of the kernel source it was taken on
            # (tools/summarize_profile.py), and a profile of another source text is stale -- the layout's compulsory bytes then
            stale = tj.get("kernel_source_sha") != kernel_source_sha()
            if stale:
                traffic_stale = "profiles/traffic_%s.json was taken on another hitl_kernels.hip (sha %s, now %s): not used, `achieved` is from the layout's compulsory bytes" % (
                    args.workload, tj.get("kernel_source_sha"), kernel_source_sha())
            if not stale and tj.get("n_obs") == n_obs and (tj.get("n_matches") in (None, int(n_matches_l))):
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = "profiles/traffic_%s.json (%s; kernel-trace average %.1f us in that run)" % (
                    args.workload, tj.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py"), tj.get("stf_avg_us", 0.0))
        except Exception:
            traffic = None
    pmc = None
    ppath = os.path.join(ROOT, "profiles", "pmc_%s.json" % args.workload)
    if os.path.exists(ppath):
        try:
            pmc = json.load(open(ppath))
        except Exception:
            pmc = None
    moved = traffic if traffic else compulsory
    achieved = moved / (avg_stf_ms * 1e-3) / 1e9 if avg_stf_ms > 0 else 0.0
    out_l = out[begin:end] if use_dist else out
    err_gt = float(np.abs(out_l[:, :2].astype(np.float64) - gt[begin:end, :2]).max())

    result = {
        "metric": "full batch re-solve throughput (observation rows x Gauss-Newton/LM iterations per second); "
                  "resolve_ms = full batch re-solve wall-clock, gn_iters_per_sec alongside",
        "value": obs_iters_all / elapsed,
        "unit": "obs-iters/s",
        "n_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling if world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": wl["desc"] + ((" -- BASELINE.json %s" % wl["baseline"]) if n_global == named_poses and not args.obs else
                                              " -- NOT a BASELINE.json size (%d poses in all)" % n_global) +
                               ((", split into %d pose-range submaps (%s scaling)" % (world, args.scaling)) if world > 1 else ""),
                   "poses_per_gpu": n_local, "obs_per_pose": M, "n_poses_total": n_global,
                   "n_obs_per_gpu": n_obs, "matches_per_gpu": int(n_matches_l), "pairs_per_gpu": int(n_pairs_l),
                   "matches_per_obs": n_matches_l / max(n_obs, 1), "window": wl["window"],
                   "human_constraint_rows": rows[1], "mode": "C (joint)", "band_poses": int(summ.band_poses),
                   "parallelism": ("%d pose-range submaps, 1 per GPU, one global LM; exchange: %s" % (world, comm_kind)) if world > 1 else "single GPU",
                   "comm": comm_kind, "comm_ranks": comm_ranks if world > 1 else 1,
                   "submap_linear_solver": {0: "none (single GPU)", 1: "interface columns", 2: "separators", 3: "distributed PCG"}.get(
                       int(summ.submap_linear_solver), "?"),
                   "cuts": cuts if world > 1 else None,
                   "ranks_share_gpus": bool(world > 1 and torch.cuda.device_count() < world)},
        "resolve_ms": ms_per_step,
        "gn_iters_per_sec": iters_all / elapsed,
        "lm_iterations_per_resolve": summ.iterations,
        "termination": int(summ.termination),
        "final_cost": summ.final_cost,
        "correspondence_search_ms": match_ms, "correspondence_search_first_call_ms": match_first_ms,
        "correspondence_search": "kd_tree_search = 1 (default): the reference's per-scan KD-trees + pruned walk (JointOptimization.cpp:561-642, "
                                 "kdtree.cpp:141-197); first call includes BuildKDTrees (once per map)",
        "residual_drift_vs_ground_truth_m": err_gt,
        "residual_drift_note": "max |xy - ground truth| after the re-solve; the synthetic odometry drift is only removed where "
                               "human constraints and scan overlap observe it, so this is a property of the input, not an error",
        # frac = bytes this kernel really moves through HBM per launch (PMC-measured when a matching profile is
        # committed, else the layout's compulsory bytes) / live HIP-event launch time / 8 TB/s.  The SURVEY 8(d)
        # algorithmic-byte figure (24 B per match) is kept beside it: this layout moves 4 B per match and serves the
        # target gathers from L2, so that figure can exceed the peak and is not a bandwidth.
        "roofline": {"bound": "hbm", "kernel": "k_linearize_stf", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "traffic_stale": traffic_stale,
                     "bytes_per_launch_used_for_achieved": moved,
                     "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_achieved": alg_achieved,
                     "algorithmic_frac": alg_achieved / HBM_PEAK_GBS,
                     "compulsory_bytes_this_layout": compulsory,
                     "avg_launch_ms": avg_stf_ms, "launches": stf_launches, "launches_bracketed_by_hip_events": stf_calls,
                     # counter digests describe ONE profiled workload: reported only when this run is that workload
                     # the contract's "bound: hbm" is a label: HBM is the unit this kernel loads least.  What it issues is fp64 VALU work:
                     # vector instructions per launch (PMC SQ_INSTS_VALU of the committed profile of THIS workload) x 4 cycles (a wave64
                     # fp64 / fp32 / integer instruction holds its SIMD for 4) / (1024 SIMDs x the live launch time x 2.4 GHz)
                     "fp64_issue_frac": ((pmc or {}).get("valu_instructions_per_launch", 0) * 4.0 / (1024 * avg_stf_ms * 1e-3 * 2.4e9)
                                         if traffic_source and avg_stf_ms > 0 and (pmc or {}).get("valu_instructions_per_launch") else None),
                     "limiter": (pmc or {}).get("limiter") if traffic_source else None,
                     "utilisation_counters": pmc if traffic_source else None,
                     "profiled": bool(traffic_source)},
        "synth_gen_s": gen_s,
    }

    # ---- side leg (untimed): every launch bracketed, linear solve of every LM iteration timed (timing_level 2) ----
    if world == 1:
        try:
            solver.set_timing_level(2)
            _, sm2 = solver.solve(H.MODE_JOINT, pose_arg)
            solver.set_timing_level(args.timing_level)
            result["phase_breakdown_timing_level_2"] = {
                "linear_solver_ms_per_resolve": sm2.linear_solver_ms, "linear_solver_ms_per_lm_iteration": sm2.linear_solver_ms / max(sm2.iterations, 1),
                "linearize_stf_ms_per_launch_all_launches": sm2.linearize_ms / max(sm2.linearize_timed_calls, 1),
                "lm_iterations": int(sm2.iterations), "resolve_ms_with_event_overhead": sm2.total_ms,
                "note": "one extra re-solve after the timed region with hitl_options.timing_level = 2 (an event record drains the "
                        "queue, so this leg is slower than ms_per_step and is not part of `value`)"}
        except Exception as e:
            result["phase_breakdown_timing_level_2"] = {"failed": repr(e)}

    # ---- measured device-copy ceiling beside the vendor peak; untimed side leg, never part of `value` or `frac` ----
    if world == 1 and not args.no_copy_ceiling:
        try:
            from hitl_slam_amd.api import device_copy_bandwidth
            nbytes, reps = 1 << 30, 10                              # 1 GiB read + 1 GiB written per launch
            copy_gbs = device_copy_bandwidth(dev_index, nbytes, reps)
            result["roofline"]["measured_copy_ceiling"] = {
                "GB/s": copy_gbs, "ms_per_copy": 2.0 * nbytes / (copy_gbs * 1e9) * 1e3 if copy_gbs > 0 else None, "bytes_per_copy": 2 * nbytes,
                "achieved_over_ceiling": (achieved / copy_gbs) if copy_gbs > 0 else None,
                "what": "hitl_device_copy_bandwidth: the library's float4 copy kernel (one float4 per thread, a workgroup per 4 KB) "
                        "over 1 GiB, read + write counted, 10 launches between two HIP events on its own stream after 2 warm-up launches, after the "
                        "timed region; `frac` stays against the 8 TB/s vendor peak"}
        except Exception as e:
            result["roofline"]["measured_copy_ceiling"] = {"failed": repr(e)}

    # ---- one interactive correction cycle on the same resident map, reported alongside ----
    if world == 1:
        try:
            def timed(fn, *a, **k):
                t0 = time.perf_counter()
                r = fn(*a, **k)
                return r, (time.perf_counter() - t0) * 1e3
            clicks = S.correction_clicks(gt, init_l, off, px, py)
            cyc = {}
            poses0 = np.ascontiguousarray(init_l)
            seen, cyc["verify_points_ms"] = timed(solver.verify_points, poses0, clicks.reshape(-1, 2), 0.05)
            (seg, inl, pas, its), cyc["fit_segments_ms"] = timed(solver.fit_segments, poses0, clicks.reshape(2, 4))
            (counts, _), cyc["observation_sets_ms"] = timed(solver.observation_sets, poses0, seg.reshape(-1), 0.03, False)
            od, cyc["order_user_input_ms"] = timed(solver.order_user_input, counts, seg.reshape(-1))
            (p2, corr_v, new_rows), cyc["explicit_correction_ms"] = timed(solver.apply_explicit_correction, 4, od["seg_xy8"], poses0,
                                                                         od["corrected"], od["anchor"])
            (p3, cov3), cyc["backprop_ms"] = timed(solver.backprop, p2, cov, od["bounds"][0], od["bounds"][1], corr_v)
            # the worst case of the O(span^2) stage: bounds at the two ends of the whole trajectory
            _, cyc["backprop_full_span_ms"] = timed(solver.backprop, poses0, cov, 1, n_local - 2, corr_v)
            _, cyc["transform_scans_ms"] = timed(solver.transform_scans, poses0)
            (_, _, pinned), _ = timed(solver.transform_scans_pinned, poses0)                       # allocates the pinned buffers
            _, cyc["transform_scans_pinned_host_ms"] = timed(solver.transform_scans_pinned, poses0, pinned)
            solver.host_free(pinned)
            _, cyc["transform_scans_device_resident_ms"] = timed(solver.transform_scans_device, poses0)
            _, cyc["pose_covariances_ms"] = timed(solver.pose_covariances, H.MODE_JOINT, out)
            pre = sum(cyc[k] for k in ("verify_points_ms", "fit_segments_ms", "observation_sets_ms", "order_user_input_ms",
                                       "explicit_correction_ms", "backprop_ms"))
            cyc["pre_solve_stages_ms"] = pre
            # what ONE interactive correction costs end to end on this map: pre-solve stages + new correspondences + the re-solve
            # + the world-frame re-projection for the display (HitLSLAM::Run, HitLSLAM.cpp:380-480)
            # (re-projection into the page-locked buffers Solver.run_correction uses by default since round 4; the pageable-memory
            # figure stays in correction_cycle.transform_scans_ms)
            result["interactive_correction_ms"] = pre + match_ms + ms_per_step + cyc["transform_scans_pinned_host_ms"]
            cyc.update(all_points_verified=bool(int(seen.sum()) == 4), fit_passes=[int(v) for v in pas], fit_inliers=[int(v) for v in inl],
                       anchors=int(od["anchor"].size), corrected=int(od["corrected"].size), new_rows=int(new_rows.size),
                       backprop_bounds=[int(od["bounds"][0]), int(od["bounds"][1])],
                       note="wall-clock per C-ABI call including its host<->device copies; the re-solve itself is ms_per_step")
            result["correction_cycle"] = cyc
        except Exception as e:
            result["correction_cycle"] = {"failed": repr(e)}

    # ---- opt-in moments mode on the same problem (NOT the headline: `value` above streams every match every iteration) ----
    if world == 1:
        try:
            opt2 = H.default_options()
            opt2.device = dev_index
            opt2.stf_moments = 1
            s2 = H.Solver(opt2)
            s2.set_map(off, px, py, nx, ny, cov)
            s2.set_human_constraints(rows)
            s2.find_correspondences(init_l, wl["window"])
            out2, summ2 = s2.solve(H.MODE_JOINT, init_l)                 # includes summing the moments once
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out2, summ2 = s2.solve(H.MODE_JOINT, init_l)             # moments of the resident correspondence set reused
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t0) * 1e3 / args.steps
            t0 = time.perf_counter()
            for _ in range(args.steps):
                # the reference rebuilds its correspondences on every solve (JointOptimization.cpp:1189-1191): search, the
                # moment sums of the new set (k_stf_moments) and the solve, all inside the timed region
                s2.find_correspondences(init_l, wl["window"])
                out2, summ2 = s2.solve(H.MODE_JOINT, init_l)
            torch.cuda.synchronize()
            ms2_full = (time.perf_counter() - t0) * 1e3 / args.steps
            d = out2.astype(np.float64) - out
            d[:, 2] = np.angle(np.exp(1j * d[:, 2]))
            result["stf_moments_mode"] = {
                "resolve_ms": ms2, "search_plus_moments_plus_resolve_ms": ms2_full,
                "lm_iterations_per_resolve": int(summ2.iterations), "final_cost": summ2.final_cost,
                "max_pose_difference_vs_streaming": float(np.abs(d).max()),
                "linearize_stf_ms_per_launch": summ2.linearize_ms / max(summ2.linearize_timed_calls, 1),
                "note": "hitl_options.stf_moments=1: scan-to-scan factors from 2x15 moments per pose pair summed once per "
                        "correspondence set; same algebra, per-iteration cost independent of the number of matches"}
            s2.close()
        except Exception as e:
            result["stf_moments_mode"] = {"failed": repr(e)}

    # ---- mode A on the same resident map: odometry + human rows only (SolveHumanConstraints, JointOptimization.cpp:1064-1138 --
    #      the solve the shipped binary runs after every correction; the headline above is mode C) ----
    if world == 1:
        try:
            outa, summa = solver.solve(H.MODE_ODOM_HC, init_l)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                outa, summa = solver.solve(H.MODE_ODOM_HC, init_l)
            torch.cuda.synchronize()
            result["mode_a_resolve_ms"] = (time.perf_counter() - t0) * 1e3 / args.steps
            result["mode_a_lm_iterations"] = int(summa.iterations)
        except Exception as e:
            result["mode_a_resolve_ms"] = None
            result["mode_a_error"] = repr(e)

    # ---- the opt-in brute-force definition of the correspondence query on the same map (kd_tree_search = 0) ----
    if world == 1:
        try:
            opt3 = H.default_options()
            opt3.device = dev_index
            opt3.kd_tree_search = 0
            s3 = H.Solver(opt3)
            s3.set_map(off, px, py, nx, ny, cov)
            s3.find_correspondences(init_l, wl["window"])
            t0 = time.perf_counter()
            s3.find_correspondences(init_l, wl["window"])
            result["correspondence_search_brute_force_ms"] = (time.perf_counter() - t0) * 1e3
            result["correspondence_search_brute_force_matches"] = int(s3.correspondence_counts()[1])
            s3.close()
        except Exception as e:
            result["correspondence_search_brute_force_ms"] = None
            result["correspondence_search_brute_force_error"] = repr(e)

    # ---- CPU baseline: the oracle (a port, not the reference binary) on the WHOLE bench workload ----------------
    # Same map, same correspondences, same human rows, same mode; threads = the cores this process may really use
    # (affinity mask capped by the cgroup CPU quota -- os.cpu_count() is neither); warm-up + best of up to 5 within
    # --cpu-budget-s; a 1-thread figure on a bounded sample beside it.  `make native` build of the oracle for this host.
    if not args.no_cpu_baseline and world == 1:
        try:
            os.environ["HITL_ORACLE_NATIVE"] = "1"
            from oracle import hitl_oracle as O
            from tools import probe_host
            cores = probe_host.usable_cores()
            host = probe_host.probe()
            corr = solver.get_correspondences()
            orc = O.OracleSolver()
            orc.set_map(off, px, py, nx, ny, cov)
            orc.set_human_constraints(rows)
            orc.set_correspondences(*corr)
            O.set_num_threads(cores)
            t0 = time.perf_counter()
            oout, so = orc.solve(H.MODE_JOINT, init_l)                       # warm-up (page faults, thread pool)
            first = time.perf_counter() - t0
            reps = int(max(1, min(5, (args.cpu_budget_s - first) // max(first, 1e-3))))
            times, phases = [], None
            for _ in range(reps):
                t0 = time.perf_counter()
                oout, so = orc.solve(H.MODE_JOINT, init_l)
                dt = time.perf_counter() - t0
                if not times or dt < min(times):
                    phases = orc.phase_ms()
                times.append(dt)
            best = min(times)
            d = oout.astype(np.float64) - out
            d[:, 2] = np.angle(np.exp(1j * d[:, 2]))
            evals = float(n_matches_l) * so.linearize_calls
            result["cpu_baseline"] = {
                "value": n_obs * so.iterations / best, "unit": "obs-iters/s", "cores": cores, "kind": "port",
                "sample": "the WHOLE bench workload (%d poses x %d obs, %d matches, %d constraint rows, mode C), oracle = C/OpenMP "
                          "restatement (analytic Jacobians, exact banded Cholesky; -O3 -march=native), %d threads, 1 warm-up + best "
                          "of %d: %.1f ms per re-solve, %d LM iterations (GPU: %d)" % (n_local, M, int(n_matches_l), rows[1], cores,
                                                                                     reps, best * 1e3, so.iterations, summ.iterations),
                "resolve_ms": best * 1e3, "resolve_ms_all_repeats": [t * 1e3 for t in times], "warmup_ms": first * 1e3,
                "lm_iterations": int(so.iterations), "lm_iterations_gpu": int(summ.iterations),
                "final_cost": so.final_cost, "final_cost_gpu": summ.final_cost,
                "max_pose_difference_vs_gpu_m_rad": [float(np.abs(d[:, :2]).max()), float(np.abs(d[:, 2]).max())],
                "phase_ms_best_run": phases,
                "match_evaluations_per_s_per_core": evals / (phases["evaluate"] * 1e-3) / cores if phases and phases["evaluate"] > 0 else None,
                "host": {"cpu_model": host.get("cpu_model"), "os_cpu_count": host.get("os_cpu_count"), "affinity": host.get("affinity"),
                         "cgroup_cpu_max": host.get("/sys/fs/cgroup/cpu.max"), "threads_used": cores,
                         "OMP_PROC_BIND": host.get("OMP_PROC_BIND"), "OMP_NUM_THREADS_env": host.get("OMP_NUM_THREADS")},
            }
            result["speedup_vs_cpu_port"] = result["value"] / result["cpu_baseline"]["value"]
            result["resolve_ms_cpu_port_over_gpu"] = best * 1e3 / ms_per_step
            # 1 thread, bounded sample: the first ns poses of the same workload (pairs inside the sample)
            ns = min(args.cpu_sample_poses, n_local)
            ns = max((ns // ppf) * ppf, min(ppf, n_local))
            if ns > 0:
                ps, pt, mo, i0, i1 = corr
                keep_idx = np.nonzero((ps < ns) & (pt < ns))[0]
                lens = (mo[1:] - mo[:-1])[keep_idx]
                mo_s = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
                take = np.concatenate([np.arange(mo[k], mo[k + 1]) for k in keep_idx]) if len(keep_idx) else np.zeros(0, np.int64)
                rows_s = S.constraints(p, gt, max(1, int(round(wl["groups"] * ns / n_local))), kind=wl["kind"], begin=0, end=ns)
                o1 = O.OracleSolver()
                o1.set_map(off[:ns + 1], px[:off[ns]], py[:off[ns]], nx[:off[ns]], ny[:off[ns]], cov[:ns])
                o1.set_human_constraints(rows_s)
                o1.set_correspondences(ps[keep_idx], pt[keep_idx], mo_s, i0[take], i1[take])
                O.set_num_threads(1)
                t0 = time.perf_counter()
                _, s1 = o1.solve(H.MODE_JOINT, np.ascontiguousarray(init_l[:ns]))
                t1 = time.perf_counter() - t0
                ph1 = o1.phase_ms()
                result["cpu_baseline"]["one_thread_sample"] = {
                    "poses": ns, "matches": int(mo_s[-1]), "resolve_ms": t1 * 1e3, "lm_iterations": int(s1.iterations),
                    "obs_iters_per_s": float(off[ns]) * s1.iterations / t1,
                    "match_evaluations_per_s": float(mo_s[-1]) * s1.linearize_calls / (ph1["evaluate"] * 1e-3) if ph1["evaluate"] > 0 else None}
                o1.close()
                O.set_num_threads(cores)
            orc.close()
        except Exception as e:  # the baseline is a reported extra; never fail the bench line on it
            result["cpu_baseline"] = {"value": None, "unit": "obs-iters/s", "cores": None, "kind": "port",
                                      "sample": "failed: %r" % (e,)}
    print(json.dumps(result))
    sys.stdout.flush()
    solver.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
