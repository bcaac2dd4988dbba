"""Worker of tests/test_submap_gpu.py: one rank of a 2-submap solve.  Both ranks may share one
GPU (the exchange goes through the host-buffer all-gather callback over gloo), which is how the
N > 1 path is exercised on a 1-GPU box; on a multi-GPU node each rank takes its own device."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_path, mode, ppf = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import hitl_slam_amd as H
    from hitl_slam_amd.api import torch_allgather_ops
    from conftest import Problem

    n_per, halo = (60 if world == 2 else 40), 6     # halo = 2 x matching window
    # variants of tests/test_submap_offband_gpu.py: poses per rank, halo depth, search window (0 = all pairs), band cap
    n_per = int(os.environ.get("HITL_TEST_NPER", n_per))
    halo = int(os.environ.get("HITL_TEST_HALO", halo))
    window = int(os.environ.get("HITL_TEST_WINDOW", 3))
    if "HITL_TEST_WINDOWS" in os.environ:          # a search window per rank ("1,3,3"): a rank whose own pairs are all short
        window = int(os.environ["HITL_TEST_WINDOWS"].split(",")[rank])
    prob = Problem(n_per * world, 48, ppf=ppf, groups=2, kind=0)
    b, e = rank * n_per, (rank + 1) * n_per
    if "HITL_TEST_CUTS" in os.environ:            # uneven submaps: "0,50,120" = rank 0 owns [0, 50), rank 1 [50, 120)
        cuts = [int(x) for x in os.environ["HITL_TEST_CUTS"].split(",")]
        assert len(cuts) == world + 1 and cuts[-1] == n_per * world
        b, e = cuts[rank], cuts[rank + 1]
    lo, hi = b - (halo if rank > 0 else 0), e + (halo if rank < world - 1 else 0)
    replicated = halo < 0                         # every rank holds the whole map (hitl_solver_set_partition_range, halo < 0)
    if replicated:
        lo, hi = 0, n_per * world
    M = 48
    if os.environ.get("HITL_TEST_COLUMNS_RANK") == str(rank):       # ONE rank asks for the interface-column form: all must follow
        os.environ["HITL_DIST_COLUMNS"] = "1"
    # failure injection on ONE rank (tests/test_submap_gpu.py::test_a_failure_on_one_rank_...): "hook=value" for the library's test
    # hooks (hitl_test_set_switch; per process), or "nan_pose" = this rank hands in a NaN at one of its own poses
    inject = os.environ.get("HITL_TEST_INJECT", "")
    inject_here = inject and os.environ.get("HITL_TEST_INJECT_RANK") == str(rank)
    if inject_here and "=" in inject:
        k, _, v = inject.partition("=")
        H.set_test_switch(k, int(v))
    opt = H.default_options()
    opt.rank, opt.world_size = rank, world
    opt.device = rank % max(torch.cuda.device_count(), 1)
    opt.stf_moments = int(os.environ.get("HITL_TEST_STF_MOMENTS", "0"))
    if "HITL_TEST_BAND" in os.environ:
        opt.max_band_poses = int(os.environ["HITL_TEST_BAND"])
    s = H.Solver(opt)
    ops, keep = torch_allgather_ops()
    s.set_comm(ops, keep)
    if replicated:
        s.set_partition_range(b, e, prob.n, -1)
    else:
        s.set_partition(b, prob.n, halo)
    sl = slice(lo * M, hi * M)
    s.set_map(np.arange(hi - lo + 1, dtype=np.int64) * M, prob.px[sl], prob.py[sl], prob.nx[sl], prob.ny[sl], prob.cov[lo:hi])
    s.set_human_constraints(prob.rows)
    s.find_correspondences(prob.init, window)
    corr = s.get_correspondences()
    own = (corr[0] >= b) & (corr[0] < e)             # every pair is reported by the rank owning its source pose
    lens = corr[2][1:] - corr[2][:-1]
    if os.environ.get("HITL_TEST_EXPECT_FAIL"):
        # a solve every rank must refuse together (no rank may be left waiting in a collective)
        pin = np.array(prob.init, np.float32)
        if inject_here and inject == "nan_pose":
            pin[(b + e) // 2, 0] = np.nan
        status, err = 0, ""
        try:
            s.solve(mode, pin)
        except H.HitlError as ex:
            status, err = ex.status, str(ex)
        json.dump(dict(rank=rank, error=err, status=status), open("%s.%d" % (out_path, rank), "w"))
        s.close()
        dist.destroy_process_group()
        return
    ecost, egrad = s.evaluate(mode, prob.init)        # collective: global cost, this rank's owned gradient rows
    cov = s.pose_covariances(mode, prob.init) if window <= 4 and "HITL_TEST_BAND" not in os.environ and mode != 1 else None   # collective; needs every pair inside the band (mode B: J^T J may be singular)
    poses, summ = s.solve(mode, prob.init)
    # observation scans on a submap rank: over its resident scans (global poses [lo, hi)), global pose array in
    wx, wy = s.transform_scans(prob.init)
    probe = np.array([[wx[5], wy[5]], [wx[-7], wy[-7]], [1e6, 1e6]], np.float32)
    seen = s.verify_points(prob.init, probe, 0.05)
    seg = np.array([wx[3], wy[3], wx[40], wy[40], wx[60], wy[60], wx[90], wy[90]], np.float32)
    counts, mask = s.observation_sets(prob.init, seg, 0.03)
    # endpoint fit across the ranks: every rank passes the SAME two clicks (rank 1's, broadcast: two points of its first
    # owned scan, i.e. of the pose right above the first cut -- a wall piece seen from poses on both sides of the cut where
    # the corridor continues) and must come back with the same fitted segment
    fb = (b - lo) * M                                   # first owned scan of this rank
    clicks = torch.tensor([wx[fb + 3], wy[fb + 3], wx[fb + 40], wy[fb + 40]], dtype=torch.float32)
    dist.broadcast(clicks, 1)
    fseg, finl, fpas, fits = s.fit_segments(prob.init, clicks.numpy().reshape(1, 4))
    res = dict(rank=rank, begin=b, end=e, poses=poses.tolist(), summary=summ.as_dict(),
               pairs=[corr[0][own].tolist(), corr[1][own].tolist()], n_matches=int(lens[own].sum()),
               scans=dict(lo=lo, hi=hi, wx=wx.tolist(), wy=wy.tolist(), probe=probe.tolist(), seen=seen.tolist(), seg=seg.tolist(),
                          counts=counts.tolist(), mask_sum=int(mask.sum())),
               evaluate=dict(cost=ecost, grad=egrad.tolist()), cov=(cov.tolist() if cov is not None else None),
               fit=dict(clicks=clicks.numpy().tolist(), seg=fseg.tolist(), inliers=finl.tolist(), passes=fpas.tolist(), its=fits.tolist()))
    json.dump(res, open("%s.%d" % (out_path, rank), "w"))
    s.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
