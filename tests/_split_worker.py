"""Worker of tests/test_gpu_splits_fullsize.py: one rank of a BASELINE.json configuration run as the split it names --
bench.py's own partition plan (bench.plan_partition: configs[3] = the 50k x 400 map cut into 4 pose-range submaps,
configs[4] = the 200k x 512 map cut into 8), the ranks sharing this box's GPU and exchanging through the host-buffer
all-gather callback over gloo.  WORLD_SIZE unset or 1: the same map on one rank (the single-GPU solve the split is held to).
Writes <out>.<rank>.npz: the rank's owned poses after the re-solve, its hitl_summary, its exchange count."""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    workload, out_path = sys.argv[1], sys.argv[2]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    import torch
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
    import hitl_slam_amd as H
    from hitl_slam_amd.api import torch_allgather_ops
    from tools import hitl_synth as S
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    plan = bench.plan_partition(workload, world)
    wl, M, ppf, n_global, cuts = plan["wl"], plan["obs"], plan["ppf"], plan["n_global"], plan["cuts"]
    p = S.params(n_global, M, seed=plan["seed"], poses_per_floor=ppf)
    gt, init, cov = S.poses(p)
    begin, end = cuts[rank], cuts[rank + 1]
    halo = 2 * wl["window"] if world > 1 else 0
    lo, hi = begin - (halo if rank > 0 else 0), end + (halo if rank < world - 1 else 0)
    off, px, py, nx, ny = S.scans(p, lo, hi)
    rows = S.constraints(p, gt, wl["groups"], kind=wl["kind"], begin=0, end=n_global)      # the whole map's rows, global pose ids
    opt = H.default_options()
    opt.device = rank % max(torch.cuda.device_count(), 1)
    counter = [0, 0]
    if world > 1:
        opt.rank, opt.world_size = rank, world
        s = H.Solver(opt)
        ops, keep = torch_allgather_ops(counter=counter)
        s.set_comm(ops, keep)
        s.set_partition(begin, n_global, halo)
        s.set_map(off, px, py, nx, ny, cov[lo:hi])
        pose_arg = init
    else:
        s = H.Solver(opt)
        s.set_map(off, px, py, nx, ny, cov)
        pose_arg = init
    s.set_human_constraints(rows)
    s.find_correspondences(pose_arg, wl["window"])
    n_pairs, n_matches = s.correspondence_counts()
    out, summ = s.solve(H.MODE_JOINT, pose_arg)                 # warm-up (allocations, first launches)
    before = list(counter)
    t0 = time.perf_counter()
    out, summ = s.solve(H.MODE_JOINT, pose_arg)
    ms = (time.perf_counter() - t0) * 1e3
    d = summ.as_dict()
    np.savez("%s.%d.npz" % (out_path, rank), poses=np.asarray(out, np.float32).reshape(-1, 3)[begin:end], begin=begin, end=end,
             exchanges=counter[0] - before[0], exchange_bytes=counter[1] - before[1], resolve_ms=ms, n_pairs=n_pairs, n_matches=n_matches,
             **{"s_" + k: v for k, v in d.items() if isinstance(v, (int, float))})
    s.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
