"""hostcheck, pose-range submaps (CPU only): W processes, each one rank of a submap session on the sanitizer build of the library's
HOST code (hostcheck/libhitlgpu_hostcheck.so: AddressSanitizer + UBSan, a stub device layer underneath), joined by the caller
all-gather callback of the C ABI (hitl_solver_set_comm) over shared memory.  What this covers that tests/hostcheck_driver.py
(one rank) cannot: the partition bookkeeping, halo and owner tables, the exchange records of prepare() (interface widths, the
agreement on the linear-solve form, the separator / interface-column buffers), the LM loop's collectives and the agreement on
failure -- every staging buffer of the rank code path under the sanitizers, and the ORDER of the collectives on every rank: a
rank that leaves a collective the others still wait in shows up here as a barrier timeout, not as a hung GPU box.
No kernel runs under the stub: numerical results are meaningless and not looked at.  Run by tests/hostcheck_run.sh."""
import multiprocessing as mp
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "hostcheck" in os.environ.get("HITL_LIBHITLGPU", ""), "run through tests/hostcheck_run.sh"

SLOT = 1 << 20            # bytes per rank in the shared exchange buffer
WAIT_S = 120.0            # a collective nobody completes within this is a deadlock


def shm_allgather_ops(rank, world, barrier, shm):
    """hitl_comm_ops over a shared buffer and a barrier: write the own record, wait, read all, wait."""
    import ctypes as C
    from hitl_slam_amd.api import ALLGATHER_FN, CommOps
    base = C.addressof(C.c_char.from_buffer(shm))
    stats = {"exchanges": 0, "bytes": 0}

    def _ag(ctx, send, recv, nbytes):
        try:
            stats["exchanges"] += 1
            stats["bytes"] += int(nbytes)
            if nbytes > SLOT:
                sys.stderr.write("exchange record of %d bytes exceeds the test's slot\n" % nbytes)
                return 1
            C.memmove(base + rank * SLOT, send, nbytes)
            barrier.wait(WAIT_S)
            for r in range(world):
                C.memmove(recv + r * nbytes, base + r * SLOT, nbytes)
            barrier.wait(WAIT_S)
            return 0
        except Exception as e:      # BrokenBarrierError = some rank never arrived: reported as a failed exchange
            sys.stderr.write("rank %d: all-gather callback failed: %r\n" % (rank, e))
            return 1

    fn = ALLGATHER_FN(_ag)
    return CommOps(fn, None), (fn, stats)


def tolerant(fn, *a, **k):
    from hitl_slam_amd.api import HitlError
    try:
        return fn(*a, **k)
    except (HitlError, AssertionError) as e:       # a status code is a fine outcome under the stub; a sanitizer report aborts
        return e


def rank_main(rank, world, barrier, shm, case, q):
    try:
        for k, v in case.get("env", {}).items():
            os.environ[k] = v
        import hitl_slam_amd as H
        from hitl_slam_amd import abi
        from hitl_slam_amd.api import HitlError
        if case.get("inject_rank") == rank:
            H.set_test_switch(case["inject"][0], case["inject"][1])
        rng = np.random.default_rng(11)                 # the same stream on every rank: one global map
        n, m, window = case["n"], case["m"], case["window"]
        cuts = case["cuts"]
        assert len(cuts) == world + 1 and cuts[-1] == n
        halo = case.get("halo", 2 * window)
        b, e = cuts[rank], cuts[rank + 1]
        replicated = halo < 0
        lo, hi = (0, n) if replicated else (b - (halo if rank > 0 else 0), e + (halo if rank < world - 1 else 0))
        lo, hi = max(lo, 0), min(hi, n)
        counts = rng.integers(max(1, m - 4), m + 1, n)
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        O = int(off[-1])
        px, py = rng.normal(0, 3, O).astype(np.float32), rng.normal(0, 3, O).astype(np.float32)
        ang = rng.uniform(-3, 3, O)
        nx, ny = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
        cov = np.tile(np.eye(3, dtype=np.float32).reshape(-1) * 1e-4, (n, 1))
        poses = np.stack([np.arange(n) * 0.3, rng.normal(0, 0.01, n), rng.normal(0, 0.01, n)], 1).astype(np.float32)
        rows = np.zeros(9, abi.HC_DTYPE)
        rows["type"] = [2, 4, 5, 6, 2, 4, 5, 6, 2]
        rows["constrained_pose_id"] = rng.integers(1, n, 9)
        rows["anchor_pose_id"] = rng.integers(0, n, 9)
        rows["delta_parallel"] = 0.1

        o = H.default_options()
        o.rank, o.world_size = rank, world
        for k, v in case.get("opts", {}).items():
            setattr(o, k, v)
        s = H.Solver(o)
        ops, keep = shm_allgather_ops(rank, world, barrier, shm)
        s.set_comm(ops, keep)
        if replicated or case.get("range_call"):
            s.set_partition_range(b, e, n, halo)
        else:
            s.set_partition(b, n, halo)
        sl = slice(int(off[lo]), int(off[hi]))
        s.set_map(off[lo:hi + 1] - off[lo], px[sl], py[sl], nx[sl], ny[sl], cov[lo:hi])
        s.set_human_constraints(rows)                   # global pose ids
        done = 0
        tolerant(s.comm_selftest); done += 1
        tolerant(s.find_correspondences, poses, window); done += 1
        # a caller-provided pair list in GLOBAL pose ids: pairs inside this rank's range, across each of its cuts (both
        # directions), and -- with replicated scans -- one far across (the off-band / distributed-PCG bookkeeping)
        ps, pt = [], []
        for c in sorted(set([b, e])):
            for d in range(1, max(window, 1) + 1):
                if c - d >= lo and c + d - 1 < hi and c - d >= 0 and c + d - 1 < n:
                    ps += [c - d, c + d - 1]; pt += [c + d - 1, c - d]
        if e - b >= 3:
            ps += [b, b + 1]; pt += [b + 2, b]
        if replicated and n > 3 * max(window, 1) + 4:
            ps += [b]; pt += [(b + n // 2) % n]
        keepm = [(a_, b_) for a_, b_ in zip(ps, pt) if a_ != b_ and b <= a_ < e]      # a pair is listed by the rank owning its source
        if keepm:
            ps = np.array([a_ for a_, _ in keepm], np.int32); pt = np.array([b_ for _, b_ in keepm], np.int32)
            lens = rng.integers(11, 16, len(ps))
            mo = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            i0 = np.concatenate([rng.integers(0, counts[a_], l) for a_, l in zip(ps, lens)]).astype(np.int32)
            i1 = np.concatenate([rng.integers(0, counts[b_], l) for b_, l in zip(pt, lens)]).astype(np.int32)
            r = tolerant(s.set_correspondences, ps, pt, mo, i0, i1); done += 1
            assert not isinstance(r, Exception), "the pair list across the cuts was refused: %r" % (r,)
        tolerant(s.get_correspondences); done += 1
        outcomes = []
        for mode in case.get("modes", (H.MODE_ODOM_HC, H.MODE_STF_ONLY, H.MODE_JOINT)):
            r = tolerant(s.solve, mode, poses, allow_failure=True)
            outcomes.append(r.status if isinstance(r, HitlError) else 0)
            r = tolerant(s.evaluate, mode, poses)
            outcomes.append(r.status if isinstance(r, HitlError) else 0)
            done += 2
        r = tolerant(s.pose_covariances, H.MODE_ODOM_HC, poses); done += 1
        outcomes.append(r.status if isinstance(r, HitlError) else 0)
        tolerant(s.transform_scans, poses); done += 1
        tolerant(s.verify_points, poses, np.zeros((4, 2), np.float32), 0.05); done += 1
        tolerant(s.observation_sets, poses, np.array([0, 0, 1, 0, 0, 1, 1, 1], np.float32), 0.03, True); done += 1
        tolerant(s.fit_segments, poses, np.array([[0, 0, 1, 0], [0, 1, 1, 1]], np.float32)); done += 1
        tolerant(s.backprop, poses, cov, 1, n - 2, np.array([0.1, 0.0, 0.01], np.float32)); done += 1
        # a second session step with another constraint list (undo = a shorter list) and the same map
        s.set_human_constraints(rows[:4])
        r = tolerant(s.solve, H.MODE_ODOM_HC, poses, allow_failure=True); done += 1
        outcomes.append(r.status if isinstance(r, HitlError) else 0)
        s.close()
        outcomes.append(keep[1]["exchanges"])           # the number of collectives is part of what must agree
        q.put((rank, "ok", done, outcomes))
    except BaseException:
        q.put((rank, "exception", traceback.format_exc(), []))


CASES = [
    dict(name="two ranks, separator form", world=2, n=64, m=20, window=3, cuts=[0, 30, 64]),
    dict(name="two ranks, interface columns", world=2, n=64, m=20, window=3, cuts=[0, 30, 64], env={"HITL_DIST_COLUMNS": "1"}),
    dict(name="three ranks, uneven, band of one pose", world=3, n=90, m=12, window=2, cuts=[0, 20, 61, 90], opts=dict(max_band_poses=1)),
    dict(name="three ranks, replicated scans, all pairs", world=3, n=48, m=10, window=0, cuts=[0, 16, 32, 48], halo=-1),
    dict(name="two ranks, moments mode, range call", world=2, n=40, m=16, window=1, cuts=[0, 19, 40], opts=dict(stf_moments=1), range_call=True),
    dict(name="two ranks, timing level 2", world=2, n=40, m=16, window=4, cuts=[0, 24, 40], opts=dict(timing_level=2)),
    dict(name="four ranks (the C4 split of BASELINE configs[3])", world=4, n=120, m=10, window=3, cuts=[0, 30, 60, 90, 120]),
    dict(name="eight ranks (the C5 split), uneven, one rank of 9 poses", world=8, n=200, m=8, window=3,
         cuts=[0, 25, 50, 59, 90, 115, 140, 170, 200]),
    dict(name="eight ranks, interface columns", world=8, n=200, m=8, window=2, cuts=[0, 25, 50, 75, 100, 125, 150, 175, 200],
         env={"HITL_DIST_COLUMNS": "1"}, modes=(0, 2)),
    dict(name="two ranks, rank 1 fails in prepare", world=2, n=64, m=20, window=3, cuts=[0, 30, 64], inject_rank=1, inject=("fail_prepare", 1),
         expect_all_fail=True),
    dict(name="three ranks, rank 0 reports a bad step from iteration 2", world=3, n=90, m=12, window=2, cuts=[0, 20, 61, 90], inject_rank=0,
         inject=("poison_step_from", 2)),
]


def run_case(case):
    ctx = mp.get_context("fork")              # ranks are forked BEFORE anything loads the library
    world = case["world"]
    barrier = ctx.Barrier(world)
    shm = ctx.RawArray("b", world * SLOT)
    q = ctx.Queue()
    procs = [ctx.Process(target=rank_main, args=(r, world, barrier, shm, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(world):
            r = q.get(timeout=600)
            res[r[0]] = r
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.terminate()
    for p in procs:
        assert p.exitcode == 0, "%s: a rank ended with exit code %r (sanitizer report above?)" % (case["name"], p.exitcode)
    assert len(res) == world, "%s: ranks missing: %r" % (case["name"], sorted(res))
    for r in range(world):
        assert res[r][1] == "ok", "%s: rank %d: %s" % (case["name"], r, res[r][2])
    # every collective call ended the same way on every rank (one global LM loop, agreed failures)
    outs = [res[r][3] for r in range(world)]
    assert all(o == outs[0] for o in outs), "%s: ranks disagree on their status codes: %r" % (case["name"], outs)
    assert outs[0][-1] >= 5, "%s: the sessions made no exchanges: %r" % (case["name"], outs)
    if case.get("expect_all_fail"):
        assert all(st != 0 for st in outs[0][:2]), "%s: the injected failure was not reported by every rank: %r" % (case["name"], outs)
    return sum(res[r][2] for r in range(world)), outs[0][-1]


if __name__ == "__main__":
    total = exchanges = 0
    for case in CASES:
        calls, ex = run_case(case)
        total += calls
        exchanges += ex
        print("hostcheck ranks: %-62s ok (%d exchanges per rank)" % (case["name"], ex), flush=True)
    print("hostcheck ranks: %d calls and %d exchanges over %d multi-rank sessions, sanitizers silent, no rank left waiting" % (total, exchanges, len(CASES)))
