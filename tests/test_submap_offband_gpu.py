"""Pose-range submaps whose correspondence sets leave the pose band (the reference's search,
JointOptimization.cpp:575-635, pairs every pose with every other): the ranks then solve the LM step together by
preconditioned CG around their own band factorisations instead of the exact interface system.  Same harness as
test_submap_gpu.py (ranks share the single GPU of the test box, exchanges go through the all-gather callback over gloo);
the distributed result must equal the oracle's GLOBAL solve on the same correspondence definition.

  band1   : window 3 with the band capped at 1 pose -> pairs at distance 2 and 3 are off the band, inside and across the cuts
  allpairs: two ranks, each holding the whole map (halo = the other rank's range), all-pairs search: the reference's own
            configuration on a single corridor, where poses see each other far beyond any band
  *_replicated_3ranks: three ranks with REPLICATED SCANS (hitl_solver_set_partition_range, halo < 0): every rank holds the
            whole map and owns the rows of its pose range, so a pair may reach any distance across both cuts"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import Problem
from test_submap_gpu import _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    # name: (world, poses per rank, halo, window, band cap, poses per floor)
    "band1_2ranks": (2, 60, 6, 3, 1, 120),
    "band1_3ranks": (3, 40, 6, 3, 1, 120),
    "band1_floor_cut": (2, 60, 6, 3, 1, 30),
    "allpairs_replicated": (2, 12, 12, 0, 4, 24),
    "band1_uneven_3ranks": (3, 40, 6, 3, 1, 120),      # ranks own 23, 60 and 37 poses (HITL_TEST_CUTS below)
    # replicated scans (halo < 0, round 4): pose pairs that reach deeper across the cuts than any halo, on three ranks --
    # the reference's all-pairs loop over a single corridor, and a window of 8 poses with the band capped at 1
    "allpairs_replicated_3ranks": (3, 8, -1, 0, 4, 24),
    "window8_replicated_3ranks": (3, 30, -1, 8, 1, 90),
}
CUTS = {"band1_uneven_3ranks": "0,23,83,120"}


@pytest.mark.parametrize("case", sorted(CASES))
def test_offband_submaps_equal_the_global_solve(tmp_path, case):
    from oracle import hitl_oracle as O
    world, n_per, halo, window, band, ppf = CASES[case]
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HITL_TEST_NPER=str(n_per), HITL_TEST_HALO=str(halo),
               HITL_TEST_WINDOW=str(window), HITL_TEST_BAND=str(band))
    if case in CUTS:
        env["HITL_TEST_CUTS"] = CUTS[case]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_submap_worker.py"), out, "2", str(ppf)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [json.load(open("%s.%d" % (out, k))) for k in range(world)]

    prob = Problem(n_per * world, 48, ppf=ppf, groups=2, kind=0)
    o = prob.load(O.OracleSolver())
    o.find_correspondences(prob.init, window, 1)
    ops = o.get_correspondences()
    src = np.concatenate([np.array(x["pairs"][0], np.int32) for x in res])
    tgt = np.concatenate([np.array(x["pairs"][1], np.int32) for x in res])
    assert np.array_equal(src, ops[0]) and np.array_equal(tgt, ops[1])       # the ranks' pairs tile the global set
    assert np.abs(ops[0] - ops[1]).max() > band                              # and some of them are off the band
    po, so = o.solve(2, prob.init)
    for x in res:
        sm = x["summary"]
        assert sm["num_offband_pairs"] > 0 and sm["linear_iterations"] > sm["iterations"]     # the ranks really iterated
        assert sm["termination"] == so.termination
        assert abs(sm["iterations"] - so.iterations) <= 1          # inexact steps: the last tolerance test may fall one iteration apart
        assert abs(sm["initial_cost"] - so.initial_cost) <= 1e-10 * so.initial_cost
        assert abs(sm["final_cost"] - so.final_cost) <= 1e-6 * so.final_cost
        got = np.array(x["poses"], np.float32)[x["begin"]:x["end"]]
        assert np.abs(got[:, :2] - po[x["begin"]:x["end"], :2]).max() <= 1e-4
        assert np.abs(got[:, 2] - po[x["begin"]:x["end"], 2]).max() <= 1e-4
    print("%s: oracle %d LM iterations, cost %.9g | ranks: %s" % (case, so.iterations, so.final_cost, [(x["summary"]["iterations"],
          x["summary"]["linear_iterations"], x["summary"]["num_offband_pairs"], "%.9g" % x["summary"]["final_cost"]) for x in res]))
    assert len({(x["summary"]["iterations"], x["summary"]["termination"], x["summary"]["final_cost"]) for x in res}) == 1   # identical on all ranks


def test_a_pair_reaching_past_the_halos_is_refused_on_every_rank(tmp_path):
    """Window 3 with halos of 2 poses: the pair (57, 60) straddles the cut at 60 and reaches 3 poses into rank 0, where rank 1
    cannot see it.  Both ranks must come back with HITL_ERR_UNSUPPORTED (agreed in prepare's exchange) instead of one of them
    waiting in the first collective of the LM loop."""
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HITL_TEST_NPER="60", HITL_TEST_HALO="2",
               HITL_TEST_WINDOW="3", HITL_TEST_BAND="1", HITL_TEST_EXPECT_FAIL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_submap_worker.py"), out, "2", "120"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [json.load(open("%s.%d" % (out, k))) for k in range(2)]
    for x in res:
        assert "status 6" in x["error"] or "UNSUPPORTED" in x["error"].upper(), x["error"]
    assert any("halo" in x["error"] for x in res)
