"""BASELINE.json configs[3] and configs[4] run as the splits they name, at full size: the 50 000 x 400
multi-floor map cut into 4 pose-range submaps and the 200 000 x 512 campus map cut into 8 -- bench.py's own partition plan
(cuts on multiples of 12 poses: a cut of configs[3] falls four poses inside a floor, so scan-to-scan pairs straddle it) -- the ranks
sharing this box's one GPU and exchanging through the all-gather callback over gloo.  Each is held to the single-GPU solve of the
same map, which tests/test_gpu_fullsize.py and tests/test_gpu_configs.py hold to the oracle: the submaps run ONE global
Levenberg-Marquardt loop with an exact linear solve (JointOptimization.cpp:542-558, 743-823: every factor touches at most two
poses, so the rows of the normal equations a rank owns are complete from its own factors), hence the same termination, the same
iteration and accepted-step counts, the same final cost and the same poses."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_split_worker.py")


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run(workload, world, out):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world == 1:
        cmd = [sys.executable, WORKER, workload, out]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), WORKER, workload, out]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [np.load("%s.%d.npz" % (out, k)) for k in range(world)]


@pytest.mark.parametrize("workload,world,poses,obs", [("c4", 4, 50000, 400), ("c5", 8, 200000, 512)], ids=["configs3_4_submaps", "configs4_8_submaps"])
def test_baseline_config_runs_as_the_split_it_names(tmp_path, workload, world, poses, obs):
    one = _run(workload, 1, str(tmp_path / "one"))[0]
    ranks = _run(workload, world, str(tmp_path / "split"))
    assert int(one["end"]) - int(one["begin"]) == poses
    # the ranks tile the map, in order, on cuts that are multiples of the 12-pose alignment
    assert int(ranks[0]["begin"]) == 0 and int(ranks[-1]["end"]) == poses
    for a, b in zip(ranks[:-1], ranks[1:]):
        assert int(a["end"]) == int(b["begin"]) and int(a["end"]) % 12 == 0
    # every rank solved its linear systems in the separator form and made the same number of exchanges (a rank that took another
    # path would have left the others waiting: the run would have timed out rather than returned)
    assert all(float(r["s_submap_linear_solver"]) == 2.0 for r in ranks)
    assert len({int(r["exchanges"]) for r in ranks}) == 1 and int(ranks[0]["exchanges"]) > 0
    # one global LM loop: every rank reports the same path, and it is the single-GPU path
    for key in ("s_termination", "s_iterations", "s_successful_steps", "s_unsuccessful_steps"):
        assert len({int(r[key]) for r in ranks}) == 1, key
        assert int(ranks[0][key]) == int(one[key]), (key, int(ranks[0][key]), int(one[key]))
    cost1 = float(one["s_final_cost"])
    for r in ranks:
        assert abs(float(r["s_final_cost"]) - cost1) <= 1e-8 * cost1
    # the problem is the same problem: pairs, matches and residual rows (hitl_solver_solve sums the ranks' owned counts -- each directed
    # pair is counted by the rank owning its source pose -- so every rank reports the global figures)
    for key in ("s_num_pairs", "s_num_matches", "s_num_residuals", "s_num_hc_residuals", "s_band_poses"):
        assert {int(r[key]) for r in ranks} == {int(one[key])}, key
    got = np.concatenate([r["poses"] for r in ranks]).astype(np.float64)
    d = got - one["poses"].astype(np.float64)
    d[:, 2] = np.angle(np.exp(1j * d[:, 2]))
    assert got.shape == (poses, 3)
    assert np.abs(d[:, :2]).max() <= 1e-4 and np.abs(d[:, 2]).max() <= 1e-4, (np.abs(d[:, :2]).max(), np.abs(d[:, 2]).max())
    print("%s on %d submaps: %d LM iterations, cost %.12g (single GPU %.12g), max |dxy| %.3g m, max |dtheta| %.3g rad, %d exchanges per rank, "
          "%.1f ms per re-solve with the ranks sharing one GPU (single GPU %.1f ms)"
          % (workload, world, int(one["s_iterations"]), float(ranks[0]["s_final_cost"]), cost1, np.abs(d[:, :2]).max(), np.abs(d[:, 2]).max(),
             int(ranks[0]["exchanges"]), max(float(r["resolve_ms"]) for r in ranks), float(one["resolve_ms"])))
