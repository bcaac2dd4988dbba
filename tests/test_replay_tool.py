"""tools/hitl_replay.c: the reference binary's batch path (load pose graph, load recording, replay, save poses) as a
plain C program over the C ABI.  Builds with gcc, fails loudly without a GPU, reproduces the Python-driven cycle on one."""
import os
import subprocess

import numpy as np
import pytest

from hitl_slam_amd import io as hio
from tests.cycle_scenario import Scenario

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hitl_slam_amd", "csrc")


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("replay") / "hitl_replay")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "hitl_replay.c"),
                           "-o", exe, "-L", CSRC, "-lhitlgpu", "-lm", "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


@pytest.fixture(scope="module")
def session(tmp_path_factory):
    sc = Scenario()
    sc2 = Scenario(first_piece=(10.5, 17.5))
    tmp = tmp_path_factory.mktemp("replaycase")
    pr = sc.prob
    mpath, lpath = str(tmp / "corridor.stfs.covars"), str(tmp / "session.log")
    hio.save_stfs_covars(mpath, "corridor", 1.0, sc.poses, pr.off, pr.px, pr.py, pr.nx, pr.ny, sc.cov)
    hio.save_input_log(lpath, [dict(type=4, undone=0, points=sc.selected.reshape(4, 2)), dict(type=3, undone=0, points=np.zeros((8, 2), np.float32)),
                               dict(type=4, undone=0, points=sc2.selected.reshape(4, 2))])
    return mpath, lpath, str(tmp / "out.txt")


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_replay_tool_builds_and_fails_loudly_without_a_gpu(tool, session):
    mpath, lpath, out = session
    r = subprocess.run([tool, mpath, lpath, out], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr
    assert not os.path.exists(out)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 2])
def test_replay_tool_matches_the_python_driven_cycle(tool, session, mode):
    from hitl_slam_amd import Solver
    mpath, lpath, out = session
    r = subprocess.run([tool, mpath, lpath, out, str(mode), "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "entry 1: type 3 with 8 points is not supported" in r.stdout
    got = np.loadtxt(out, dtype=np.float32).reshape(-1, 3)
    g = hio.load_stfs_covars(mpath, fix_normals=True)
    s = Solver()
    s.set_map(g["obs_offset"], g["px"], g["py"], g["nx"], g["ny"], g["cov9"])
    poses, cov, groups, statuses = s.replay_log(hio.load_input_log(lpath), g["poses"], g["cov9"], mode=mode, stf_window=3)
    s.close()
    assert statuses[0] == "ok" and statuses[1] == "unsupported"
    assert r.stdout.count("iterations") == statuses.count("ok")
    assert np.abs(got - poses).max() <= 2e-6 + 1e-6 * np.abs(poses).max()       # "%f" in the results file
