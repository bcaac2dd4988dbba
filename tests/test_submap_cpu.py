"""N > 1 path on CPU (gloo, world_size 2): the product's pose-range partitioner plus a numpy
mirror of the submap exchange (interface width 1: only the odometry edge couples the ranks) --
one all-gather of the interface rows and cut blocks, the 6*world interface system, rank-ordered
scalar sums -- on systems assembled by the oracle.  The
HIP implementation of the same algebra is checked end-to-end in tests/test_submap_gpu.py."""
import json
import os
import subprocess
import sys


def _free_port():
    """A port nobody is listening on (rendezvous of the worker processes on 127.0.0.1)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_interface_solve_reproduces_the_global_step(tmp_path):
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_submap_cpu_worker.py"), out]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [json.load(open("%s.%d" % (out, k))) for k in range(2)]
    for x in res:
        assert x["err"] < 1e-9                      # local step == rows of the global solve
        assert x["cov_err"] < 1e-8                  # local selected inverse - interface correction == blocks of the global inverse
        assert abs(x["tot"][0] - x["ref"][0]) <= 1e-9 * abs(x["ref"][0])
        assert abs(x["tot"][1] - x["ref"][1]) <= 1e-9 * abs(x["ref"][1])
    assert res[0]["tot"] == res[1]["tot"]           # bit-identical decisions on every rank


def test_separator_solve_reproduces_the_global_step(tmp_path):
    """Round 4: the separator form of the submap linear solve on three ranks over gloo (a middle rank with a separator pair on both
    sides), cuts from hitl_partition_range_aligned inside ONE corridor, so scan pairs three poses long cross both of them: the
    ranks' steps equal the rows of the global solve, the neighbours' separators deliver the halo steps, the sums agree bit for bit."""
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_submap_sep_cpu_worker.py"), out]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [json.load(open("%s.%d" % (out, k))) for k in range(3)]
    assert [x["begin"] for x in res] == [0, 24, 48] and res[2]["end"] == 73
    for x in res:
        assert x["err"] < 1e-9 and x["halo_err"] < 1e-9
        assert abs(x["tot"][0] - x["ref"][0]) <= 1e-9 * abs(x["ref"][0])
        assert abs(x["tot"][1] - x["ref"][1]) <= 1e-9 * abs(x["ref"][1])
    assert res[0]["tot"] == res[1]["tot"] == res[2]["tot"]


def test_distributed_pcg_reproduces_the_global_step(tmp_path):
    """Off-band pose pairs on submap ranks: the ranks iterate together (block-Jacobi PCG, two all-gathers per iteration).
    numpy mirror of hitl_lm.cpp's dist_pcg_solve over gloo; the HIP implementation is checked end to end in
    tests/test_submap_offband_gpu.py."""
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_submap_pcg_cpu_worker.py"), out]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [json.load(open("%s.%d" % (out, k))) for k in range(2)]
    for x in res:
        assert x["err"] < 1e-9 and 1 < x["iters"] < 200     # owned rows of the global solve, after a real iteration
    assert res[0]["decisions"] == res[1]["decisions"]       # bit-identical scalars -> identical decisions on every rank
