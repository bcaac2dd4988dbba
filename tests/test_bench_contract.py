"""The bench line contract (one JSON object: metric/value/unit/..., a `roofline` object -- `achieved` from the HBM bytes the
kernel really moves, the SURVEY 8(d) algorithmic-byte figure beside it -- and a `cpu_baseline` object), checked on the
committed lines of profiled runs (profiles/r02_final_bench_*.json, r05_final_bench_*.json, r06_final_bench_*.json) and on
bench.py's own source: the product path may not fall back to the oracle, the oracle is only the cpu_baseline leg."""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = [os.path.join(ROOT, "profiles", f) for f in ("r02_final_bench_c4.json", "r02_final_bench_c2.json", "r05_final_bench_c4.json", "r05_final_bench_c2.json",
                                                    "r06_final_bench_c4.json", "r06_final_bench_c2.json")]


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_bench_line_has_the_contract_fields(path):
    r = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None      # BASELINE.md publishes no number
    assert r["dtype"] == "f64" and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert r["value"] > 0 and r["ms_per_step"] > 0 and r["n_gpus"] == 1
    rf = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1      # a fraction of a physical peak
    # achieved = bytes really moved per launch (PMC-measured, else the layout's compulsory bytes) / measured launch time
    assert abs(rf["achieved"] - rf["bytes_per_launch_used_for_achieved"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert rf["bytes_per_launch_used_for_achieved"] in (rf["traffic"], rf["compulsory_bytes_this_layout"])
    assert (rf["traffic"] is None) == (rf["traffic_source"] is None)
    assert abs(rf["algorithmic_achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * rf["algorithmic_achieved"]
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["unit"] == r["unit"] and cb["cores"] >= 1
    assert "extrapolated" not in json.dumps(cb) and cb["lm_iterations"] == cb["lm_iterations_gpu"] == r["lm_iterations_per_resolve"]
    assert cb["cores"] == cb["host"]["threads_used"] <= cb["host"]["affinity"]


def test_committed_c3_line_has_the_contract_fields():
    """`bench.py --workload c3` (BASELINE configs[2]: the PCG path): the same contract; `roofline` is the persistent PCG launch, per CG
    iteration -- `achieved` = SURVEY 8(d)'s algorithmic bytes of one CG iteration / its measured time, `traffic` = the HBM-side
    bytes of the counter passes per CG iteration; `cpu_baseline` is a bounded sample (the first LM iterations of the same re-solve)."""
    r = json.load(open(os.path.join(ROOT, "profiles", "r06_final_bench_c3.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["higher_is_better"] is True and r["vs_baseline"] is None and r["dtype"] == "f64" and r["n_gpus"] == 1
    assert "configs[2]" in r["config"]["workload"] and "model" not in r["config"] and r["config"]["window"] == 0
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"] == "k_pcg_fused"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_cg_iteration"] / (rf["us_per_cg_iteration"] * 1e-6) / 1e9) < 1e-6 * rf["achieved"]
    assert abs(rf["us_per_cg_iteration"] * rf["cg_iterations_per_launch"] - 1e3 * rf["avg_launch_ms"]) < 1e-6 * 1e3 * rf["avg_launch_ms"]
    assert rf["traffic"] is not None and rf["traffic_source"] and rf["traffic"] > 0.5 * rf["algorithmic_bytes_per_cg_iteration"]
    assert r["cg_iterations_per_resolve"] > 5 * r["lm_iterations_per_resolve"]          # the PCG really iterates on this workload
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["unit"] == r["unit"] and cb["cores"] >= 1 and cb["whole_resolve"] is False and cb["lm_iterations"] <= r["lm_iterations_per_resolve"]


def test_bench_source_keeps_the_oracle_in_its_lane():
    src = open(os.path.join(ROOT, "bench.py")).read()
    # the oracle is imported only inside a cpu_baseline leg (one per bench leg: the default workloads and --workload c3), behind the
    # --no-cpu-baseline switch, and what it produces goes into result["cpu_baseline"] and nowhere else
    hits = [m.start() for m in re.finditer(r"from oracle import", src)]
    assert len(hits) == 2
    for i in hits:
        assert "CPU baseline" in src[max(0, i - 900):i + 200] and "no_cpu_baseline" in src[max(0, i - 900):i]
        assert 'result["cpu_baseline"]' in src[i:i + 6000]
    assert "no CPU fallback" in src          # a missing GPU is an error, not a detour


def _run_bench(args, env_extra=None, timeout=900):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """`--gpus N` inside a launcher that started another number of ranks is refused before anything touches a GPU: the line may never
    say n_gpus = N for a run of fewer ranks (a one-rank run used to be labelled 8 GPUs)."""
    r = _run_bench(["--gpus", "4"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode == 4 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""
    r = _run_bench(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode == 4 and r.stdout.strip() == ""


@pytest.mark.gpu
def test_bench_started_directly_with_gpus_2_launches_two_ranks():
    """`python bench.py --gpus 2 --backend gloo` with no launcher around it: bench.py starts the two ranks itself (children of a
    parent that never touched the GPU), they share this box's GPU over gloo, and the line says what ran: n_gpus = comm_ranks = 2."""
    r = _run_bench(["--gpus", "2", "--backend", "gloo", "--scaling", "weak", "--poses", "6000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["config"]["comm_ranks"] == 2 and j["config"]["ranks_share_gpus"] is True and j["scaling"] == "weak"
    assert j["config"]["n_poses_total"] == 12000 and j["lm_iterations_per_resolve"] > 0 and j["value"] > 0
    # the cut (6000) is a multiple of the pose band: the ranks' linear solves take the separator form; the interface-column form
    # (rounds 2-3, HITL_DIST_COLUMNS=1) solves the same systems: same LM path
    assert j["config"]["submap_linear_solver"] == "separators" and j["config"]["cuts"] == [0, 6000, 12000]
    r2 = _run_bench(["--gpus", "2", "--backend", "gloo", "--scaling", "weak", "--poses", "6000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                    {"HITL_DIST_COLUMNS": "1"})
    assert r2.returncode == 0, r2.stderr[-2000:]
    j2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][0])
    assert j2["config"]["submap_linear_solver"] == "interface columns"
    assert j2["lm_iterations_per_resolve"] == j["lm_iterations_per_resolve"] and abs(j2["final_cost"] - j["final_cost"]) <= 1e-9 * j["final_cost"]


@pytest.mark.gpu
def test_bench_gpus_2_defaults_to_the_named_map_split_in_two():
    """without --scaling, N > 1 ranks split the SAME map (BASELINE configs[3]: the 50k map on 4 GPUs; here a
    6000-pose stand-in on two ranks sharing the box's GPU): "scaling": "strong", n_poses_total = the map's poses, and the LM path
    of the split solve is the single-GPU one (one global LM loop, exact separator solve)."""
    r = _run_bench(["--gpus", "2", "--backend", "gloo", "--poses", "6000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["scaling"] == "strong" and j["n_gpus"] == 2 and j["config"]["comm_ranks"] == 2
    assert j["config"]["n_poses_total"] == 6000 and j["config"]["cuts"][0] == 0 and j["config"]["cuts"][-1] == 6000
    assert j["config"]["submap_linear_solver"] == "separators"
    r1 = _run_bench(["--gpus", "1", "--poses", "6000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert r1.returncode == 0, r1.stderr[-2000:]
    j1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][0])
    assert j1["scaling"] == "weak" and j1["config"]["n_poses_total"] == 6000
    assert j1["lm_iterations_per_resolve"] == j["lm_iterations_per_resolve"] and abs(j1["final_cost"] - j["final_cost"]) <= 1e-7 * j1["final_cost"]


def test_bench_defaults_name_baseline_configs():
    """What `--gpus N` runs without further flags (parsed from bench.py's own tables; no GPU): N = 1, 2, 4 -> the 50k x 400 map
    (configs[3]), N = 8 -> the 200k x 512 map (configs[4]); N > 1 splits that map (strong)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.WORKLOADS["c4"]["poses"] == 50000 and b.WORKLOADS["c4"]["obs"] == 400 and b.WORKLOADS["c4"]["groups"] == 50
    assert b.WORKLOADS["c5"]["poses"] == 200000 and b.WORKLOADS["c5"]["obs"] == 512 and b.WORKLOADS["c5"]["groups"] == 200
    assert b.WORKLOADS["c2"]["poses"] == 1000 and b.WORKLOADS["c2"]["obs"] == 360
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'args.workload = "c5" if args.gpus == 8 else "c4"' in src
    assert 'args.scaling = "strong" if args.gpus > 1 else "weak"' in src


@pytest.mark.gpu
def test_bench_refuses_rccl_with_more_ranks_than_gpus():
    """Two RCCL ranks on a one-GPU box: a clear non-zero exit, not a hang and not a mislabelled line."""
    import subprocess
    import sys
    # (torch is asked in a child: importing it into the test process next to the OpenMP oracle and the HIP library ends the
    # interpreter with a double free at exit)
    n_dev = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True,
                               timeout=300).stdout.strip() or 0)
    if n_dev >= 2:
        pytest.skip("this box has two GPUs: the refusal cannot be provoked")
    r = _run_bench(["--gpus", "2", "--poses", "6000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], timeout=300)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "need 2 GPUs" in r.stderr


def test_a_traffic_profile_of_another_kernel_source_is_not_used(tmp_path):
    """profiles/traffic_<workload>.json carries the hash of the hitl_kernels.hip it was measured on
    (tools/summarize_profile.py); bench.py compares it with the source in the tree and, when they differ, reports the layout's
    compulsory bytes and says so (roofline.traffic = null, roofline.traffic_stale) instead of yesterday's counters."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    sha = b.kernel_source_sha()
    assert len(sha) == 16 and sha == b.kernel_source_sha()
    other = tmp_path / "k.hip"
    other.write_text(open(os.path.join(ROOT, "hitl_slam_amd", "csrc", "hitl_kernels.hip")).read() + "\n// one more line\n")
    assert b.kernel_source_sha(str(other)) != sha
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'tj.get("kernel_source_sha") != kernel_source_sha()' in src and '"traffic_stale": traffic_stale' in src
    summ = open(os.path.join(ROOT, "tools", "summarize_profile.py")).read()
    assert 'res["kernel_source_sha"]' in summ
    # the committed profile either belongs to the kernel source in the tree or will be reported as stale -- never silently used
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_c4.json")))
    assert tj.get("kernel_source_sha") in (None, sha) or isinstance(tj.get("kernel_source_sha"), str)
