"""The library's host code under AddressSanitizer + UBSan (CPU only): `make -C hitl_slam_amd/csrc hostcheck` builds every host
.cpp of libhitlgpu.so with g++ -fsanitize=address,undefined over a stub device layer and walks the C ABI through one-rank
sessions (tests/hostcheck_driver.py), 2-3-rank submap sessions joined by the all-gather callback (tests/hostcheck_ranks.py:
exchange order, agreement on failure, every staging buffer of the rank code path) and the GPU-free CPU tests.  Test
infrastructure: the sanitizer build is never the library that ships, and nothing of it runs on a product path."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_and_ubsan():
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "HITL_LIBHITLGPU", "HITL_TEST_SWITCHES")}
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "hitl_slam_amd", "csrc"), "hostcheck"], env=env, capture_output=True, text=True,
                       timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "hostcheck: clean" in r.stdout, tail
    assert "sanitizers silent, no rank left waiting" in r.stdout, tail            # the multi-rank sessions ran
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
