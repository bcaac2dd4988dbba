"""The C-ABI shared library loads without a GPU, exports every symbol include/hitl_gpu.h
declares, agrees with the ctypes mirror on struct layout, runs its host-only entry points,
and fails loudly (never falls back) when asked to compute without a device."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hitl_gpu.h")


@pytest.fixture(scope="module")
def lib():
    import hitl_slam_amd as H
    if not os.path.exists(H.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return H.load_library()


def test_every_declared_symbol_is_exported(lib):
    src = open(HEADER).read()
    names = sorted(set(re.findall(r"\b(hitl_[a-z_0-9]+)\s*\(", src)))
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libhitlgpu.so does not export %s" % n


def test_struct_layout_matches_ctypes_mirror(lib):
    from hitl_slam_amd import abi
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "hitl_gpu.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(hitl_human_constraint_t), sizeof(hitl_options), sizeof(hitl_summary),
         offsetof(hitl_options, function_tolerance), offsetof(hitl_options, rccl_unique_id),
         offsetof(hitl_summary, initial_cost), offsetof(hitl_summary, linearize_calls),
         offsetof(hitl_summary, hbm_bytes), offsetof(hitl_summary, pcie_bytes));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    got = [int(v) for v in out]
    assert got[0] == C.sizeof(abi.HumanConstraint) == 28       # == struct HumanConstraint (human_constraints.h:29-41)
    assert got[1] == C.sizeof(abi.Options)
    assert got[2] == C.sizeof(abi.Summary)
    assert got[3] == abi.Options.function_tolerance.offset
    assert got[4] == abi.Options.rccl_unique_id.offset
    assert got[5] == abi.Summary.initial_cost.offset
    assert got[6] == abi.Summary.linearize_calls.offset
    assert got[7] == abi.Summary.hbm_bytes.offset and got[8] == abi.Summary.pcie_bytes.offset


def test_defaults_agree_between_c_and_python(lib):
    from hitl_slam_amd import abi
    o = abi.Options()
    lib.hitl_options_default(C.byref(o))
    p = abi.default_options()
    for name, _ in abi.Options._fields_:
        if name in ("reserved", "rccl_unique_id"):
            continue
        assert getattr(o, name) == getattr(p, name), name
    assert lib.hitl_abi_version() == abi.ABI_VERSION


def test_partition_range_covers_and_balances(lib):
    from hitl_slam_amd.api import partition_range
    rng = np.random.default_rng(0)
    counts = rng.integers(0, 500, 1000)
    off = np.concatenate([[0], np.cumsum(counts)])
    for w in (1, 2, 3, 4, 8):
        rs = [partition_range(off, r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == 1000
        for a, b in zip(rs[:-1], rs[1:]):
            assert a[1] == b[0]
        loads = [off[e] - off[b] for b, e in rs]
        assert all(e > b for b, e in rs)
        assert max(loads) - min(loads) <= 2 * counts.max()
    # more ranks than poses is the caller's problem, but must not crash
    assert partition_range(np.array([0, 5, 9]), 0, 2) == (0, 1)
    # cuts on multiples of `align` poses (the separator form of the submap solve needs every rank but the last to own a multiple
    # of the pose band): still a tiling, still balanced to within one block, at least two blocks per rank; maps too short for that
    # keep the unaligned cuts
    for w in (2, 3, 8):
        for align in (3, 12):
            rs = [partition_range(off, r, w, align=align) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == 1000 and all(a[1] == b[0] for a, b in zip(rs[:-1], rs[1:]))
            assert all(e % align == 0 for _, e in rs[:-1]) and all(e - b >= 2 * align for b, e in rs)
            plain = [partition_range(off, r, w) for r in range(w)]
            assert all(abs(a[1] - p[1]) <= align for a, p in zip(rs, plain))
    assert [partition_range(np.arange(41) * 7, r, 4, align=12) for r in range(4)] == [partition_range(np.arange(41) * 7, r, 4) for r in range(4)]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_silent_cpu_fallback(lib):
    import hitl_slam_amd as H
    with pytest.raises(H.HitlError) as e:
        H.Solver()
    assert e.value.status == H.abi.ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.hitl_last_error() or b"HIP" in lib.hitl_last_error()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hitl_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "hitl_oracle" not in txt and "orc_" not in txt, "%s references the oracle" % f


def test_libm_pairing_assumption_holds_on_this_host(lib):
    """The correspondence search computes one sincosf per candidate PAIR (hitl_corr.cpp): that is exact only if sincosf agrees
    with sinf / cosf and the odd / even symmetry holds bit for bit in the C library at hand.  The library checks a sample at
    first use (and falls back otherwise); this asserts the fast path is the one in use here, and repeats the check in numpy
    terms through ctypes' libm on a different sample."""
    import ctypes.util
    assert lib.hitl_libm_sincos_exact() == 1
    m = C.CDLL(ctypes.util.find_library("m"))
    for f in (m.sinf, m.cosf):
        f.restype = C.c_float
        f.argtypes = [C.c_float]
    m.sincosf.restype = None
    m.sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-7.0, 7.0, 4000), rng.uniform(-1e-3, 1e-3, 500), [0.0, -0.0, np.pi, -np.pi, 1e9]]).astype(np.float32)
    for x in xs:
        s, c = C.c_float(), C.c_float()
        m.sincosf(float(x), C.byref(s), C.byref(c))
        s1, c1, s2, c2 = m.sinf(float(x)), m.cosf(float(x)), m.sinf(float(-x)), m.cosf(float(-x))
        assert np.float32(s.value).tobytes() == np.float32(s1).tobytes() and np.float32(c.value).tobytes() == np.float32(c1).tobytes()
        assert np.float32(s2).tobytes() == np.float32(-np.float32(s1)).tobytes() and np.float32(c2).tobytes() == np.float32(c1).tobytes()


def test_header_is_self_contained_c():
    """include/hitl_gpu.h compiles on its own as C (a maintainer's first include; round 3 shipped a `size_t` without <stddef.h>
    for an hour, which only the `make native` build of the oracle on the GPU box noticed)."""
    import subprocess
    for std in ("-std=c99", "-std=gnu11"):
        subprocess.check_call(["gcc", std, "-fsyntax-only", "-Wall", "-Werror", "-x", "c", HEADER])
    subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Werror", "-x", "c++", HEADER])
