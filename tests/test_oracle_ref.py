"""oracle/_ref: the reference's OWN code, as far as it compiles in this image.  Of the files on the path only shared/math/util.h
builds from its own source (everything else needs Ceres / Eigen / glog: DESIGN.md "Oracle"); `make -C oracle ref` compiles it
where it lies under /root/reference into oracle/_ref/libref_mathutil.so (git-ignored, travels to the GPU box), and this file
pins the restatements of its angle_mod -- CopyParams' wrap (JointOptimization.cpp:388, double) and the rotation of the odometry
measurement (JointOptimization.cpp:757,765, float argument, double arithmetic, float result) -- bit for bit: the C oracle's
orc_angle_mod and orc_odom_setup, and the Python mirror that generated the golden fixtures."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_mathutil.so")
REF_SRC = "/root/reference/HitL-SLAM/src/shared/math/util.h"


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(REF_SO) and os.path.exists(REF_SRC):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libref_mathutil.so absent and /root/reference not on this box")
    lib = C.CDLL(REF_SO)
    lib.ref_angle_mod_f64.argtypes, lib.ref_angle_mod_f64.restype = [C.c_double], C.c_double
    lib.ref_angle_mod_f32.argtypes, lib.ref_angle_mod_f32.restype = [C.c_float], C.c_float
    lib.ref_angle_diff_f64.argtypes, lib.ref_angle_diff_f64.restype = [C.c_double, C.c_double], C.c_double
    lib.ref_angle_diff_f32.argtypes, lib.ref_angle_diff_f32.restype = [C.c_float, C.c_float], C.c_float
    return lib


def _doubles():
    rng = np.random.default_rng(20261004)
    v = [rng.uniform(-50, 50, 20000), rng.normal(0, 1e3, 5000), rng.normal(0, 1e-3, 2000)]
    sp = []
    for k in range(-9, 10):
        for base in (k * math.pi, (k + 0.5) * 2 * math.pi, k * 2 * math.pi):        # the wrap points and the ties of rint()
            x = base
            sp.append(x)
            up = dn = x
            for _ in range(3):
                up, dn = np.nextafter(up, np.inf), np.nextafter(dn, -np.inf)
                sp += [float(up), float(dn)]
    sp += [0.0, -0.0, 5e-324, -5e-324, 1e9, -1e9, 1e15 + 0.5, 3.141592653589793, -3.141592653589793]
    return np.concatenate(v + [np.array(sp)])


def _same_bits64(a, b):
    return np.array_equal(np.asarray(a, np.float64).view(np.uint64), np.asarray(b, np.float64).view(np.uint64))


def test_oracle_angle_mod_is_the_references_bit_for_bit(ref):
    from oracle import hitl_oracle as O
    orc = O.lib().orc_angle_mod
    orc.argtypes, orc.restype = [C.c_double], C.c_double
    xs = _doubles()
    got = np.array([orc(float(x)) for x in xs])
    want = np.array([ref.ref_angle_mod_f64(float(x)) for x in xs])
    assert _same_bits64(got, want)
    for bad in (float("nan"), float("inf"), float("-inf")):
        assert math.isnan(orc(bad)) and math.isnan(ref.ref_angle_mod_f64(bad))
    # angle_diff(a, b) is angle_mod(a - b) in the reference as well
    for a, b in zip(xs[:4000], xs[4000:8000]):
        assert orc(float(a) - float(b)) == ref.ref_angle_diff_f64(float(a), float(b)) or (math.isnan(orc(float(a) - float(b))))


def test_odometry_rotation_is_the_references_float_instantiation(ref):
    """angle_mod<float>(poses_[i].angle - poses_[i-1].angle): float difference, double arithmetic against the double macro M_2PI, the
    assignment narrows (JointOptimization.cpp:757,765).  orc_odom_setup's third output must be exactly that."""
    from oracle import hitl_oracle as O
    rng = np.random.default_rng(7)
    prev = rng.uniform(-7, 7, 6000).astype(np.float32)
    cur = rng.uniform(-7, 7, 6000).astype(np.float32)
    # differences that sit on the wrap points after the float subtraction
    prev[:64] = np.float32(0.0)
    cur[:64] = (np.arange(64) - 32).astype(np.float32) * np.float32(math.pi) / np.float32(4)
    for p, c in zip(prev, cur):
        _, _, rot = O.odom_setup([0.0, 0.0, float(p)], [1.0, 0.5, float(c)])
        want = ref.ref_angle_mod_f32(float(np.float32(c) - np.float32(p)))
        assert np.float32(rot).view(np.uint32) == np.float32(want).view(np.uint32), (p, c, rot, want)
        assert np.float32(want).view(np.uint32) == np.float32(ref.ref_angle_diff_f32(float(c), float(p))).view(np.uint32)


def test_mirror_angle_mod_is_the_references_too(ref):
    """tests/golden/mirror.py generated the golden fixtures: its angle_mod (both instantiations) against the reference's."""
    from tests.golden import mirror
    xs = _doubles()[::7]
    for x in xs:
        assert np.float64(mirror.angle_mod(float(x))).view(np.uint64) == np.float64(ref.ref_angle_mod_f64(float(x))).view(np.uint64)
    fs = np.concatenate([np.random.default_rng(3).uniform(-20, 20, 3000), xs[:500]]).astype(np.float32)
    for f in fs:
        got = mirror.angle_mod(np.float32(f))
        assert isinstance(got, np.float32)
        assert got.view(np.uint32) == np.float32(ref.ref_angle_mod_f32(float(f))).view(np.uint32)


def test_copy_params_rounding_of_a_solve_output(ref):
    """CopyParams: poses_[i].angle = angle_mod(pose_array_[j + 2]) -- double in, the float member narrows (JointOptimization.cpp:388);
    the oracle's solve writes (float)orc_angle_mod(x).  Same thing through the reference's function."""
    from oracle import hitl_oracle as O
    orc = O.lib().orc_angle_mod
    orc.argtypes, orc.restype = [C.c_double], C.c_double
    for x in _doubles()[::5]:
        a, b = np.float32(orc(float(x))), np.float32(ref.ref_angle_mod_f64(float(x)))
        assert a.view(np.uint32) == b.view(np.uint32) or (np.isnan(a) and np.isnan(b))
