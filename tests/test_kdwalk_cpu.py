"""The per-lane state machine the GPU's KD-tree search runs (hitl_slam_amd/csrc/hitl_kdwalk.h: `kd_step`) against the
literal explicit-stack unrolling of KDTree::FindNearestPointNormal (perception_tools/kdtree.cpp:141-197), on the CPU
(the header carries no intrinsics; g++ -ffp-contract=off).  Same winner and the same distance bits on every query --
realistic scan-to-scan queries, and adversarial point sets that force the rare branches: queries exactly on split
planes (pd == 0: both children are 'near'), duplicate points, the FLT_MIN early return, gates wider than the scan."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

from conftest import Problem

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "kdwalk_harness.cpp")
HDR = os.path.join(HERE, "..", "hitl_slam_amd", "csrc", "hitl_kdwalk.h")
LIB = os.path.join(HERE, "_build", "libkdwalk_harness.so")


@pytest.fixture(scope="module")
def kdw():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", LIB, SRC])
    lib = C.CDLL(LIB)
    lib.kdw_check.restype = C.c_int64
    lib.kdw_check2.restype = C.c_int64
    lib.kdw_sim.restype = C.c_int64
    lib.kdw_sim2.restype = C.c_int64
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def build_tree(lib, x, y, nx, ny):
    m = len(x)
    pts = np.zeros((max(m, 1), 4), np.float32)
    meta = np.zeros(max(m, 1), np.uint32)
    x, y, nx, ny = [np.ascontiguousarray(a, np.float32) for a in (x, y, nx, ny)]
    lib.kdw_build(C.c_int(m), _p(x), _p(y), _p(nx), _p(ny), _p(pts), _p(meta))
    return pts, meta


def check(lib, pts, meta, m, qx, qy, thr, cap=62, fast=0):
    """fast: kd_step_fast calls behind every kd_step (1 = the schedule the kernels run)"""
    nq = len(qx)
    qx, qy = np.ascontiguousarray(qx, np.float32), np.ascontiguousarray(qy, np.float32)
    pn, po = np.zeros(nq, np.int32), np.zeros(nq, np.int32)
    dn, do = np.zeros(nq, np.float32), np.zeros(nq, np.float32)
    ev = np.zeros(nq, np.int32)
    bad = lib.kdw_check2(C.c_int(m), _p(pts), _p(meta), C.c_int64(nq), _p(qx), _p(qy), C.c_float(thr), _p(pn), _p(dn), _p(po), _p(do), _p(ev), C.c_int(cap),
                         C.c_int(fast))
    return bad, pn, dn, po, do, ev


def scan_queries(prob, i, j):
    """source scan i in the frame of pose j (fp32; rounding details do not matter here: both walks get the same queries)"""
    f = np.float32
    pi, pj = prob.init[i], prob.init[j]
    ci, si, cj, sj = f(math.cos(pi[2])), f(math.sin(pi[2])), f(math.cos(pj[2])), f(math.sin(pj[2]))
    M = np.array([[cj * ci + sj * si, sj * ci - cj * si], [cj * si - sj * ci, sj * si + cj * ci]], f)
    d = (pi[:2] - pj[:2]).astype(f)
    t = np.array([cj * d[0] + sj * d[1], cj * d[1] - sj * d[0]], f)
    a, b = prob.off[i], prob.off[i + 1]
    q = np.stack([prob.px[a:b], prob.py[a:b]], 1) @ M.T + t
    return q[:, 0].astype(f), q[:, 1].astype(f)


def test_state_machine_equals_the_recursion_on_scan_queries(kdw):
    prob = Problem(40, 400, ppf=20)
    total = 0
    evals, evals_fast = [], []
    for j in range(0, prob.n, 3):
        a, b = prob.off[j], prob.off[j + 1]
        pts, meta = build_tree(kdw, prob.px[a:b], prob.py[a:b], prob.nx[a:b], prob.ny[a:b])
        assert sorted(meta & 0xffff) == list(range(b - a))
        for i in range(max(0, j - 3), min(prob.n, j + 4)):
            if i == j:
                continue
            qx, qy = scan_queries(prob, i, j)
            bad, pn, dn, po, do, ev = check(kdw, pts, meta, b - a, qx, qy, 0.15)
            assert bad == 0, (i, j, np.nonzero((pn != po) | (dn.view(np.uint32) != do.view(np.uint32)))[0][:5])
            for fast in (1, 2):       # kd_step_fast behind every kd_step: the same walk in fewer rounds
                badf, pnf, dnf, _, _, evf = check(kdw, pts, meta, b - a, qx, qy, 0.15, fast=fast)
                assert badf == 0 and np.array_equal(pnf, pn) and np.array_equal(dnf.view(np.uint32), dn.view(np.uint32))
                if fast == 1:
                    evals_fast.append(evf)
            total += len(qx)
            evals.append(ev)
    evals = np.concatenate(evals)
    steps, deep = evals & 0xffff, (evals >> 16) & 0x3fff
    assert total > 20000 and not (evals >> 30).any()
    # design figures, not a parity statement: steps per query (a 400-point tree has 9 levels) and how deep the stack of
    # real frames gets (frames that may still take the far side of their split plane; the rest fold into the incumbent)
    print("\nkd_step + kd_step_fast alternating: %.1f rounds per query" % (np.concatenate(evals_fast) & 0xffff).mean())
    print("\nkd_step: %d queries, steps per query mean %.1f, p50 %d, p99 %d, max %d; frame-stack depth: %s"
          % (total, steps.mean(), np.median(steps), np.percentile(steps, 99), steps.max(),
             ", ".join("%d: %.4f" % (d, (deep == d).mean()) for d in range(deep.max() + 1))))


@pytest.mark.parametrize("seed", range(6))
def test_state_machine_equals_the_recursion_on_adversarial_sets(kdw, seed):
    rng = np.random.default_rng(seed)
    for m in (0, 1, 2, 3, 4, 5, 7, 8, 31, 64, 100, 257):
        # coarse grid coordinates: many duplicates, many queries exactly on split planes, many exact ties of the score
        x = (rng.integers(0, 12, m) * 0.0625).astype(np.float32)
        y = (rng.integers(0, 12, m) * 0.0625).astype(np.float32)
        ang = rng.integers(0, 8, m) * (math.pi / 4)
        nx, ny = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
        if seed % 2:
            nx[:], ny[:] = 0.0, 1.0           # one wall: equal scores everywhere, the walk order decides
        pts, meta = build_tree(kdw, x, y, nx, ny)
        nq = 4000
        qx = (rng.integers(-2, 14, nq) * 0.0625).astype(np.float32)
        qy = (rng.integers(-2, 14, nq) * 0.0625).astype(np.float32)
        off = rng.random(nq) < 0.5
        qx[off] += (rng.standard_normal(off.sum()) * 0.02).astype(np.float32)
        qy[off] += (rng.standard_normal(off.sum()) * 0.02).astype(np.float32)
        for thr in (0.15, 0.07, 3.0, 1e-3):
            bad, pn, dn, po, do, ev = check(kdw, pts, meta, m, qx, qy, thr)
            assert bad == 0, (m, thr, np.nonzero((pn != po) | (dn.view(np.uint32) != do.view(np.uint32)))[0][:5])
            badf, pnf, dnf, _, _, _ = check(kdw, pts, meta, m, qx, qy, thr, fast=1)
            assert badf == 0 and np.array_equal(pnf, pn) and np.array_equal(dnf.view(np.uint32), dn.view(np.uint32))
            if m >= 31 and thr == 0.15:
                assert (dn == 0).any()        # the FLT_MIN early return was exercised
            # a frame stack that is too small must say so, never answer wrongly (the kernels size it for the tree's depth).
            # cap = LDS entries below the register-resident top frame: depth d needs d - 1 of them
            bad1, _, _, _, _, ev1 = check(kdw, pts, meta, m, qx, qy, thr, cap=1)
            assert bad1 == 0
            deep = (ev >> 16) & 0x3fff
            assert np.array_equal(ev1 >> 30, (deep > 2).astype(np.int32))


def test_wave_schedule_figures(kdw):
    """Lock-step simulation of one wave over a target scan's queue (6 neighbouring source scans): how many wave iterations
    the refill rule costs against the ideal (busy lane-iterations / 64).  Printed for DESIGN.md; asserts only sanity."""
    prob = Problem(16, 400, ppf=8)
    j = 8
    a, b = prob.off[j], prob.off[j + 1]
    pts, meta = build_tree(kdw, prob.px[a:b], prob.py[a:b], prob.nx[a:b], prob.ny[a:b])
    qs = [scan_queries(prob, i, j) for i in range(j - 3, j + 4) if i != j]
    qx = np.ascontiguousarray(np.concatenate([q[0] for q in qs]))
    qy = np.ascontiguousarray(np.concatenate([q[1] for q in qs]))
    rows = []
    for lanes, rmin in ((64, 1), (64, 8), (64, 16), (64, 32), (64, 64), (128, 1), (128, 16), (256, 1), (256, 32)):
        busy, rf = C.c_int64(0), C.c_int64(0)
        it = kdw.kdw_sim(C.c_int(b - a), _p(pts), _p(meta), C.c_int64(len(qx)), _p(qx), _p(qy), C.c_float(0.15), C.c_int(lanes), C.c_int(rmin),
                         C.byref(busy), C.byref(rf))
        rows.append((lanes, rmin, it, busy.value, rf.value))
        assert it * lanes >= busy.value > 0
    print()
    for lanes, rmin, it, busy, rf in rows:
        print("lanes %3d refill>=%2d: %5d iterations x %d waves, lane utilisation %.2f, %d refills" % (lanes, rmin, it, lanes // 64, busy / (it * lanes), rf))
