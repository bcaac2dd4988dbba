"""Development helper: times k_linearize_stf alone (one linearisation per call through hitl_solver_linearize at timing level 2) for
HITL_STF_VARIANT ablations whose results are deliberately wrong.  Usage: HITL_STF_VARIANT=1001 python tools/stf_ablate.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = 50000, 400
p = S.params(n, m, seed=20264102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
opt = H.default_options(); opt.timing_level = 2
s = H.Solver(opt)
s.set_map(off, px, py, nx, ny, cov)
s.find_correspondences(init, 3)
x = init.astype(np.float64)
opt2 = H.default_options(); opt2.max_num_iterations = 6; opt2.timing_level = 2
s.close()
s = H.Solver(opt2)
s.set_map(off, px, py, nx, ny, cov)
s.find_correspondences(init, 3)
import ctypes as C
from hitl_slam_amd import abi
summ = abi.Summary()
pin = np.ascontiguousarray(init, np.float32).reshape(-1)
pout = np.zeros_like(pin)
f = s._lib.hitl_solver_solve
st = f(s._h, 2, pin.ctypes.data_as(C.POINTER(C.c_float)), pout.ctypes.data_as(C.POINTER(C.c_float)), C.byref(summ))
if True:
    print("variant %s: stf %.1f us/launch over %d launches" % (os.environ.get("HITL_STF_VARIANT"), summ.linearize_ms / max(summ.linearize_timed_calls, 1) * 1e3, summ.linearize_timed_calls))
