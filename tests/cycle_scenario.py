"""A synthetic user correction for the correction-cycle tests (tests/test_cycle_*.py, tests/test_gpu_cycle.py).

One corridor floor driven once.  The +y side wall is broken by alcoves into pieces; piece 1 (x in [2.5, 9.5]) is seen by the
early poses, later pieces (x in [10.5, 17.5], [18.5, 25.5]) by later poses, and all are colinear in truth.  The 'user' selects a later piece first
(the observations to move) and piece 1 second (where they should go), clicking on points of the drifted map.
"""
import numpy as np

from tests.conftest import Problem


class Scenario(object):
    def __init__(self, n_poses=110, obs=120, seed=20260102, first_piece=(18.5, 25.5)):
        from oracle import hitl_oracle as O
        from tools import hitl_synth as S
        self.prob = Problem(n_poses, obs, seed=seed, ppf=n_poses, groups=0)
        pr = self.prob
        self.n = n_poses
        self.poses = np.array(pr.init, np.float32).reshape(-1, 3)
        self.cov = np.array(pr.cov, np.float32).reshape(-1, 9)
        gt = np.array(pr.gt, np.float64).reshape(-1, 3)
        o = O.OracleSolver()
        pr.load(o)
        wx, wy = o.transform_scans(self.poses)
        gx, gy = o.transform_scans(gt.astype(np.float32))
        o.close()
        off = np.asarray(pr.off)
        pose_of = np.repeat(np.arange(n_poses), np.diff(off))
        # floor-local coordinates of every observation (ground truth): x along the corridor from the first pose
        x0, y0 = gt[0, 0] - 0.15, gt[0, 1] + 0.7          # local_pose(0) = (0.15, -0.7): LANE_AMP 0.7 (tools/hitl_synth.c)
        lx, ly = gx - x0, gy - y0
        self.local = (lx, ly)

        def pick(pose, xa, xb):
            sel = np.where((pose_of == pose) & (np.abs(ly - 1.2) < 0.04) & (lx > xa) & (lx < xb))[0]
            assert sel.size >= 6, (pose, sel.size)
            a, b = sel[np.argmin(lx[sel])], sel[np.argmax(lx[sel])]
            return [wx[a], wy[a], wx[b], wy[b]]

        fa, fb = first_piece
        self.pose_first = int(round((0.5 * (fa + fb) - 0.15) / 0.3))      # stands in front of the first-selected piece
        self.pose_second = int(round((6.0 - 0.15) / 0.3))                 # in front of piece 1
        first = pick(self.pose_first, fa + 0.3, fb - 0.3)
        second = pick(self.pose_second, 2.8, 9.2)
        self.selected = np.array(first + second, np.float32)   # selected_points_[0..3]
        self.world = (wx, wy)
