"""An independent numpy restatement of hitl-slam's batch re-solve, written from the reference sources alone.

TEST INFRASTRUCTURE.  This module must not import `oracle/` (nor the product): it exists so that the C oracle and the HIP
path are BOTH held to a second reading of the same reference text (tests/test_golden.py, tests/test_mirror.py), and it is
what emits the fixtures in this directory (make_golden.py).  It shares no code with `oracle/hitl_oracle.c`:

  * the three live functors are transcribed literally -- world-frame algebra, the operation order of the C++ -- over a
    small forward-mode dual-number class (`Dual`: what ceres::AutoDiffCostFunction<..., Jet<double, 6>> computes), where
    the oracle states analytic Jacobians and the kernels use a source-frame reformulation;
  * the linear algebra is dense numpy (normal equations by np.linalg.solve on the full 3(N-1) system), where the oracle
    runs a banded Cholesky and the product a block cyclic reduction;
  * the correspondence search is the recursion of kdtree.cpp as written (Python recursion, pointer-style nodes) and a
    brute-force scan; the source->target transform follows Eigen's operation order for
    `target_tf.inverse(Eigen::Affine) * source_tf` (general 2x2 inverse, Transform * Transform, Transform * vector).

fp32 quantities are computed with numpy float32 scalars in the order of the C++ expressions; float trigonometry goes
through the C library (`cosf`, `sinf`, `atan2f` via ctypes: "the float overloads of a GCC build").

Reference lines (HitL-SLAM/src/...):
  human_in_the_loop_slam/residual_functors.h:768-848   PointToPointGlobConstraint
  human_in_the_loop_slam/residual_functors.h:1054-1133 PoseConstraint
  human_in_the_loop_slam/residual_functors.h:1299-1415 Colocation / Colinear / Perpendicular / Parallel rows
  human_in_the_loop_slam/JointOptimization.cpp:296-305 RelativePoseTransform
  human_in_the_loop_slam/JointOptimization.cpp:380-402 CopyParams / SetParams
  human_in_the_loop_slam/JointOptimization.cpp:539-559 AddSTFConstraints
  human_in_the_loop_slam/JointOptimization.cpp:561-642 FindSTFCorrespondences
  human_in_the_loop_slam/JointOptimization.cpp:736-825 AddOdometryConstraints
  human_in_the_loop_slam/JointOptimization.cpp:969-1054 AddHumanConstraints
  human_in_the_loop_slam/JointOptimization.cpp:1064-1138, 1156-1256 the two ceres::Solve call sites and their options
  perception_tools/kdtree.cpp:37-197                   GetSplittingPlane, BuildKDTree, FindNearestPointNormal
  shared/math/util.h:433-439                           angle_mod
Ceres' trust-region loop is restated from its documentation (TrustRegionMinimizer + LevenbergMarquardtStrategy, 1.x):
the version the reference links is not pinned (CMakeLists.txt:83), so this is a specification, not a transcription.
"""
import ctypes
import math

import numpy as np

f32 = np.float32
_libm = ctypes.CDLL("libm.so.6")
for _n in ("cosf", "sinf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
_libm.atan2f.restype = ctypes.c_float
_libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
_libm.rint.restype = ctypes.c_double
_libm.rint.argtypes = [ctypes.c_double]


def cosf(x):
    return f32(_libm.cosf(float(f32(x))))


def sinf(x):
    return f32(_libm.sinf(float(f32(x))))


def atan2f(y, x):
    return f32(_libm.atan2f(float(f32(y)), float(f32(x))))


M_2PI = 6.28318530717958647693


def angle_mod(a):
    """shared/math/util.h:433-439 (instantiated for double at JointOptimization.cpp:388, for float at :752)."""
    if isinstance(a, np.float32):
        return f32(float(a) - M_2PI * _libm.rint(float(a) / M_2PI))    # float angle: M_2PI is a double macro, the arithmetic is double, the assignment narrows
    return a - M_2PI * _libm.rint(a / M_2PI)


# ---------------------------------------------------------------------------------------------------------------------
# forward-mode duals: value [...], partials [..., 6]
# ---------------------------------------------------------------------------------------------------------------------
class Dual(object):
    __slots__ = ("v", "d")

    def __init__(self, v, d=None, nvar=6):
        self.v = np.asarray(v, np.float64)
        self.d = np.zeros(self.v.shape + (nvar,)) if d is None else d

    @staticmethod
    def var(v, k, nvar=6):
        x = Dual(v, nvar=nvar)
        x.d[..., k] = 1.0
        return x

    def _c(self, o):
        return o if isinstance(o, Dual) else Dual(np.broadcast_to(np.asarray(o, np.float64), self.v.shape), np.zeros(self.d.shape))

    def __add__(self, o):
        o = self._c(o)
        return Dual(self.v + o.v, self.d + o.d)
    __radd__ = __add__

    def __sub__(self, o):
        o = self._c(o)
        return Dual(self.v - o.v, self.d - o.d)

    def __rsub__(self, o):
        return self._c(o) - self

    def __neg__(self):
        return Dual(-self.v, -self.d)

    def __mul__(self, o):
        o = self._c(o)
        return Dual(self.v * o.v, self.d * o.v[..., None] + o.d * self.v[..., None])
    __rmul__ = __mul__

    def __truediv__(self, o):
        o = self._c(o)
        q = self.v / o.v
        return Dual(q, (self.d - o.d * q[..., None]) / o.v[..., None])

    def sum(self):
        return Dual(self.v.sum(-1), self.d.sum(-2))


def dsin(a):
    return Dual(np.sin(a.v), a.d * np.cos(a.v)[..., None])


def dcos(a):
    return Dual(np.cos(a.v), -a.d * np.sin(a.v)[..., None])


def dsqrt(a):
    r = np.sqrt(a.v)
    return Dual(r, a.d / (2.0 * r)[..., None])


def datan2(y, x):
    den = x.v * x.v + y.v * y.v
    return Dual(np.arctan2(y.v, x.v), (y.d * x.v[..., None] - x.d * y.v[..., None]) / den[..., None])


def _pose_vars(p, first):
    return [Dual.var(p[k], first + k) for k in range(3)]


# ---------------------------------------------------------------------------------------------------------------------
# functors
# ---------------------------------------------------------------------------------------------------------------------
def pose_constraint(pose1, pose2, axis_transform, sig, radial_translation, rotation):
    """residual_functors.h:1081-1125.  axis_transform 2x2 float32, sig 3 float32 (radial, tangential, angular).
    Returns (r[3], J[3, 6])."""
    p1, p2 = _pose_vars(pose1, 0), _pose_vars(pose2, 3)
    tx, ty = p2[0] - p1[0], p2[1] - p1[1]
    a = -p1[2]                                            # Eigen::Rotation2D<T>(-pose1[2])
    c, s = dcos(a), dsin(a)
    rx, ry = c * tx - s * ty, s * tx + c * ty
    A = np.asarray(axis_transform, np.float32).astype(np.float64)
    ux, uy = rx * float(A[0, 0]) + ry * float(A[0, 1]), rx * float(A[1, 0]) + ry * float(A[1, 1])
    r0 = (ux - float(f32(radial_translation))) / float(f32(sig[0]))
    r1 = uy / float(f32(sig[1]))
    e = p2[2] - p1[2] - float(f32(rotation))
    r2 = datan2(dsin(e), dcos(e)) / float(f32(sig[2]))
    return np.array([r0.v, r1.v, r2.v]), np.stack([r0.d, r1.d, r2.d])


def point_to_point_glob(pose0, pose1, p0, p1, n0, n1, std_dev, corr):
    """residual_functors.h:786-830.  p0, p1, n0, n1: float32 [m, 2].  Returns (r[2], J[2, 6])."""
    q0, q1 = _pose_vars(pose0, 0), _pose_vars(pose1, 3)
    c0, s0, c1, s1 = dcos(q0[2]), dsin(q0[2]), dcos(q1[2]), dsin(q1[2])
    P0, P1, N0, N1 = (np.asarray(a, np.float32).astype(np.float64) for a in (p0, p1, n0, n1))
    m = P0.shape[0]

    def lift(s):                                           # a per-pose scalar against m points
        return Dual(np.broadcast_to(s.v, (m,)), np.broadcast_to(s.d, (m, 6)))
    c0, s0, c1, s1 = lift(c0), lift(s0), lift(c1), lift(s1)
    t0x, t0y, t1x, t1y = lift(q0[0]), lift(q0[1]), lift(q1[0]), lift(q1[1])
    p0gx, p0gy = (c0 * P0[:, 0] - s0 * P0[:, 1]) + t0x, (s0 * P0[:, 0] + c0 * P0[:, 1]) + t0y
    p1gx, p1gy = (c1 * P1[:, 0] - s1 * P1[:, 1]) + t1x, (s1 * P1[:, 0] + c1 * P1[:, 1]) + t1y
    n0gx, n0gy = c0 * N0[:, 0] - s0 * N0[:, 1], s0 * N0[:, 0] + c0 * N0[:, 1]
    n1gx, n1gy = c1 * N1[:, 0] - s1 * N1[:, 1], s1 * N1[:, 0] + c1 * N1[:, 1]
    dx, dy = p1gx - p0gx, p1gy - p0gy
    k_c, k_s = float(f32(corr)), float(f32(std_dev))
    e0 = (n0gx * dx + n0gy * dy) * k_c / k_s               # T(n.dot(delta)) * T(correlation_factor) / T(std_dev)
    e1 = (n1gx * dx + n1gy * dy) * k_c / k_s
    out_r, out_J = np.zeros(2), np.zeros((2, 6))
    for row, e in enumerate((e0, e1)):
        tot = (e * e).sum()                                # std::accumulate of sq(...)
        if tot.v != 0.0:
            res = dsqrt(tot / float(m))
            out_r[row], out_J[row] = res.v, res.d
    return out_r, out_J


def human_target(row, poses_f32):
    """JointOptimization.cpp:981-989 / :1003-1012 / :1028-1030 / :1042-1044, float arithmetic.  row = (type, constrained,
    anchor, delta_parallel, delta_perpendicular, delta_angle, relative_penalty_dir).  Returns (x*, y*, theta*, penalty_dir) as
    float32 (the functor constructors take double(...) of exactly these)."""
    typ, anchor = int(row[0]), int(row[2])
    dpar, dperp, dang, rel = f32(row[3]), f32(row[4]), f32(row[5]), f32(row[6])
    ax, ay, aa = f32(poses_f32[anchor, 0]), f32(poses_f32[anchor, 1]), f32(poses_f32[anchor, 2])
    para = (cosf(aa), sinf(aa))
    perp = (f32(-para[1]), para[0])
    tx = f32(f32(ax + f32(dpar * para[0])) + f32(dperp * perp[0]))
    ty = f32(f32(ay + f32(dpar * para[1])) + f32(dperp * perp[1]))
    t_angle = f32(aa + dang)
    target_angle = atan2f(sinf(t_angle), cosf(t_angle))
    pen = f32(aa + rel)
    return typ, tx, ty, target_angle, pen


def human_row(typ, tgt, pose):
    """residual_functors.h:1305-1411: (r[k], J[k, 3]), k = 3 / 2 / 1 / 1 for types 2 / 4 / 5 / 6, 0 otherwise."""
    x = [Dual.var(pose[k], k, nvar=3) for k in range(3)]
    tx, ty, tt, pen = (float(v) for v in tgt)
    if typ == 2:
        rs = [tx - x[0], ty - x[1], tt - x[2]]
    elif typ == 4:
        cp, sp = math.cos(pen), math.sin(pen)              # T(cos(penalty_dir_)): double cos of the float-valued angle
        rs = [cp * (tx - x[0]) + sp * (ty - x[1]), tt - x[2]]
    elif typ in (5, 6):
        rs = [tt - x[2]]
    else:
        rs = []
    if not rs:
        return np.zeros(0), np.zeros((0, 3))
    return np.array([float(r.v) for r in rs]), np.stack([r.d for r in rs])


def odometry_setup(prev, cur):
    """JointOptimization.cpp:748-769 in float: (axis_transform 2x2, radial_translation, rotation)."""
    tx, ty = f32(f32(cur[0]) - f32(prev[0])), f32(f32(cur[1]) - f32(prev[1]))
    k_eps = f32(1e-6)
    c90, s90 = cosf(f32(math.pi / 2)), sinf(f32(math.pi / 2))            # Rotation2Df(M_PI_2)
    if abs(tx) < k_eps and abs(ty) < k_eps:
        rad = (cosf(cur[2]), sinf(cur[2]))
        radial_translation = f32(0.0)
    else:
        a = f32(-f32(prev[2]))
        c, s = cosf(a), sinf(a)
        vx, vy = f32(f32(c * tx) - f32(s * ty)), f32(f32(s * tx) + f32(c * ty))
        nrm = f32(np.sqrt(f32(f32(vx * vx) + f32(vy * vy))))
        rad = (f32(vx / nrm), f32(vy / nrm))                               # .normalized()
        radial_translation = f32(np.sqrt(f32(f32(tx * tx) + f32(ty * ty))))   # translation.norm()
    tan = (f32(f32(c90 * rad[0]) - f32(s90 * rad[1])), f32(f32(s90 * rad[0]) + f32(c90 * rad[1])))
    rotation = angle_mod(f32(f32(cur[2]) - f32(prev[2])))
    A = np.array([[rad[0], rad[1]], [tan[0], tan[1]]], np.float32)
    return A, radial_translation, rotation


# ---------------------------------------------------------------------------------------------------------------------
# correspondence search
# ---------------------------------------------------------------------------------------------------------------------
def relative_pose_transform(src, tgt):
    """JointOptimization.cpp:296-305 with Eigen's arithmetic: Translation2Df * Rotation2Df, Transform::inverse(Affine)
    (linear().inverse() by the 2x2 cofactor formula, translation = -inverse * translation), Transform * Transform.
    Returns (L 2x2, t 2) as float32."""
    def tf(p):
        c, s = cosf(p[2]), sinf(p[2])
        return np.array([[c, f32(-s)], [s, c]], np.float32), (f32(p[0]), f32(p[1]))
    Ls, ts = tf(src)
    Lt, tt = tf(tgt)
    det = f32(f32(Lt[0, 0] * Lt[1, 1]) - f32(Lt[1, 0] * Lt[0, 1]))
    invdet = f32(f32(1.0) / det)
    Li = np.array([[f32(Lt[1, 1] * invdet), f32(f32(-Lt[0, 1]) * invdet)],
                   [f32(f32(-Lt[1, 0]) * invdet), f32(Lt[0, 0] * invdet)]], np.float32)
    ti = (f32(f32(f32(-Li[0, 0]) * tt[0]) + f32(f32(-Li[0, 1]) * tt[1])), f32(f32(f32(-Li[1, 0]) * tt[0]) + f32(f32(-Li[1, 1]) * tt[1])))
    L = np.array([[f32(f32(Li[r, 0] * Ls[0, c]) + f32(Li[r, 1] * Ls[1, c])) for c in range(2)] for r in range(2)], np.float32)
    t = (f32(f32(f32(Li[0, 0] * ts[0]) + f32(Li[0, 1] * ts[1])) + ti[0]), f32(f32(f32(Li[1, 0] * ts[0]) + f32(Li[1, 1] * ts[1])) + ti[1]))
    return L, t


class KDNode(object):
    __slots__ = ("pt", "nrm", "index", "split", "left", "right")


def kd_build(values):
    """kdtree.cpp:106-139 with GetSplittingPlane :37-69.  values: list of (x, y, nx, ny, index) with float32 coordinates.
    std::sort's order among equal keys is unspecified; the stable order is taken (as in the oracle and the product)."""
    if not values:
        return None
    n = len(values)
    mx, my = f32(0), f32(0)
    for v in values:
        mx, my = f32(mx + v[0]), f32(my + v[1])
    mx, my = f32(mx / f32(n)), f32(my / f32(n))
    sx, sy = f32(0), f32(0)
    for v in values:
        sx = f32(sx + f32(f32(v[0] - mx) * f32(v[0] - mx)))
        sy = f32(sy + f32(f32(v[1] - my) * f32(v[1] - my)))
    plane, best = 0, f32(0)
    if sx > best:
        plane, best = 0, sx
    if sy > best:
        plane, best = 1, sy
    values = sorted(values, key=lambda v: v[plane])
    ind = n // 2
    nd = KDNode()
    nd.pt, nd.nrm, nd.index, nd.split = (values[ind][0], values[ind][1]), (values[ind][2], values[ind][3]), values[ind][4], plane
    nd.left = kd_build(values[:ind]) if ind > 0 else None
    nd.right = kd_build(values[ind + 1:]) if ind < n - 1 else None
    return nd


FLT_MAX = f32(3.402823466e+38)
FLT_MIN = f32(1.175494351e-38)


def kd_find_nearest_point_normal(nd, q, thr, out):
    """kdtree.cpp:141-197.  out: one-element list standing in for *neighbor_node."""
    best = FLT_MAX
    dx, dy = f32(nd.pt[0] - q[0]), f32(nd.pt[1] - q[1])
    if f32(f32(dx * dx) + f32(dy * dy)) < f32(thr * thr):
        out[0] = nd
        best = f32(abs(f32(f32(nd.nrm[0] * f32(q[0] - nd.pt[0])) + f32(nd.nrm[1] * f32(q[1] - nd.pt[1])))))
        if best < FLT_MIN:
            return f32(0.0)
    pd = f32(q[nd.split] - nd.pt[nd.split])
    other = None
    if pd <= 0.0 and nd.left is not None:
        c = [None]
        d = kd_find_nearest_point_normal(nd.left, q, thr, c)
        if d < best:
            best, out[0] = d, c[0]
        other = nd.right
    if pd >= 0.0 and nd.right is not None:
        c = [None]
        d = kd_find_nearest_point_normal(nd.right, q, thr, c)
        if d < best:
            best, out[0] = d, c[0]
        other = nd.left
    if other is not None and pd != 0.0 and f32(abs(pd)) < (best if best < thr else thr):
        c = [None]
        d = kd_find_nearest_point_normal(other, q, thr, c)
        if d < best:
            best, out[0] = d, c[0]
    return best


def find_stf_correspondences(poses_f32, off, px, py, nx, ny, window, method, thr=0.15, max_angle=math.radians(25.0), cap=6, skip=1,
                             min_pair=10):
    """JointOptimization.cpp:561-642 restricted to |i - j| <= window (window <= 0: all pairs, as written).
    method 1 = the KD-tree walk, 0 = the brute-force definition of the same query.
    Returns (pair_src, pair_tgt, match_offset, src_idx, tgt_idx)."""
    n = len(off) - 1
    thr = f32(thr)
    min_cos = cosf(f32(max_angle))
    if window <= 0 or window > n:
        window = n
    trees = None
    if method == 1:
        trees = [kd_build([(f32(px[o]), f32(py[o]), f32(nx[o]), f32(ny[o]), o - off[i]) for o in range(off[i], off[i + 1])]) for i in range(n)]
    ps, pt, mo, i0, i1 = [], [], [0], [], []
    for i in range(n):
        mi = off[i + 1] - off[i]
        counts = [0] * mi
        for j in range(max(0, i - window), min(n - 1, i + window) + 1):
            if j == i:
                continue
            L, t = relative_pose_transform(poses_f32[i], poses_f32[j])
            dth = f32(float(poses_f32[j][2]) - float(poses_f32[i][2]))      # double difference, Rotation2Df(double) narrows
            cr, sr = cosf(dth), sinf(dth)
            k0, k1 = [], []
            oj, mj = off[j], off[j + 1] - off[j]
            for k in range(0, mi, skip):
                if counts[k] >= cap:
                    continue
                a = (f32(px[off[i] + k]), f32(py[off[i] + k]))
                an = (f32(nx[off[i] + k]), f32(ny[off[i] + k]))
                q = (f32(f32(f32(L[0, 0] * a[0]) + f32(L[0, 1] * a[1])) + t[0]), f32(f32(f32(L[1, 0] * a[0]) + f32(L[1, 1] * a[1])) + t[1]))
                nq = (f32(f32(cr * an[0]) - f32(sr * an[1])), f32(f32(sr * an[0]) + f32(cr * an[1])))
                if method == 1:
                    if mj == 0:
                        continue
                    out = [None]
                    dist = kd_find_nearest_point_normal(trees[j], q, thr, out)
                    if out[0] is None:
                        continue                       # neighbor_point never assigned: dist == FLT_MAX fails the test below
                    bn, bi = out[0].nrm, out[0].index
                else:
                    dist, bi = FLT_MAX, -1
                    for tix in range(mj):
                        ddx, ddy = f32(q[0] - f32(px[oj + tix])), f32(q[1] - f32(py[oj + tix]))
                        if f32(f32(ddx * ddx) + f32(ddy * ddy)) < f32(thr * thr):
                            sc = f32(abs(f32(f32(f32(nx[oj + tix]) * ddx) + f32(f32(ny[oj + tix]) * ddy))))
                            if sc < dist:
                                dist, bi = sc, tix
                    if bi < 0:
                        continue
                    bn = (f32(nx[oj + bi]), f32(ny[oj + bi]))
                cosang = f32(f32(bn[0] * nq[0]) + f32(bn[1] * nq[1]))
                if dist < thr and cosang > min_cos:
                    k0.append(k)
                    k1.append(bi)
                    counts[k] += 1
            if len(k0) > min_pair:
                ps.append(i)
                pt.append(j)
                i0 += k0
                i1 += k1
                mo.append(len(i0))
    return (np.array(ps, np.int32), np.array(pt, np.int32), np.array(mo, np.int64), np.array(i0, np.int32), np.array(i1, np.int32))


# ---------------------------------------------------------------------------------------------------------------------
# the problem: residual vector and Jacobian over the free parameters (pose 0 is constant: JointOptimization.cpp:824, :1197)
# ---------------------------------------------------------------------------------------------------------------------
MODE_ODOM_HC, MODE_STF_ONLY, MODE_JOINT = 0, 1, 2


class Problem(object):
    def __init__(self, mode, poses_f32, hc_rows, corr, scans, std_dev=0.05, corr_factor=1.0 / 40.0, sig=(0.03, 0.03, 0.01)):
        self.mode = mode
        self.ref = np.asarray(poses_f32, np.float32)
        self.n = len(self.ref)
        self.blocks = []                                   # (kind, poses touched, payload)
        if mode != MODE_STF_ONLY:
            for i in range(1, self.n):                     # AddOdometryConstraints
                self.blocks.append(("odom", (i - 1, i), odometry_setup(self.ref[i - 1], self.ref[i]) + (np.asarray(sig, np.float32),)))
            for row in hc_rows:                            # AddHumanConstraints
                typ, tx, ty, tt, pen = human_target(row, self.ref)
                if typ in (2, 4, 5, 6):
                    self.blocks.append(("hc", (int(row[1]),), (typ, (tx, ty, tt, pen))))
        self.n_hc_res = sum({2: 3, 4: 2, 5: 1, 6: 1}[b[2][0]] for b in self.blocks if b[0] == "hc")
        self.pairs = []
        if mode != MODE_ODOM_HC and corr is not None:      # AddSTFConstraints
            off, px, py, nx, ny = scans
            ps, pt, mo, i0, i1 = corr
            P = np.stack([px, py], 1).astype(np.float32)
            N = np.stack([nx, ny], 1).astype(np.float32)
            for p in range(len(ps)):
                a, b = off[ps[p]] + i0[mo[p]:mo[p + 1]], off[pt[p]] + i1[mo[p]:mo[p + 1]]
                self.blocks.append(("stf", (int(ps[p]), int(pt[p])), (P[a], P[b], N[a], N[b], f32(std_dev), f32(corr_factor))))
                self.pairs.append(len(self.blocks) - 1)
        self.n_res = sum(self._rows(b) for b in self.blocks)

    @staticmethod
    def _rows(b):
        return 3 if b[0] == "odom" else 2 if b[0] == "stf" else {2: 3, 4: 2, 5: 1, 6: 1}[b[2][0]]

    def evaluate(self, x, jac=True):
        """x: [n, 3] float64.  Returns (r, J over ALL 3n columns or None, per-pair r [P, 2], per-pair J [P, 2, 6])."""
        x = np.asarray(x, np.float64).reshape(self.n, 3)
        r = np.zeros(self.n_res)
        J = np.zeros((self.n_res, 3 * self.n)) if jac else None
        pr, pj = np.zeros((len(self.pairs), 2)), np.zeros((len(self.pairs), 2, 6))
        row = 0
        npair = 0
        for kind, poses, payload in self.blocks:
            if kind == "odom":
                A, rho, rot, sig = payload
                rb, Jb = pose_constraint(x[poses[0]], x[poses[1]], A, sig, rho, rot)
            elif kind == "stf":
                rb, Jb = point_to_point_glob(x[poses[0]], x[poses[1]], *payload)
                pr[npair], pj[npair] = rb, Jb
                npair += 1
            else:
                rb, Jb = human_row(payload[0], payload[1], x[poses[0]])
            k = len(rb)
            r[row:row + k] = rb
            if jac:
                for slot, p in enumerate(poses):
                    J[row:row + k, 3 * p:3 * p + 3] += Jb[:, 3 * slot:3 * slot + 3]
            row += k
        return r, J, pr, pj


def linearize(problem, x):
    """cost = 1/2 |r|^2, gradient J^T r over all 3n parameters with the constant pose's entries zeroed, per-pair r and J."""
    r, J, pr, pj = problem.evaluate(x)
    g = J.T @ r
    g[:3] = 0.0
    return 0.5 * float(r @ r), g, pr, pj


CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2


def solve(problem, tolerances_after_accept=False, max_num_iterations=100, function_tolerance=1e-6, gradient_tolerance=1e-10,
          parameter_tolerance=1e-8, initial_radius=1e4, max_radius=1e16, min_radius=1e-32, min_relative_decrease=1e-3,
          min_lm_diagonal=1e-6, max_lm_diagonal=1e32, max_consecutive_invalid=5):
    """Ceres' TRUST_REGION / LEVENBERG_MARQUARDT loop with default options + max_num_iterations = 100
    (JointOptimization.cpp:1066-1070, :145-162), exact linear solves, Jacobi scaling.

    tolerances_after_accept = False: parameter and function tolerance are tested on the candidate BEFORE the accept test and
    end the solve at the pre-step state (the order of Ceres' TrustRegionMinimizer); True: they are tested after an accepted
    step, on the updated state (SURVEY.md A.4 as first written).
    Returns dict(x [n, 3] float64, poses float32 [n, 3] as CopyParams leaves them, termination, iterations, successful_steps,
    initial_cost, final_cost)."""
    n = problem.n
    x = problem.ref.astype(np.float64).copy()                        # SetParams
    free = np.arange(3, 3 * n)                                       # pose 0 constant
    r, J, _, _ = problem.evaluate(x)
    J = J[:, free]
    cost = 0.5 * float(r @ r)
    initial_cost = cost
    g = J.T @ r
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))                    # jacobi_scaling: once, from the initial Jacobian
    radius, decrease_factor = initial_radius, 2.0
    iterations, successful, invalid = 0, 0, 0
    termination = NO_CONVERGENCE
    if np.abs(g).max(initial=0.0) <= gradient_tolerance:
        termination = CONVERGENCE
    else:
        while True:
            if iterations >= max_num_iterations:
                termination = NO_CONVERGENCE
                break
            iterations += 1
            Js = J * scale
            H = Js.T @ Js
            diag = np.clip(np.diag(H), min_lm_diagonal, max_lm_diagonal)
            try:
                step_s = np.linalg.solve(H + np.diag(diag / radius), -(Js.T @ r))
            except np.linalg.LinAlgError:
                termination = FAILURE
                break
            model = Js @ step_s
            model_cost_change = -float(model @ (r + 0.5 * model))
            if not np.all(np.isfinite(step_s)) or model_cost_change < 0.0:
                invalid += 1
                if invalid >= max_consecutive_invalid:
                    termination = FAILURE
                    break
                radius *= 0.5                                        # StepIsInvalid
                continue
            invalid = 0
            delta = step_s * scale
            xc = x.copy()
            xc.reshape(-1)[free] += delta
            rc = problem.evaluate(xc, jac=False)[0]
            new_cost = 0.5 * float(rc @ rc)
            step_norm = float(np.linalg.norm(delta))
            x_norm = float(np.linalg.norm(x.reshape(-1)[free]))
            cost_change = cost - new_cost
            if not tolerances_after_accept:
                if step_norm <= parameter_tolerance * (x_norm + parameter_tolerance):
                    termination = CONVERGENCE
                    break
                if abs(cost_change) <= function_tolerance * cost:
                    termination = CONVERGENCE
                    break
            relative_decrease = cost_change / model_cost_change
            if relative_decrease > min_relative_decrease:
                successful += 1
                radius = min(max_radius, radius / max(1.0 / 3.0, 1.0 - (2.0 * relative_decrease - 1.0) ** 3))
                decrease_factor = 2.0
                old_cost = cost
                x = xc
                r, J, _, _ = problem.evaluate(x)
                J = J[:, free]
                cost = 0.5 * float(r @ r)
                g = J.T @ r
                if np.abs(g).max(initial=0.0) <= gradient_tolerance:
                    termination = CONVERGENCE
                    break
                if tolerances_after_accept:
                    if step_norm <= parameter_tolerance * (x_norm + parameter_tolerance):
                        termination = CONVERGENCE
                        break
                    if abs(cost_change) <= function_tolerance * old_cost:
                        termination = CONVERGENCE
                        break
            else:
                radius /= decrease_factor
                decrease_factor *= 2.0
                if radius < min_radius:
                    termination = CONVERGENCE
                    break
    poses = np.zeros((n, 3), np.float32)                              # CopyParams
    poses[:, 0], poses[:, 1] = x[:, 0].astype(np.float32), x[:, 1].astype(np.float32)
    poses[:, 2] = np.array([angle_mod(float(t)) for t in x[:, 2]]).astype(np.float32)
    return dict(x=x, poses=poses, termination=termination, iterations=iterations, successful_steps=successful,
                initial_cost=initial_cost, final_cost=cost, num_residuals=problem.n_res, num_hc_residuals=problem.n_hc_res)
