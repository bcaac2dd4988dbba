"""An independent Python restatement of one user correction (HitLSLAM::Run) and of the covariance recovery, written from the
reference sources alone -- the second reading of SURVEY.md section 8f rows 3 and 4, next to `mirror.py` (the batch re-solve).

TEST INFRASTRUCTURE.  This module must not import `oracle/` (nor the product): it emits tests/golden/cycle_corridor.npz
(make_golden_cycle.py) and both the C oracle and the HIP path are held to that fixture (tests/test_golden.py).  It shares no
code with oracle/hitl_oracle.c: literal Python loops over pointer-style lists where the oracle uses closed forms and explicit
index arithmetic, numpy float32 SCALARS in the order of the C++ / Eigen expressions, a dual-number evaluation of the endpoint
functor with a literal one-parameter trust-region loop, a dense pseudo-inverse for the covariance blocks.

Reference lines (HitL-SLAM/src/...):
  human_in_the_loop_slam/HitLSLAM.cpp:218-244      verifyUserInput
  human_in_the_loop_slam/HitLSLAM.cpp:245-254      transformPointCloudsToWorldFrame
  human_in_the_loop_slam/HitLSLAM.cpp:380-480      Run (the order of the stages; angle normalisation :442-447)
  human_in_the_loop_slam/EMinput.cpp:107-150       segDistResidualEM
  human_in_the_loop_slam/EMinput.cpp:153-191       SegFitEM
  human_in_the_loop_slam/EMinput.cpp:195-250       AutomaticEndpointAdjustment
  human_in_the_loop_slam/EMinput.cpp:265-322       distToLineSeg, EstablishObservationSets
  human_in_the_loop_slam/EMinput.cpp:325-455       OrderAndFilterUserInput
  human_in_the_loop_slam/ApplyExplicitCorrection.cpp:150-325   the four Add*HumanConstraint
  human_in_the_loop_slam/ApplyExplicitCorrection.cpp:357-415   FindContiguousGroups, ApplyExplicitCorrections
  human_in_the_loop_slam/ApplyExplicitCorrection.cpp:417-487   AppExpCorrections, calculateConstraintTargets
  human_in_the_loop_slam/Backprop.cpp:98-200       BackPropagateError
  shared/math/eigen_helper.h:26-34, 67-81          ScalarCross, Perp2, DistanceToLineSegment
  episodic_non_markov_localization/vector_mapping.cpp:2772-2812   ceres::Covariance blocks, rotated into the robot frame

Readings this file shares with the oracle because the reference text does not decide them (stated, not hidden):
  * the build uses `-march=native -O3` (CMakeLists.txt:12-16): whether `a*b + c` is contracted into an FMA depends on the
    host; no contraction is assumed, every fp32 operation rounds;
  * unqualified `acos / atan2 / sin / cos / sqrt` on float arguments are the float overloads (a GCC >= 6 build);
  * Eigen evaluates a fixed-size 2x2 * 2 product as `m(i,0) * v(0) + m(i,1) * v(1)`, `Transform * v` as `linear * v + translation`,
    `Translation * Rotation` as (linear = R, translation = t), `Transform * Translation(o)` as `translation += linear * o`,
    `v.normalized()` as a division of every coefficient by `sqrt(v.squaredNorm())`.
"""
import ctypes
import math

import numpy as np

from mirror import Dual, atan2f, cosf, dcos, dsin, dsqrt, f32, sinf

_libm = ctypes.CDLL("libm.so.6")
_libm.acosf.restype = ctypes.c_float
_libm.acosf.argtypes = [ctypes.c_float]


def acosf(x):
    return f32(_libm.acosf(float(f32(x))))


def sqrtf(x):
    return f32(np.sqrt(f32(x)))


class V2(object):
    """Eigen::Vector2f: two float32 scalars, coefficient-wise arithmetic, every operation rounds to float."""
    __slots__ = ("x", "y")

    def __init__(self, x, y):
        self.x, self.y = f32(x), f32(y)

    def __add__(self, o):
        return V2(self.x + o.x, self.y + o.y)

    def __sub__(self, o):
        return V2(self.x - o.x, self.y - o.y)

    def __neg__(self):
        return V2(-self.x, -self.y)

    def scaled(self, s):                       # float scalar * vector
        s = f32(s)
        return V2(s * self.x, s * self.y)

    def divided(self, s):                      # vector / float scalar
        s = f32(s)
        return V2(self.x / s, self.y / s)

    def dot(self, o):
        return f32(self.x * o.x + self.y * o.y)

    def norm(self):
        return sqrtf(self.dot(self))

    def normalized(self):
        return self.divided(self.norm())

    def eq(self, o):
        return bool(self.x == o.x and self.y == o.y)


def rot_apply(angle_f32, v):
    """Rotation2Df(angle) * v: toRotationMatrix() = (cos, -sin; sin, cos) with the float sin / cos, then a 2x2 * 2 product."""
    c, s = cosf(angle_f32), sinf(angle_f32)
    return V2(c * v.x + (-s) * v.y, s * v.x + c * v.y)


# ---------------------------------------------------------------------------------------------------------------------
# world-frame clouds, verifyUserInput
# ---------------------------------------------------------------------------------------------------------------------
def transform_clouds(poses, off, px, py):
    """HitLSLAM.cpp:245-254: pose_transform = Translation2f(t) * Rotation2Df(angle); world = pose_transform * point."""
    poses = np.asarray(poses, np.float32).reshape(-1, 3)
    wx, wy = np.zeros(len(px), np.float32), np.zeros(len(py), np.float32)
    for i in range(len(poses)):
        c, s = cosf(poses[i, 2]), sinf(poses[i, 2])
        a, b = int(off[i]), int(off[i + 1])
        x, y = np.asarray(px[a:b], np.float32), np.asarray(py[a:b], np.float32)
        wx[a:b] = (c * x + (-s) * y) + poses[i, 0]          # float32 arrays: every operation rounds to float
        wy[a:b] = (s * x + c * y) + poses[i, 1]
    return wx, wy


def verify_points(wx, wy, clicks, thr):
    """HitLSLAM.cpp:222-235, per clicked point: is some cloud point closer than local_select_thresh (a float)?"""
    clicks = np.asarray(clicks, np.float32).reshape(-1, 2)
    seen = np.zeros(len(clicks), np.int32)
    thr = f32(thr)
    for i, (cx, cy) in enumerate(clicks):
        dx, dy = wx - cx, wy - cy
        nrm = np.sqrt(dx * dx + dy * dy)                    # (a - b).norm() in float
        seen[i] = 1 if bool((nrm < thr).any()) else 0
    return seen


# ---------------------------------------------------------------------------------------------------------------------
# EMInput: endpoint adjustment
# ---------------------------------------------------------------------------------------------------------------------
def distance_to_line_segment(p0, p1, x, y):
    """eigen_helper.h:67-81 over arrays of points (x, y) -- including the comparison of a LENGTH t with 1.0."""
    delta = p1 - p0
    d = delta.normalized()
    qx, qy = x - p0.x, y - p0.y                             # p - p0
    t = qx * d.x + qy * d.y
    n0 = np.sqrt(qx * qx + qy * qy)
    rx, ry = x - p1.x, y - p1.y
    n1 = np.sqrt(rx * rx + ry * ry)
    perp = np.abs((-d.y) * qx + d.x * qy)                   # fabs(Perp2(dir).dot(p - p0))
    return np.where(t < f32(0), n0, np.where(t > f32(1.0), n1, perp)).astype(np.float32)


def seg_dist_residual_em(theta, px, py, cmx, cmy, length):
    """EMinput.cpp:107-150 on duals (one parameter): residual and d residual / d theta for arrays of points."""
    th = Dual.var(np.float64(theta), 0, nvar=1)
    ax, ay = dcos(th), dsin(th)
    nrm = dsqrt(ax * ax + ay * ay)                          # alpha.normalize()
    ax, ay = ax / nrm, ay / nrm
    p1x, p1y = (ax * length) + cmx, (ay * length) + cmy     # T(cmx_) + T(len_) * alpha[0]
    p2x, p2y = -((ax * length) - cmx), -((ay * length) - cmy)   # T(cmx_) - T(len_) * alpha[0]  (a - b == -(b - a) exactly)
    px, py = np.asarray(px, np.float64), np.asarray(py, np.float64)
    P = lambda a: Dual(np.broadcast_to(a.v, px.shape).copy(), np.broadcast_to(a.d, px.shape + (1,)).copy(), nvar=1)
    p1x, p1y, p2x, p2y = P(p1x), P(p1y), P(p2x), P(p2y)
    C = lambda a, b: Dual(a - b.v, -b.d, nvar=1)            # T(constant) - jet
    qx, qy = C(px, p1x), C(py, p1y)                         # T(px_) - p1[0], ...
    ex, ey = p2x - p1x, p2y - p1y
    t = (qx * ex + qy * ey) / (ex * ex + ey * ey)
    r_a = dsqrt(qx * qx + qy * qy)
    fx, fy = C(px, p2x), C(py, p2y)
    r_b = dsqrt(fx * fx + fy * fy)
    gx, gy = C(px, p1x + t * ex), C(py, p1y + t * ey)
    r_c = dsqrt(gx * gx + gy * gy)
    with np.errstate(all="ignore"):
        v = np.where(t.v < 0.0, r_a.v, np.where(t.v > 1.0, r_b.v, r_c.v))
        d = np.where(t.v < 0.0, r_a.d[..., 0], np.where(t.v > 1.0, r_b.d[..., 0], r_c.d[..., 0]))
    return v, d


def seg_fit_em(p1, p2, data):
    """EMinput.cpp:153-191: ceres::Solve (DENSE_QR, max 25 iterations, defaults otherwise) on the one angle; Ceres'
    TrustRegionMinimizer restated for one parameter (the rules of mirror.solve).  Returns (ep1 V2, ep2 V2, iterations)."""
    icx, icy = (float(p1[0]) + float(p2[0])) / 2.0, (float(p1[1]) + float(p2[1])) / 2.0
    hy = math.sqrt((float(p1[0]) - float(p2[0])) ** 2 + (float(p1[1]) - float(p2[1])) ** 2)
    ad = abs(float(p1[0]) - float(p2[0]))
    theta0 = math.acos(ad / hy)
    length = hy / 2.0
    dx, dy = data[:, 0], data[:, 1]

    def evaluate(th):
        with np.errstate(all="ignore"):
            r, dr = seg_dist_residual_em(th, dx, dy, icx, icy, length)
            return 0.5 * float(np.sum(r * r)), float(np.sum(dr * r)), float(np.sum(dr * dr))

    theta = theta0
    cost, g, h = evaluate(theta)
    # an exactly-zero residual has the derivative 0 / 0 under a Jet sqrt: the evaluation fails, Ceres returns FAILURE with the
    # parameter untouched, and SegFitEM never reads the summary
    failed = not (math.isfinite(cost) and math.isfinite(g) and math.isfinite(h))
    iterations = 0
    if not failed and abs(g) > 1e-10:
        scale = 1.0 / (1.0 + math.sqrt(h))                  # jacobi scaling from the initial Jacobian
        radius, decrease_factor, invalid = 1e4, 2.0, 0
        while True:
            if iterations >= 25:
                break
            iterations += 1
            hs = h * scale * scale
            diag = min(max(hs, 1e-6), 1e32)
            step_s = -(scale * g) / (hs + diag / radius)
            model_cost_change = -step_s * (scale * g) - 0.5 * step_s * step_s * hs
            if not math.isfinite(step_s) or model_cost_change < 0.0:
                invalid += 1
                if invalid >= 5:
                    failed = True
                    break
                radius *= 0.5
                continue
            invalid = 0
            delta = step_s * scale
            cand = theta + delta
            new_cost, cg, ch = evaluate(cand)
            if abs(delta) <= 1e-8 * (abs(theta) + 1e-8):
                break
            if not math.isfinite(new_cost):
                new_cost = float(np.finfo(np.float64).max)
            cost_change = cost - new_cost
            if abs(cost_change) <= 1e-6 * cost:
                break
            rho = cost_change / model_cost_change
            if rho > 1e-3:
                theta, cost, g, h = cand, new_cost, cg, ch
                if not (math.isfinite(g) and math.isfinite(h)):
                    failed = True
                    break
                if abs(g) <= 1e-10:
                    break
                radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
                decrease_factor = 2.0
            else:
                radius /= decrease_factor
                decrease_factor *= 2.0
                if radius < 1e-32:
                    break
    if failed:
        theta = theta0
    ax, ay = math.cos(theta), math.sin(theta)
    nrm = math.sqrt(ax * ax + ay * ay)
    ax, ay = ax / nrm, ay / nrm                             # Vector2d alpha; alpha.normalize()
    return (V2(icx + (hy / 2.0) * ax, icy + (hy / 2.0) * ay), V2(icx - (hy / 2.0) * ax, icy - (hy / 2.0) * ay), iterations)


def fit_segments(wx, wy, clicks):
    """EMinput.cpp:195-250 for the two selections.  Returns (fitted [2, 4] float32, inliers of the last pass, passes, LM iterations)."""
    sp = np.asarray(clicks, np.float32).reshape(2, 4).copy()
    inl, pas, its = np.zeros(2, np.int32), np.zeros(2, np.int32), np.zeros(2, np.int32)
    for k in range(2):
        thresh = 0.05
        adj1 = adj2 = 2 * thresh
        while adj1 > thresh or adj2 > thresh:
            a, b = V2(sp[k, 0], sp[k, 1]), V2(sp[k, 2], sp[k, 3])
            dst = distance_to_line_segment(a, b, wx, wy).astype(np.float64)       # double dst = DistanceToLineSegment(...)
            sel = np.nonzero(dst < 0.03)[0]                                       # pose order, reading order
            data = np.stack([wx[sel].astype(np.float64), wy[sel].astype(np.float64)], 1)
            e1, e2, it = seg_fit_em((a.x, a.y), (b.x, b.y), data)
            adj1, adj2 = float((a - e1).norm()), float((b - e2).norm())
            sp[k] = [e1.x, e1.y, e2.x, e2.y]
            inl[k] = len(sel)
            pas[k] += 1
            its[k] += it
    return sp, inl, pas, its


# ---------------------------------------------------------------------------------------------------------------------
# EMInput: observation sets, ordering
# ---------------------------------------------------------------------------------------------------------------------
def dist_to_line_seg(p1, p2, x, y):
    """EMinput.cpp:265-276 over arrays of points; float arithmetic, the result widened to double by the caller."""
    ex, ey = p2.x - p1.x, p2.y - p1.y
    qx, qy = x - p1.x, y - p1.y
    t = (qx * ex + qy * ey) / f32(ex * ex + ey * ey)
    d1 = np.sqrt(qx * qx + qy * qy)
    rx, ry = x - p2.x, y - p2.y
    d2 = np.sqrt(rx * rx + ry * ry)
    jx, jy = p1.x + t * ex, p1.y + t * ey                   # proj = p1 + t * (p2 - p1)
    gx, gy = x - jx, y - jy
    d3 = np.sqrt(gx * gx + gy * gy)
    return np.where(t < f32(0.0), d1, np.where(t > f32(1.0), d2, d3)).astype(np.float32)


def observation_sets(wx, wy, off, seg8, thr=0.03):
    """EMinput.cpp:278-322: per pose the number of its points inside the 'pill' around each selection (counts [n, 2])."""
    s = np.asarray(seg8, np.float32).reshape(4, 2)
    pts = [V2(*s[k]) for k in range(4)]
    d_first = dist_to_line_seg(pts[0], pts[1], wx, wy).astype(np.float64) < thr
    d_second = dist_to_line_seg(pts[2], pts[3], wx, wy).astype(np.float64) < thr
    n = len(off) - 1
    counts = np.zeros((n, 2), np.int32)
    for i in range(n):
        a, b = int(off[i]), int(off[i + 1])
        counts[i, 0], counts[i, 1] = int(d_first[a:b].sum()), int(d_second[a:b].sum())
    return counts


def order_user_input(counts, seg8):
    """EMinput.cpp:325-455 with std::vector semantics spelled out.  Returns dict(corrected, anchor, bounds, seg_xy8)."""
    sel = [V2(*p) for p in np.asarray(seg8, np.float32).reshape(4, 2)]
    first = [i for i in range(len(counts)) if counts[i][0] > 5]           # selected_poses.first[i].first
    second = [i for i in range(len(counts)) if counts[i][1] > 5]
    first_sel, second_sel = list(first), list(second)
    overlaps = []
    for b in second_sel:
        for a in first_sel:
            if b == a:
                overlaps.append(a)
    start, end = 0, 0
    if len(overlaps) == len(first_sel) and len(overlaps) == len(second_sel):
        start = end = -1
    elif len(overlaps) == len(first_sel):
        for o in overlaps:
            second_sel = [v for v in second_sel if v != o]
    elif len(overlaps) == len(second_sel):
        for o in overlaps:
            first_sel = [v for v in first_sel if v != o]
    elif len(overlaps) > 0:
        for o in overlaps:
            first_sel = [v for v in first_sel if v != o]
            second_sel = [v for v in second_sel if v != o]
    if not first_sel or not second_sel:                                  # the reference reads [0] of an empty vector here: undefined
        return dict(corrected=np.zeros(0, np.int32), anchor=np.zeros(0, np.int32), bounds=(-1, -1),
                    seg_xy8=np.array([c for p in sel for c in (p.x, p.y)], np.float32))
    first_min, first_max = first_sel[0], first_sel[-1]
    second_min, second_max = second_sel[0], second_sel[-1]
    new_first = [j for i in first_sel for j in first if i == j]
    new_second = [j for i in second_sel for j in second if i == j]
    corrected, anchor = [], []
    if first_min > second_max:                                           # "user was good"
        corrected, anchor = new_first, new_second
        start, end = second_max + 1, first_min - 1
    elif first_max < second_min:                                         # "bad user": the selections swap
        sel = sel[2:] + sel[:2]
        corrected, anchor = new_second, new_first
        start, end = first_max + 1, second_min - 1
    else:
        start = end = -1
    return dict(corrected=np.array(corrected, np.int32), anchor=np.array(anchor, np.int32), bounds=(int(start), int(end)),
                seg_xy8=np.array([c for p in sel for c in (p.x, p.y)], np.float32))


# ---------------------------------------------------------------------------------------------------------------------
# AppExpCorrect
# ---------------------------------------------------------------------------------------------------------------------
def explicit_corrections(ctype, sel, poses, corrected):
    """The four Add*HumanConstraint (ApplyExplicitCorrection.cpp:150-325): per corrected pose (index, (Tx, Ty, theta))."""
    sp = [V2(*p) for p in np.asarray(sel, np.float32).reshape(4, 2)]
    out = []
    if ctype == 2:                                                       # kLineSegmentCorrection
        cmA, cmB = (sp[1] + sp[0]).divided(2.0), (sp[3] + sp[2]).divided(2.0)
        A, B = (sp[1] - sp[0]).normalized(), (sp[3] - sp[2]).normalized()
        theta = float(acosf(A.dot(B)))                                   # double theta = acos(float)
        if f32(A.x * B.y - A.y * B.x) < 0.0:                             # A3.cross(B3)(2)
            theta = -theta
        th = f32(theta)                                                  # Rotation2Df R(theta)
        for i in corrected:
            p0 = V2(poses[i][0], poses[i][1])
            p1 = cmB + rot_apply(th, p0 - cmA)
            T = p1 - p0
            out.append((int(i), (T.x, T.y, f32(theta))))
    elif ctype == 4:                                                     # kColinearCorrection
        cmA, cmB = (sp[1] + sp[0]).scaled(0.5), (sp[3] + sp[2]).scaled(0.5)
        A, B = (sp[1] - sp[0]).normalized(), (sp[3] - sp[2]).normalized()
        theta = acosf(A.dot(B)) if f32(A.x * B.y - A.y * B.x) >= 0.0 else -acosf(A.dot(B))
        alpha = (cmA - cmB).dot(B)
        new_cmA = cmB + B.scaled(alpha)
        for i in corrected:
            p0 = V2(poses[i][0], poses[i][1])
            p1 = new_cmA + rot_apply(theta, p0 - cmA)
            T = p1 - p0
            out.append((int(i), (T.x, T.y, theta)))
    elif ctype == 5:                                                     # kPerpendicularCorrection
        cmA = (sp[1] + sp[0]).divided(2.0)
        A, B = (sp[1] - sp[0]).normalized(), (sp[3] - sp[2]).normalized()
        theta = -float(acosf(A.dot(B))) if f32(A.x * B.y - A.y * B.x) < 0.0 else float(acosf(A.dot(B)))
        if theta == math.pi / 2.0 or theta == -math.pi / 2.0:
            theta = 0.0
        elif theta > 0.0:
            theta = -(-theta + math.pi / 2.0)
        else:
            theta = -(-theta - math.pi / 2.0)
        th = f32(theta)
        for i in corrected:
            p0 = V2(poses[i][0], poses[i][1])
            p1 = cmA + rot_apply(th, p0 - cmA)
            T = p1 - p0
            out.append((int(i), (T.x, T.y, f32(theta))))
    elif ctype == 6:                                                     # kParallelCorrection
        cmA = (sp[1] + sp[0]).scaled(0.5)
        A, B = (sp[1] - sp[0]).normalized(), (sp[3] - sp[2]).normalized()
        theta = acosf(A.dot(B)) if f32(A.x * B.y - A.y * B.x) >= 0.0 else -acosf(A.dot(B))
        for i in corrected:
            p0 = V2(poses[i][0], poses[i][1])
            p1 = cmA + rot_apply(theta, p0 - cmA)
            T = p1 - p0
            out.append((int(i), (T.x, T.y, theta)))
    return out


def apply_explicit_correction(ctype, sel, poses, corrected, anchor):
    """AppExpCorrect::Run (ApplyExplicitCorrection.cpp:489-492): poses after the explicit correction, the correction C, and the
    rows of calculateConstraintTargets as a list of (type, constrained, anchor, d_par, d_perp, d_angle, rel_pen_dir)."""
    P = np.asarray(poses, np.float32).reshape(-1, 3).copy()
    corrections = explicit_corrections(ctype, sel, P, corrected)
    groups, one = [], []
    for i in range(len(P)):                                              # FindContiguousGroups(0, n - 1)
        hit = None
        for j, c in enumerate(corrections):
            if c[0] == i:
                hit = j                                                  # the LAST entry with this pose wins
        if hit is not None:
            one.append(corrections[hit])
        elif one:
            groups.append(one)
            one = []
    if one:
        groups.append(one)
    C = np.zeros(3, np.float32)
    if groups:
        C = np.array(groups[0][0][1], np.float32)                        # only the first contiguous group is applied
        for idx, (tx, ty, th) in groups[0]:
            P[idx, 0] = P[idx, 0] + tx
            P[idx, 1] = P[idx, 1] + ty
            P[idx, 2] = P[idx, 2] + th
        last_pose = groups[0][-1][0]
        lx, ly, lth = groups[0][-1][1]
        for k in range(last_pose + 1, len(P)):                           # the implicit motion of everything behind the group,
            P[k, 2] = P[k, 2] + lth                                      # about the ALREADY MOVED last corrected pose
            base = V2(P[last_pose, 0], P[last_pose, 1])
            ab = V2(P[k, 0], P[k, 1]) - base
            new_ab = rot_apply(lth, ab)
            t = (base + new_ab) + V2(lx, ly)
            P[k, 0], P[k, 1] = t.x, t.y
    # calculateConstraintTargets (:417-487) on the corrected poses
    sp = [V2(*p) for p in np.asarray(sel, np.float32).reshape(4, 2)]
    direction = sp[3] - sp[2]
    correction_angle = atan2f(direction.y, direction.x)
    rows = []
    for a in anchor:
        anchor_angle = f32(P[a, 2])
        d = f32(correction_angle - anchor_angle)
        rel_pen_dir = f32(float(atan2f(sinf(d), cosf(d))) + math.pi / 2.0)      # float + double M_PI / 2, narrowed on assignment
        anchor_loc = V2(P[a, 0], P[a, 1])
        for c in corrected:
            pose_rel = V2(P[c, 0], P[c, 1]) - anchor_loc
            p = V2(cosf(anchor_angle), sinf(anchor_angle))
            n = V2(-p.y, p.x)
            para, perp = p.dot(pose_rel), n.dot(pose_rel)
            dd = f32(f32(P[c, 2]) - anchor_angle)
            angle_dist = atan2f(sinf(dd), cosf(dd))
            rows.append((int(ctype), int(c), int(a), para, perp, angle_dist, rel_pen_dir))
    return P, C, rows


# ---------------------------------------------------------------------------------------------------------------------
# Backprop (COP-SLAM style distribution of the correction over the poses between the bounds)
# ---------------------------------------------------------------------------------------------------------------------
def backprop(poses, cov9, lo, hi, correction):
    """Backprop::Run + BackPropagateError (Backprop.cpp:98-213), the literal O(span^2) float loops."""
    P = np.asarray(poses, np.float32).reshape(-1, 3).copy()
    Cv = np.asarray(cov9, np.float32).reshape(-1, 3, 3).copy()
    if not lo < hi:
        return P, Cv.reshape(-1, 9)
    corr = np.asarray(correction, np.float32)
    dest = V2(P[hi, 0], P[hi, 1]) + V2(corr[0], corr[1])
    dest_rot_var, dest_trans_var = f32(0.0001), f32(0.001)
    rot_sig = [f32(Cv[i, 2, 2]) for i in range(len(Cv))]
    trans_sig = [f32(float(f32(Cv[i, 0, 0] + Cv[i, 1, 1])) / 2.0) for i in range(len(Cv))]   # float + float, / 2.0 (double), push_back narrows
    sum_rot, sum_trans = f32(0.0), f32(0.0)
    for i in range(lo, hi + 1):
        sum_rot = f32(sum_rot + rot_sig[i])
        sum_trans = f32(sum_trans + trans_sig[i])
    sum_rot = f32(sum_rot + dest_rot_var)
    sum_trans = f32(sum_trans + dest_trans_var)
    rot_w = [f32(rot_sig[i] / sum_rot) for i in range(lo, hi + 1)]
    trans_w = [f32(trans_sig[i] / sum_trans) for i in range(lo, hi + 1)]
    rot_beta = f32(f32(1) / f32(f32(1) + f32(rot_sig[hi - 1] / dest_rot_var)))
    trans_beta = f32(f32(1) / f32(f32(1) + f32(trans_sig[hi - 1] / dest_trans_var)))
    for i in range(lo, hi):
        Cv[i, 0, 0] = Cv[i, 0, 0] * trans_beta
        Cv[i, 0, 1] = Cv[i, 0, 1] * trans_beta
        Cv[i, 1, 0] = Cv[i, 1, 0] * trans_beta
        Cv[i, 1, 1] = Cv[i, 1, 1] * trans_beta
        Cv[i, 0, 2] = Cv[i, 0, 2] * rot_beta
        Cv[i, 0, 2] = Cv[i, 0, 2] * rot_beta                   # (0,2) twice, (1,2) never: as written (:164-165)
        Cv[i, 2, 0] = Cv[i, 2, 0] * rot_beta
        Cv[i, 2, 1] = Cv[i, 2, 1] * rot_beta
        Cv[i, 2, 2] = Cv[i, 2, 2] * rot_beta
    theta = f32(corr[2])
    for i in range(lo, hi):                                    # APPLY ROTATION
        dth = f32(rot_w[i - lo] * theta)
        c, s = cosf(dth), sinf(dth)
        pi_ = V2(P[i, 0], P[i, 1])
        o = -pi_                                               # Translation2Df(-poses_[i].translation)
        tr = pi_ + V2(c * o.x + (-s) * o.y, s * o.x + c * o.y)     # (T * R) * T': translation += linear * other
        P[i, 2] = P[i, 2] + dth
        for k in range(i + 1, hi + 1):
            P[k, 2] = P[k, 2] + dth
            v = V2(P[k, 0], P[k, 1])
            w = V2(c * v.x + (-s) * v.y, s * v.x + c * v.y) + tr
            P[k, 0], P[k, 1] = w.x, w.y
    trans = dest - V2(P[hi, 0], P[hi, 1])                      # APPLY TRANSLATION
    for i in range(lo, hi):
        d = trans.scaled(trans_w[i - lo])
        for k in range(i + 1, hi + 1):
            P[k, 0] = P[k, 0] + d.x
            P[k, 1] = P[k, 1] + d.y
    return P, Cv.reshape(-1, 9)


def normalize_angles(poses):
    """HitLSLAM.cpp:442-447: angle = atan2(sin(angle), cos(angle)) in float."""
    P = np.asarray(poses, np.float32).reshape(-1, 3).copy()
    for i in range(len(P)):
        P[i, 2] = atan2f(sinf(P[i, 2]), cosf(P[i, 2]))
    return P


# ---------------------------------------------------------------------------------------------------------------------
# covariance recovery (vector_mapping.cpp:2772-2812)
# ---------------------------------------------------------------------------------------------------------------------
def pose_covariances(problem, poses_f32):
    """ceres::Covariance (DENSE_SVD) blocks (i, i) of the problem evaluated at the given poses, rotated into each pose's frame:
    one_cov = T one_cov T^T with T = blockdiag(Rotation2Df(-angle), 1).  Pose 0 is constant: its block stays zero."""
    poses = np.asarray(poses_f32, np.float32).reshape(-1, 3)
    n = len(poses)
    _, J, _, _ = problem.evaluate(poses.astype(np.float64))
    Jf = J[:, 3:]
    cov = np.linalg.pinv(Jf.T @ Jf)
    out = np.zeros((n, 3, 3))
    for i in range(1, n):
        blk = cov[3 * (i - 1):3 * i, 3 * (i - 1):3 * i]
        a = f32(-poses[i, 2])
        c, s = float(cosf(a)), float(sinf(a))                  # Rotation2Df(-angle).toRotationMatrix().cast<double>()
        T = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        out[i] = T @ blk @ T.T
    return out
