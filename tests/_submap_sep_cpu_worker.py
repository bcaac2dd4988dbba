"""Worker of tests/test_submap_cpu.py (gloo, no GPU): the separator form of the pose-range submap solve (hitl_lm.cpp:
linear_solve_and_step_dist with dist_sep; hitl_cr.hip: k_sep_record / k_sep_build / k_sep_scatter), mirrored in numpy on systems
assembled by the oracle.  Each rank keeps its first and last owned super-block (F, R), eliminates everything between them,
all-gathers the Schur complement onto the two, solves the chain of 2 * world separators and back-substitutes."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_path = sys.argv[1]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from conftest import Problem, oracle_corr
    from hitl_slam_amd.api import partition_range
    from oracle import hitl_oracle as O

    Wb, B = 3, 9
    n = 23 * world + 4                                      # not a multiple of anything: the aligned partitioner has to move the cuts
    prob = Problem(n, 24, ppf=n, groups=1, kind=0)          # one corridor: scan pairs up to three poses long cross every cut
    corr = oracle_corr(prob)
    # the product's own partitioner (host-only C ABI call): cuts on multiples of 12 poses, every rank at least two blocks
    b, e = partition_range(prob.off, rank, world, align=12)
    assert b % 12 == 0 and (e % 12 == 0 or rank == world - 1) and e - b >= 24
    cuts = [partition_range(prob.off, r, world, align=12) for r in range(world)]
    assert cuts[0][0] == 0 and cuts[-1][1] == n and all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
    o = prob.load(O.OracleSolver(), corr=corr)
    rng = np.random.default_rng(11)
    x = prob.init.astype(np.float64) + 0.02 * rng.standard_normal(prob.init.shape)
    H, g, _ = o.assemble_dense(2, prob.init, x)            # global J^T J, gradient (pose 0 = identity rows)
    d = 3 * n
    scale = np.ones(d)
    scale[3:] = 1.0 / (1.0 + np.sqrt(np.diag(H)[3:]))
    mu = 1e4
    A = H * scale[:, None] * scale[None, :]
    dg = np.clip(np.diag(A).copy(), 1e-6, 1e32) / mu
    dg[:3] = 0.0
    A = A + np.diag(dg)
    rhs = -scale * g
    ref = np.linalg.solve(A, rhs)                          # what the single-GPU path computes
    src, tgt = corr[0], corr[1]
    assert np.abs(src - tgt).max() == Wb                   # the band really is three poses wide

    lo, hi = 3 * b, 3 * e
    nloc = hi - lo
    nblk = -(-(e - b) // Wb)
    pad = nblk * B - nloc                                  # the last rank's last block may be short: identity padding rows
    M = np.eye(nblk * B)
    M[:nloc, :nloc] = A[lo:hi, lo:hi]
    w = np.zeros(nblk * B)
    w[:nloc] = rhs[lo:hi]
    F, R = slice(0, B), slice((nblk - 1) * B, nblk * B)
    I = slice(B, (nblk - 1) * B)
    # what the separator form requires: the interior couples to nothing outside the rank, F only to the block below the cut
    # and R only to the block above it
    outside = A[lo:hi, :].copy()
    outside[:, lo:hi] = 0
    assert np.abs(outside[B:nloc - (B - pad)]).max() == 0.0
    if rank > 0:
        blk = outside[:B].copy(); blk[:, lo - B:lo] = 0
        assert np.abs(blk).max() == 0.0 and np.abs(outside[:B, lo - B:lo]).max() > 0.0
    # ---- forward: Schur complement of the interior onto (F, R)
    Mii = M[I, I]
    XiF, XiR, Xiw = np.linalg.solve(Mii, M[I, F]), np.linalg.solve(Mii, M[I, R]), np.linalg.solve(Mii, w[I])
    DF = M[F, F] - M[F, I] @ XiF
    DR = M[R, R] - M[R, I] @ XiR
    K = M[R, F] - M[R, I] @ XiF                            # coupling R <-> F (rows R): the direct block (two-block ranks) + fill-in
    wF = w[F] - M[F, I] @ Xiw
    wR = w[R] - M[R, I] @ Xiw
    Clo = np.zeros((B, B))
    if rank > 0:
        Clo = A[lo:lo + B, lo - B:lo].copy()               # F x the previous rank's R
    send = torch.from_numpy(np.concatenate([DF.ravel(), DR.ravel(), K.ravel(), Clo.ravel(), wF, wR]))
    allr = [torch.zeros_like(send) for _ in range(world)]
    dist.all_gather(allr, send)                            # ONE all-gather of 4 B^2 + 2 B doubles, rank order
    allr = [t.numpy() for t in allr]
    # ---- the chain of 2 * world separators (k_sep_build): node 2g = F_g, node 2g + 1 = R_g
    T = 2 * world
    C = np.zeros((T * B, T * B))
    y = np.zeros(T * B)
    for gk in range(world):
        rg = allr[gk]
        f, r = slice(2 * gk * B, (2 * gk + 1) * B), slice((2 * gk + 1) * B, (2 * gk + 2) * B)
        C[f, f] = rg[0:81].reshape(B, B)
        C[r, r] = rg[81:162].reshape(B, B)
        C[r, f] = rg[162:243].reshape(B, B)
        C[f, r] = C[r, f].T
        if gk > 0:
            p = slice((2 * gk - 1) * B, 2 * gk * B)
            C[f, p] = rg[243:324].reshape(B, B)
            C[p, f] = C[f, p].T
        y[f], y[r] = rg[324:333], rg[333:342]
    xs = np.linalg.solve(C, y)
    # ---- back-substitution between this rank's own two separators
    xF, xR = xs[2 * rank * B:(2 * rank + 1) * B], xs[(2 * rank + 1) * B:(2 * rank + 2) * B]
    xI = Xiw - XiF @ xF - XiR @ xR
    dt = np.concatenate([xF, xI, xR])[:nloc]
    err = float(np.abs(dt - ref[lo:hi]).max() / np.abs(ref).max())
    # the neighbours' separators cover this rank's halo poses (k_sep_scatter): the last Wb poses below, the first Wb above the cut
    halo_err = 0.0
    if rank > 0:
        halo_err = max(halo_err, float(np.abs(xs[(2 * rank - 1) * B:2 * rank * B] - ref[lo - B:lo]).max() / np.abs(ref).max()))
    if rank < world - 1:
        halo_err = max(halo_err, float(np.abs(xs[(2 * rank + 2) * B:(2 * rank + 3) * B] - ref[hi:hi + B]).max() / np.abs(ref).max()))
    part = torch.tensor([float(dt @ rhs[lo:hi]), float(np.sum((scale[lo:hi] * dt) ** 2))], dtype=torch.float64)
    parts = [torch.zeros_like(part) for _ in range(world)]
    dist.all_gather(parts, part)
    tot = sum(p.numpy() for p in parts)
    json.dump(dict(rank=rank, begin=b, end=e, err=err, halo_err=halo_err, tot=tot.tolist(),
                   ref=[float(ref @ rhs), float(np.sum((scale * ref)[3:] ** 2))]), open("%s.%d" % (out_path, rank), "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
