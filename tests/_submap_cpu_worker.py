"""Worker of tests/test_submap_cpu.py (gloo, no GPU): the exchange pattern and interface
algebra of the pose-range submap solve (hitl_lm.cpp: linear_solve_and_step_dist), mirrored
in numpy on systems assembled by the oracle."""
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

    n_per = 15
    prob = Problem(n_per * world, 24, ppf=n_per, groups=1, kind=0)
    corr = oracle_corr(prob)
    # the product's own partitioner (host-only C ABI call, runs without a GPU)
    b, e = partition_range(prob.off, rank, world)
    assert (b, e) == (rank * n_per, (rank + 1) * n_per)
    o = prob.load(O.OracleSolver(), corr=corr)
    rng = np.random.default_rng(11)
    x = prob.init.astype(np.float64) + 0.02 * rng.standard_normal(prob.init.shape)
    H, g, _ = o.assemble_dense(2, prob.init, x)            # global J^T J, gradient (pose 0 = identity rows)
    d = 3 * prob.n
    scale = np.ones(d)
    scale[3:] = 1.0 / (1.0 + np.sqrt(np.diag(H)[3:]))
    mu = 1e4
    A = H * scale[:, None] * scale[None, :]
    dg = np.clip(np.diag(A).copy(), 1e-6, 1e32) / mu
    dg[:3] = 0.0
    A = A + np.diag(dg)
    rhs = -scale * g
    ref = np.linalg.solve(A, rhs)                          # what the single-GPU path computes

    # ---- what rank r holds: its diagonal block, its right-hand side, the coupling to its neighbours
    lo, hi = 3 * b, 3 * e
    M, br = A[lo:hi, lo:hi], rhs[lo:hi]
    nloc = hi - lo
    E = np.zeros((nloc, 6))
    E[[0, 1, 2], [0, 1, 2]] = 1.0
    E[[nloc - 3, nloc - 2, nloc - 1], [3, 4, 5]] = 1.0
    z0 = np.linalg.solve(M, br)
    Z = np.linalg.solve(M, E)
    rows = [0, 1, 2, nloc - 3, nloc - 2, nloc - 1]
    rec = np.concatenate([z0[rows], Z[rows, :].T.ravel()])   # 6 + 36, column-major like k_gather_iface
    C_own = A[lo - 3:lo, lo:lo + 3].ravel() if rank > 0 else np.zeros(9)   # block (last of r-1, first of r)
    send = torch.from_numpy(np.concatenate([rec, C_own]))
    allr = [torch.zeros_like(send) for _ in range(world)]
    dist.all_gather(allr, send)                            # ONE tiny all-gather, rank order
    allr = [t.numpy() for t in allr]
    # off-band structure the submap solve requires: nothing but the cut blocks couples the ranks
    other = A[lo:hi, :].copy()
    other[:, lo:hi] = 0
    if rank > 0:
        other[0:3, lo - 3:lo] = 0
    if rank < world - 1:
        other[nloc - 3:nloc, hi:hi + 3] = 0
    assert np.abs(other).max() == 0.0

    NI = 6 * world
    S, y = np.eye(NI), np.zeros(NI)
    for gk in range(world):
        rg = allr[gk]
        y[6 * gk:6 * gk + 6] = rg[:6]
        Zg = rg[6:42].reshape(6, 6)                        # Zg[c][r]
        if gk >= 1:
            Ct = allr[gk][42:].reshape(3, 3)
            for r in range(6):
                for k in range(3):
                    S[6 * gk + r, 6 * (gk - 1) + 3 + k] += sum(Zg[c][r] * Ct[k][c] for c in range(3))
        if gk + 1 < world:
            Ct = allr[gk + 1][42:].reshape(3, 3)
            for r in range(6):
                for k in range(3):
                    S[6 * gk + r, 6 * (gk + 1) + k] += sum(Zg[3 + c][r] * Ct[c][k] for c in range(3))
    S_iface = S.copy()                                     # the interface matrix (np.linalg.solve below leaves it alone, the copy is for the covariance check)
    y = np.linalg.solve(S, y)
    coef = np.zeros(6)
    if rank >= 1:
        coef[:3] = allr[rank][42:].reshape(3, 3).T @ y[6 * (rank - 1) + 3:6 * (rank - 1) + 6]
    if rank + 1 < world:
        coef[3:] = allr[rank + 1][42:].reshape(3, 3) @ y[6 * (rank + 1):6 * (rank + 1) + 3]
    dt = z0 - Z @ coef
    err = float(np.abs(dt - ref[lo:hi]).max() / np.abs(ref).max())
    # ---- pose covariances on the ranks (hitl_lm.cpp: dist_cov_correction): with only this rank's right-hand side non-zero the
    # interface solve gives x_g = M^-1 e - Z (B A^-1 P) Z^T e, i.e. the diagonal block of the GLOBAL inverse over this rank's
    # rows is M^-1 - Z T Z^T with T = B A^-1 P (6 x 6); compare its 3x3 pose blocks with the dense global inverse
    T = np.zeros((6, 6))
    for c in range(6):
        rhs_i = np.zeros(NI)
        rhs_i[6 * rank + c] = 1.0
        yy = np.linalg.solve(S_iface, rhs_i)
        if rank >= 1:
            T[:3, c] = allr[rank][42:].reshape(3, 3).T @ yy[6 * (rank - 1) + 3:6 * (rank - 1) + 6]
        if rank + 1 < world:
            T[3:, c] = allr[rank + 1][42:].reshape(3, 3) @ yy[6 * (rank + 1):6 * (rank + 1) + 3]
    Minv = np.linalg.inv(M)
    Sig_local = Minv - Z @ T @ Z.T
    Sig_global = np.linalg.inv(A)[lo:hi, lo:hi]
    cov_err = 0.0
    for k in range(nloc // 3):
        blk = slice(3 * k, 3 * k + 3)
        if lo + 3 * k == 0:
            continue                                       # global pose 0: identity rows
        cov_err = max(cov_err, float(np.abs(Sig_local[blk, blk] - Sig_global[blk, blk]).max() / np.abs(Sig_global[blk, blk]).max()))
    # rank-ordered scalar sums (cost, dt.rhs, ...) are identical on every rank
    part = torch.tensor([float(dt @ br), float(np.sum((scale[lo:hi] * dt) ** 2))], dtype=torch.float64)
    parts = [torch.zeros_like(part) for _ in range(world)]
    dist.all_gather(parts, part)
    tot = sum(p.numpy() for p in parts)
    json.dump(dict(rank=rank, err=err, cov_err=cov_err, tot=tot.tolist(), ref=[float(ref @ rhs), float(np.sum((scale * ref)[3:] ** 2))]),
              open("%s.%d" % (out_path, rank), "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
