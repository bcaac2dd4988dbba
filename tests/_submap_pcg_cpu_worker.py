"""Worker of tests/test_submap_cpu.py::test_distributed_pcg_* (gloo, no GPU): the exchange pattern and algebra of the
distributed PCG of the submap solve (hitl_lm.cpp: dist_gather_owned, dist_pcg_solve), mirrored in numpy on a system
assembled by the oracle from an ALL-PAIRS correspondence set (pose pairs far outside any band, inside and across the cut).

Per CG iteration: one all-gather carrying (r.z, r.r, b.b) and the owned part of z -- the halo rows of z are filled from
it --, one scalar all-gather for p.q; block-Jacobi preconditioner = each rank's own diagonal block; scalar sums in rank
order so that all ranks take identical decisions."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def gather(rank_vec):
    t = torch.from_numpy(np.ascontiguousarray(rank_vec, dtype=np.float64))
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.numpy() for o in out]


def main():
    out_path = sys.argv[1]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from conftest import Problem, oracle_corr
    from oracle import hitl_oracle as O

    n_per = 12
    prob = Problem(n_per * world, 64, ppf=n_per * world, groups=1)
    corr = oracle_corr(prob, window=0)                      # all pairs: the reference's search
    assert np.abs(corr[0] - corr[1]).max() > 4
    o = prob.load(O.OracleSolver(), corr=corr)
    rng = np.random.default_rng(13)
    x = prob.init.astype(np.float64) + 0.02 * rng.standard_normal(prob.init.shape)
    H, g, _ = o.assemble_dense(2, prob.init, x)
    d = 3 * prob.n
    scale = np.ones(d)
    scale[3:] = 1.0 / (1.0 + np.sqrt(np.diag(H)[3:]))
    mu = 1e4
    A = H * scale[:, None] * scale[None, :]
    dg = np.clip(np.diag(A).copy(), 1e-6, 1e32) / mu
    dg[:3] = 0.0
    A = A + np.diag(dg)
    rhs = -scale * g
    ref = np.linalg.solve(A, rhs)

    # what rank r holds: the rows of its owned poses (all columns: the whole map is its halo), its right-hand side
    b, e = rank * n_per, (rank + 1) * n_per
    lo, hi = 3 * b, 3 * e
    rows, M, br = A[lo:hi, :], A[lo:hi, lo:hi], rhs[lo:hi]
    far = rows.copy()
    far[:, max(lo - 12, 0):hi + 12] = 0.0
    assert np.abs(far).max() > 0                            # the ranks are coupled far beyond a 4-pose band
    xk = np.zeros(hi - lo)
    r = br.copy()
    p_full = np.zeros(d)                                    # owned rows + halo rows (= every other rank's rows)
    rz_old, target, iters = 0.0, 0.0, 0
    decisions = []
    for it in range(200):
        z = np.linalg.solve(M, r)                           # block-Jacobi: the rank's own factorisation
        rec = gather(np.concatenate([[float(r @ z), float(r @ r), float(br @ br)], z]))   # exchange 1
        rz = sum(x_[0] for x_ in rec); rr = sum(x_[1] for x_ in rec); bb = sum(x_[2] for x_ in rec)   # rank order
        z_full = np.concatenate([x_[3:] for x_ in rec])     # halo rows of z: the owners' values
        if it == 0:
            target = (1e-12 ** 2) * bb
        decisions.append((rz, rr))
        if rr <= target:
            break
        beta = 0.0 if it == 0 else rz / rz_old
        p_full = z_full + beta * p_full                     # halo rows: owners' z + this rank's copy of their p
        q = rows @ p_full
        pq = sum(x_[0] for x_ in gather(np.array([float(p_full[lo:hi] @ q)])))      # exchange 2
        alpha = rz / pq
        xk += alpha * p_full[lo:hi]
        r -= alpha * q
        rz_old = rz
        iters += 1
    err = float(np.abs(xk - ref[lo:hi]).max() / np.abs(ref).max())
    json.dump(dict(rank=rank, err=err, iters=iters, decisions=decisions), open("%s.%d" % (out_path, rank), "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
