"""Test-infrastructure helper (not collected by pytest): how the CPU oracle's solve scales with OpenMP threads on THIS
host.  Run on the GPU box to choose and justify the thread count of bench.py's cpu_baseline leg."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from tools import hitl_synth as S, probe_host
    from oracle import hitl_oracle as O
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    print(json.dumps(probe_host.probe()))
    p = S.params(n, 400, seed=20264102, poses_per_floor=200)
    gt, init, cov = S.poses(p)
    off, px, py, nx, ny = S.scans(p)
    rows = S.constraints(p, gt, max(1, n // 1000), kind=0)
    o = O.OracleSolver()
    o.set_map(off, px, py, nx, ny, cov)
    o.set_human_constraints(rows)
    O.set_num_threads(probe_host.usable_cores())
    t = time.time()
    o.find_correspondences(init, 3, 0)
    print("match %.2f s" % (time.time() - t))
    P, Cn = o.correspondence_counts()
    cores = probe_host.usable_cores()
    for th in [1, 2, 4, 8, 16, 32, 64, 128, 256]:
        if th > max(cores, 8):
            break
        O.set_num_threads(th)
        t = time.time()
        out, s = o.solve(2, init)
        dt = time.time() - t
        print("threads %3d solve %.3f s iters %d linearize %.1f ms/call linsolve %.1f ms/iter  match-evals/s/thread %.3g total %.3g"
              % (th, dt, s.iterations, s.linearize_ms / s.linearize_calls, s.linear_solver_ms / max(s.iterations, 1),
                 Cn * s.linearize_calls / (s.linearize_ms / 1e3) / th, Cn * s.linearize_calls / (s.linearize_ms / 1e3)))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
