#!/usr/bin/env python3
"""Expected per-LM-iteration time of the pose-range submap path on 1 / 2 / 4 / 8 GPUs, built from the single-GPU kernel traces
(profiles/r03b_bench_c4_kernel_stats.csv, the forced-dist trace of tools/dev/dist_trace.sh quoted in DESIGN.md section 6) and one
free parameter, the cost of a small all-gather + host synchronisation over RCCL (unmeasured: no multi-GPU node was available).
It exists so that the first hardware curve has something to be held to.  Prints a markdown table.

Model (microseconds per LM iteration, a rank holding n_r poses; C4 = 50 000 poses x 400 observations, C5 = 512 observations):
  linearise   255 * (n_r / 50k) * (obs / 400)      floor 15     k_linearize_stf_r, proportional to the matches of the rank
  assemble     23 * (n_r / 50k)                    floor 6
  band pass    11 * (n_r / 50k)                    floor 5
  forward 1    30 + 45 * (n_r / 50k)                            first cyclic-reduction stage with the 6 interface columns riding (75 at 50k)
  forward 2+   5 per level, levels = ceil(log2(max(n_r / 48, 2)))   second stage + top: dependent-latency chains (61 at 50k)
  back         20 + 17 * (n_r / 50k)                            7 right-hand sides (37 at 50k)
  interface    27 + 0.2 * (6 * world)                           gather / cross-cut blocks / combine / copies + the dense host solve
  exchanges    2 * t_ex                                         interface record, step scalars
  host         28                                               launch gaps + decision
A single GPU runs the plain path: 437 us per iteration at C4 (measured), scaled by matches for C5 on one GPU.
"""
import math
import sys


def rank_iteration_us(n_r, obs, world, t_ex):
    f = n_r / 50000.0
    lin = max(15.0, 255.0 * f * obs / 400.0)
    asm = max(6.0, 23.0 * f)
    band = max(5.0, 11.0 * f)
    fwd1 = 30.0 + 45.0 * f
    fwd2 = 5.0 * math.ceil(math.log2(max(n_r / 48.0, 2.0)))
    back = 20.0 + 17.0 * f
    iface = 27.0 + 0.2 * 6 * world
    return lin + asm + band + fwd1 + fwd2 + back + iface + 2.0 * t_ex + 28.0


def single_gpu_us(n, obs):
    f = n / 50000.0
    return 255.0 * f * obs / 400.0 + (437.0 - 255.0) * (0.45 + 0.55 * f)      # the solve side grows sub-linearly (log-depth stages)


def main():
    t_ex = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0
    print("| workload | GPUs | poses per rank | us per LM iteration | speed-up / efficiency |")
    print("|---|---|---|---|---|")
    for name, n, obs in (("strong, C4 (50 000 x 400)", 50000, 400), ("strong, C5 (200 000 x 512)", 200000, 512)):
        t1 = single_gpu_us(n, obs)
        for w in (1, 2, 4, 8):
            t = t1 if w == 1 else rank_iteration_us(n / w, obs, w, t_ex)
            print("| %s | %d | %d | %.0f | %.2fx |" % (name, w, n // w, t, t1 / t))
    t1 = single_gpu_us(50000, 400)
    for w in (1, 2, 4, 8):
        t = t1 if w == 1 else rank_iteration_us(50000, 400, w, t_ex)
        print("| weak, C4 per GPU | %d | 50000 | %.0f | %.0f %% |" % (w, t, 100.0 * t1 / t))
    print("\n(t_ex = %.0f us per all-gather + host synchronisation; python tools/scaling_model.py <t_ex> for another value)" % t_ex)


if __name__ == "__main__":
    main()
