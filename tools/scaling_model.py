#!/usr/bin/env python3
"""Expected per-LM-iteration time of the pose-range submap path on 1 / 2 / 4 / 8 GPUs, built from the single-GPU kernel traces
(the forced-dist traces of tools/dev/dist_trace.sh: profiles/r04_dist_trace.txt) and two free parameters, the cost of a small
all-gather over RCCL with a host synchronisation behind it (t_ex) and without one (t_dev) -- unmeasured: no multi-GPU node was
available.  It exists so that the first hardware curve has something to be held to.  Prints a markdown table per form.

Model (microseconds per LM iteration, a rank holding n_r poses; C4 = 50 000 poses x 400 observations, C5 = 512 observations):
  linearise   245 * (n_r / 50k) * (obs / 400)      floor 15     k_linearize_stf_r, proportional to the matches of the rank
  assemble     23 * (n_r / 50k)                    floor 6
  band pass    10 * (n_r / 50k)                    floor 5
 separator form (round 4; every rank but the last owns a multiple of the pose band):
  forward 1    18 + 32 * (n_r / 50k)                            first cyclic-reduction stage, one right-hand side (50 at 50k)
  forward 2+   5 per level, levels = ceil(log2(max(n_r / 48, 2)))   second stage + the last chunk: dependent-latency chains (50 at 50k)
  separators   14 + 6 + 2 * ceil(log2(2 * world)) + 5           record / build / scatter kernels, the chain of 2 * world separators (forward, back)
  back         12 + 10 * (n_r / 50k)
  exchanges    t_dev + t_ex                                     the record (device to device, on the stream) + the step scalars
 interface-column form (rounds 2-3; any partition):
  forward 1    30 + 45 * (n_r / 50k)                            the 6 interface columns riding (75 at 50k; 18 columns when pairs cross the cut)
  forward 2+   as above + 11
  back         20 + 17 * (n_r / 50k)                            7 right-hand sides (37 at 50k)
  interface    27 + 0.2 * (6 * world)                           gather / cross-cut blocks / combine / copies + the dense host solve
  exchanges    2 * t_ex
  host         28                                               launch gaps + decision (both forms)
A single GPU runs the plain path: 437 us per iteration at C4 (measured), scaled by matches for C5 on one GPU.
"""
import math
import sys


def rank_iteration_us(n_r, obs, world, t_ex, t_dev, form):
    f = n_r / 50000.0
    lin = max(15.0, 245.0 * f * obs / 400.0)
    asm = max(6.0, 23.0 * f)
    band = max(5.0, 10.0 * f)
    levels = math.ceil(math.log2(max(n_r / 48.0, 2.0)))
    if form == "separators":
        fwd1 = 18.0 + 32.0 * f
        fwd2 = 5.0 * levels
        sep = 14.0 + 6.0 + 2.0 * math.ceil(math.log2(2 * world)) + 5.0
        back = 12.0 + 10.0 * f
        return lin + asm + band + fwd1 + fwd2 + sep + back + t_dev + t_ex + 28.0
    fwd1 = 30.0 + 45.0 * f
    fwd2 = 5.0 * levels + 11.0
    back = 20.0 + 17.0 * f
    iface = 27.0 + 0.2 * 6 * world
    return lin + asm + band + fwd1 + fwd2 + back + iface + 2.0 * t_ex + 28.0


def single_gpu_us(n, obs):
    f = n / 50000.0
    return 245.0 * f * obs / 400.0 + (437.0 - 245.0) * (0.45 + 0.55 * f)      # the solve side grows sub-linearly (log-depth stages)


def main():
    t_ex = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0
    t_dev = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
    for form in ("separators", "interface columns"):
        print("**%s**\n" % form)
        print("| workload | GPUs | poses per rank | us per LM iteration | speed-up / efficiency |")
        print("|---|---|---|---|---|")
        for name, n, obs in (("strong, C4 (50 000 x 400)", 50000, 400), ("strong, C5 (200 000 x 512)", 200000, 512)):
            t1 = single_gpu_us(n, obs)
            for w in (1, 2, 4, 8):
                t = t1 if w == 1 else rank_iteration_us(n / w, obs, w, t_ex, t_dev, form)
                print("| %s | %d | %d | %.0f | %.2fx |" % (name, w, n // w, t, t1 / t))
        t1 = single_gpu_us(50000, 400)
        for w in (1, 2, 4, 8):
            t = t1 if w == 1 else rank_iteration_us(50000, 400, w, t_ex, t_dev, form)
            print("| weak, C4 per GPU | %d | 50000 | %.0f | %.0f %% |" % (w, t, 100.0 * t1 / t))
        print()
    print("(t_ex = %.0f us per all-gather + host synchronisation, t_dev = %.0f us per all-gather on the stream; "
          "python tools/scaling_model.py <t_ex> <t_dev> for other values)" % (t_ex, t_dev))


if __name__ == "__main__":
    main()
