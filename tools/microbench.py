"""Per-kernel timings of the device pipeline on one GPU (development aid; the graded number is bench.py).
usage: python tools/microbench.py [--fs 48000] [--utts 64] [--seconds 10] [--iters 3] [--stages hcds]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import world_class_amd as w  # noqa: E402
from world_class_amd.synth import make_utterance, true_f0  # noqa: E402

KERNELS = ["harvest_decimate", "harvest_bandpass", "harvest_raw", "harvest_refine", "harvest_contour",
           "cheaptrick_frames", "d4c_lovetrain", "d4c_frames", "d4c_bands", "synthesis_timebase", "synthesis_pulses"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fs", type=int, default=48000)
    ap.add_argument("--utts", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--stages", default="hcds")
    a = ap.parse_args()
    L = w.lib()
    fs = a.fs
    nd = min(a.utts, 8)
    base = [make_utterance(fs, a.seconds, 3000 + u) for u in range(nd)]
    tf = [true_f0(fs, a.seconds, 3000 + u) for u in range(nd)]
    xs = [base[u % nd] for u in range(a.utts)]
    xl = [len(x) for x in xs]
    fl = [w.get_samples(fs, n, 5.0) for n in xl]
    yl = [w.synthesis_out_length(n, 5.0, fs) for n in fl]
    frames = sum(fl)
    hv, ct, d4 = w.Harvest(fs), w.CheapTrick(fs), w.D4C(fs)
    sy = w.Synthesis(fs, ct.fft_size, 5.0)
    d_x = w.DeviceArray.from_host(np.concatenate(xs))
    d_t = w.DeviceArray.from_host(np.concatenate([tf[u % nd][0] for u in range(a.utts)]))
    d_f = w.DeviceArray.from_host(np.concatenate([tf[u % nd][1] for u in range(a.utts)]))
    d_sp = w.DeviceArray(frames * ct.bins)
    d_ap = w.DeviceArray(frames * ct.bins)
    d_y = w.DeviceArray(sum(yl))
    L.wc_set_kernel_timing(1)

    def step():
        pos = [0] * a.utts
        if "h" in a.stages:
            hv.compute_device(d_x, xl, d_t, d_f)
        if "c" in a.stages:
            pos = ct.compute_device(d_x, xl, d_t, d_f, fl, d_sp, rng_pos=pos)
        if "d" in a.stages:
            pos = d4.compute_device(d_x, xl, d_t, d_f, fl, ct.fft_size, d_ap, rng_pos=pos)
        if "s" in a.stages:
            if "c" not in a.stages or "d" not in a.stages:
                raise SystemExit("synthesis needs c and d in --stages")
            sy.compute_device(d_f, fl, d_sp, d_ap, yl, d_y, rng_pos=pos)
        L.wc_synchronize()

    step()
    best = {}
    wall = []
    for _ in range(a.iters):
        t0 = time.perf_counter()
        step()
        wall.append(time.perf_counter() - t0)
        for k in KERNELS:
            ms = L.wc_last_kernel_ms(k.encode())
            if ms >= 0:
                best[k] = min(best.get(k, 1e9), ms)
    t = min(wall)
    print(f"fs={fs} utts={a.utts} frames={frames} stages={a.stages}: wall {t*1e3:.1f} ms -> {frames/t/1e6:.3f} Mframes/s")
    for k in KERNELS:
        if k in best:
            print(f"  {k:20s} {best[k]:8.2f} ms")
    print(f"  {'sum of kernels':20s} {sum(best.values()):8.2f} ms")


if __name__ == "__main__":
    main()
