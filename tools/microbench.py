"""Stage micro-benchmarks on one GPU (not the graded bench: see bench.py).
usage: python tools/microbench.py cheaptrick --fs 48000 --utts 64 --seconds 10"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import world_class_amd as w  # noqa: E402
from world_class_amd.synth import make_utterance, true_f0  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stage")
    ap.add_argument("--fs", type=int, default=48000)
    ap.add_argument("--utts", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    L = w.lib()
    fs = a.fs
    base = [make_utterance(fs, a.seconds, 3000 + u) for u in range(min(a.utts, 8))]
    tf = [true_f0(fs, a.seconds, 3000 + u) for u in range(min(a.utts, 8))]
    xs = [base[u % len(base)] for u in range(a.utts)]
    tfs = [tf[u % len(tf)] for u in range(a.utts)]
    xl = [len(x) for x in xs]
    fl = [len(t) for t, _ in tfs]
    d_x = w.DeviceArray.from_host(np.concatenate(xs))
    d_t = w.DeviceArray.from_host(np.concatenate([t for t, _ in tfs]))
    d_f = w.DeviceArray.from_host(np.concatenate([f for _, f in tfs]))
    frames = sum(fl)
    L.wc_set_kernel_timing(1)
    if a.stage == "cheaptrick":
        st = w.CheapTrick(fs)
        d_o = w.DeviceArray(frames * st.bins)
        run = lambda: st.compute_device(d_x, xl, d_t, d_f, fl, d_o)
        kname = b"cheaptrick_frames"
        bytes_per_frame = 8 * st.fft_size + 8 * st.bins
    else:
        raise SystemExit("unknown stage")
    run(); L.wc_synchronize()
    ts, ks = [], []
    for _ in range(a.iters):
        t0 = time.perf_counter(); run(); L.wc_synchronize(); ts.append(time.perf_counter() - t0)
        ks.append(L.wc_last_kernel_ms(kname))
    t = min(ts); k = min(ks)
    print(f"{a.stage} fs={fs} utts={a.utts} frames={frames}: wall {t*1e3:.2f} ms -> {frames/t/1e6:.3f} Mframes/s; "
          f"kernel {k:.2f} ms -> {frames/k/1e3:.3f} Mframes/s, {frames*bytes_per_frame/k/1e6:.1f} GB/s algorithmic")


if __name__ == "__main__":
    main()
