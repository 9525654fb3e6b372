"""Parity sweep beyond the committed goldens (development aid, run on a GPU box): N seeded 48 kHz utterances through the fused
pipeline against the CPU oracle; prints the worst deviations and any voiced/unvoiced disagreement.
    python tools/parity_sweep.py [--n 32] [--seconds 10] [--fs 48000] [--first-seed 9000] [--floor 71] [--frame-period 5] [--ragged]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import world_class_amd as w  # noqa: E402
from oracle import port  # noqa: E402  (tools/ may use the oracle as the checker, like tests/)
from world_class_amd.synth import make_utterance  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--fs", type=int, default=48000)
    ap.add_argument("--first-seed", type=int, default=9000)
    ap.add_argument("--floor", type=float, default=71.0)
    ap.add_argument("--frame-period", type=float, default=5.0)
    ap.add_argument("--ragged", action="store_true", help="utterance i lasts seconds * (0.2 + 0.8 * ((i * 7) % 10) / 9)")
    a = ap.parse_args()
    dur = [a.seconds * (0.2 + 0.8 * ((i * 7) % 10) / 9) if a.ragged else a.seconds for i in range(a.n)]
    xs = [make_utterance(a.fs, dur[i], a.first_seed + i) for i in range(a.n)]
    res = w.Pipeline(a.fs, frame_period=a.frame_period, harvest_f0_floor=a.floor).run_batch(xs)
    P = port.Port()
    P.set_threads(os.cpu_count() or 1)
    worst = dict(f0=0.0, sp=0.0, ap=0.0, y=0.0)
    flips = 0
    for i, (x, r) in enumerate(zip(xs, res)):
        o = P.pipeline(x, a.fs, harvest_floor=a.floor, frame_period=a.frame_period)
        fl = int(np.sum((r["f0"] == 0) != (o["f0"] == 0)))
        flips += fl
        same = (r["f0"] == 0) == (o["f0"] == 0)
        e = dict(f0=np.abs(r["f0"] - o["f0"])[same].max(), sp=(np.abs(r["sp"] - o["sp"]) / o["sp"]).max(),
                 ap=np.abs(r["ap"] - o["ap"]).max(), y=np.abs(r["y"] - o["y"]).max())
        for k in worst:
            worst[k] = max(worst[k], float(e[k]))
        if fl or e["f0"] > 1e-6 or e["sp"] > 1e-7 or e["ap"] > 1e-7 or e["y"] > 1e-8:
            print("seed", a.first_seed + i, "V/UV flips", fl, {k: "%.2e" % v for k, v in e.items()})
    print("fs", a.fs, "floor", a.floor, "hop", a.frame_period, "utterances", a.n, "V/UV flips", flips, "worst", {k: "%.2e" % v for k, v in worst.items()})


if __name__ == "__main__":
    main()
