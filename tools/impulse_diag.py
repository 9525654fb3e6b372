"""Impulse trains (period a whole number of decimated samples) through Harvest with the sliding-DFT band-pass (default) and with the
direct FIR evaluation of the same filter (WC_HARVEST_BANDPASS=fir) against the CPU oracle: voicing flips and largest F0 deviation
(development aid, round 5: where do the flips of profiles/r05_parity_sweep_third_seeds.txt come from?)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import world_class_amd as w  # noqa: E402
from oracle import port  # noqa: E402
from world_class_amd.synth import SIGNAL_KINDS, make_signal  # noqa: E402

fs = 16000
P = port.Port()
seeds = [1440023, 1440043, 1340043, 1340003, 230003, 230013, 230023, 230033, 1440003, 1440013, 1440033, 1440053, 1440063, 1440073, 1440083, 1440093]
seeds = [s for s in seeds if SIGNAL_KINDS[s % len(SIGNAL_KINDS)] == "impulses"]
xs = [make_signal(fs, 3.0, s) for s in seeds]
refs = [P.harvest(x, fs)[1] for x in xs]
for mode in ("sdft_no_rerun", "default", "fir"):
    os.environ.pop("WC_HARVEST_BANDPASS", None)
    os.environ.pop("WC_HARVEST_TIES", None)
    if mode == "fir":
        os.environ["WC_HARVEST_BANDPASS"] = "fir"
    if mode == "sdft_no_rerun":
        os.environ["WC_HARVEST_TIES"] = "ignore"
    h = w.Harvest(fs)
    for s, x, r in zip(seeds, xs, refs):
        _, f = h.compute(x)
        flips = int(((f == 0) != (r == 0)).sum())
        v = (f > 0) & (r > 0)
        dev = float(np.abs(f - r)[v].max()) if v.any() else 0.0
        raw = h.debug_fetch("raw")
        print(mode, s, "flips", flips, "max dev %.3e" % dev, "frames", len(f), flush=True)

# the fused pipeline (its own retry loop): the flagged group runs again on the FIR twin
os.environ.pop("WC_HARVEST_BANDPASS", None)
os.environ.pop("WC_HARVEST_TIES", None)
res = w.Pipeline(fs).run_batch(xs)
for s_, r, o in zip(seeds, refs, res):
    f = o["f0"]
    v = (f > 0) & (r > 0)
    print("pipeline", s_, "flips", int(((f == 0) != (r == 0)).sum()), "max dev %.3e" % (float(np.abs(f - r)[v].max()) if v.any() else 0.0), flush=True)
# natural utterances must not raise the flag: the stage call's time with and without acting on it
import time
from world_class_amd.synth import make_utterance
ux = [make_utterance(48000, 10.0, 3000 + i) for i in range(8)]
for mode in ("default", "ignore"):
    if mode == "ignore":
        os.environ["WC_HARVEST_TIES"] = "ignore"
    h = w.Harvest(48000)
    h.compute_batch(ux)
    t0 = time.perf_counter()
    for _ in range(3):
        h.compute_batch(ux)
    print("8 x 10 s natural utterances, ties", mode, "%.2f ms per call" % ((time.perf_counter() - t0) / 3 * 1e3), "twin created:", "n/a", flush=True)
