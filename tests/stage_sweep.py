"""Stage-level parity sweep (development aid, run on a GPU box): CheapTrick, D4C and Synthesis through the C-ABI against the CPU
oracle on F0 contours Harvest would never produce -- random plateaus between 30 and 1300 Hz (below the floors, above the
ceiling), single voiced frames, all-unvoiced and all-voiced stretches -- at several rates and hops.
    python tests/stage_sweep.py [--n 40] [--first-seed 70000]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import world_class_amd as w  # noqa: E402
from oracle import port  # noqa: E402  (a checker, which is why it lives under tests/)
from world_class_amd.synth import make_utterance  # noqa: E402


def contour(rng, n):
    f0 = np.zeros(n)
    i = 0
    while i < n:
        ln = int(rng.integers(1, 60))
        kind = rng.integers(0, 5)
        if kind == 0:
            v = 0.0
        elif kind == 1:
            v = rng.uniform(30.0, 90.0)
        elif kind == 2:
            v = rng.uniform(700.0, 1300.0)
        else:
            v = rng.uniform(80.0, 500.0)
        seg = v * (1 + 0.01 * rng.normal(size=ln)) if v > 0 and rng.uniform() < 0.5 else np.full(ln, v)
        f0[i:i + ln] = seg[:n - i]
        i += ln
    return f0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=40)
    ap.add_argument("--first-seed", type=int, default=70000)
    a = ap.parse_args()
    P = port.Port()
    P.set_threads(os.cpu_count() or 1)
    worst = dict(sp=0.0, ap=0.0, y=0.0)
    for c in range(a.n):
        seed = a.first_seed + c
        rng = np.random.default_rng(seed)
        fs = int(rng.choice([8000, 16000, 22050, 24000, 44100, 48000]))
        fp = float(rng.choice([1.0, 5.0, 5.0, 10.0]))
        x = make_utterance(fs, float(rng.uniform(0.3, 1.5)), seed)
        nfr = w.get_samples(fs, len(x), fp)
        tpos = np.arange(nfr) * fp / 1000.0
        f0 = contour(rng, nfr)
        start = int(rng.integers(0, 10 ** 6))
        # CheapTrick
        P.rng_seek(start)
        sp_o = P.cheaptrick(x, fs, tpos, f0)
        w.rng_set_position(start)
        sp_g = w.CheapTrick(fs).compute(x, tpos, f0)
        assert w.rng_get_position() == P.rng_position(), "CheapTrick draw count"
        n = (sp_o.shape[1] - 1) * 2
        # D4C
        P.rng_seek(start)
        ap_o = P.d4c(x, fs, tpos, f0, n)
        w.rng_set_position(start)
        ap_g = w.D4C(fs).compute(x, tpos, f0, n)
        assert w.rng_get_position() == P.rng_position(), "D4C draw count"
        # Synthesis (the oracle's parameters on both sides)
        P.rng_seek(start)
        y_o = P.synthesis(f0, sp_o, ap_o, fs, fp)
        w.rng_set_position(start)
        y_g = w.Synthesis(fs, n, fp).compute(f0, sp_o, ap_o)
        assert w.rng_get_position() == P.rng_position(), "Synthesis draw count"
        e = dict(sp=float((np.abs(sp_g - sp_o) / sp_o).max()), ap=float(np.abs(ap_g - ap_o).max()), y=float(np.abs(y_g - y_o).max()))
        bad = [k for k in e if not np.isfinite(e[k])] + ([] if np.isfinite(sp_g).all() and np.isfinite(ap_g).all() and np.isfinite(y_g).all() else ["nonfinite"])
        for k in worst:
            worst[k] = max(worst[k], e[k]) if np.isfinite(e[k]) else float("inf")
        if bad or e["sp"] > 1e-7 or e["ap"] > 1e-7 or e["y"] > 1e-8:
            print("seed", seed, "fs", fs, "hop", fp, "frames", nfr, {k: "%.2e" % v for k, v in e.items()}, bad)
    P.rng_reset()
    print("cases", a.n, "worst", {k: "%.2e" % v for k, v in worst.items()})


if __name__ == "__main__":
    main()
