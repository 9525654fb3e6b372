"""Parity sweep beyond the committed goldens (development aid, run on a GPU box): N seeded 48 kHz utterances through the fused
pipeline against the CPU oracle; prints the worst deviations and any voiced/unvoiced disagreement.
    python tests/parity_sweep.py [--n 32] [--seconds 10] [--fs 48000] [--first-seed 9000] [--floor 71] [--frame-period 5] [--ragged] [--zoo]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import world_class_amd as w  # noqa: E402
from oracle import port  # noqa: E402  (a checker, which is why it lives under tests/)
from world_class_amd.synth import SIGNAL_KINDS as ZOO, SIGNAL_KINDS2 as ZOO2, make_signal as zoo_signal, make_signal2 as zoo2_signal, make_utterance  # noqa: E402


def sp_dev(a, b, f0, fs, f0_floor=None):
    """Largest deviation of a spectral envelope from the checker's, relative, where the algorithm itself resolves the value.
    LinearSmoothing returns (c[hi] - c[lo]) / width of a SEQUENTIAL cumulative sum c of the power spectrum (reference
    src/world_common.cpp:47-51, :82-116): every one of its additions rounds at ulp(c), so a bin whose smoothed power is within a
    few thousand ulp(sum) / width of nothing carries whichever way those roundings fell -- the real reference and its own CPU
    restatement disagree by 1.9e-3 on such a bin (tests/golden/ref_self_spread.json: "restatement_vs_reference", a frame of an
    undithered synthetic signal whose envelope falls 150 dB).  The deviation is therefore measured against
    |b| + 3 ulp(sum b fs / N) / (width 1e-7): at the stated 1e-7 this allows three roundings of the sum on top of the relative
    tolerance, which only bins ~120 dB and more below their frame's total ever notice."""
    a, b = np.asarray(a), np.asarray(b)
    n = (b.shape[1] - 1) * 2
    floor = 3.0 * fs / (n - 3.0) if f0_floor is None else f0_floor  # reference src/cheaptrick.cpp:102-105
    f0c = np.where(f0 <= floor, 500.0, f0)
    total = np.where(np.isfinite(b), b, 0.0).sum(axis=1) * fs / n
    quantum = 3.0 * np.spacing(np.maximum(total, 1e-300)) / (f0c * 2.0 / 3.0)
    fa, fb = np.isfinite(a), np.isfinite(b)
    if not np.array_equal(fa, fb):
        return float("inf")
    d = np.abs(a - b) / (np.abs(b) + quantum[:, None] / 1e-7)
    return float(d[fa].max()) if fa.any() else 0.0


NAN_TOLERANT = False  # --nan-tolerant: where the checker returns NaN (the reference's D4C on noise-free frames: 0 / 0) nothing is compared


def dev(a, b, rel=False):
    """largest deviation; NaN / inf have to sit in the same places with the same sign"""
    fa, fb = np.isfinite(a), np.isfinite(b)
    if NAN_TOLERANT:
        a, b = a[fb], b[fb]
        fa, fb = np.isfinite(a), np.isfinite(b)
    if not np.array_equal(fa, fb) or not np.array_equal(np.nan_to_num(a[~fa], nan=7.0), np.nan_to_num(b[~fb], nan=7.0)):
        return float("inf")
    if not fa.any():
        return 0.0
    d = np.abs(a[fa] - b[fa])
    if rel:
        d = d / np.maximum(np.abs(b[fa]), 1e-300)
    return float(d.max()) if d.size else 0.0


def tie_counts():
    """(utterances that went through a Harvest refinement, utterances of those that raised the tie flag) in this process so far"""
    import ctypes as C
    fn = w.lib().wc_harvest_tie_counts
    fn.restype, fn.argtypes = None, [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int]
    seen, flagged = C.c_ulonglong(0), C.c_ulonglong(0)
    fn(C.byref(seen), C.byref(flagged), 0)
    return int(seen.value), int(flagged.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--fs", type=int, default=48000)
    ap.add_argument("--first-seed", type=int, default=9000)
    ap.add_argument("--seed-step", type=int, default=1, help="10 with --zoo / --zoo2: one kind only")
    ap.add_argument("--floor", type=float, default=71.0)
    ap.add_argument("--zoo", action="store_true", help="signals of other kinds (noise, chirps, impulse trains, ...) instead of utterances")
    ap.add_argument("--zoo2", action="store_true", help="the second set of kinds (speech-like, clipped, level stairs, quantised tones, ...)")
    ap.add_argument("--frame-period", type=float, default=5.0)
    ap.add_argument("--dither", type=float, default=0.0, help="rms of white noise added to every signal (noise-free bands make "
                    "CheapTrick and D4C ill-conditioned in any implementation, the reference included)")
    ap.add_argument("--ragged", action="store_true", help="utterance i lasts seconds * (0.2 + 0.8 * ((i * 7) % 10) / 9)")
    ap.add_argument("--checker", choices=("port", "ref"), default="port",
                    help="port: the CPU restatement (oracle/port.py); ref: the real reference (oracle/_ref, a fresh process per signal; "
                         "Harvest only where its Synthesis crashes) -- the restatement is pinned at 16 / 24 / 48 kHz only")
    ap.add_argument("--nan-tolerant", action="store_true", help="compare only where the checker's value is finite (and count the rest)")
    a = ap.parse_args()
    global NAN_TOLERANT
    NAN_TOLERANT = a.nan_tolerant
    dur = [a.seconds * (0.2 + 0.8 * ((i * 7) % 10) / 9) if a.ragged else a.seconds for i in range(a.n)]
    gen = zoo2_signal if a.zoo2 else zoo_signal if a.zoo else make_utterance
    kinds = ZOO2 if a.zoo2 else ZOO
    xs = [gen(a.fs, dur[i], (a.first_seed + i * a.seed_step)) for i in range(a.n)]
    if a.dither > 0:
        xs = [x + a.dither * np.random.default_rng((a.first_seed + i * a.seed_step) + 10 ** 6).normal(size=len(x)) for i, x in enumerate(xs)]
    res = w.Pipeline(a.fs, frame_period=a.frame_period, harvest_f0_floor=a.floor).run_batch(xs)
    ties = tie_counts()
    P = port.Port()
    P.set_threads(os.cpu_count() or 1)
    worst = dict(f0=0.0, sp=0.0, ap=0.0, y=0.0)
    flips = 0
    crashed = harvest_only = 0
    nan_ref = 0
    for i, (x, r) in enumerate(zip(xs, res)):
        if a.checker == "ref":
            from oracle import ref
            try:
                o = ref.run_fresh("pipeline", x, a.fs, harvest_floor=a.floor, frame_period=a.frame_period)
            except Exception:  # (the reference's Synthesis overflows its pulse arrays on some inputs: DESIGN_HISTORY.md section 7)
                try:
                    hv = ref.run_fresh("harvest", x, a.fs, f0_floor=a.floor, frame_period=a.frame_period)
                except Exception:  # (and its Harvest corrupts its heap on others)
                    crashed += 1
                    print("seed", (a.first_seed + i * a.seed_step), kinds[((a.first_seed + i * a.seed_step)) % len(kinds)] if (a.zoo or a.zoo2) else "", "the reference's Harvest crashed: nothing to compare with;",
                          "ours: voiced %d of %d frames, all finite: %s" % (int((r["f0"] > 0).sum()), len(r["f0"]), bool(all(np.isfinite(r[k]).all() for k in ("f0", "sp", "ap", "y")))))
                    continue
                o = dict(r, f0=hv[1])  # (the later stages then compare the kernels with themselves: only F0 is checked)
                harvest_only += 1
        else:
            o = P.pipeline(x, a.fs, harvest_floor=a.floor, frame_period=a.frame_period)
        nan_ref += int((~np.isfinite(o["ap"])).sum())
        fl = int(np.sum((r["f0"] == 0) != (o["f0"] == 0)))
        flips += fl
        same = (r["f0"] == 0) == (o["f0"] == 0)
        e = dict(f0=dev(r["f0"][same], o["f0"][same]), sp=dev(r["sp"], o["sp"], rel=True), ap=dev(r["ap"], o["ap"]),
                 y=dev(r["y"], o["y"]) / max(1.0, float(np.abs(x).max())))
        for k in worst:
            worst[k] = max(worst[k], float(e[k]))
        if fl or e["f0"] > 1e-6 or e["sp"] > 1e-7 or e["ap"] > 1e-7 or e["y"] > 1e-8:
            print("seed", (a.first_seed + i * a.seed_step), kinds[((a.first_seed + i * a.seed_step)) % len(kinds)] if (a.zoo or a.zoo2) else "", "V/UV flips", fl, {k: "%.2e" % v for k, v in e.items()})
    print("fs", a.fs, "floor", a.floor, "hop", a.frame_period, "utterances", a.n, "V/UV flips", flips, "worst", {k: "%.2e" % v for k, v in worst.items()},
          "; flagged for a tie (and run again with FIR sums): %d of %d" % (ties[1], ties[0]),
          ("; reference crashed on %d, its Synthesis on %d more (F0 only); non-finite aperiodicities from the checker: %d" % (crashed, harvest_only, nan_ref)) if a.checker == "ref" else "")


if __name__ == "__main__":
    main()
