"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/stage_options.npz from the REAL reference (oracle/_ref/libworld_ref.so):
CheapTrickOption (q1, f0_floor, fft_size; reference include/cheaptrick.hpp) and D4COption (threshold; include/d4c.hpp) away
from their defaults, each stage in a fresh process (noise stream at its seed), on one 16 kHz utterance with the F0 contour
of the reference's Harvest.  Every 16th row and all row sums are kept.  Run in the build container only:

    make -C oracle ref && python oracle/gen_golden_stage_options.py
"""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
from oracle import ref  # noqa: E402
from world_class_amd.synth import make_utterance  # noqa: E402

FS, SECONDS, SEED, STRIDE = 16000, 1.0, 4321, 16
CHEAPTRICK = [("q1_-0.09", dict(q1=-0.09)), ("floor40", dict(f0_floor=40.0)), ("fft2048", dict(fft_size=2048)),
              ("q1_-0.3_floor100", dict(q1=-0.3, f0_floor=100.0)), ("fft4096", dict(fft_size=4096))]
D4C = [("thr0", 0.0), ("thr0.5", 0.5), ("thr0.95", 0.95)]


def main():
    x = make_utterance(FS, SECONDS, SEED)
    tpos, f0 = ref.run_fresh("harvest", x, FS)
    out = {"f0": f0}
    for name, kw in CHEAPTRICK:
        sp = ref.run_fresh("cheaptrick", x, FS, tpos, f0, **kw)
        out["ct/" + name + "/rows"], out["ct/" + name + "/rowsum"] = sp[::STRIDE], sp.sum(axis=1)
        print("cheaptrick", kw, sp.shape)
    for name, thr in D4C:
        ap = ref.run_fresh("d4c", x, FS, tpos, f0, 1024, threshold=thr)
        out["d4c/" + name + "/rows"], out["d4c/" + name + "/rowsum"] = ap[::STRIDE], ap.sum(axis=1)
        print("d4c threshold", thr, "gated rows", int((ap[:, 5] < 0.999999).sum()))
    np.savez_compressed(os.path.join(_ROOT, "tests", "golden", "stage_options.npz"), **out)


if __name__ == "__main__":
    main()
