"""Harvest's raw F0 candidates of single (band, 1 ms frame) pairs in 80-bit arithmetic -- the band-pass as a direct sum of the
filter, the four zero-crossing passes and the interpolation of reference src/harvest.cpp:1098-1255 restated in numpy long double
-- beside the real reference's own values (oracle/_ref, CPU only; development aid, round 5).  Where a stretch of the signal lies
200 dB and more below the utterance's maximum the reference's FFT convolution leaks 1e-16 of the loud part into it, coherently:
its raw candidates there sit 1e-5 relative off what its own algorithm gives in exact arithmetic, the kernels' do not (DESIGN.md
section 7 (v)).
    [ZOO2=1] python tools/raw_exact.py fs seed seconds band frame [band frame ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref  # noqa: E402
from world_class_amd.synth import make_signal, make_signal2  # noqa: E402

LD = np.longdouble


def raw_exact(y, band, frame, f0_floor=71.0, fs_internal=8000.0):
    fb = 0.9 * f0_floor * 2.0 ** ((band + 1) / 40.0)
    hl = int(np.floor(2.0 * fs_internal / fb + 0.5))
    k = np.arange(-hl, hl + 1).astype(LD)
    pos = np.arange(2 * hl + 1).astype(LD) / (2 * hl)
    pi = LD(np.pi)
    nuttall = 0.355768 - 0.487396 * np.cos(2 * pi * pos) + 0.144232 * np.cos(4 * pi * pos) - 0.012604 * np.cos(6 * pi * pos)
    taps = nuttall * np.cos(2 * pi * LD(fb) * k / LD(fs_internal))
    n = len(y)
    ypad = np.concatenate([np.zeros(hl + 2, dtype=LD), y.astype(LD), np.zeros(hl + 2, dtype=LD)])
    out = np.zeros(n, dtype=LD)  # out[i] = sum_q tap[q] y[i + 1 - hl + q]: the reference rotates by hl + 1 (src/harvest.cpp:1299-1304)
    for q in range(2 * hl + 1):
        out += taps[q] * ypad[3 + q:3 + q + n]

    def intervals(s):
        i = np.nonzero((s[:-1] > 0) & (s[1:] <= 0))[0]
        fine = (i + 1) - s[i] / (s[i + 1] - s[i])
        return (fine[:-1] + fine[1:]) / 2 / LD(fs_internal), LD(fs_internal) / (fine[1:] - fine[:-1])

    d = np.diff(out)
    t = LD(frame) / 1000
    vals = []
    for s in (out, -out, d, -d):
        loc, itv = intervals(s)
        j = min(max(int(np.searchsorted(loc, t, side="right")), 1), len(loc) - 1)
        vals.append(itv[j - 1] + (t - loc[j - 1]) / (loc[j] - loc[j - 1]) * (itv[j] - itv[j - 1]))
    return float(sum(vals) / 4), fb


if __name__ == "__main__":
    fs, seed, sec = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    x = (make_signal2 if os.environ.get("ZOO2") else make_signal)(fs, sec, seed)
    taps = ref.harvest_taps(x, fs)
    rest = [int(v) for v in sys.argv[4:]]
    for band, frame in zip(rest[0::2], rest[1::2]):
        v, fb = raw_exact(taps["y"], band, frame)
        gated = v if 0.9 * fb <= v <= 1.1 * fb and 71.0 <= v <= 800.0 else 0.0
        print("band %3d (%.3f Hz, gate %.4f .. %.4f) frame %5d: exact %.9f -> %.9f after the gate; the reference's %.9f" %
              (band, fb, 0.9 * fb, 1.1 * fb, frame, v, gated, taps["raw"][band, frame]))
