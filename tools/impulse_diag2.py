"""Which stage of Harvest first tells the sliding-DFT band-pass from the direct FIR one on an impulse train? (development aid, round 5)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import world_class_amd as w  # noqa: E402
from world_class_amd.synth import make_signal  # noqa: E402

fs = 16000
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1440023
x = make_signal(fs, 3.0, seed)
out = {}
for mode in ("sdft", "fir"):
    os.environ["WC_HARVEST_TIES"] = "ignore"
    if mode == "fir":
        os.environ["WC_HARVEST_BANDPASS"] = "fir"
    else:
        os.environ.pop("WC_HARVEST_BANDPASS", None)
    h = w.Harvest(fs)
    _, f = h.compute(x)
    d = {"f0": f}
    for name in ("y", "raw", "cand0", "cand1", "score1", "cand", "score", "base", "s1", "s2", "s3", "fixed", "f0_1ms"):
        d[name] = h.debug_fetch(name)
    out[mode] = d
a, b = out["sdft"], out["fir"]
L1 = len(a["f0_1ms"])
for name in ("y", "raw", "cand0", "cand1", "score1", "cand", "score", "base", "s1", "s2", "s3", "fixed", "f0_1ms", "f0"):
    u, v = a[name], b[name]
    nz = int(((u == 0) != (v == 0)).sum())
    both = (u != 0) & (v != 0)
    dev = float(np.abs(u - v)[both].max()) if both.any() else 0.0
    rel = float((np.abs(u - v)[both] / np.abs(v[both])).max()) if both.any() else 0.0
    print("%-8s size %8d  zero/non-zero mismatches %6d  max abs dev %.3e  max rel dev %.3e" % (name, u.size, nz, dev, rel))
# the frames whose final voicing differs, and what their rows look like
fl = np.nonzero((a["f0_1ms"] == 0) != (b["f0_1ms"] == 0))[0]
print("1 ms frames with different voicing:", fl[:20], len(fl))
nb = a["raw"].size // L1
nc = a["cand1"].size // L1
for i in fl[:3]:
    for name, wd in (("cand0", a["cand0"].size // L1), ("cand1", nc), ("score1", nc), ("cand", a["cand"].size // L1), ("score", a["score"].size // L1)):
        ra, rb = a[name].reshape(L1, wd)[i], b[name].reshape(L1, wd)[i]
        k = np.nonzero((ra != 0) | (rb != 0))[0]
        print(" frame", i, name, "sdft", np.array2string(ra[k], precision=6, max_line_width=200), "| fir", np.array2string(rb[k], precision=6, max_line_width=200))
    print(" frame", i, "base", a["base"][i], b["base"][i], "fixed", a["fixed"][i], b["fixed"][i])
ra, rb = a["raw"].reshape(nb, L1), b["raw"].reshape(nb, L1)
print("NaN in raw: sdft", int(np.isnan(ra).sum()), "fir", int(np.isnan(rb).sum()))
mm = np.argwhere((ra == 0) != (rb == 0))
print("bands of the mismatches:", np.unique(mm[:, 0]), "frames", mm[:, 1].min(), "..", mm[:, 1].max())
for bnd in np.unique(mm[:, 0])[:6]:
    fr = mm[mm[:, 0] == bnd][:, 1]
    print(" band", bnd, "frames", fr[:8], "... sdft", ra[bnd, fr[:4]], "fir", rb[bnd, fr[:4]])
bb = np.unique(mm[mm[:, 1] == 2043][:, 0]) if (mm[:, 1] == 2043).any() else []
for bnd in bb:
    print(" frame 2035..2060 band", bnd, "sdft", np.array2string(ra[bnd, 2035:2060], precision=3, max_line_width=250))
    print("                         fir ", np.array2string(rb[bnd, 2035:2060], precision=3, max_line_width=250))
