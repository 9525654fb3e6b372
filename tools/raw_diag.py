"""Where do the GPU's raw F0 candidates part from the real reference's?  (development aid, round 5)
    [ZOO2=1] python tools/raw_diag.py fs seed seconds [fir]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fs, seed, sec = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
if len(sys.argv) > 4 and sys.argv[4] == "fir":
    os.environ["WC_HARVEST_BANDPASS"] = "fir"
os.environ["WC_HARVEST_TIES"] = "ignore"
import world_class_amd as w  # noqa: E402
from oracle import ref  # noqa: E402
from world_class_amd.synth import make_signal, make_signal2  # noqa: E402

x = (make_signal2 if os.environ.get("ZOO2") else make_signal)(fs, sec, seed)
taps = ref.harvest_taps(x, fs)
h = w.Harvest(fs, frame_period=5.0)
f0 = h.compute(x)[1]
nb, L1 = taps["raw"].shape
g = h.debug_fetch("raw").reshape(nb, L1)
r = taps["raw"]
bad = np.argwhere((np.abs(g - r) > 1e-6 * np.maximum(1.0, np.abs(r))))
print("raw candidates: %d of %d entries differ by more than 1e-6 relative" % (len(bad), g.size))
bands = sorted(set(int(b) for b, _ in bad))
for b in bands[:40]:
    fr = [int(i) for bb, i in bad if bb == b]
    print("  band %3d (%.1f Hz): frames %d..%d (%d)  e.g. frame %d gpu %.9f ref %.9f" % (b, 0.9 * 71.0 * 2 ** ((b + 1) / 40.0), fr[0], fr[-1], len(fr), fr[0], g[b, fr[0]], r[b, fr[0]]))
for name, key in (("cand1", "cand_refined"), ("cand", "cand"), ("base", "f0_base"), ("fixed", "f0_fixed"), ("f0_1ms", "f0_1ms")):
    gg = h.debug_fetch(name)
    rr = taps[key]
    if rr.ndim == 2 and name in ("cand1", "cand"):
        print("  ", name, "shapes", gg.shape, rr.shape, "n_cand", taps["n_cand"])
        continue
    d = np.abs(gg[:len(rr)] - rr)
    idx = np.nonzero(d > 1e-6)[0]
    print("  ", name, "frames off by more than 1e-6:", len(idx), idx[:10], "max", float(d.max()))
# the refined candidates and scores around the first frame whose base contour differs
gb, rb = h.debug_fetch("base"), taps["f0_base"]
off = np.nonzero(np.abs(gb[:len(rb)] - rb) > 1e-6)[0]
if len(off):
    c1 = h.debug_fetch("cand1").reshape(L1, -1)
    s1 = h.debug_fetch("score1").reshape(L1, -1)
    c2 = h.debug_fetch("cand").reshape(L1, -1)
    s2 = h.debug_fetch("score").reshape(L1, -1)
    for i in range(max(0, off[0] - 3), min(L1, off[0] + 4)):
        def pairs(c, s):
            k = np.nonzero(c > 0)[0]
            return sorted(((round(float(s[j]), 6), round(float(c[j]), 6)) for j in k), reverse=True)[:6]
        print("frame", i, "base gpu %.6f ref %.6f" % (gb[i], rb[i]))
        print("    refined  gpu", pairs(c1[i], s1[i]))
        print("    refined  ref", pairs(taps["cand_refined"][i], taps["score_refined"][i]))
        print("    reliable gpu", pairs(c2[i], s2[i]))
        print("    reliable ref", pairs(taps["cand"][i], taps["score"][i]))
    nz = np.argwhere((g == 0) != (r == 0))
    print("raw zero / non-zero mismatches:", [(int(b), int(i), float(g[b, i]), float(r[b, i])) for b, i in nz[:12]])
