"""GPU Harvest against the CPU restatement AND the real reference (oracle/_ref) on single zoo signals (development aid, round 5):
    python tools/zoo_diag.py fs seed seconds frame_period [...more quadruples]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import world_class_amd as w  # noqa: E402
from oracle import port, ref  # noqa: E402
from world_class_amd.synth import SIGNAL_KINDS, SIGNAL_KINDS2, make_signal, make_signal2  # noqa: E402

if os.environ.get("ZOO2"):  # the second set of kinds (tests/parity_sweep.py --zoo2)
    SIGNAL_KINDS, make_signal = SIGNAL_KINDS2, make_signal2

P = port.Port()
args = sys.argv[1:]


def cmp(a, b):
    v = (a > 0) & (b > 0)
    return int(((a == 0) != (b == 0)).sum()), (float(np.abs(a - b)[v].max()) if v.any() else 0.0)


for k in range(0, len(args), 4):
    fs, seed, sec, fp = int(args[k]), int(args[k + 1]), float(args[k + 2]), float(args[k + 3])
    x = make_signal(fs, sec, seed)
    f_port = P.harvest(x, fs, frame_period=fp)[1]
    f_ref = ref.run_fresh("harvest", x, fs, frame_period=fp)[1]
    out = {}
    for mode in ("sdft", "fir", "acting_on_ties", "no_quiet_chunks"):
        os.environ["WC_HARVEST_TIES"] = "ignore"
        os.environ.pop("WC_HARVEST_QUIET", None)
        if mode == "acting_on_ties":
            os.environ.pop("WC_HARVEST_TIES", None)
        if mode == "no_quiet_chunks":
            os.environ["WC_HARVEST_QUIET"] = "sliding"
        if mode == "fir":
            os.environ["WC_HARVEST_BANDPASS"] = "fir"
        else:
            os.environ.pop("WC_HARVEST_BANDPASS", None)
        h = w.Harvest(fs, frame_period=fp)
        out[mode] = h.compute(x)[1]
        if mode == "sdft":
            taps = ref.harvest_taps(x, fs) if ref.taps_available() else None
            if taps is not None:
                L1 = len(taps["f0_1ms"])
                for name, key in (("y", "y"), ("raw", "raw"), ("base", "f0_base"), ("fixed", "f0_fixed"), ("f0_1ms", "f0_1ms")):
                    g = h.debug_fetch(name)
                    r = taps[key].ravel()
                    n = min(len(g), len(r))
                    g, r = g[:n], r[:n]
                    both = (g != 0) & (r != 0)
                    print("   gpu vs reference taps %-7s zero/non-zero mismatches %6d  max abs dev %.3e" %
                          (name, int(((g == 0) != (r == 0)).sum()), float(np.abs(g - r)[both].max()) if both.any() else 0.0))
    print(fs, seed, SIGNAL_KINDS[seed % len(SIGNAL_KINDS)], "hop", fp, "frames", len(f_ref))
    print("   port vs ref      flips %d max dev %.3e" % cmp(f_port, f_ref))
    print("   gpu(sdft) vs ref flips %d max dev %.3e" % cmp(out["sdft"], f_ref))
    print("   gpu(fir)  vs ref flips %d max dev %.3e" % cmp(out["fir"], f_ref))
    print("   gpu(sdft) vs port flips %d max dev %.3e" % cmp(out["sdft"], f_port))
    print("   gpu(sdft, acting on ties) vs ref flips %d max dev %.3e" % cmp(out["acting_on_ties"], f_ref))
    print("   gpu(sdft, quiet chunks left to the sliding sums) vs ref flips %d max dev %.3e" % cmp(out["no_quiet_chunks"], f_ref))
    d = np.abs(out["sdft"] - f_ref)
    idx = np.nonzero(d > 1e-3)[0]
    print("   frames with |gpu - ref| > 1e-3:", idx[:12], "..." if len(idx) > 12 else "", "gpu", out["sdft"][idx[:6]], "ref", f_ref[idx[:6]], flush=True)
