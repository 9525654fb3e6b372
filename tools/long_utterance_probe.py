"""Long utterances through the fused pipeline against the CPU oracle (development aid, round 5): minutes of signal in ONE utterance,
alone and beside short ones.   python tools/long_utterance_probe.py fs seconds [fs seconds ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import world_class_amd as w  # noqa: E402
from oracle import port  # noqa: E402
from parity_sweep import dev  # noqa: E402
from world_class_amd.synth import make_utterance  # noqa: E402

P = port.Port()
P.set_threads(os.cpu_count() or 1)
args = sys.argv[1:]
for k in range(0, len(args), 2):
    fs, sec = int(args[k]), float(args[k + 1])
    n_seg = max(1, int(sec / 6.0))
    x = np.concatenate([make_utterance(fs, 6.0, 7000 + i) for i in range(n_seg)])
    short = [make_utterance(fs, 0.7, 7100), make_utterance(fs, 2.0, 7101)]
    pipe = w.Pipeline(fs)
    t0 = time.perf_counter()
    res = pipe.run_batch([short[0], x, short[1]])
    t1 = time.perf_counter()
    for name, sig, r in (("short", short[0], res[0]), ("long", x, res[1]), ("short2", short[1], res[2])):
        o = P.pipeline(sig, fs)
        fl = int(((r["f0"] == 0) != (o["f0"] == 0)).sum())
        same = (r["f0"] == 0) == (o["f0"] == 0)
        print("fs %d  %-6s %7.1f s  %8d frames  V/UV flips %d  f0 %.2e  sp %.2e  ap %.2e  y %.2e" %
              (fs, name, len(sig) / fs, len(r["f0"]), fl, dev(r["f0"][same], o["f0"][same]), dev(r["sp"], o["sp"], rel=True), dev(r["ap"], o["ap"]), dev(r["y"], o["y"])), flush=True)
        if name == "long":
            # the waveform as a stage: the checker's Synthesis on the parameters the kernels produced, from the same place in the noise
            # stream (an F0 that differs by 1e-9 Hz moves every later pulse: end to end the deviation grows with the length)
            P.rng_seek(o["syn_start"])
            y2 = P.synthesis(r["f0"], r["sp"], r["ap"], fs, 5.0)
            P.rng_reset()
            print("          Synthesis as a stage on the kernels' own parameters: y %.2e" % dev(r["y"], y2), flush=True)
    print("   (the batch took %.1f ms on the device side of the call)" % ((t1 - t0) * 1e3))
