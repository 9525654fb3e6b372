"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/harvest_long.npz from the REAL reference (oracle/_ref/libworld_ref.so):
F0 contours of whole 10 s utterances, among them one whose contour depends on how std::sort orders voiced sections that
start on the same frame (reference src/harvest.cpp:508-517; seed 9033 has two sections extended back to frame 0 and more
than 16 sections, so libstdc++'s introsort decides).  Run in the build container only:

    make -C oracle ref && python oracle/gen_golden_harvest.py

Only data travels: the inputs are regenerated from their seeds by world_class_amd.synth.make_utterance (a checksum of
the samples is stored), the expected F0 comes from the reference.
"""
import hashlib
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
from oracle import ref  # noqa: E402
from world_class_amd.synth import make_utterance  # noqa: E402

# name, fs, seconds, seed, f0_floor
CASES = [
    ("tie_48k_10s_9033", 48000, 10.0, 9033, 71.0),
    ("plain_48k_10s_9001", 48000, 10.0, 9001, 71.0),
    ("plain_16k_10s_12003_floor40", 16000, 10.0, 12003, 40.0),
]


def main():
    out = {}
    for name, fs, sec, seed, floor in CASES:
        x = make_utterance(fs, sec, seed)
        tpos, f0 = ref.run_fresh("harvest", x, fs, f0_floor=floor)
        out[name + "/meta"] = np.array([fs, sec, seed, floor], dtype=np.float64)
        out[name + "/x_sha256"] = np.frombuffer(hashlib.sha256(x.tobytes()).digest(), dtype=np.uint8)
        out[name + "/f0"] = f0
        print(name, "frames", len(f0), "voiced", int((f0 > 0).sum()))
    path = os.path.join(_ROOT, "tests", "golden", "harvest_long.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
