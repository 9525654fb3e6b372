"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/harvest_options.npz from the REAL reference (oracle/_ref/libworld_ref.so):
F0 contours for the HarvestOption fields beyond floor / ceil / frame period (reference include/harvest.hpp:16-24) --
target_fs, channels_in_octave and use_cos_table (the main window from an 8001-entry cosine table, src/harvest.cpp:152-170,
:779-787).  Run in the build container only:

    make -C oracle ref && python oracle/gen_golden_harvest_options.py

Only data travels: the utterances are regenerated from their seeds by world_class_amd.synth.make_utterance (tests check
them against tests/golden/world_golden.npz's pinned generator); expected F0 comes from the reference.
"""
import json
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
from oracle import ref  # noqa: E402
from world_class_amd.synth import make_utterance  # noqa: E402

# name, fs, seconds, seed, options
CASES = [
    ("target4k_16k", 16000, 1.5, 777, dict(target_fs=4000.0)),
    ("target16k_48k", 48000, 1.0, 777, dict(target_fs=16000.0)),
    ("target12k_table_48k", 48000, 1.0, 778, dict(target_fs=12000.0, use_cos_table=True)),
    ("octave60_16k", 16000, 1.5, 777, dict(channels_in_octave=60.0)),
    ("octave30_floor40_16k", 16000, 1.5, 779, dict(channels_in_octave=30.0, f0_floor=40.0)),
    ("table_16k", 16000, 1.5, 777, dict(use_cos_table=True)),
    ("table_1ms_24k", 24000, 1.0, 780, dict(use_cos_table=True, frame_period=1.0)),
]


def main():
    out, meta = {}, {}
    R = ref.Ref()
    for name, fs, sec, seed, opts in CASES:
        x = make_utterance(fs, sec, seed)
        tpos, f0 = R.harvest_opt(x, fs, **opts)
        out[name + "/f0"] = f0
        meta[name] = dict(fs=fs, seconds=sec, seed=seed, options=opts)
        print(name, "frames", len(f0), "voiced", int((f0 > 0).sum()))
    # the table changes the result: keep the exact-cosine contour of one case next to it
    x = make_utterance(16000, 1.5, 777)
    out["table_16k/f0_exact_cosines"] = R.harvest_opt(x, 16000)[1]
    print("table vs exact cosines: max", np.abs(out["table_16k/f0"] - out["table_16k/f0_exact_cosines"]).max(), "Hz")
    d = os.path.join(_ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(d, "harvest_options.npz"), **out)
    with open(os.path.join(d, "harvest_options.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
