"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/config2_16k_10s.npz from the REAL reference (oracle/_ref/libworld_ref.so):
the full pipeline Harvest -> CheapTrick -> D4C -> Synthesis (demo order, reference test/test.cpp:288-384, Harvest defaults) on
the eight distinct utterances of BASELINE config 2 as bench.py builds it (stage_config2: make_utterance(16000, 10.0, 2000 + u)),
i.e. a 16 kHz utterance at the configuration's full size through every stage.  Run in the build container only:

    make -C oracle ref && python oracle/gen_golden_config2.py

Same layout as headline_48k_10s.npz (oracle/gen_golden_headline.py): only data travels, the utterances are regenerated from
their seeds (a checksum of the samples is stored)."""
import hashlib
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
from oracle import ref  # noqa: E402
from oracle.gen_golden_headline import windows, BLOCK, WIN  # noqa: E402
from world_class_amd.synth import make_utterance  # noqa: E402

FS, SECONDS = 16000, 10.0


def main():
    out = {}
    for u in range(8):
        seed = 2000 + u
        stride, nwin = (50, 16) if u < 2 else (250, 4)
        x = make_utterance(FS, SECONDS, seed)
        r = ref.run_fresh("pipeline", x, FS, harvest_floor=71.0)
        k = "u%d/" % u
        out[k + "meta"] = np.array([FS, SECONDS, seed, stride, BLOCK, WIN], dtype=np.float64)
        out[k + "x_sha256"] = np.frombuffer(hashlib.sha256(x.tobytes()).digest(), dtype=np.uint8)
        out[k + "f0"] = r["f0"]
        out[k + "sp_rowsum"] = r["sp"].sum(1)
        out[k + "ap_rowsum"] = r["ap"].sum(1)
        out[k + "sp_rows"] = r["sp"][::stride]
        out[k + "ap_rows"] = r["ap"][::stride]
        y = r["y"]
        nb = len(y) // BLOCK
        out[k + "y_len"] = np.array([len(y)])
        out[k + "y_blocksum"] = y[:nb * BLOCK].reshape(nb, BLOCK).sum(1)
        out[k + "y_win_start"] = np.array(windows(len(y), nwin))
        out[k + "y_win"] = np.stack([y[s:s + WIN] for s in windows(len(y), nwin)])
        print(k, "frames", len(r["f0"]), "voiced", int((r["f0"] > 0).sum()), "y", len(y))
    path = os.path.join(_ROOT, "tests", "golden", "config2_16k_10s.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
