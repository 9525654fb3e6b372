"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/headline_48k_10s.npz from the REAL reference (oracle/_ref/libworld_ref.so):
the full pipeline Harvest -> CheapTrick -> D4C -> Synthesis (demo order, reference test/test.cpp:288-384, Harvest defaults) on
all eight distinct utterances of the benchmark's own workload (bench.py: make_utterance(48000, 10.0, 3000 + u)), i.e. the
headline configuration at its full utterance size.  Run in the build container only:

    make -C oracle ref && python oracle/gen_golden_headline.py

Only data travels: the utterances are regenerated from their seeds (a checksum of the samples is stored).  Stored per
utterance: the whole F0 contour; of the spectrogram and the aperiodicity every row's sum and every 50th row in full (every
250th for utterances 2 .. 7, to keep the fixture small); of the resynthesised waveform the sums of 480-sample blocks and sixteen
(four) windows of 4096 samples in full."""
import hashlib
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
from oracle import ref  # noqa: E402
from world_class_amd.synth import make_utterance  # noqa: E402

FS, SECONDS, STRIDE, BLOCK, WIN, NWIN = 48000, 10.0, 50, 480, 4096, 16


def windows(n, nwin=NWIN):
    return [int(k * (n - WIN) / (nwin - 1)) for k in range(nwin)]


def main():
    out = {}
    for u in range(8):
        seed = 3000 + u
        stride, nwin = (STRIDE, NWIN) if u < 2 else (250, 4)
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
    path = os.path.join(_ROOT, "tests", "golden", "headline_48k_10s.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
