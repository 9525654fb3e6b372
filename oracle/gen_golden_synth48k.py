"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/synth_only_48k_10s.npz from the REAL reference (oracle/_ref/libworld_ref.so):
Synthesis alone (reference src/synthesis.cpp:77-177) from GIVEN {f0, spectrogram, aperiodicity} at BASELINE config 4's size --
sixteen utterances of 48 kHz x 10 s (2001 frames, 2048-point FFT), each in a process of its own (noise stream at its seed
state).  Sixteen, so that the product's Synthesis stage call takes its two-halves path (n_utt >= 16) against the reference.
Run in the build container only:

    make -C oracle ref && python oracle/gen_golden_synth48k.py

Only data travels: the parameters are regenerated from their seeds by oracle/gen_golden.synth_params (seeded numpy arithmetic);
of every waveform the sums of 480-sample blocks and eight windows of 2048 samples are stored in full."""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
from oracle import ref  # noqa: E402
from oracle.gen_golden import synth_params  # noqa: E402

FS, FFT, FRAMES, FP, N_UTT, FIRST_SEED, BLOCK, WIN, NWIN = 48000, 2048, 2001, 5.0, 16, 7100, 480, 2048, 8


def windows(n):
    return [int(k * (n - WIN) / (NWIN - 1)) for k in range(NWIN)]


def main():
    out = {"meta": np.array([FS, FFT, FRAMES, FP, N_UTT, FIRST_SEED, BLOCK, WIN], dtype=np.float64)}
    for u in range(N_UTT):
        f0, sp, ap = synth_params(FS, FFT, FRAMES, FIRST_SEED + u)
        y = ref.run_fresh("synthesis", f0, sp, ap, FS, FP)
        k = "u%d/" % u
        nb = len(y) // BLOCK
        out[k + "y_len"] = np.array([len(y)])
        out[k + "y_blocksum"] = y[:nb * BLOCK].reshape(nb, BLOCK).sum(1)
        out[k + "y_win_start"] = np.array(windows(len(y)))
        out[k + "y_win"] = np.stack([y[s:s + WIN] for s in windows(len(y))])
        out[k + "param_sums"] = np.array([f0.sum(), sp.sum(), ap.sum()])  # (the regenerated parameters are checked against these)
        print(k, "voiced", int((f0 > 0).sum()), "y", len(y), "max", float(np.abs(y).max()))
    path = os.path.join(_ROOT, "tests", "golden", "synth_only_48k_10s.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
