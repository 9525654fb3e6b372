"""TEST INFRASTRUCTURE ONLY (build container).  Runs the reference's codec (src/codec.cpp compiled from where it lies into
oracle/_ref/libworld_ref_tools.so) on rows of the committed pipeline goldens and writes tests/golden/io/codec_golden.npz.

    python oracle/gen_golden_codec.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
_dp = C.POINTER(C.c_double)


def rows(mat):
    arr = (_dp * mat.shape[0])()
    for i in range(mat.shape[0]):
        arr[i] = mat[i].ctypes.data_as(_dp)
    return arr


def main():
    L = C.CDLL(os.path.join(HERE, "_ref", "libworld_ref_tools.so"))
    R = C.POINTER(_dp)
    L.CodeSpectralEnvelope.argtypes = [R, C.c_int, C.c_int, C.c_int, C.c_int, R]
    L.DecodeSpectralEnvelope.argtypes = [R, C.c_int, C.c_int, C.c_int, C.c_int, R]
    L.CodeAperiodicity.argtypes = [R, C.c_int, C.c_int, C.c_int, R]
    L.DecodeAperiodicity.argtypes = [R, C.c_int, C.c_int, C.c_int, R]
    G = np.load(os.path.join(ROOT, "tests", "golden", "world_golden.npz"))
    g = {}
    for case, fs, fft in (("c1_16k_2s_floor71", 16000, 1024), ("m24k_1s_1ms", 24000, 1024), ("m48k_1s", 48000, 2048)):
        sp = np.ascontiguousarray(G[case + "/sp_rows"][:16])
        ap = np.ascontiguousarray(G[case + "/ap_rows"][:16])
        n = sp.shape[0]
        g[f"{case}/fs"], g[f"{case}/fft"] = fs, fft
        g[f"{case}/sp"], g[f"{case}/ap"] = sp, ap
        g[f"{case}/n_ap"] = L.GetNumberOfAperiodicities(fs)
        for nd in (25, 60, fft // 4 + 1):
            coded = np.full((n, nd), np.nan)
            L.CodeSpectralEnvelope(rows(sp), n, fs, fft, nd, rows(coded))
            dec = np.full((n, fft // 2 + 1), np.nan)
            L.DecodeSpectralEnvelope(rows(coded), n, fs, fft, nd, rows(dec))
            g[f"{case}/sp_coded_{nd}"], g[f"{case}/sp_decoded_{nd}"] = coded, dec
        n_ap = int(g[f"{case}/n_ap"])
        cap = np.full((n, n_ap), np.nan)
        L.CodeAperiodicity(rows(ap), n, fs, fft, rows(cap))
        dap = np.full((n, fft // 2 + 1), np.nan)
        L.DecodeAperiodicity(rows(cap), n, fs, fft, rows(dap))
        g[f"{case}/ap_coded"], g[f"{case}/ap_decoded"] = cap, dap
    g["n_ap_table_fs"] = np.array([8000, 12000, 16000, 22050, 24000, 32000, 44100, 48000, 96000])
    g["n_ap_table"] = np.array([L.GetNumberOfAperiodicities(int(f)) for f in g["n_ap_table_fs"]])
    out = os.path.join(ROOT, "tests", "golden", "io", "codec_golden.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, f"{os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
