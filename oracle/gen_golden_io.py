"""TEST INFRASTRUCTURE ONLY (runs in the build container, where /root/reference exists).

Writes tests/golden/io/: small files produced BY THE REFERENCE'S OWN tools (tools/audioio.cpp, tools/parameterio.cpp,
compiled from where they lie into oracle/_ref/libworld_ref_tools.so) and io_golden.npz with what the reference reads
back from them and from hand-made WAV variants, plus the outputs of the demo's ParameterModification.

    python oracle/gen_golden_io.py
"""
import ctypes as C
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "io")
_dp = C.POINTER(C.c_double)


def rows(mat):
    arr = (_dp * mat.shape[0])()
    for i in range(mat.shape[0]):
        arr[i] = mat[i].ctypes.data_as(_dp)
    return arr


def main():
    os.makedirs(OUT, exist_ok=True)
    L = C.CDLL(os.path.join(HERE, "_ref", "libworld_ref_tools.so"))
    L.GetHeaderInformation.restype = C.c_double
    L.GetHeaderInformation.argtypes = [C.c_char_p, C.c_char_p]
    L.WriteF0.argtypes = [C.c_char_p, C.c_int, C.c_double, _dp, _dp, C.c_int]
    for f in (L.WriteSpectralEnvelope, L.WriteAperiodicity):
        f.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(_dp)]
    L.ref_parameter_modification.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.POINTER(_dp), C.c_int, C.c_double, C.c_double]

    def P(name):
        return os.path.join(OUT, name).encode()

    rng = np.random.default_rng(20240229)
    g = {}

    # ---- WAV written by the reference: in-range samples, clipping on both sides, exact +-1, tiny values ----
    fs = 16000
    x = 0.6 * np.sin(2 * np.pi * 220.0 * np.arange(3000) / fs) + rng.normal(0, 0.05, 3000)
    x[:8] = [1.0, -1.0, 1.5, -1.5, 0.99999, -0.99999, 1e-6, -1e-6]
    x[8:12] = [32767.4 / 32767, -32767.9 / 32767, 0.5 / 32767, -0.5 / 32767]
    x = np.ascontiguousarray(x)
    L.wavwrite(x.ctypes.data_as(_dp), len(x), fs, 16, P("ref_written_16k.wav"))
    g["wav_x"] = x

    def ref_read(name):
        n = L.GetAudioLength(P(name))
        out = np.full(max(n, 1), np.nan)
        fs_, nb = C.c_int(-1), C.c_int(-1)
        L.wavread(P(name), C.byref(fs_), C.byref(nb), out.ctypes.data_as(_dp))
        return n, fs_.value, nb.value, out[:max(n, 0)]

    n, fs_r, nb, y = ref_read("ref_written_16k.wav")
    g["wav_read_len"], g["wav_read_fs"], g["wav_read_nbit"], g["wav_read_x"] = n, fs_r, nb, y

    # ---- hand-made variants: extra chunk before "data", 8 / 24 / 32-bit samples, broken headers ----
    def make_wav(name, nbit, samples, extra=b"", fmt_size=16, channels=1, fmt_id=1, riff=b"RIFF"):
        qb = nbit // 8
        data = b"".join(int(int(s) & ((1 << (8 * qb)) - 1)).to_bytes(qb, "little") for s in samples)
        fmt = struct.pack("<HHIIHH", fmt_id, channels, 22050, 22050 * qb, qb, nbit)
        body = b"WAVE" + b"fmt " + struct.pack("<I", fmt_size) + fmt + extra + b"data" + struct.pack("<I", len(data)) + data
        with open(os.path.join(OUT, name), "wb") as f:
            f.write(riff + struct.pack("<I", len(body)) + body)

    s16 = rng.integers(-32768, 32768, 64)
    s16[:4] = [-32768, 32767, 0, -1]
    make_wav("list_chunk_16bit.wav", 16, s16, extra=b"LIST" + struct.pack("<I", 12) + b"INFOdat\x00ISFT")  # a false "dat" start
    s24 = rng.integers(-(1 << 23), 1 << 23, 48)
    s24[:4] = [-(1 << 23), (1 << 23) - 1, 0, -1]
    make_wav("pcm_24bit.wav", 24, s24)
    s8 = rng.integers(-128, 128, 40)
    make_wav("pcm_8bit.wav", 8, s8)
    s32 = rng.integers(-(1 << 31), 1 << 31, 24)
    make_wav("pcm_32bit.wav", 32, s32)
    make_wav("bad_stereo.wav", 16, s16[:8], channels=2)
    make_wav("bad_fmt_size.wav", 16, s16[:8], fmt_size=18)
    make_wav("bad_riff.wav", 16, s16[:8], riff=b"RIFX")
    # sample sizes the reference must never be shown (it indexes a 4-byte scratch with nbit / 8 - 1): fixtures for OUR reader only
    make_wav("bad_nbit.wav", 40, s16[:8])
    make_wav("bad_nbit12.wav", 12, s16[:8])
    for name in ("list_chunk_16bit", "pcm_24bit", "pcm_8bit", "pcm_32bit", "bad_stereo", "bad_fmt_size", "bad_riff"):
        n, fs_r, nb, y = ref_read(name + ".wav")
        g[name + "_len"], g[name + "_fs"], g[name + "_nbit"], g[name + "_x"] = n, fs_r, nb, y
    g["missing_len"] = L.GetAudioLength(P("does_not_exist.wav"))

    # ---- parameter files written by the reference ----
    nf, fft = 9, 64
    tpos = np.arange(nf) * 5.0 / 1000.0
    f0 = np.where(rng.random(nf) > 0.3, rng.uniform(80, 400, nf), 0.0)
    sp = np.ascontiguousarray(rng.uniform(1e-8, 2.0, (nf, fft // 2 + 1)))
    ap = np.ascontiguousarray(rng.uniform(0.001, 0.999, (nf, fft // 2 + 1)))
    L.WriteF0(P("ref.f0"), nf, 5.0, tpos.ctypes.data_as(_dp), f0.ctypes.data_as(_dp), 0)
    L.WriteF0(P("ref_f0.txt"), nf, 5.0, tpos.ctypes.data_as(_dp), f0.ctypes.data_as(_dp), 1)
    L.WriteSpectralEnvelope(P("ref.sp"), 16000, nf, 5.0, fft, 0, rows(sp))
    L.WriteAperiodicity(P("ref.ap"), 16000, nf, 5.0, fft, 0, rows(ap))
    L.WriteSpectralEnvelope(P("ref_nod20.sp"), 16000, nf, 5.0, fft, 20, rows(sp))  # coded dimensions: 20 per frame
    g.update(par_tpos=tpos, par_f0=f0, par_sp=sp, par_ap=ap)
    t2, f2 = np.full(nf, np.nan), np.full(nf, np.nan)
    assert L.ReadF0(P("ref.f0"), t2.ctypes.data_as(_dp), f2.ctypes.data_as(_dp)) == 1
    g["read_f0_tpos"], g["read_f0"] = t2, f2
    sp2 = np.full_like(sp, np.nan)
    assert L.ReadSpectralEnvelope(P("ref.sp"), rows(sp2)) == 1
    ap2 = np.full_like(ap, np.nan)
    assert L.ReadAperiodicity(P("ref.ap"), rows(ap2)) == 1
    sp20 = np.full((nf, 20), np.nan)
    assert L.ReadSpectralEnvelope(P("ref_nod20.sp"), rows(sp20)) == 1
    g["read_sp"], g["read_ap"], g["read_sp20"] = sp2, ap2, sp20
    hdr = {}
    for fname in ("ref.f0", "ref.sp", "ref_nod20.sp"):
        for tag in ("NOF ", "FP  ", "FFT ", "NOD ", "FS  "):
            hdr[f"{fname}|{tag}"] = L.GetHeaderInformation(P(fname), tag.encode())
    g["hdr_keys"] = np.array(sorted(hdr))
    g["hdr_vals"] = np.array([hdr[k] for k in sorted(hdr)])

    # ---- the demo's ParameterModification (fresh copies per call) ----
    fsm, fftm, nfm = 16000, 1024, 6
    f0m = np.where(rng.random(nfm) > 0.3, rng.uniform(80, 400, nfm), 0.0)
    spm = np.ascontiguousarray(np.exp(rng.normal(-8.0, 2.0, (nfm, fftm // 2 + 1))))
    g.update(mod_f0=f0m, mod_sp=spm, mod_fs=fsm, mod_fft=fftm)
    for tag, n_args, shift, ratio in (("scale_only", 1, 1.25, 0.0), ("up", 2, 1.0, 1.2), ("down", 2, 0.8, 0.8),
                                      ("down_small", 2, 1.0, 0.37)):
        f = f0m.copy()
        s = spm.copy()
        L.ref_parameter_modification(fsm, nfm, fftm, f.ctypes.data_as(_dp), rows(s), n_args, shift, ratio)
        g[f"mod_{tag}_f0"], g[f"mod_{tag}_sp"] = f, s
        g[f"mod_{tag}_args"] = np.array([n_args, shift, ratio])
    np.savez_compressed(os.path.join(OUT, "io_golden.npz"), **g)
    size = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("wrote", OUT, f"{size / 1024:.0f} KiB,", len(os.listdir(OUT)), "files")


if __name__ == "__main__":
    main()
