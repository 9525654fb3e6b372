"""TEST INFRASTRUCTURE ONLY.  numpy restatement of the reference's feature codec (src/codec.cpp:12-325), pinned to the real
reference by tests/test_codec_oracle.py through tests/golden/io/codec_golden.npz (made by oracle/gen_golden_codec.py)."""
import numpy as np

from .port_io import interp1

M0, F0, FLOOR, CEIL = 1127.01048, 700.0, 40.0, 20000.0
INTERVAL, UPPER, SAFE = 3000.0, 15000.0, 0.000000000001


def freq_to_mel(f):
    return M0 * np.log(f / F0 + 1.0)


def mel_to_freq(m):
    return F0 * (np.exp(m / M0) - 1.0)


def number_of_aperiodicities(fs):
    return int(min(UPPER, fs / 2.0 - INTERVAL) / INTERVAL)


def code_spectral_envelope(sp, fs, fft_size, nd):
    """reference :269-296, :109-123, :47-61 (DCT through a real FFT with the reference's e^{+i} convention)"""
    md = fft_size // 2
    floor_mel, ceil_mel = freq_to_mel(FLOOR), freq_to_mel(min(fs / 2.0, CEIL))
    i = np.arange(md, dtype=np.float64)
    mel_axis = (ceil_mel - floor_mel) * i / md + floor_mel
    w = (2.0 * np.cos(i * np.pi / fft_size) / np.sqrt(fft_size)) + 1j * (2.0 * np.sin(i * np.pi / fft_size) / np.sqrt(fft_size))
    w[0] = w[0].real / np.sqrt(2.0) + 1j * w[0].imag
    # the reference leaves the last axis point unset and never reaches it (every mel point lies below the one before)
    freq_axis = freq_to_mel(np.arange(md + 1, dtype=np.float64) * fs / fft_size)
    out = np.empty((sp.shape[0], nd))
    half = np.arange(md // 2)
    for f in range(sp.shape[0]):
        mel = interp1(freq_axis, np.log(sp[f]), mel_axis)
        wave = np.empty(md)
        wave[half] = mel[2 * half]
        wave[half + md // 2] = mel[md - 2 * half - 1]
        spec = np.conj(np.fft.rfft(wave))  # reference r2c = conjugate of the textbook DFT
        out[f] = (spec.real[:nd] * w.real[:nd] - spec.imag[:nd] * w.imag[:nd]) / np.sqrt(md)
    return out


def decode_spectral_envelope(coded, fs, fft_size):
    """reference :298-325, :87-107, :63-85"""
    md, nd = fft_size // 2, coded.shape[1]
    floor_mel, ceil_mel = freq_to_mel(FLOOR), freq_to_mel(min(fs / 2.0, CEIL))
    i = np.arange(nd, dtype=np.float64)
    w = np.cos(i * np.pi / fft_size) * np.sqrt(fft_size) + 1j * (np.sin(i * np.pi / fft_size) * np.sqrt(fft_size))
    w[0] = w[0].real / np.sqrt(2.0) + 1j * w[0].imag
    mel_axis = np.empty(md + 2)
    mel_axis[1:md + 1] = mel_to_freq((ceil_mel - floor_mel) * np.arange(md, dtype=np.float64) / md + floor_mel)
    mel_axis[0], mel_axis[md + 1] = 0.0, fs / 2.0
    freq_axis = np.arange(md + 1, dtype=np.float64) * fs / fft_size
    out = np.empty((coded.shape[0], md + 1))
    half = np.arange(md // 2)
    for f in range(coded.shape[0]):
        inp = np.zeros(md, dtype=np.complex128)
        inp[:nd] = coded[f] * w.real * np.sqrt(md) - 1j * (coded[f] * w.imag * np.sqrt(md))
        o = np.fft.fft(inp)  # c2c BACKWARD of the reference = e^{-i}, unnormalised
        mel = np.empty(md + 2)
        mel[1 + 2 * half] = o.real[half]
        mel[2 + 2 * half] = o.real[md - half - 1]
        mel[0], mel[md + 1] = mel[1], mel[md]
        out[f] = np.exp(interp1(mel_axis, mel, freq_axis) / md)
    return out


def code_aperiodicity(ap, fs, fft_size):
    """reference :216-236 with interp1Q (src/world_matlabfunctions.cpp:220-241)"""
    n_ap = number_of_aperiodicities(fs)
    xi = INTERVAL * (np.arange(n_ap) + 1.0)
    dx = fs / fft_size
    base = ((xi - 0) / dx).astype(np.int64)
    frac = (xi - 0) / dx - base
    out = np.empty((ap.shape[0], n_ap))
    for f in range(ap.shape[0]):
        y = 20 * np.log10(ap[f])
        dy = np.append(np.diff(y), 0.0)
        out[f] = y[base] + dy[base] * frac
    return out


def decode_aperiodicity(coded, fs, fft_size):
    """reference :238-267"""
    n_ap = number_of_aperiodicities(fs)
    bins = fft_size // 2 + 1
    freq = fs / fft_size * np.arange(bins)
    coarse_axis = np.append(np.arange(n_ap + 1) * INTERVAL, fs / 2.0)
    out = np.full((coded.shape[0], bins), 1.0 - SAFE)
    for f in range(coded.shape[0]):
        tmp = 0.0
        for v in coded[f]:
            tmp += v
        if tmp / n_ap > -0.5:
            continue
        coarse = np.concatenate([[-60.0], coded[f], [-SAFE]])
        out[f] = np.power(10.0, interp1(coarse_axis, coarse, freq) / 20.0)
    return out
