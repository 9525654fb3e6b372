"""Python mirror of include/world_class_codec.h: the reference's feature codec (include/codec.hpp) on the GPU."""
import ctypes as C

import numpy as np

from . import _check, _ptr, lib

_dp = C.POINTER(C.c_double)
_rows_t = C.POINTER(_dp)

CODEC_SIGNATURES = {
    "GetNumberOfAperiodicities": (C.c_int, [C.c_int]),
    "CodeAperiodicity": (None, [_rows_t, C.c_int, C.c_int, C.c_int, _rows_t]),
    "DecodeAperiodicity": (None, [_rows_t, C.c_int, C.c_int, C.c_int, _rows_t]),
    "CodeSpectralEnvelope": (None, [_rows_t, C.c_int, C.c_int, C.c_int, C.c_int, _rows_t]),
    "DecodeSpectralEnvelope": (None, [_rows_t, C.c_int, C.c_int, C.c_int, C.c_int, _rows_t]),
    "wc_code_spectral_envelope_device": (C.c_int, [C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]),
    "wc_decode_spectral_envelope_device": (C.c_int, [C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]),
    "wc_code_aperiodicity_device": (C.c_int, [C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]),
    "wc_decode_aperiodicity_device": (C.c_int, [C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]),
}

_bound = False


def _L():
    global _bound
    L = lib()
    if not _bound:
        for name, (res, args) in CODEC_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _bound = True
    return L


def _rows(mat):
    from . import _rows as rows  # (a table of row addresses built by numpy, see world_class_amd/__init__.py)
    return rows(mat)


def number_of_aperiodicities(fs):
    return _L().GetNumberOfAperiodicities(int(fs))


def code_spectral_envelope(sp, fs, fft_size, number_of_dimensions):
    sp = np.ascontiguousarray(sp, dtype=np.float64)
    out = np.full((sp.shape[0], number_of_dimensions), np.nan)
    _L().CodeSpectralEnvelope(_rows(sp), sp.shape[0], int(fs), int(fft_size), int(number_of_dimensions), _rows(out))
    return out


def decode_spectral_envelope(coded, fs, fft_size):
    coded = np.ascontiguousarray(coded, dtype=np.float64)
    out = np.full((coded.shape[0], fft_size // 2 + 1), np.nan)
    _L().DecodeSpectralEnvelope(_rows(coded), coded.shape[0], int(fs), int(fft_size), coded.shape[1], _rows(out))
    return out


def code_aperiodicity(ap, fs, fft_size):
    ap = np.ascontiguousarray(ap, dtype=np.float64)
    out = np.full((ap.shape[0], number_of_aperiodicities(fs)), np.nan)
    _L().CodeAperiodicity(_rows(ap), ap.shape[0], int(fs), int(fft_size), _rows(out))
    return out


def decode_aperiodicity(coded, fs, fft_size):
    coded = np.ascontiguousarray(coded, dtype=np.float64)
    out = np.full((coded.shape[0], fft_size // 2 + 1), np.nan)
    _L().DecodeAperiodicity(_rows(coded), coded.shape[0], int(fs), int(fft_size), _rows(out))
    return out


def code_spectral_envelope_device(fs, fft_size, n_frames, number_of_dimensions, d_sp, d_coded):
    _check(_L().wc_code_spectral_envelope_device(int(fs), int(fft_size), int(n_frames), int(number_of_dimensions), _ptr(d_sp), _ptr(d_coded)))


def decode_spectral_envelope_device(fs, fft_size, n_frames, number_of_dimensions, d_coded, d_sp):
    _check(_L().wc_decode_spectral_envelope_device(int(fs), int(fft_size), int(n_frames), int(number_of_dimensions), _ptr(d_coded), _ptr(d_sp)))


def code_aperiodicity_device(fs, fft_size, n_frames, d_ap, d_coded):
    _check(_L().wc_code_aperiodicity_device(int(fs), int(fft_size), int(n_frames), _ptr(d_ap), _ptr(d_coded)))


def decode_aperiodicity_device(fs, fft_size, n_frames, d_coded, d_ap):
    _check(_L().wc_decode_aperiodicity_device(int(fs), int(fft_size), int(n_frames), _ptr(d_coded), _ptr(d_ap)))
