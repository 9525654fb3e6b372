"""Python mirror of include/world_class_io.h: the reference's WAV / parameter-file functions (same names as reference
tools/audioio.hpp and tools/parameterio.hpp), device-side PCM conversion and parameter modification."""
import ctypes as C

import numpy as np

from . import WorldClassError, _c, _check, _ptr, lib

_dp = C.POINTER(C.c_double)
_i16p = C.POINTER(C.c_int16)
_rows_t = C.POINTER(_dp)

IO_SIGNATURES = {
    "wavwrite": (None, [_dp, C.c_int, C.c_int, C.c_int, C.c_char_p]),
    "GetAudioLength": (C.c_int, [C.c_char_p]),
    "wavread": (None, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp]),
    "WriteF0": (None, [C.c_char_p, C.c_int, C.c_double, _dp, _dp, C.c_int]),
    "ReadF0": (C.c_int, [C.c_char_p, _dp, _dp]),
    "GetHeaderInformation": (C.c_double, [C.c_char_p, C.c_char_p]),
    "WriteSpectralEnvelope": (None, [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _rows_t]),
    "ReadSpectralEnvelope": (C.c_int, [C.c_char_p, _rows_t]),
    "WriteAperiodicity": (None, [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _rows_t]),
    "ReadAperiodicity": (C.c_int, [C.c_char_p, _rows_t]),
    "wc_wavread_pcm16": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), _i16p, C.c_int]),
    "wc_pcm16_to_double_device": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p]),
    "wc_float_to_double_device": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p]),
    "wc_double_to_pcm16_device": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p]),
    "wc_modify_parameters_device": (C.c_int, [C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p, C.c_double, C.c_double]),
}

_bound = False


def _io():
    global _bound
    L = lib()
    if not _bound:
        for name, (res, args) in IO_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _bound = True
    return L


def _path(p):
    return str(p).encode()


def _rows(mat):
    from . import _rows as rows  # (a table of row addresses built by numpy, see world_class_amd/__init__.py)
    return rows(mat)


def wavwrite(x, fs, filename, nbit=16):
    x = _c(x)
    _io().wavwrite(x.ctypes.data_as(_dp), len(x), int(fs), int(nbit), _path(filename))


def audio_length(filename):
    return _io().GetAudioLength(_path(filename))


def wavread(filename):
    """(x, fs, nbit) like the reference's wavread; raises if the reference would have rejected the file."""
    n = audio_length(filename)
    if n <= 0:
        raise WorldClassError(f"cannot read {filename} (GetAudioLength = {n})")
    x = np.empty(n)
    fs, nbit = C.c_int(0), C.c_int(0)
    _io().wavread(_path(filename), C.byref(fs), C.byref(nbit), x.ctypes.data_as(_dp))
    return x, fs.value, nbit.value


def wavread_pcm16(filename):
    """(int16 samples as stored, fs) -- upload these and expand on the device with pcm16_to_double_device."""
    n = audio_length(filename)
    if n <= 0:
        raise WorldClassError(f"cannot read {filename} (GetAudioLength = {n})")
    pcm = np.empty(n, dtype=np.int16)
    fs = C.c_int(0)
    got = _io().wc_wavread_pcm16(_path(filename), C.byref(fs), pcm.ctypes.data_as(_i16p), n)
    if got < 0:
        raise WorldClassError(f"{filename} is not 16-bit PCM")
    return pcm[:got], fs.value


def header_information(filename, parameter):
    return _io().GetHeaderInformation(_path(filename), parameter.encode())


def write_f0(filename, temporal_positions, f0, frame_period, text=False):
    t, f = _c(temporal_positions), _c(f0)
    _io().WriteF0(_path(filename), len(f), float(frame_period), t.ctypes.data_as(_dp), f.ctypes.data_as(_dp), 1 if text else 0)


def read_f0(filename):
    n = int(header_information(filename, "NOF "))
    t, f = np.empty(n), np.empty(n)
    if _io().ReadF0(_path(filename), t.ctypes.data_as(_dp), f.ctypes.data_as(_dp)) != 1:
        raise WorldClassError(f"cannot read {filename}")
    return t, f


def _write_matrix(fn, filename, mat, fs, frame_period, fft_size, number_of_dimensions):
    mat = np.ascontiguousarray(mat, dtype=np.float64)
    fn(_path(filename), int(fs), mat.shape[0], float(frame_period), int(fft_size), int(number_of_dimensions), _rows(mat))


def _read_matrix(fn, filename):
    n = int(header_information(filename, "NOF "))
    fft_size = int(header_information(filename, "FFT "))
    nd = int(header_information(filename, "NOD ")) or fft_size // 2 + 1
    mat = np.empty((n, nd))
    if fn(_path(filename), _rows(mat)) != 1:
        raise WorldClassError(f"cannot read {filename}")
    return mat


def write_spectral_envelope(filename, sp, fs, frame_period, fft_size, number_of_dimensions=0):
    _write_matrix(_io().WriteSpectralEnvelope, filename, sp, fs, frame_period, fft_size, number_of_dimensions)


def read_spectral_envelope(filename):
    return _read_matrix(_io().ReadSpectralEnvelope, filename)


def write_aperiodicity(filename, ap, fs, frame_period, fft_size, number_of_dimensions=0):
    _write_matrix(_io().WriteAperiodicity, filename, ap, fs, frame_period, fft_size, number_of_dimensions)


def read_aperiodicity(filename):
    return _read_matrix(_io().ReadAperiodicity, filename)


def pcm16_to_double_device(d_pcm, n, d_x):
    _check(_io().wc_pcm16_to_double_device(_ptr(d_pcm), int(n), _ptr(d_x)))


def float_to_double_device(d_f, n, d_x):
    _check(_io().wc_float_to_double_device(_ptr(d_f), int(n), _ptr(d_x)))


def double_to_pcm16_device(d_y, n, d_pcm):
    _check(_io().wc_double_to_pcm16_device(_ptr(d_y), int(n), _ptr(d_pcm)))


def modify_parameters_device(fs, fft_size, n_frames, d_f0, d_sp, f0_scale=1.0, spectral_ratio=0.0):
    """reference test/test.cpp:201-243 on device-resident parameters (0 = leave the spectra alone)"""
    _check(_io().wc_modify_parameters_device(int(fs), int(fft_size), int(n_frames), _ptr(d_f0), _ptr(d_sp), float(f0_scale),
                                             float(spectral_ratio)))
