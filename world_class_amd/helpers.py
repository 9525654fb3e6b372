"""Python mirror of include/world_matlabfunctions.hpp: the reference's free helper functions as host functions of
libworldclass_hip.so (same names as reference include/world_matlabfunctions.hpp / world_common.hpp)."""
import ctypes as C

import numpy as np

from . import _c, lib

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

HELPER_SIGNATURES = {
    "fftshift": (None, [_dp, C.c_int, _dp]),
    "histc": (None, [_dp, C.c_int, _dp, C.c_int, _ip]),
    "interp1": (None, [_dp, _dp, C.c_int, _dp, C.c_int, _dp]),
    "decimate": (None, [_dp, C.c_int, C.c_int, _dp]),
    "matlab_round": (C.c_int, [C.c_double]),
    "diff": (None, [_dp, C.c_int, _dp]),
    "interp1Q": (None, [C.c_double, C.c_double, _dp, C.c_int, _dp, C.c_int, _dp]),
    "randn": (C.c_double, []),
    "matlab_std": (C.c_double, [_dp, C.c_int]),
    "GetSuitableFFTSize": (C.c_int, [C.c_int]),
    "DCCorrection": (None, [_dp, C.c_double, C.c_int, C.c_int, _dp]),
    "LinearSmoothing": (None, [_dp, C.c_double, C.c_int, C.c_int, _dp]),
    "NuttallWindow": (None, [C.c_int, _dp]),
}

_bound = False


def _L():
    global _bound
    L = lib()
    if not _bound:
        for name, (res, args) in HELPER_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _bound = True
    return L


def _p(a):
    return a.ctypes.data_as(_dp)


def fftshift(x):
    x = _c(x)
    y = np.empty_like(x)
    _L().fftshift(_p(x), len(x), _p(y))
    return y


def histc(x, edges):
    x, e = _c(x), _c(edges)
    idx = np.zeros(len(e), dtype=np.int32)
    _L().histc(_p(x), len(x), _p(e), len(e), idx.ctypes.data_as(_ip))
    return idx


def interp1(x, y, xi):
    x, y, xi = _c(x), _c(y), _c(xi)
    yi = np.empty(len(xi))
    _L().interp1(_p(x), _p(y), len(x), _p(xi), len(xi), _p(yi))
    return yi


def interp1Q(x0, shift, y, xi):
    y, xi = _c(y), _c(xi)
    yi = np.empty(len(xi))
    _L().interp1Q(float(x0), float(shift), _p(y), len(y), _p(xi), len(xi), _p(yi))
    return yi


def decimate(x, r):
    x = _c(x)
    y = np.zeros(len(x) // r + 32)  # the reference writes a few values past x_length / r + 1
    _L().decimate(_p(x), len(x), int(r), _p(y))
    return y[:len(x) // r + 1]


def matlab_round(x):
    return _L().matlab_round(float(x))


def diff(x):
    x = _c(x)
    y = np.empty(max(len(x) - 1, 0))
    _L().diff(_p(x), len(x), _p(y))
    return y


def randn(n=None):
    L = _L()
    if n is None:
        return L.randn()
    return np.array([L.randn() for _ in range(n)])


def matlab_std(x):
    x = _c(x)
    return _L().matlab_std(_p(x), len(x))


def suitable_fft_size(sample):
    return _L().GetSuitableFFTSize(int(sample))


def dc_correction(spectrum, f0, fs, fft_size):
    s = _c(spectrum)
    out = s.copy()
    _L().DCCorrection(_p(s), float(f0), int(fs), int(fft_size), _p(out))
    return out


def linear_smoothing(spectrum, width, fs, fft_size):
    s = _c(spectrum)
    out = np.empty(fft_size // 2 + 1)
    _L().LinearSmoothing(_p(s), float(width), int(fs), int(fft_size), _p(out))
    return out


def nuttall_window(n):
    y = np.empty(n)
    _L().NuttallWindow(int(n), _p(y))
    return y
