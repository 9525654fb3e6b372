"""world_class_amd -- MI355X-native WORLD analysis/synthesis hot path.

Python host-side mirror of the reference's four stage classes (reference include/harvest.hpp:16-44,
include/cheaptrick.hpp:14-38, include/d4c.hpp:16-36, include/synthesis.hpp:29-51) over the C-ABI of
include/world_class_c.h (libworldclass_hip.so, hand-written HIP for gfx950).

There is no CPU fallback: importing works anywhere (so the build can be checked without a GPU), but
creating a stage object or calling compute without the HIP library / a HIP device raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# WC_LIB_PATH: another build of the same library (development: A/B runs of kernel variants, tools/ab_build.py)
LIB_PATH = os.environ.get("WC_LIB_PATH") or os.path.join(_HERE, "libworldclass_hip.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_u64p = C.POINTER(C.c_uint64)
_vp = C.c_void_p

_SIGNATURES = {
    "wc_last_error": (C.c_char_p, []),
    "wc_version": (C.c_char_p, []),
    "wc_build_hash": (C.c_char_p, []),
    "wc_device_count": (C.c_int, []),
    "wc_set_device": (C.c_int, [C.c_int]),
    "wc_get_device": (C.c_int, []),
    "wc_set_stream": (C.c_int, [_vp]),
    "wc_synchronize": (C.c_int, []),
    "wc_rng_get_position": (C.c_uint64, []),
    "wc_rng_set_position": (None, [C.c_uint64]),
    "wc_release_scratch": (C.c_int, []),
    "wc_get_samples": (C.c_int, [C.c_int, C.c_int, C.c_double]),
    "wc_cheaptrick_fft_size": (C.c_int, [C.c_int, C.c_double]),
    "wc_cheaptrick_f0_floor": (C.c_double, [C.c_int, C.c_int]),
    "wc_synthesis_out_length": (C.c_int, [C.c_int, C.c_double, C.c_int]),
    "wc_harvest_create": (_vp, [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]),
    "wc_harvest_destroy": (None, [_vp]),
    "wc_harvest_compute": (C.c_int, [_vp, _dp, C.c_int, _dp, _dp]),
    "wc_harvest_compute_device": (C.c_int, [_vp, C.c_int, _vp, _ip, _vp, _vp]),
    "wc_harvest_get_samples": (C.c_int, [_vp, C.c_int]),
    "wc_harvest_compute_batch": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _ip, C.POINTER(_vp), C.POINTER(_vp)]),
    "wc_cheaptrick_compute_batch": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _ip, C.POINTER(_vp), C.POINTER(_vp), _ip, C.POINTER(_vp), _u64p]),
    "wc_d4c_compute_batch": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _ip, C.POINTER(_vp), C.POINTER(_vp), _ip, C.c_int, C.POINTER(_vp), _u64p]),
    "wc_synthesis_compute_batch": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _ip, C.c_int, C.POINTER(_vp), C.POINTER(_vp), _ip, C.POINTER(_vp), _u64p]),
    "wc_cheaptrick_create": (_vp, [C.c_int, C.c_double, C.c_double, C.c_int]),
    "wc_cheaptrick_destroy": (None, [_vp]),
    "wc_cheaptrick_get_fft_size": (C.c_int, [_vp]),
    "wc_cheaptrick_compute": (C.c_int, [_vp, _dp, C.c_int, _dp, _dp, C.c_int, C.POINTER(_dp)]),
    "wc_cheaptrick_compute_device": (C.c_int, [_vp, C.c_int, _vp, _ip, _vp, _vp, _ip, _vp, _u64p]),
    "wc_d4c_create": (_vp, [C.c_int, C.c_double]),
    "wc_d4c_destroy": (None, [_vp]),
    "wc_d4c_compute": (C.c_int, [_vp, _dp, C.c_int, _dp, _dp, C.c_int, C.c_int, C.POINTER(_dp)]),
    "wc_d4c_compute_device": (C.c_int, [_vp, C.c_int, _vp, _ip, _vp, _vp, _ip, C.c_int, _vp, _u64p]),
    "wc_synthesis_create": (_vp, [C.c_int, C.c_int, C.c_double]),
    "wc_synthesis_destroy": (None, [_vp]),
    "wc_synthesis_get_fft_size": (C.c_int, [_vp]),
    "wc_synthesis_compute": (C.c_int, [_vp, _dp, C.c_int, C.POINTER(_dp), C.POINTER(_dp), C.c_int, _dp]),
    "wc_synthesis_compute_device": (C.c_int, [_vp, C.c_int, _vp, _ip, _vp, _vp, _ip, _vp, _u64p]),
    "wc_pipeline_create": (_vp, [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double]),
    "wc_pipeline_destroy": (None, [_vp]),
    "wc_pipeline_get_fft_size": (C.c_int, [_vp]),
    "wc_pipeline_set_option": (C.c_int, [_vp, C.c_char_p, C.c_char_p]),
    "wc_pipeline_run_device": (C.c_int, [_vp, C.c_int, _vp, _ip, _vp, _vp, _vp, _vp, _vp, _u64p]),
    "wc_pipeline_run_batch_host": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.c_int, _ip, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                             C.POINTER(_vp), C.POINTER(_vp), C.c_int, _u64p]),
    "wc_pipeline_run_batch_host_coded": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.c_int, _ip, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                                   C.c_int, C.POINTER(_vp), C.POINTER(_vp), C.c_int, _u64p]),
    "wc_device_malloc": (_vp, [C.c_uint64]),
    "wc_device_free": (None, [_vp]),
    "wc_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_uint64]),
    "wc_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_uint64]),
    "wc_set_kernel_timing": (C.c_int, [C.c_int]),
    "wc_last_kernel_ms": (C.c_float, [C.c_char_p]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))

_lib = None


class WorldClassError(RuntimeError):
    pass


def lib():
    """The loaded C-ABI library; raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WorldClassError(
                f"{LIB_PATH} is missing: build it with `python -m world_class_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback.")
        try:  # share torch's HIP runtime when torch is in the process (same SONAME)
            import torch  # noqa: F401
        except Exception:  # pragma: no cover
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return lib().wc_last_error().decode()


def _check(rc):
    if rc != 0:
        raise WorldClassError(f"world_class_amd: error {rc}: {last_error()}")


def _handle(h):
    if not h:
        raise WorldClassError(f"world_class_amd: {last_error()}")
    return h


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def _out_matrix(out, rows, cols, what):
    """a caller's result buffer: exactly (rows, cols) float64, C-contiguous -- anything else would be a heap overwrite in C"""
    if out is None:
        return np.empty((rows, cols))
    if not isinstance(out, np.ndarray) or out.dtype != np.float64 or out.shape != (rows, cols) or not out.flags.c_contiguous:
        raise ValueError(f"{what}: out must be a C-contiguous float64 array of shape ({rows}, {cols})")
    return out


def _row_tables(mats):
    """(array of row-pointer tables, keep-alive list) for a list of 2-D float64 arrays: the reference's double** per utterance"""
    tabs = [_rows(m) for m in mats]
    arr = (_vp * len(mats))(*[C.cast(t, _vp) for t in tabs])
    return arr, tabs


def _ptr_array(arrs):
    return (_vp * len(arrs))(*[a.ctypes.data for a in arrs])


def _rows(mat):
    """the reference's double** for a 2-D float64 array: a table of row addresses (built by numpy -- a Python loop of ctypes casts
    over 2001 rows took 1 ms per call and 36 ms the first time, more than the stage's own work)"""
    addr = np.uintp(mat.ctypes.data) + np.arange(mat.shape[0], dtype=np.uintp) * np.uintp(mat.strides[0])
    return addr.ctypes.data_as(C.POINTER(_dp))  # (the pointer object keeps `addr` alive)


# ---- size helpers ---------------------------------------------------------------------------
def get_samples(fs, x_length, frame_period=5.0):
    return lib().wc_get_samples(fs, x_length, frame_period)


def cheaptrick_fft_size(fs, f0_floor=71.0):
    return lib().wc_cheaptrick_fft_size(fs, f0_floor)


def synthesis_out_length(f0_length, frame_period, fs):
    return lib().wc_synthesis_out_length(f0_length, frame_period, fs)


def rng_set_position(pos=0):
    lib().wc_rng_set_position(int(pos))


def rng_get_position():
    return int(lib().wc_rng_get_position())


# ---- device buffers (thin RAII over wc_device_malloc) ------------------------------------------
class DeviceArray:
    """A packed float64 (or raw byte) array in HBM owned by the library's allocator."""

    def __init__(self, n, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self.n = int(n)
        self.ptr = _handle(lib().wc_device_malloc(max(1, self.n) * self.dtype.itemsize))

    @classmethod
    def from_host(cls, arr, dtype=np.float64):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        d = cls(arr.size, dtype)
        if arr.size:
            _check(lib().wc_memcpy_h2d(d.ptr, arr.ctypes.data, arr.nbytes))
        return d

    def to_host(self, shape=None):
        out = np.empty(self.n, dtype=self.dtype)
        if self.n:
            _check(lib().wc_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if self.ptr:
            lib().wc_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ints(v):
    return (C.c_int * len(v))(*[int(i) for i in v])


def _ptr(obj):
    """Device pointer of a DeviceArray, a torch tensor, or a raw integer address."""
    if isinstance(obj, DeviceArray):
        return obj.ptr
    if hasattr(obj, "data_ptr"):
        return obj.data_ptr()
    return int(obj)


def _rng_arg(rng_pos, n):
    if rng_pos is None:
        return None, None
    arr = (C.c_uint64 * n)(*[int(v) for v in rng_pos])
    return arr, arr


# ---- stage classes ----------------------------------------------------------------------------
class CheapTrick:
    """reference include/cheaptrick.hpp:14-38 (CheapTrickOption defaults: q1 -0.15, f0_floor 71, fft_size 0=auto)"""

    def __init__(self, fs, q1=-0.15, f0_floor=71.0, fft_size=0):
        self.fs = fs
        self._h = _handle(lib().wc_cheaptrick_create(fs, q1, f0_floor, fft_size))
        self.fft_size = lib().wc_cheaptrick_get_fft_size(self._h)
        self.bins = self.fft_size // 2 + 1

    def compute(self, x, temporal_positions, f0, out=None):
        """out: a (frames, bins) float64 array of an earlier call to be written again (the caller's buffer, as in the reference
        demo, which allocates its rows once: reference test/test.cpp:92-101)"""
        x, t, f = _c(x), _c(temporal_positions), _c(f0)
        sp = _out_matrix(out, len(f), self.bins, "CheapTrick.compute")
        _check(lib().wc_cheaptrick_compute(self._h, _p(x), len(x), _p(t), _p(f), len(f), _rows(sp)))
        return sp

    def compute_device(self, d_x, x_lengths, d_tpos, d_f0, f0_lengths, d_sp, rng_pos=None):
        n = len(x_lengths)
        arr, arg = _rng_arg(rng_pos, n)
        _check(lib().wc_cheaptrick_compute_device(self._h, n, _ptr(d_x), _ints(x_lengths), _ptr(d_tpos), _ptr(d_f0),
                                                  _ints(f0_lengths), _ptr(d_sp), arg))
        return list(arr) if arr is not None else None

    def compute_batch(self, xs, tposs, f0s, rng_pos=None):
        """host lists in, host list of spectrograms out: wc_cheaptrick_compute_batch (one trip over PCIe each way, one batch)"""
        xs, ts, fs_ = [_c(v) for v in xs], [_c(v) for v in tposs], [_c(v) for v in f0s]
        _check_batch_shapes("CheapTrick.compute_batch", xs, ts, fs_)
        out = [np.empty((len(f), self.bins)) for f in fs_]
        tabs, keep = _row_tables(out)
        arr, arg = _rng_arg(rng_pos, len(xs))
        _check(lib().wc_cheaptrick_compute_batch(self._h, len(xs), _ptr_array(xs), _ints([len(v) for v in xs]), _ptr_array(ts), _ptr_array(fs_),
                                                 _ints([len(v) for v in fs_]), tabs, arg))
        return (out, list(arr)) if rng_pos is not None else out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().wc_cheaptrick_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _check_batch_shapes(who, xs, ts, fs_):
    """the per-utterance lists of a host batch call must agree: the C side strides by the lengths it is given"""
    if not (len(xs) == len(ts) == len(fs_)) or len(xs) == 0:
        raise ValueError(f"{who}: xs, tposs and f0s must be non-empty lists of the same length")
    for u, (x, t, f) in enumerate(zip(xs, ts, fs_)):
        if x.ndim != 1 or t.ndim != 1 or f.ndim != 1 or len(t) != len(f):
            raise ValueError(f"{who}: utterance {u}: x, temporal positions and f0 must be vectors, the last two of one length")


class D4C:
    """reference include/d4c.hpp:16-36 (D4COption default threshold 0.85)"""

    def __init__(self, fs, threshold=0.85):
        self.fs = fs
        self._h = _handle(lib().wc_d4c_create(fs, threshold))

    def compute(self, x, temporal_positions, f0, fft_size, out=None):
        x, t, f = _c(x), _c(temporal_positions), _c(f0)
        ap = _out_matrix(out, len(f), fft_size // 2 + 1, "D4C.compute")
        _check(lib().wc_d4c_compute(self._h, _p(x), len(x), _p(t), _p(f), len(f), fft_size, _rows(ap)))
        return ap

    def compute_device(self, d_x, x_lengths, d_tpos, d_f0, f0_lengths, fft_size, d_ap, rng_pos=None):
        n = len(x_lengths)
        arr, arg = _rng_arg(rng_pos, n)
        _check(lib().wc_d4c_compute_device(self._h, n, _ptr(d_x), _ints(x_lengths), _ptr(d_tpos), _ptr(d_f0),
                                           _ints(f0_lengths), fft_size, _ptr(d_ap), arg))
        return list(arr) if arr is not None else None

    def compute_batch(self, xs, tposs, f0s, fft_size, rng_pos=None):
        """host lists in, host list of aperiodicity matrices out: wc_d4c_compute_batch"""
        bins = fft_size // 2 + 1
        xs, ts, fs_ = [_c(v) for v in xs], [_c(v) for v in tposs], [_c(v) for v in f0s]
        _check_batch_shapes("D4C.compute_batch", xs, ts, fs_)
        out = [np.empty((len(f), bins)) for f in fs_]
        tabs, keep = _row_tables(out)
        arr, arg = _rng_arg(rng_pos, len(xs))
        _check(lib().wc_d4c_compute_batch(self._h, len(xs), _ptr_array(xs), _ints([len(v) for v in xs]), _ptr_array(ts), _ptr_array(fs_),
                                          _ints([len(v) for v in fs_]), fft_size, tabs, arg))
        return (out, list(arr)) if rng_pos is not None else out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().wc_d4c_destroy(self._h)
                self._h = None
        except Exception:
            pass


from ._synthesis import Synthesis  # noqa: E402,F401
from ._harvest import Harvest  # noqa: E402,F401
from ._pipeline import Pipeline  # noqa: E402,F401
