"""Harvest stage wrapper (reference include/harvest.hpp:16-44) over the C-ABI."""
import ctypes as C

import numpy as np

from . import _c, _check, _handle, _ints, _p, _ptr, _ptr_array, get_samples, lib


class Harvest:
    """HarvestOption defaults of reference src/harvest.cpp:52-56: f0_floor 71, f0_ceil 800, frame_period 5,
    target_fs 8000, channels_in_octave 40, use_cos_table False (True: the reference's tabulated refinement window)."""

    def __init__(self, fs, f0_floor=71.0, f0_ceil=800.0, frame_period=5.0, target_fs=8000.0,
                 channels_in_octave=40.0, use_cos_table=False):
        self.fs, self.frame_period = fs, frame_period
        self._h = _handle(lib().wc_harvest_create(fs, f0_floor, f0_ceil, frame_period, target_fs,
                                                  channels_in_octave, int(use_cos_table)))

    def get_samples(self, x_length):
        return get_samples(self.fs, x_length, self.frame_period)

    def compute(self, x):
        x = _c(x)
        n = self.get_samples(len(x))
        tpos, f0 = np.zeros(n), np.zeros(n)
        _check(lib().wc_harvest_compute(self._h, _p(x), len(x), _p(tpos), _p(f0)))
        return tpos, f0

    def compute_device(self, d_x, x_lengths, d_tpos, d_f0):
        _check(lib().wc_harvest_compute_device(self._h, len(x_lengths), _ptr(d_x), _ints(x_lengths), _ptr(d_tpos),
                                               _ptr(d_f0)))

    def compute_batch(self, xs):
        """host list in, list of (temporal positions, F0) out: wc_harvest_compute_batch (one trip over PCIe each way, one batch)"""
        xs = [_c(v) for v in xs]
        fl = [self.get_samples(len(x)) for x in xs]
        ts, fs_ = [np.zeros(n) for n in fl], [np.zeros(n) for n in fl]
        _check(lib().wc_harvest_compute_batch(self._h, len(xs), _ptr_array(xs), _ints([len(x) for x in xs]), _ptr_array(ts), _ptr_array(fs_)))
        return list(zip(ts, fs_))

    def debug_fetch(self, name, utt=0):
        """Development hook: an intermediate of the most recent call (y, raw, cand0, cand1, score1, cand, score, base,
        s1, s2, s3, fixed, f0_1ms) as a flat float64 array."""
        fn = lib().wc_harvest_debug_fetch
        fn.restype = C.c_longlong
        fn.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p]
        n = fn(self._h, name.encode(), utt, None)
        if n < 0:
            _check(int(n))
        out = np.zeros(n)
        _check(min(0, int(fn(self._h, name.encode(), utt, out.ctypes.data))))
        return out

    def debug_refine(self, cand0, by_slots=False):
        """Development hook: the refinement kernel alone on the candidate rows `cand0` ([1 ms frames of the most recent call][S],
        0 = empty slot); returns (refined candidates, scores), each [frames][7 S].  by_slots: the slot layout instead of the packed one."""
        fn = lib().wc_harvest_debug_refine
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        cand0 = np.ascontiguousarray(cand0, dtype=np.float64)
        frames, S = cand0.shape
        c1 = np.zeros((frames, 7 * S))
        s1 = np.zeros((frames, 7 * S))
        _check(fn(self._h, cand0.ctypes.data, int(by_slots), c1.ctypes.data, s1.ctypes.data))  # 0 default (frames in groups), 1 slots, 2 packed
        return c1, s1

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().wc_harvest_destroy(self._h)
                self._h = None
        except Exception:
            pass
