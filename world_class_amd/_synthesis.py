"""Synthesis stage wrapper (reference include/synthesis.hpp:29-51) over the C-ABI."""
import ctypes as C

import numpy as np

from . import (_c, _check, _handle, _ints, _p, _ptr, _ptr_array, _rng_arg, _row_tables, _rows, lib,
               synthesis_out_length)


class Synthesis:
    """Synthesis(fs, fft_size, frame_period_ms); compute(f0, spectrogram, aperiodicity, out_length)"""

    def __init__(self, fs, fft_size, frame_period=5.0):
        self.fs, self.fft_size, self.frame_period = fs, fft_size, frame_period
        self.bins = fft_size // 2 + 1
        self._h = _handle(lib().wc_synthesis_create(fs, fft_size, frame_period))

    def out_length(self, f0_length):
        return synthesis_out_length(f0_length, self.frame_period, self.fs)  # reference test/test.cpp:362-363

    def compute(self, f0, spectrogram, aperiodicity, out_length=None, out=None):
        f = _c(f0)
        sp, ap = _c(spectrogram), _c(aperiodicity)
        if sp.shape != (len(f), self.bins) or ap.shape != (len(f), self.bins):
            raise ValueError(f"Synthesis.compute: spectrogram and aperiodicity must be ({len(f)}, {self.bins})")
        if out_length is None:
            out_length = self.out_length(len(f)) if out is None else len(out)
        if out is None:
            out = np.zeros(out_length)
        elif not isinstance(out, np.ndarray) or out.dtype != np.float64 or out.ndim != 1 or not out.flags.c_contiguous or len(out) < out_length:
            raise ValueError(f"Synthesis.compute: out must be a contiguous float64 vector of at least {out_length} samples")
        _check(lib().wc_synthesis_compute(self._h, _p(f), len(f), _rows(sp), _rows(ap), out_length, _p(out)))
        return out

    def compute_device(self, d_f0, f0_lengths, d_sp, d_ap, out_lengths, d_out, rng_pos=None):
        n = len(f0_lengths)
        arr, arg = _rng_arg(rng_pos, n)
        _check(lib().wc_synthesis_compute_device(self._h, n, _ptr(d_f0), _ints(f0_lengths), _ptr(d_sp), _ptr(d_ap),
                                                 _ints(out_lengths), _ptr(d_out), arg))
        return list(arr) if arr is not None else None

    def compute_batch(self, f0s, sps, aps, out_lengths=None, rng_pos=None):
        fl = [len(v) for v in f0s]
        if out_lengths is None:
            out_lengths = [self.out_length(n) for n in fl]
        fs_, sps, aps = [_c(v) for v in f0s], [_c(v) for v in sps], [_c(v) for v in aps]
        if not (len(fs_) == len(sps) == len(aps) == len(out_lengths)) or len(fs_) == 0:
            raise ValueError("Synthesis.compute_batch: f0s, sps, aps (and out_lengths) must be non-empty lists of the same length")
        for u, (f, sp, ap) in enumerate(zip(fs_, sps, aps)):
            if sp.shape != (len(f), self.bins) or ap.shape != (len(f), self.bins):
                raise ValueError(f"Synthesis.compute_batch: utterance {u}: spectrogram and aperiodicity must be ({len(f)}, {self.bins})")
        ys = [np.zeros(n) for n in out_lengths]
        stab, keep1 = _row_tables(sps)
        atab, keep2 = _row_tables(aps)
        arr, arg = _rng_arg(rng_pos, len(fl))
        _check(lib().wc_synthesis_compute_batch(self._h, len(fl), _ptr_array(fs_), _ints(fl), self.fft_size, stab, atab, _ints(out_lengths),
                                                _ptr_array(ys), arg))
        return (ys, list(arr)) if rng_pos is not None else ys

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().wc_synthesis_destroy(self._h)
                self._h = None
        except Exception:
            pass
