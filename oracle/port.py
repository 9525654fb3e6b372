"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/_build/libwc_oracle.so (our CPU restatement,
oracle/wc_oracle.cpp).  Same method names as oracle/ref.py's Ref so tests can swap them.
Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def lib_path():
    return os.path.join(_HERE, "_build", "libwc_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("wc_oracle.cpp", "wc_oracle.h")]
    out = lib_path()
    if force or not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in src):
        subprocess.run(["make", "-C", _HERE, "oracle"], check=True, stdout=subprocess.DEVNULL)
    return out


def _p(a):
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Port:
    def __init__(self, threads=0):
        self.lib = C.CDLL(build())
        L = self.lib
        L.wco_rng_seek.argtypes = [C.c_uint64]
        L.wco_rng_position.restype = C.c_uint64
        L.wco_randn_fill.argtypes = [C.c_int, _dp]
        L.wco_matlab_round.restype = C.c_int
        L.wco_matlab_round.argtypes = [C.c_double]
        L.wco_suitable_fft_size.restype = C.c_int
        L.wco_suitable_fft_size.argtypes = [C.c_int]
        L.wco_interp1.argtypes = [_dp, _dp, C.c_int, _dp, C.c_int, _dp]
        L.wco_interp1Q.argtypes = [C.c_double, C.c_double, _dp, C.c_int, _dp, C.c_int, _dp]
        L.wco_histc.argtypes = [_dp, C.c_int, _dp, C.c_int, _ip]
        L.wco_decimate.argtypes = [_dp, C.c_int, C.c_int, _dp]
        L.wco_dc_correction.argtypes = [_dp, C.c_double, C.c_int, C.c_int, _dp]
        L.wco_linear_smoothing.argtypes = [_dp, C.c_double, C.c_int, C.c_int, _dp]
        L.wco_nuttall.argtypes = [C.c_int, _dp]
        L.wco_fft_r2c.argtypes = [C.c_int, _dp, _dp]
        L.wco_fft_c2r.argtypes = [C.c_int, _dp, _dp]
        L.wco_fft_c2c.argtypes = [C.c_int, C.c_int, _dp, _dp]
        L.wco_minimum_phase.argtypes = [C.c_int, _dp, _dp]
        L.wco_set_threads.argtypes = [C.c_int]
        L.wco_get_samples.restype = C.c_int
        L.wco_get_samples.argtypes = [C.c_int, C.c_int, C.c_double]
        L.wco_harvest.argtypes = [_dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp]
        L.wco_cheaptrick_fft_size.restype = C.c_int
        L.wco_cheaptrick_fft_size.argtypes = [C.c_int, C.c_double]
        L.wco_cheaptrick_f0_floor.restype = C.c_double
        L.wco_cheaptrick_f0_floor.argtypes = [C.c_int, C.c_int]
        L.wco_cheaptrick.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_double,
                                     C.c_double, C.c_int, _dp]
        L.wco_d4c.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double, _dp]
        L.wco_synthesis.argtypes = [_dp, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_int, _dp]
        L.wco_cheaptrick_draws.restype = C.c_uint64
        L.wco_cheaptrick_draws.argtypes = [C.c_int, _dp, C.c_int, C.c_double, C.c_int]
        L.wco_harvest_debug.restype = C.c_int
        L.wco_harvest_debug.argtypes = [_dp, C.c_int, C.c_int, C.c_double, C.c_double, _ip,
                                        _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.wco_synthesis_pulses.restype = C.c_int
        L.wco_synthesis_pulses.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _ip]
        L.wco_set_threads(threads)

    def set_harvest_options(self, target_fs=8000.0, channels_in_octave=40.0, use_cos_table=False):
        """the HarvestOption fields beyond floor / ceil / frame period, process-wide until set back (call without arguments)"""
        self.lib.wco_set_harvest_options.argtypes = [C.c_double, C.c_double, C.c_int]
        self.lib.wco_set_harvest_options(target_fs, channels_in_octave, int(use_cos_table))

    # ---- RNG ----------------------------------------------------------------------------
    def rng_reset(self):
        self.lib.wco_rng_reset()

    def rng_seek(self, pos):
        self.lib.wco_rng_seek(int(pos))

    def rng_position(self):
        return int(self.lib.wco_rng_position())

    def randn(self, n):
        out = np.zeros(n)
        self.lib.wco_randn_fill(n, _p(out))
        return out

    def set_threads(self, t):
        self.lib.wco_set_threads(int(t))

    # ---- stages -------------------------------------------------------------------------
    def get_samples(self, fs, n, frame_period=5.0):
        return self.lib.wco_get_samples(fs, n, frame_period)

    def harvest(self, x, fs, f0_floor=71.0, f0_ceil=800.0, frame_period=5.0):
        x = _c(x)
        L = self.get_samples(fs, len(x), frame_period)
        tpos = np.zeros(L)
        f0 = np.zeros(L)
        self.lib.wco_harvest(_p(x), len(x), fs, f0_floor, f0_ceil, frame_period, _p(tpos), _p(f0))
        return tpos, f0

    def harvest_debug(self, x, fs, f0_floor=71.0, f0_ceil=800.0):
        x = _c(x)
        dims = np.zeros(4, dtype=np.int32)
        L1 = self.lib.wco_harvest_debug(_p(x), len(x), fs, f0_floor, f0_ceil,
                                        dims.ctypes.data_as(_ip), None, None, None, None, None, None, None)
        yl, nb, mc, nc = [int(v) for v in dims]
        y = np.zeros(yl)
        raw = np.zeros((nb, L1))
        cand = np.zeros((L1, mc))
        score = np.zeros((L1, mc))
        base = np.zeros(L1)
        fixed = np.zeros(L1)
        f1 = np.zeros(L1)
        self.lib.wco_harvest_debug(_p(x), len(x), fs, f0_floor, f0_ceil, dims.ctypes.data_as(_ip),
                                   _p(y), _p(raw), _p(cand), _p(score), _p(base), _p(fixed), _p(f1))
        return dict(y=y, raw=raw, cand=cand, score=score, f0_base=base, f0_fixed=fixed, f0_1ms=f1,
                    n_cand=nc)

    def cheaptrick_fft_size(self, fs, f0_floor=71.0):
        return self.lib.wco_cheaptrick_fft_size(fs, f0_floor)

    def cheaptrick_f0_floor(self, fs, fft_size):
        return self.lib.wco_cheaptrick_f0_floor(fs, fft_size)

    def cheaptrick(self, x, fs, tpos, f0, q1=-0.15, f0_floor=71.0, fft_size=0):
        x, tpos, f0 = _c(x), _c(tpos), _c(f0)
        nfft = fft_size or self.cheaptrick_fft_size(fs, f0_floor)
        sp = np.zeros((len(f0), nfft // 2 + 1))
        self.lib.wco_cheaptrick(_p(x), len(x), fs, _p(tpos), _p(f0), len(f0), q1, f0_floor,
                                fft_size, _p(sp))
        return sp

    def cheaptrick_draws(self, fs, f0, f0_floor=71.0, fft_size=0):
        f0 = _c(f0)
        return int(self.lib.wco_cheaptrick_draws(fs, _p(f0), len(f0), f0_floor, fft_size))

    def d4c(self, x, fs, tpos, f0, fft_size, threshold=0.85):
        x, tpos, f0 = _c(x), _c(tpos), _c(f0)
        ap = np.zeros((len(f0), fft_size // 2 + 1))
        self.lib.wco_d4c(_p(x), len(x), fs, _p(tpos), _p(f0), len(f0), fft_size, threshold, _p(ap))
        return ap

    def synthesis(self, f0, sp, ap, fs, frame_period=5.0, out_length=None):
        f0, sp, ap = _c(f0), _c(sp), _c(ap)
        fft_size = (sp.shape[1] - 1) * 2
        if out_length is None:
            out_length = int((len(f0) - 1) * frame_period / 1000.0 * fs) + 1
        y = np.zeros(out_length)
        self.lib.wco_synthesis(_p(f0), len(f0), _p(sp), _p(ap), fft_size, fs, frame_period,
                               out_length, _p(y))
        return y

    def synthesis_pulses(self, f0, fft_size, fs, frame_period=5.0, out_length=None):
        """(number of pulses, capacity the reference allocates) -- count > capacity overflows the reference."""
        f0 = _c(f0)
        if out_length is None:
            out_length = int((len(f0) - 1) * frame_period / 1000.0 * fs) + 1
        cap = C.c_int(0)
        n = self.lib.wco_synthesis_pulses(_p(f0), len(f0), fft_size, fs, frame_period, out_length, C.byref(cap))
        return n, cap.value

    def pipeline(self, x, fs, harvest_floor=71.0, frame_period=5.0, given_f0=None):
        """Demo order of reference test/test.cpp:288-384, RNG at its seed state."""
        self.rng_reset()
        if given_f0 is None:
            tpos, f0 = self.harvest(x, fs, f0_floor=harvest_floor, frame_period=frame_period)
        else:
            tpos, f0 = given_f0
        sp = self.cheaptrick(x, fs, tpos, f0)
        fft_size = (sp.shape[1] - 1) * 2
        ap = self.d4c(x, fs, tpos, f0, fft_size)
        syn_start = self.rng_position()  # (where Synthesis starts in the noise stream: to repeat that stage on other parameters)
        y = self.synthesis(f0, sp, ap, fs, frame_period)
        return dict(tpos=tpos, f0=f0, sp=sp, ap=ap, y=y, syn_start=syn_start)

    # ---- helpers ------------------------------------------------------------------------
    def matlab_round(self, x):
        return self.lib.wco_matlab_round(float(x))

    def suitable_fft_size(self, n):
        return self.lib.wco_suitable_fft_size(int(n))

    def interp1(self, x, y, xi):
        x, y, xi = _c(x), _c(y), _c(xi)
        yi = np.zeros(len(xi))
        self.lib.wco_interp1(_p(x), _p(y), len(x), _p(xi), len(xi), _p(yi))
        return yi

    def interp1Q(self, x0, dx, y, xi):
        y, xi = _c(y), _c(xi)
        yi = np.zeros(len(xi))
        self.lib.wco_interp1Q(x0, dx, _p(y), len(y), _p(xi), len(xi), _p(yi))
        return yi

    def histc(self, x, edges):
        x, edges = _c(x), _c(edges)
        idx = np.zeros(len(edges), dtype=np.int32)
        self.lib.wco_histc(_p(x), len(x), _p(edges), len(edges), idx.ctypes.data_as(_ip))
        return idx

    def decimate(self, x, r):
        x = _c(x)
        y = np.zeros(len(x) + 32)  # the reference writes n/r + ceil(9/r) values
        self.lib.wco_decimate(_p(x), len(x), r, _p(y))
        return y[:len(x) // r + 1].copy()

    def dc_correction(self, spec, f0, fs, fft_size):
        spec = _c(spec)
        out = spec.copy()
        self.lib.wco_dc_correction(_p(spec), f0, fs, fft_size, _p(out))
        return out

    def linear_smoothing(self, spec, width, fs, fft_size):
        spec = _c(spec)
        out = np.zeros(fft_size // 2 + 1)
        self.lib.wco_linear_smoothing(_p(spec), width, fs, fft_size, _p(out))
        return out

    def nuttall(self, n):
        y = np.zeros(n)
        self.lib.wco_nuttall(n, _p(y))
        return y

    def fft_r2c(self, x):
        x = _c(x)
        out = np.zeros((len(x) // 2 + 1, 2))
        self.lib.wco_fft_r2c(len(x), _p(x), _p(out))
        return out[:, 0] + 1j * out[:, 1]

    def fft_c2r(self, X, n):
        a = np.zeros((n // 2 + 1, 2))
        a[:, 0], a[:, 1] = X.real, X.imag
        out = np.zeros(n)
        self.lib.wco_fft_c2r(n, _p(a), _p(out))
        return out

    def fft_c2c(self, X, sign):
        n = len(X)
        a = np.zeros((n, 2))
        a[:, 0], a[:, 1] = X.real, X.imag
        out = np.zeros((n, 2))
        self.lib.wco_fft_c2c(n, sign, _p(a), _p(out))
        return out[:, 0] + 1j * out[:, 1]

    def minimum_phase(self, log_spectrum, n):
        ls = _c(log_spectrum)
        out = np.zeros((n // 2 + 1, 2))
        self.lib.wco_minimum_phase(n, _p(ls), _p(out))
        return out[:, 0] + 1j * out[:, 1]
