"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/_ref/libworld_ref{,_omp}.so (the real
reference compiled by oracle/Makefile).  Used by tests/, oracle/gen_golden.py and bench.py's
cpu_baseline leg; never by the product path.

The reference's randn() keeps process-global static state (reference
src/world_matlabfunctions.cpp:243-264), so anything that must start from the seed state has to run
in a fresh process: see run_fresh().
"""
import ctypes as C
import os
import pickle
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def lib_path(omp=False):
    """omp: False serial (deterministic), True the OpenMP build, "fma" the serial build under -mfma -ffp-contract=fast"""
    name = "libworld_ref_fma.so" if omp == "fma" else "libworld_ref_omp.so" if omp else "libworld_ref.so"
    return os.path.join(_HERE, "_ref", name)


def available(omp=False):
    return os.path.exists(lib_path(omp))


def _p(a):
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Ref:
    def __init__(self, omp=False):
        self.lib = C.CDLL(lib_path(omp))
        L = self.lib
        L.ref_get_samples.restype = C.c_int
        L.ref_get_samples.argtypes = [C.c_int, C.c_int, C.c_double]
        L.ref_cheaptrick_fft_size.restype = C.c_int
        L.ref_cheaptrick_fft_size.argtypes = [C.c_int, C.c_double]
        L.ref_cheaptrick_f0_floor.restype = C.c_double
        L.ref_cheaptrick_f0_floor.argtypes = [C.c_int, C.c_int]
        L.ref_harvest.argtypes = [_dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp]
        L.ref_cheaptrick.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_double,
                                     C.c_double, C.c_int, _dp]
        L.ref_d4c.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double, _dp]
        L.ref_synthesis.argtypes = [_dp, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_int, _dp]
        L.ref_randn.argtypes = [C.c_int, _dp]
        L.ref_matlab_round.restype = C.c_int
        L.ref_matlab_round.argtypes = [C.c_double]
        L.ref_suitable_fft_size.restype = C.c_int
        L.ref_suitable_fft_size.argtypes = [C.c_int]
        L.ref_interp1.argtypes = [_dp, _dp, C.c_int, _dp, C.c_int, _dp]
        L.ref_interp1Q.argtypes = [C.c_double, C.c_double, _dp, C.c_int, _dp, C.c_int, _dp]
        L.ref_histc.argtypes = [_dp, C.c_int, _dp, C.c_int, _ip]
        L.ref_decimate.argtypes = [_dp, C.c_int, C.c_int, _dp]
        L.ref_dc_correction.argtypes = [_dp, C.c_double, C.c_int, C.c_int, _dp]
        L.ref_linear_smoothing.argtypes = [_dp, C.c_double, C.c_int, C.c_int, _dp]
        L.ref_nuttall.argtypes = [C.c_int, _dp]
        L.ref_fft_r2c.argtypes = [C.c_int, _dp, _dp]
        L.ref_fft_c2r.argtypes = [C.c_int, _dp, _dp]
        L.ref_fft_c2c.argtypes = [C.c_int, C.c_int, _dp, _dp]
        L.ref_minimum_phase.argtypes = [C.c_int, _dp, _dp]

    # ---- stages -------------------------------------------------------------------------
    def get_samples(self, fs, n, frame_period=5.0):
        return self.lib.ref_get_samples(fs, n, frame_period)

    def harvest(self, x, fs, f0_floor=71.0, f0_ceil=800.0, frame_period=5.0):
        x = _c(x)
        L = self.get_samples(fs, len(x), frame_period)
        tpos = np.zeros(L)
        f0 = np.zeros(L)
        self.lib.ref_harvest(_p(x), len(x), fs, f0_floor, f0_ceil, frame_period, _p(tpos), _p(f0))
        return tpos, f0

    def harvest_opt(self, x, fs, f0_floor=71.0, f0_ceil=800.0, frame_period=5.0, target_fs=8000.0, channels_in_octave=40.0,
                    use_cos_table=False):
        """every field of HarvestOption (reference include/harvest.hpp:16-24)"""
        self.lib.ref_harvest_opt.argtypes = [_dp, C.c_int, C.c_int] + [C.c_double] * 5 + [C.c_int, _dp, _dp]
        x = _c(x)
        L = self.get_samples(fs, len(x), frame_period)
        tpos, f0 = np.zeros(L), np.zeros(L)
        self.lib.ref_harvest_opt(_p(x), len(x), fs, f0_floor, f0_ceil, frame_period, target_fs, channels_in_octave,
                                 int(use_cos_table), _p(tpos), _p(f0))
        return tpos, f0

    def cheaptrick_fft_size(self, fs, f0_floor=71.0):
        return self.lib.ref_cheaptrick_fft_size(fs, f0_floor)

    def cheaptrick(self, x, fs, tpos, f0, q1=-0.15, f0_floor=71.0, fft_size=0):
        x, tpos, f0 = _c(x), _c(tpos), _c(f0)
        nfft = fft_size or self.cheaptrick_fft_size(fs, f0_floor)
        sp = np.zeros((len(f0), nfft // 2 + 1))
        self.lib.ref_cheaptrick(_p(x), len(x), fs, _p(tpos), _p(f0), len(f0), q1, f0_floor,
                                fft_size, _p(sp))
        return sp

    def d4c(self, x, fs, tpos, f0, fft_size, threshold=0.85):
        x, tpos, f0 = _c(x), _c(tpos), _c(f0)
        ap = np.zeros((len(f0), fft_size // 2 + 1))
        self.lib.ref_d4c(_p(x), len(x), fs, _p(tpos), _p(f0), len(f0), fft_size, threshold, _p(ap))
        return ap

    def synthesis(self, f0, sp, ap, fs, frame_period=5.0, out_length=None):
        f0, sp, ap = _c(f0), _c(sp), _c(ap)
        fft_size = (sp.shape[1] - 1) * 2
        if out_length is None:
            out_length = int((len(f0) - 1) * frame_period / 1000.0 * fs) + 1  # test/test.cpp:362
        y = np.zeros(out_length)
        self.lib.ref_synthesis(_p(f0), len(f0), _p(sp), _p(ap), fft_size, fs, frame_period,
                               out_length, _p(y))
        return y

    # ---- helpers ------------------------------------------------------------------------
    def randn(self, n):
        out = np.zeros(n)
        self.lib.ref_randn(n, _p(out))
        return out

    def matlab_round(self, x):
        return self.lib.ref_matlab_round(float(x))

    def suitable_fft_size(self, n):
        return self.lib.ref_suitable_fft_size(int(n))

    def interp1(self, x, y, xi):
        x, y, xi = _c(x), _c(y), _c(xi)
        yi = np.zeros(len(xi))
        self.lib.ref_interp1(_p(x), _p(y), len(x), _p(xi), len(xi), _p(yi))
        return yi

    def interp1Q(self, x0, dx, y, xi):
        y, xi = _c(y), _c(xi)
        yi = np.zeros(len(xi))
        self.lib.ref_interp1Q(x0, dx, _p(y), len(y), _p(xi), len(xi), _p(yi))
        return yi

    def histc(self, x, edges):
        x, edges = _c(x), _c(edges)
        idx = np.zeros(len(edges), dtype=np.int32)
        self.lib.ref_histc(_p(x), len(x), _p(edges), len(edges), idx.ctypes.data_as(_ip))
        return idx

    def decimate(self, x, r):
        x = _c(x)
        y = np.zeros(len(x) + 32)  # the reference writes n/r + ceil(9/r) values
        self.lib.ref_decimate(_p(x), len(x), r, _p(y))
        return y[:len(x) // r + 1].copy()

    def dc_correction(self, spec, f0, fs, fft_size):
        spec = _c(spec)
        out = spec.copy()
        self.lib.ref_dc_correction(_p(spec), f0, fs, fft_size, _p(out))
        return out

    def linear_smoothing(self, spec, width, fs, fft_size):
        spec = _c(spec)
        out = np.zeros(fft_size // 2 + 1)
        self.lib.ref_linear_smoothing(_p(spec), width, fs, fft_size, _p(out))
        return out

    def nuttall(self, n):
        y = np.zeros(n)
        self.lib.ref_nuttall(n, _p(y))
        return y

    def fft_r2c(self, x):
        x = _c(x)
        out = np.zeros((len(x) // 2 + 1, 2))
        self.lib.ref_fft_r2c(len(x), _p(x), _p(out))
        return out[:, 0] + 1j * out[:, 1]

    def fft_c2r(self, X, n):
        a = np.zeros((n // 2 + 1, 2))
        a[:, 0], a[:, 1] = X.real, X.imag
        out = np.zeros(n)
        self.lib.ref_fft_c2r(n, _p(a), _p(out))
        return out

    def fft_c2c(self, X, sign):
        n = len(X)
        a = np.zeros((n, 2))
        a[:, 0], a[:, 1] = X.real, X.imag
        out = np.zeros((n, 2))
        self.lib.ref_fft_c2c(n, sign, _p(a), _p(out))
        return out[:, 0] + 1j * out[:, 1]

    def minimum_phase(self, log_spectrum, n):
        ls = _c(log_spectrum)
        out = np.zeros((n // 2 + 1, 2))
        self.lib.ref_minimum_phase(n, _p(ls), _p(out))
        return out[:, 0] + 1j * out[:, 1]

    # ---- whole pipeline, demo order (reference test/test.cpp:288-384) ----------------------
    def pipeline(self, x, fs, harvest_floor=71.0, frame_period=5.0, given_f0=None):
        if given_f0 is None:
            tpos, f0 = self.harvest(x, fs, f0_floor=harvest_floor, frame_period=frame_period)
        else:
            tpos, f0 = given_f0
        sp = self.cheaptrick(x, fs, tpos, f0)
        fft_size = (sp.shape[1] - 1) * 2
        ap = self.d4c(x, fs, tpos, f0, fft_size)
        y = self.synthesis(f0, sp, ap, fs, frame_period)
        return dict(tpos=tpos, f0=f0, sp=sp, ap=ap, y=y)


def _synthesis_behind_analysis(ref, x, fs, tpos, f0_own, f0, sp, ap, frame_period=5.0):
    """Synthesis of given parameters from where the noise stream stands behind CheapTrick and D4C on (x, tpos, f0_own) -- the place
    the reference's own pipeline reaches its Synthesis at (tests: a stage check that an upstream rounding cannot cascade into)"""
    sp_own = ref.cheaptrick(x, fs, tpos, f0_own)
    ref.d4c(x, fs, tpos, f0_own, (sp_own.shape[1] - 1) * 2)
    return ref.synthesis(f0, sp, ap, fs, frame_period)


def _at(ref, start, method, *args, **kwargs):
    """Ref().<method> with the process-global noise stream `start` draws from its seed state"""
    done = 0
    while done < start:
        n = min(1 << 20, start - done)
        ref.randn(n)
        done += n
    return getattr(ref, method)(*args, **kwargs)


Ref.synthesis_behind_analysis = _synthesis_behind_analysis
Ref.at = _at


def _pipeline_timed(ref, xs, fs, harvest_floor=71.0, frame_period=5.0):
    """bench.py's cpu_baseline leg: the demo-order pipeline over the utterances `xs` one after the other in THIS process; returns
    (frames, wall-clock stamp before the first call, stamp after the last) -- stamps of time.time(), comparable across the
    processes of one host, taken inside the process so that interpreter start-up and result pickling stay outside"""
    import time
    frames = 0
    t0 = time.time()
    for x in xs:
        frames += len(ref.pipeline(x, fs, harvest_floor=harvest_floor, frame_period=frame_period)["f0"])
    return frames, t0, time.time()


Ref.pipeline_timed = _pipeline_timed


def taps_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libworld_ref_taps.so"))


def harvest_taps(x, fs, f0_floor=71.0, f0_ceil=800.0):
    """Intermediates of the real reference's Harvest (oracle/ref_harvest_taps.cpp): y, raw [band][frame], candidates and
    scores after refinement and after removeUnreliableCandidates [frame][max_candidates], base / fixed / smoothed 1 ms
    contours.  Harvest draws no random numbers, so no fresh process is needed."""
    lib = C.CDLL(os.path.join(_HERE, "_ref", "libworld_ref_taps.so"))
    lib.ref_harvest_taps.restype = C.c_int
    lib.ref_harvest_taps.argtypes = [_dp, C.c_int, C.c_int, C.c_double, C.c_double, _ip] + [_dp] * 9
    x = _c(x)
    dims = np.zeros(4, dtype=np.int32)
    nul = [None] * 9
    L1 = lib.ref_harvest_taps(_p(x), len(x), fs, f0_floor, f0_ceil, dims.ctypes.data_as(_ip), *nul)
    yl, nb, mc, nc = [int(v) for v in dims]
    out = dict(y=np.zeros(yl), raw=np.zeros((nb, L1)), cand_refined=np.zeros((L1, mc)), score_refined=np.zeros((L1, mc)),
               cand=np.zeros((L1, mc)), score=np.zeros((L1, mc)), f0_base=np.zeros(L1), f0_fixed=np.zeros(L1), f0_1ms=np.zeros(L1))
    lib.ref_harvest_taps(_p(x), len(x), fs, f0_floor, f0_ceil, dims.ctypes.data_as(_ip), *[_p(out[k]) for k in
                         ("y", "raw", "cand_refined", "score_refined", "cand", "score", "f0_base", "f0_fixed", "f0_1ms")])
    out["n_cand"] = nc
    return out


def run_fresh(method, *args, omp=False, **kwargs):
    """Run Ref().<method>(*args, **kwargs) in a brand-new process (RNG at its seed state)."""
    payload = pickle.dumps((method, args, kwargs, omp))
    out = subprocess.run([sys.executable, os.path.abspath(__file__)], input=payload,
                         stdout=subprocess.PIPE, check=True)
    return pickle.loads(out.stdout)


if __name__ == "__main__":
    method, args, kwargs, omp = pickle.loads(sys.stdin.buffer.read())
    res = getattr(Ref(omp), method)(*args, **kwargs)
    sys.stdout.buffer.write(pickle.dumps(res))
