"""-m gpu: the one-wavefront transforms of wc_wavefft.hpp (2048-point real FFTs held in registers, two LDS exchanges, no
barrier) against the real reference's transforms (fft/* goldens, reference src/world_fft.cpp:31-167) and numpy, and the
lean log / exp the frame kernels use instead of libm's against numpy."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hooks():
    import world_class_amd as w
    L = w.lib()
    dp = C.POINTER(C.c_double)
    L.wc_debug_wave_fft.restype = C.c_int
    L.wc_debug_wave_fft.argtypes = [C.c_int, C.c_int, dp, dp]
    L.wc_debug_logexp.restype = C.c_int
    L.wc_debug_logexp.argtypes = [C.c_int, C.c_longlong, dp, dp]

    class H:
        @staticmethod
        def fft(kind, x):
            x = np.ascontiguousarray(x, dtype=np.float64)
            n_in, n_out = {0: (2048, 2050), 1: (2050, 2048), 2: (2048, 2050), 3: (4096, 4098),
                           4: (1024, 1026), 5: (1026, 1024), 6: (1024, 1026), 7: (1024, 1026), 8: (1025, 1025)}[kind]
            batch = x.size // n_in
            out = np.empty(batch * n_out)
            rc = L.wc_debug_wave_fft(kind, batch, x.ctypes.data_as(dp), out.ctypes.data_as(dp))
            assert rc == 0, w.last_error()
            return out.reshape(batch, n_out)

        @staticmethod
        def logexp(kind, x):
            x = np.ascontiguousarray(x, dtype=np.float64)
            out = np.empty_like(x)
            rc = L.wc_debug_logexp(kind, x.size, x.ctypes.data_as(dp), out.ctypes.data_as(dp))
            assert rc == 0, w.last_error()
            return out
    return H


def test_wave_transforms_match_the_reference(golden, hooks):
    n = 2048
    x = golden[f"fft/r2c_in_{n}"]
    want = golden[f"fft/r2c_out_{n}"]
    got = hooks.fft(0, x).reshape(n // 2 + 1, 2)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() < 1e-13 * scale
    assert got[0, 1] == 0.0 and got[-1, 1] == 0.0
    X = (want[:, 0] + 1j * want[:, 1]) * (1 + 0.5j)
    back = hooks.fft(1, np.stack([X.real, X.imag], 1))[0]
    want_back = golden[f"fft/c2r_out_{n}"]
    assert np.abs(back - want_back).max() < 1e-13 * np.abs(want_back).max()


def test_wave_transforms_batched_against_numpy(hooks):
    rng = np.random.default_rng(7)
    batch, n = 37, 2048
    x = rng.standard_normal((batch, n))
    got = hooks.fft(0, x).reshape(batch, n // 2 + 1, 2)
    want = np.conj(np.fft.rfft(x, axis=1))  # the reference's r2c is the conjugate of the textbook transform
    assert np.abs(got[..., 0] + 1j * got[..., 1] - want).max() < 1e-12
    Y = rng.standard_normal((batch, n // 2 + 1)) + 1j * rng.standard_normal((batch, n // 2 + 1))
    back = hooks.fft(1, np.stack([Y.real, Y.imag], 2))
    Yh = Y.copy()
    Yh[:, 0] = Yh[:, 0].real
    Yh[:, -1] = Yh[:, -1].real
    want_back = np.fft.irfft(np.conj(Yh), n=n, axis=1) * n
    assert np.abs(back - want_back).max() < 1e-12 * np.abs(want_back).max()
    # the pruned leading stage: input zero beyond its first quarter
    xz = x.copy()
    xz[:, n // 4:] = 0.0
    got = hooks.fft(2, xz).reshape(batch, n // 2 + 1, 2)
    want = np.conj(np.fft.rfft(xz, axis=1))
    assert np.abs(got[..., 0] + 1j * got[..., 1] - want).max() < 1e-12


def test_eight_points_per_lane_transforms_match_the_reference(golden, hooks):
    """the 1024-point real transforms of CheapTrick / Synthesis at 16 and 24 kHz: 512 complex points, eight per lane (wf8_*)"""
    n = 1024
    x = golden[f"fft/r2c_in_{n}"]
    want = golden[f"fft/r2c_out_{n}"]
    got = hooks.fft(4, x).reshape(n // 2 + 1, 2)
    assert np.abs(got - want).max() < 1e-13 * np.abs(want).max()
    assert got[0, 1] == 0.0 and got[-1, 1] == 0.0
    X = (want[:, 0] + 1j * want[:, 1]) * (1 + 0.5j)
    back = hooks.fft(5, np.stack([X.real, X.imag], 1))[0]
    want_back = golden[f"fft/c2r_out_{n}"]
    assert np.abs(back - want_back).max() < 1e-13 * np.abs(want_back).max()


def test_eight_points_per_lane_transforms_batched_against_numpy(hooks):
    rng = np.random.default_rng(17)
    batch, n = 41, 1024
    x = rng.standard_normal((batch, n))
    got = hooks.fft(4, x).reshape(batch, n // 2 + 1, 2)
    want = np.conj(np.fft.rfft(x, axis=1))
    assert np.abs(got[..., 0] + 1j * got[..., 1] - want).max() < 1e-12
    Y = rng.standard_normal((batch, n // 2 + 1)) + 1j * rng.standard_normal((batch, n // 2 + 1))
    back = hooks.fft(5, np.stack([Y.real, Y.imag], 2))
    Yh = Y.copy()
    Yh[:, 0] = Yh[:, 0].real
    Yh[:, -1] = Yh[:, -1].real
    want_back = np.fft.irfft(np.conj(Yh), n=n, axis=1) * n
    assert np.abs(back - want_back).max() < 1e-12 * np.abs(want_back).max()
    for kind, live in ((6, n // 4), (7, n // 2)):  # the pruned leading stages
        xz = x.copy()
        xz[:, live:] = 0.0
        got = hooks.fft(kind, xz).reshape(batch, n // 2 + 1, 2)
        want = np.conj(np.fft.rfft(xz, axis=1))
        assert np.abs(got[..., 0] + 1j * got[..., 1] - want).max() < 1e-12


def test_real_even_transform_at_half_cost(hooks):
    """wf_even2048: the DFT of a real even sequence of 2048 points (CheapTrick's cepstral transforms, reference
    src/cheaptrick.cpp:230-276; the first transform of MinimumPhaseAnalysis, src/world_common.cpp:196-205) through a 512-point
    complex transform -- against numpy on log-spectrum-like and on random inputs"""
    rng = np.random.default_rng(23)
    batch = 29
    k = np.arange(1025)
    xs = [rng.standard_normal((batch, 1025)),
          -12.0 + 4.0 * np.cos(2 * np.pi * k / 137.0)[None, :] + rng.standard_normal((batch, 1025)),   # a log spectrum: large mean
          np.exp(-0.01 * k)[None, :] * rng.standard_normal((batch, 1025))]                                 # a cepstrum: decaying
    for x in xs:
        full = np.concatenate([x, x[:, -2:0:-1]], axis=1)
        assert full.shape[1] == 2048
        want = np.fft.fft(full, axis=1).real[:, :1025]
        got = hooks.fft(8, x)
        assert np.abs(got - want).max() < 2e-13 * np.abs(want).max()


def test_two_wavefront_transform_of_4096_points(golden, hooks):
    n = 4096
    got = hooks.fft(3, golden[f"fft/r2c_in_{n}"]).reshape(n // 2 + 1, 2)
    want = golden[f"fft/r2c_out_{n}"]
    assert np.abs(got - want).max() < 1e-13 * np.abs(want).max()
    rng = np.random.default_rng(3)
    x = rng.standard_normal((19, n))
    got = hooks.fft(3, x).reshape(19, n // 2 + 1, 2)
    assert np.abs(got[..., 0] + 1j * got[..., 1] - np.conj(np.fft.rfft(x, axis=1))).max() < 2e-12


def test_lean_log_and_exp(hooks):
    rng = np.random.default_rng(11)
    x = np.concatenate([10.0 ** rng.uniform(-300, 300, 200000), rng.uniform(0.5, 2.0, 200000), 1.0 + rng.uniform(-1e-3, 1e-3, 50000),
                        [1.0, 0.5, 2.0, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308]])
    got = hooks.logexp(0, x)
    want = np.log(x)
    assert np.abs(got - want).max() < 4e-16 * np.maximum(1.0, np.abs(want)).max()
    assert (np.abs(got - want) <= 4e-16 * np.maximum(1.0, np.abs(want))).all()
    special = hooks.logexp(0, np.array([0.0, -1.0, np.inf, np.nan]))
    assert special[0] == -np.inf and np.isnan(special[1]) and special[2] == np.inf and np.isnan(special[3])
    y = np.concatenate([rng.uniform(-700, 700, 300000), rng.uniform(-1, 1, 100000), [0.0, -745.0, 709.7, -800.0, 800.0]])
    got = hooks.logexp(1, y)
    with np.errstate(over="ignore"):
        want = np.exp(y)
    ok = np.isfinite(want) & (want > 1e-300)
    assert (np.abs(got[ok] - want[ok]) <= 4e-16 * want[ok]).all()
    assert got[-1] == np.inf and got[-2] == 0.0
    assert np.isnan(hooks.logexp(1, np.array([np.nan])))[0]
