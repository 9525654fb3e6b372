"""-m gpu: the device building blocks of wc_device.hpp on their own, through the development hooks of wc_testhooks.hip:
the in-LDS Stockham FFT against the transforms of the real reference (fft/* goldens, reference src/world_fft.cpp:31-167)
and the order-faithful cumulative sum of LinearSmoothing (reference src/world_common.cpp:47-51) against a sequential loop,
bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hooks():
    import world_class_amd as w
    L = w.lib()
    dp = C.POINTER(C.c_double)
    L.wc_debug_fft.restype = C.c_int
    L.wc_debug_fft.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp]
    L.wc_debug_seq_cumsum.restype = C.c_int
    L.wc_debug_seq_cumsum.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp]

    class H:
        @staticmethod
        def fft(kind, n, x):
            x = np.ascontiguousarray(x, dtype=np.float64)
            n_in = {0: n, 1: n + 2, 2: 2 * n, 3: 2 * n}[kind]
            n_out = {0: n + 2, 1: n, 2: 2 * n, 3: 2 * n}[kind]
            batch = x.size // n_in
            out = np.empty(batch * n_out)
            rc = L.wc_debug_fft(kind, n, batch, x.ctypes.data_as(dp), out.ctypes.data_as(dp))
            assert rc == 0, w.last_error()
            return out.reshape(batch, n_out)

        @staticmethod
        def cumsum(v, threads):
            v = np.ascontiguousarray(v, dtype=np.float64)
            batch, n = v.shape
            out = np.empty_like(v)
            rc = L.wc_debug_seq_cumsum(v.ctypes.data_as(dp), n, batch, threads, out.ctypes.data_as(dp))
            assert rc == 0, w.last_error()
            return out
    return H


@pytest.mark.parametrize("n", [128, 1024, 2048, 4096])
def test_real_transforms_match_the_reference(golden, hooks, n):
    x = golden[f"fft/r2c_in_{n}"]
    want = golden[f"fft/r2c_out_{n}"]  # [n/2+1][2], e^{+i} convention, from the real reference
    got = hooks.fft(0, n, x).reshape(n // 2 + 1, 2)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() < 1e-13 * scale
    assert got[0, 1] == 0.0 and got[-1, 1] == 0.0
    # c2r of X * (1 + 0.5j): not Hermitian at bins 0 and n/2, whose imaginary parts the reference ignores; unnormalised
    X = (want[:, 0] + 1j * want[:, 1]) * (1 + 0.5j)
    back = hooks.fft(1, n, np.stack([X.real, X.imag], 1))[0]
    want_back = golden[f"fft/c2r_out_{n}"]
    assert np.abs(back - want_back).max() < 1e-13 * np.abs(want_back).max()


def test_complex_transform_matches_the_reference(golden, hooks):
    z = golden["fft/c2c_in_1024"]
    for kind, key in ((2, "fft/c2c_out_1024_sign1"), (3, "fft/c2c_out_1024_sign2")):
        got = hooks.fft(kind, 1024, z).reshape(1024, 2)
        want = golden[key]
        assert np.abs(got - want).max() < 1e-13 * np.abs(want).max(), key


@pytest.mark.parametrize("n", [64, 256, 512, 2048, 4096])
def test_complex_sizes_against_numpy(hooks, n):
    rng = np.random.default_rng(n)
    z = rng.normal(size=(3, n)) + 1j * rng.normal(size=(3, n))
    packed = np.stack([z.real, z.imag], -1)
    fwd = hooks.fft(2, n, packed).reshape(3, n, 2)
    want = np.fft.ifft(z, axis=1) * n  # e^{+i}, unnormalised
    assert np.abs(fwd[..., 0] + 1j * fwd[..., 1] - want).max() < 1e-12 * np.abs(want).max()
    bwd = hooks.fft(3, n, packed).reshape(3, n, 2)
    want = np.fft.fft(z, axis=1)
    assert np.abs(bwd[..., 0] + 1j * bwd[..., 1] - want).max() < 1e-12 * np.abs(want).max()


@pytest.mark.parametrize("n", [256, 512, 8192])
def test_real_sizes_against_numpy(hooks, n):
    rng = np.random.default_rng(n + 1)
    x = rng.normal(size=(2, n))
    X = hooks.fft(0, n, x).reshape(2, n // 2 + 1, 2)
    want = np.conj(np.fft.rfft(x, axis=1))  # e^{+i}
    assert np.abs(X[..., 0] + 1j * X[..., 1] - want).max() < 1e-12 * np.abs(want).max()
    back = hooks.fft(1, n, np.stack([want.real, want.imag], -1))
    assert np.abs(back - n * x).max() < 1e-11 * n


def sequential(v):
    return np.add.accumulate(v, axis=1)  # out[i] = out[i-1] + v[i], strictly left to right


def cumsum_cases(n, rng):
    """sequences that exercise every branch: binade crossings at every scale, exact ties against both parities, terms
    far below and around half an ulp of the running sum, zeros, denormals, huge dynamic range"""
    cases = []
    cases.append(rng.uniform(0, 1, n))                                  # plain
    cases.append(np.exp(rng.uniform(-40, 5, n)))                        # 20 decades
    cases.append(np.full(n, 1.0))                                       # integers: exact, crossings at powers of two
    cases.append(np.full(n, 2.0 ** -53))                                # every add an exact tie against 1.0 once the sum is there
    c = np.full(n, 2.0 ** -53); c[0] = 1.0; cases.append(c)             # ties from the start: round-to-even never moves
    c = np.full(n, 2.0 ** -53); c[0] = 1.0 + 2.0 ** -52; cases.append(c)  # odd start: every tie rounds up
    c = np.full(n, 1.5 * 2.0 ** -53); c[0] = 1.0; cases.append(c)       # above half an ulp
    c = np.full(n, 0.49 * 2.0 ** -53); c[0] = 1.0; cases.append(c)      # below: the sum never moves
    c = rng.integers(0, 8, n) * 2.0 ** -54; c[0] = 1.0; cases.append(c)   # multiples of a quarter ulp: ties everywhere
    c = rng.integers(0, 5, n) * 2.0 ** -53; c[:3] = [0.75, 0.125, 0.0625]; cases.append(c)
    cases.append(np.zeros(n))
    c = np.zeros(n); c[n // 2] = 3.0; cases.append(c)
    cases.append(np.full(n, 5e-324))                                    # denormals
    c = np.exp(rng.uniform(-700, -650, n)); cases.append(c)             # tiny
    c = np.exp(rng.uniform(600, 690, n)); cases.append(c)               # huge
    c = 2.0 ** rng.integers(-30, 30, n).astype(float); cases.append(c)  # powers of two: ties and crossings galore
    # a clean synthetic voice's power spectrum: harmonics 1e12 above the floor in between
    k = np.arange(n)
    c = 1e-18 + np.exp(-0.5 * ((k % 37) - 18.0) ** 2 / 1.5) * 10.0 ** rng.uniform(-3, 3, n); cases.append(c)
    c = np.where(rng.uniform(size=n) < 0.02, 1.0, 1e-17) * rng.uniform(0.5, 1.5, n); cases.append(c)
    return np.stack(cases)


@pytest.mark.parametrize("threads,n", [(256, 1035), (256, 527), (256, 2048), (256, 4096), (256, 7), (256, 256), (256, 257),
                                       (512, 2100), (512, 4096), (512, 300),
                                       (64, 1035), (64, 1150), (64, 2291), (64, 2304), (64, 7), (64, 64), (64, 65), (64, 1153)])
def test_cumulative_sum_is_the_sequential_one_bit_for_bit(hooks, threads, n):
    v = cumsum_cases(n, np.random.default_rng(threads + n))
    got = hooks.cumsum(v, threads)
    want = sequential(v)
    bad = np.argwhere(got != want)
    assert bad.size == 0, (bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])


def test_cumulative_sum_of_many_random_spectra(hooks):
    rng = np.random.default_rng(99)
    v = np.exp(rng.normal(size=(400, 1100)) * rng.uniform(0.1, 8.0, (400, 1))) * 10.0 ** rng.uniform(-20, 10, (400, 1))
    got = hooks.cumsum(v, 256)
    assert np.array_equal(got, sequential(v))
    assert np.array_equal(hooks.cumsum(v, 64), sequential(v))  # the one-wavefront form (seq_cumsum_nonneg_wave)
    # a tree-ordered sum is NOT what the reference computes: the test above would not pass with one
    tree = np.cumsum(v.astype(np.longdouble), axis=1).astype(np.float64)
    assert not np.array_equal(tree, sequential(v))


def signed_cases(n, rng):
    """Terms of either sign: the running sum crosses zero, cancels to nothing against its own history, changes binade in both
    directions, meets ties -- everything the clean / dirty test of seq_cumsum_signed_wave has to sort out."""
    cases = [c * rng.choice([-1.0, 1.0], n) for c in cumsum_cases(n, rng)]
    cases += [-c for c in cumsum_cases(n, rng)[:6]]                                   # all negative: the mirror of the unsigned case
    c = rng.normal(size=n); cases.append(c)                                           # a random walk around zero
    c = rng.normal(size=n) + 0.3; cases.append(c)                                     # drifting away from zero
    c = rng.normal(size=n) * 1e-3; c[0] = 1.0; cases.append(c)                        # small terms on a large sum: one binade
    c = rng.normal(size=n) * 1e-3; c[0] = 1.0; c[n // 2] = -1.0; cases.append(c)      # ... which then cancels
    c = rng.normal(size=n); c[1::2] = -c[0::2][: len(c[1::2])]; cases.append(c)       # exact cancellation every other term
    c = (rng.integers(-4, 5, n) * 2.0 ** -52); c[0] = 1.0; cases.append(c)            # ties against an odd / even sum
    c = (rng.integers(-4, 5, n) * 2.0 ** -53); c[0] = -1.0; cases.append(c)
    c = np.sin(np.arange(n) * 0.05) * np.exp(rng.normal(size=n)); cases.append(c)     # slowly alternating sign
    c = rng.normal(size=n) * 10.0 ** rng.uniform(-12, 3, n); cases.append(c)          # twelve decades
    return np.stack(cases)


@pytest.mark.parametrize("n", [1035, 1150, 2291, 2304, 7, 64, 65, 1153, 1100, 1029])
def test_signed_cumulative_sum_is_the_sequential_one_bit_for_bit(hooks, n):
    """seq_cumsum_signed_wave (D4C's group-delay smoothing, reference src/d4c.cpp:440-460 through src/world_common.cpp:82-116)."""
    v = signed_cases(n, np.random.default_rng(7000 + n))
    got = hooks.cumsum(v, -64)
    want = sequential(v)
    bad = np.argwhere(got.view(np.int64) != want.view(np.int64))
    assert bad.size == 0, (bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])


def test_signed_cumulative_sum_of_many_random_group_delays(hooks):
    rng = np.random.default_rng(98)
    # shaped like the numerator D4C smooths: a signed spectrum, strong near the harmonics, of any overall scale
    k = np.arange(1100)
    v = rng.normal(size=(600, 1100)) * np.exp(2.0 * np.cos(k * rng.uniform(0.05, 0.6, (600, 1)))) * 10.0 ** rng.uniform(-20, 10, (600, 1))
    v += rng.uniform(-0.5, 0.5, (600, 1)) * np.abs(v).mean(axis=1, keepdims=True)
    got = hooks.cumsum(v, -64)
    assert np.array_equal(got, sequential(v))
    tree = np.cumsum(v.astype(np.longdouble), axis=1).astype(np.float64)
    assert not np.array_equal(tree, sequential(v))


@pytest.mark.parametrize("c", [8000.0, 1000.0, 7350.0, 11025.0, 16000.0, 4000.0, 12000.0, 5512.5, 9600.0])
def test_division_by_a_known_divisor_is_the_ieee_quotient_bit_for_bit(c):
    """div_const (wc_device.hpp): x / c as a product with the correctly rounded reciprocal, the exact remainder and one correction --
    what hv_raw's interval midpoints (reference src/harvest.cpp:1210-1213: (e[k] + e[k + 1]) / 2 / fs) and frame times (i / 1000.0)
    are divided with since round 6.  Held to the IEEE quotient on 2^24 numerators per divisor: random significands over thirty
    binades, the integers the frame times are made of, half-integers like the edge sums, significands next to a power of two."""
    import world_class_amd as w
    L = w.lib()
    dp = C.POINTER(C.c_double)
    L.wc_debug_div_const.restype = C.c_int
    L.wc_debug_div_const.argtypes = [C.c_longlong, dp, C.c_double, dp]
    rng = np.random.default_rng(int(c))
    n = 1 << 22
    parts = [rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-6, 24, n),
             np.arange(n, dtype=np.float64),
             rng.integers(0, 1 << 30, n) / 2.0 + rng.integers(0, 2, n) * 2.0 ** -20,
             (1.0 + rng.integers(-8, 9, n) * 2.0 ** -52) * 2.0 ** rng.integers(0, 20, n) * rng.choice([1.0, c, c / 3.0, 1000.0], n)]
    x = np.ascontiguousarray(np.concatenate(parts))
    out = np.empty_like(x)
    assert L.wc_debug_div_const(x.size, x.ctypes.data_as(dp), c, out.ctypes.data_as(dp)) == 0
    want = x / c
    bad = np.flatnonzero(out.view(np.int64) != want.view(np.int64))
    assert bad.size == 0, (bad.size, x[bad[:5]], out[bad[:5]], want[bad[:5]])
