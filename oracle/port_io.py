"""TEST INFRASTRUCTURE ONLY.  numpy restatement of the demo's parameter modification (reference test/test.cpp:201-243)
and of the sample conversions of the reference's WAV tools (tools/audioio.cpp:155-158, :237-250).  Pinned to the real
reference by tests/test_io_formats.py through tests/golden/io/io_golden.npz."""
import numpy as np


def interp1(x, y, xi):
    """reference src/world_matlabfunctions.cpp:157-182 with histc (:136-155): k = clamp(#{x[j] <= xi}, 1, n-1),
    linear interpolation / extrapolation on the segment [x[k-1], x[k]]"""
    n = len(x)
    k = np.clip(np.searchsorted(x, xi, side="right"), 1, n - 1)
    h = x[k] - x[k - 1]
    s = (xi - x[k - 1]) / h
    return y[k - 1] + s * (y[k] - y[k - 1])


def parameter_modification(fs, fft_size, f0, sp, f0_scale=None, ratio=None):
    """returns (f0, sp) like reference test/test.cpp:201-243 (f0_scale / ratio None = argument absent)"""
    f0 = np.array(f0, dtype=np.float64)
    sp = np.array(sp, dtype=np.float64)
    if f0_scale is not None:
        f0 = f0 * f0_scale
    if ratio is None:
        return f0, sp
    bins = fft_size // 2 + 1
    j = np.arange(bins, dtype=np.float64)
    axis1 = ratio * j / fft_size * fs
    axis2 = j / fft_size * fs
    out = np.empty_like(sp)
    cut = int(fft_size / 2.0 * ratio)
    for i in range(sp.shape[0]):
        with np.errstate(over="ignore"):  # bins extrapolated past the stretched axis overflow before the fill below
            row = np.exp(interp1(axis1, np.log(sp[i]), axis2))
        if ratio < 1.0:
            row[cut:] = row[cut - 1]
        out[i] = row
    return f0, out


def pcm16_of(x):
    """wavwrite's quantisation: clamp(int(x * 32767)) with C truncation (x86: NaN / out of range -> INT_MIN)"""
    v = np.asarray(x, dtype=np.float64) * 32767
    bad = ~((v > -2147483649.0) & (v < 2147483648.0))
    iv = np.where(bad, -2147483648, np.trunc(np.where(bad, 0.0, v))).astype(np.int64)
    return np.clip(iv, -32768, 32767).astype(np.int16)
