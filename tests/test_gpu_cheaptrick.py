"""-m gpu parity tests of the HIP CheapTrick path (through the C-ABI) against the golden vectors from the
real reference and against the CPU oracle on seeded inputs."""
import numpy as np
import pytest

from conftest import PIPELINE_CASES
from world_class_amd.synth import make_utterance

pytestmark = pytest.mark.gpu

# FP64 parity bar for the spectral envelope: relative, per bin (SURVEY.md section 8(c) proposes 1e-7;
# the measured noise floor of block-scan vs sequential prefix sums is far below that on these inputs)
SP_REL = 1e-7


@pytest.fixture(scope="module")
def wca():
    import world_class_amd as w
    w.lib()
    return w


def rel(a, b):
    return (np.abs(a - b) / np.abs(b)).max()


@pytest.mark.parametrize("name", PIPELINE_CASES)
def test_cheaptrick_golden(golden, wca, name):
    c = golden.case(name)
    ct = wca.CheapTrick(c["fs"])
    assert ct.fft_size == c["fft_size"]
    wca.rng_set_position(0)
    sp = ct.compute(c["x"], c["tpos"], c["f0"])
    s = c["stride"]
    assert np.isfinite(sp).all()
    assert rel(sp[::s], c["sp_rows"]) < SP_REL
    assert rel(sp.sum(1), c["sp_rowsum"]) < SP_REL


def test_cheaptrick_vs_oracle_and_rng_position(golden, wca, port):
    fs = 48000
    x = make_utterance(fs, 0.7, 77)
    tpos, f0 = port.harvest(x, fs)
    port.rng_seek(12345)
    ref = port.cheaptrick(x, fs, tpos, f0)
    end = port.rng_position()
    ct = wca.CheapTrick(fs)
    wca.rng_set_position(12345)
    sp = ct.compute(x, tpos, f0)
    assert wca.rng_get_position() == end
    assert rel(sp, ref) < SP_REL
    port.rng_reset()


def test_cheaptrick_ragged_batch(wca, port):
    fs = 16000
    xs = [make_utterance(fs, sec, 900 + i) for i, sec in enumerate((0.3, 1.0, 0.05, 0.6))]
    tf = [port.harvest(x, fs) for x in xs]
    ct = wca.CheapTrick(fs)
    outs, pos = ct.compute_batch(xs, [t for t, _ in tf], [f for _, f in tf], rng_pos=[0, 5, 0, 1000])
    for x, (t, f), sp, p0, p1 in zip(xs, tf, outs, [0, 5, 0, 1000], pos):
        port.rng_seek(p0)
        ref = port.cheaptrick(x, fs, t, f)
        assert port.rng_position() == p1
        assert rel(sp, ref) < SP_REL
    port.rng_reset()


def test_cheaptrick_edges(wca, port):
    fs = 16000
    ct = wca.CheapTrick(fs)
    # all-unvoiced contour, arbitrary temporal positions, very short signal (shorter than one window)
    x = make_utterance(fs, 0.01, 5)
    tpos = np.array([0.0, 0.004, 0.0099, 0.5])
    f0 = np.zeros(4)
    wca.rng_set_position(0)
    sp = ct.compute(x, tpos, f0)
    port.rng_reset()
    ref = port.cheaptrick(x, fs, tpos, f0)
    assert rel(sp, ref) < SP_REL
    # one frame, f0 just above / below the CheapTrick floor
    x = make_utterance(fs, 0.2, 6)
    for f in (47.0, 47.1, 799.0):
        wca.rng_set_position(0)
        port.rng_reset()
        assert rel(ct.compute(x, [0.1], [f]), port.cheaptrick(x, fs, [0.1], [f])) < SP_REL
    # empty contour is a no-op
    assert ct.compute(x, [], []).shape == (0, ct.bins)
    port.rng_reset()


@pytest.mark.parametrize("fs", [8000, 22050, 44100, 64000, 96000])
def test_cheaptrick_other_rates(wca, port, fs):
    x = make_utterance(fs, 0.3, fs)
    tpos, f0 = port.harvest(x, fs)
    ct = wca.CheapTrick(fs)
    wca.rng_set_position(0)
    port.rng_reset()
    assert rel(ct.compute(x, tpos, f0), port.cheaptrick(x, fs, tpos, f0)) < SP_REL
    port.rng_reset()


def test_fails_loudly_on_bad_arguments(wca):
    with pytest.raises(wca.WorldClassError):
        wca.CheapTrick(48000, fft_size=1000)
    ct = wca.CheapTrick(16000)
    with pytest.raises(wca.WorldClassError):
        ct.compute(np.zeros(0), [0.0], [100.0])


def test_cheaptrick_options_golden(wca):
    """q1, f0_floor and fft_size away from their defaults (reference include/cheaptrick.hpp) against the real reference's
    envelopes (tests/golden/stage_options.npz): FFT sizes 512, 1024 (floor 40 -> 2048), 2048 and 4096 at 16 kHz"""
    from conftest import stage_option_cases
    x, fs, tpos, f0, stride, ct, _ = stage_option_cases()
    for name, kw, rows, rowsum in ct:
        wca.rng_set_position(0)
        sp = wca.CheapTrick(fs, **kw).compute(x, tpos, f0)
        assert sp.shape[1] == rows.shape[1], name
        assert rel(sp[::stride], rows) < SP_REL and rel(sp.sum(1), rowsum) < SP_REL, name


def test_cheaptrick_one_wavefront_kernel_against_the_block_kernel_and_frames_it_leaves_out(wca, port, monkeypatch):
    """48 kHz default: one wavefront per frame (ct_wave_kernel).  Against the workgroup-per-frame kernel on the same input, and
    on a contour with F0 above what its LDS holds (~2 kHz), which the block kernel picks up behind it."""
    fs = 48000
    x = make_utterance(fs, 0.5, 98)
    tpos, f0 = port.harvest(x, fs)
    f0 = f0.copy()
    f0[10:20] = 2500.0
    f0[30:34] = 1900.0
    wca.rng_set_position(777)
    a = wca.CheapTrick(fs).compute(x, tpos, f0)
    end = wca.rng_get_position()
    monkeypatch.setenv("WC_CT_IMPL", "block")
    wca.rng_set_position(777)
    b = wca.CheapTrick(fs).compute(x, tpos, f0)
    assert wca.rng_get_position() == end
    port.rng_seek(777)
    ref = port.cheaptrick(x, fs, tpos, f0)
    port.rng_reset()
    assert rel(a, b) < SP_REL
    assert rel(a, ref) < SP_REL


@pytest.mark.parametrize("fs,hop", [(16000, 5.0), (24000, 1.0), (22050, 5.0)])
def test_cheaptrick_eight_points_per_lane_kernel_against_the_block_kernel_and_frames_it_leaves_out(wca, port, monkeypatch, fs, hop):
    """N = 1024 (16 / 22.05 / 24 kHz) default: one wavefront per frame at eight points per lane (ct_wave8_kernel, transforms
    wf8_* of wc_wavefft.hpp).  Against the workgroup-per-frame kernel on the same input, and on a contour with F0 above what its
    LDS holds (ct_wave_can<1024>), which it lists for the block kernel behind it."""
    x = make_utterance(fs, 0.5, 198)
    tpos, f0 = port.harvest(x, fs, frame_period=hop)
    f0 = f0.copy()
    n = len(f0)
    f0[n // 10:n // 10 + 10] = 2300.0   # frames the kernel leaves out
    f0[n // 3:n // 3 + 4] = 1250.0      # just inside (16 kHz: the limit is 1382 Hz)
    assert wca.cheaptrick_fft_size(fs) == 1024
    wca.rng_set_position(555)
    a = wca.CheapTrick(fs).compute(x, tpos, f0)
    end = wca.rng_get_position()
    monkeypatch.setenv("WC_CT_IMPL", "block")
    wca.rng_set_position(555)
    b = wca.CheapTrick(fs).compute(x, tpos, f0)
    assert wca.rng_get_position() == end
    port.rng_seek(555)
    ref = port.cheaptrick(x, fs, tpos, f0)
    port.rng_reset()
    assert rel(a, b) < SP_REL
    assert rel(a, ref) < SP_REL


def test_cheaptrick_split_kernel_against_the_sixteen_points_kernel(wca, port, monkeypatch):
    """N = 2048 on eight points per lane (ct_wave_split_kernel, WC_CT_IMPL=split: the forward transform as two 512-point halves one
    after the other, everything in one 9 KB buffer of LDS, three wavefronts per SIMD) against the default one-wavefront kernel and
    the oracle: windows of every pruning class (F0 at the floor: 2029 samples; 150 Hz: 961; 400 Hz: 361; unvoiced: 289), the frames
    either kernel leaves to the block kernel, and the bins lane 0 holds (multiples of 64)."""
    fs = 48000
    x = make_utterance(fs, 0.6, 398)
    tpos, f0 = port.harvest(x, fs)
    f0 = f0.copy()
    n = len(f0)
    f0[5:15] = 72.0       # just above CheapTrick's floor (70.4 Hz): the longest window
    f0[20:30] = 150.0
    f0[35:45] = 400.0
    f0[50:54] = 1900.0    # just inside ct_wave_can<2048>
    f0[60:66] = 2500.0    # left to the block kernel
    f0[70:80] = 0.0
    assert n > 100
    wca.rng_set_position(4321)
    a = wca.CheapTrick(fs).compute(x, tpos, f0)
    end = wca.rng_get_position()
    monkeypatch.setenv("WC_CT_IMPL", "split")
    wca.rng_set_position(4321)
    b = wca.CheapTrick(fs).compute(x, tpos, f0)
    assert wca.rng_get_position() == end
    monkeypatch.delenv("WC_CT_IMPL")
    port.rng_seek(4321)
    ref = port.cheaptrick(x, fs, tpos, f0)
    port.rng_reset()
    worst = np.abs(b / a - 1.0).max(axis=0)
    assert rel(b, a) < 1e-10, "bins %s" % np.argsort(worst)[-5:]
    assert rel(b, ref) < SP_REL
    # through the fused pipeline too (the handle of a group reads the same switch)
    monkeypatch.setenv("WC_CT_IMPL", "split")
    r2 = wca.Pipeline(fs).run_batch([x, x[:20000]])
    monkeypatch.delenv("WC_CT_IMPL")
    r1 = wca.Pipeline(fs).run_batch([x, x[:20000]])
    for u in range(2):
        assert np.array_equal(r1[u]["f0"], r2[u]["f0"]) and rel(r2[u]["sp"], r1[u]["sp"]) < 1e-10
