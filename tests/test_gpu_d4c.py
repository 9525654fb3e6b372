"""-m gpu parity tests of the HIP D4C path (through the C-ABI) against the golden vectors from the real
reference and against the CPU oracle on seeded inputs."""
import numpy as np
import pytest

from conftest import PIPELINE_CASES
from world_class_amd.synth import make_utterance

pytestmark = pytest.mark.gpu

# aperiodicity lives in (0, 1]; absolute tolerance (1e-7)
AP_ABS = 1e-7


@pytest.fixture(scope="module")
def wca():
    import world_class_amd as w
    w.lib()
    return w


@pytest.mark.parametrize("name", PIPELINE_CASES)
def test_d4c_golden(golden, wca, port, name):
    c = golden.case(name)
    d = wca.D4C(c["fs"])
    # the reference process had consumed CheapTrick's draws before D4C started
    wca.rng_set_position(port.cheaptrick_draws(c["fs"], c["f0"]))
    ap = d.compute(c["x"], c["tpos"], c["f0"], c["fft_size"])
    s = c["stride"]
    assert np.isfinite(ap).all()
    assert np.abs(ap[::s] - c["ap_rows"]).max() < AP_ABS
    assert np.abs(ap.sum(1) - c["ap_rowsum"]).max() < AP_ABS * ap.shape[1]


def test_d4c_vs_oracle_and_rng_position(wca, port):
    fs = 48000
    x = make_utterance(fs, 0.7, 78)
    tpos, f0 = port.harvest(x, fs)
    port.rng_seek(999)
    ref = port.d4c(x, fs, tpos, f0, 2048)
    end = port.rng_position()
    d = wca.D4C(fs)
    wca.rng_set_position(999)
    ap = d.compute(x, tpos, f0, 2048)
    assert wca.rng_get_position() == end
    assert np.abs(ap - ref).max() < AP_ABS
    # gate decisions identical: rows at the 1 - 1e-12 sentinel are exactly the same rows
    assert np.array_equal(ap[:, 0] == 1.0 - 1e-12, ref[:, 0] == 1.0 - 1e-12)
    port.rng_reset()


def test_d4c_ragged_batch_threshold_and_other_grid(wca, port):
    fs = 16000
    xs = [make_utterance(fs, sec, 700 + i) for i, sec in enumerate((0.4, 1.0, 0.08))]
    tf = [port.harvest(x, fs) for x in xs]
    d = wca.D4C(fs, threshold=0.5)
    start = [0, 77, 123456]
    outs, pos = d.compute_batch(xs, [t for t, _ in tf], [f for _, f in tf], 512, rng_pos=start)
    for x, (t, f), ap, p0, p1 in zip(xs, tf, outs, start, pos):
        port.rng_seek(p0)
        ref = port.d4c(x, fs, t, f, 512, threshold=0.5)
        assert port.rng_position() == p1
        assert np.abs(ap - ref).max() < AP_ABS
    port.rng_reset()


@pytest.mark.parametrize("fs", [8000, 22050, 44100, 64000, 88200, 96000])
def test_d4c_other_rates(wca, port, fs):
    x = make_utterance(fs, 0.3, fs + 1)
    tpos, f0 = port.harvest(x, fs)
    n = port.cheaptrick_fft_size(fs)
    d = wca.D4C(fs)
    wca.rng_set_position(0)
    port.rng_reset()
    assert np.abs(d.compute(x, tpos, f0, n) - port.d4c(x, fs, tpos, f0, n)).max() < AP_ABS
    port.rng_reset()


def test_d4c_edges(wca, port):
    fs = 16000
    d = wca.D4C(fs)
    x = make_utterance(fs, 0.2, 9)
    # all unvoiced -> every row is the sentinel; low f0 is floored at 47 Hz; frames off both ends
    ap = d.compute(x, [0.0, 0.1], [0.0, 0.0], 1024)
    assert np.array_equal(ap, np.full((2, 513), 1.0 - 1e-12))
    tpos = np.array([0.0, 0.05, 0.1995, 0.3])
    f0 = np.array([30.0, 45.0, 200.0, 150.0])
    wca.rng_set_position(0)
    port.rng_reset()
    assert np.abs(d.compute(x, tpos, f0, 1024) - port.d4c(x, fs, tpos, f0, 1024)).max() < AP_ABS
    assert d.compute(x, [], [], 1024).shape == (0, 513)
    port.rng_reset()


@pytest.mark.parametrize("fs", [16000, 24000, 48000])
def test_d4c_band_selection_on_high_words_is_bit_identical(wca, port, fs, monkeypatch):
    """The band kernels need the sum of the K smallest powers of a band (reference src/d4c.cpp:466-503 sorts them) and find the
    threshold by bisecting the keys' bit patterns: since round 6 on the keys' high words first (a 32-bit compare per key and step),
    with 64-bit compares only where neighbours of the ranking share a high word; WC_D4C_SELECT=64 is the search of rounds 3-5.  The K
    smallest keys are the same set whichever threshold separates them, so the rows are the same bit for bit -- on speech, on an
    impulse train (runs of equal powers) and on a signal with digital silence (all powers zero)."""
    from world_class_amd.synth import make_signal
    fft = wca.cheaptrick_fft_size(fs)
    gaps = make_utterance(fs, 1.2, 99)
    gaps[len(gaps) // 3: len(gaps) // 2] = 0.0
    gated = 0
    for x in (make_utterance(fs, 1.0, 4321), make_signal(fs, 1.0, 40003), gaps):
        tpos, f0 = port.harvest(x, fs)
        if not (f0 > 0).any():
            f0 = np.where(np.arange(len(f0)) % 7 < 5, 150.0, 0.0)
        wca.rng_set_position(0)
        a = wca.D4C(fs).compute(x, tpos, f0, fft)
        monkeypatch.setenv("WC_D4C_SELECT", "64")
        d = wca.D4C(fs)
        monkeypatch.delenv("WC_D4C_SELECT")
        wca.rng_set_position(0)
        b = d.compute(x, tpos, f0, fft)
        assert np.array_equal(a, b, equal_nan=True)
        gated += int((a < 0.999).any(axis=1).sum())  # frames that went through the band kernels
    assert gated > 100
    wca.rng_set_position(0)


@pytest.mark.parametrize("fs", [16000, 48000])
def test_d4c_fused_and_split_schedules_agree(wca, port, fs, monkeypatch):
    """workgroup-per-frame kernels (WC_D4C_IMPL=block): frames / band / rows kernels against WC_D4C_SPLIT=0, one fused kernel.
    Same arithmetic, so identical rows (the pruned band FFT of the split kernel at 48 kHz is bit-identical to running all
    passes)."""
    monkeypatch.setenv("WC_D4C_IMPL", "block")
    x = make_utterance(fs, 0.6, 4321)
    tpos, f0 = port.harvest(x, fs)
    fft = wca.cheaptrick_fft_size(fs)
    wca.rng_set_position(0)
    a = wca.D4C(fs).compute(x, tpos, f0, fft)
    monkeypatch.setenv("WC_D4C_SPLIT", "0")
    d = wca.D4C(fs)
    monkeypatch.delenv("WC_D4C_SPLIT")
    wca.rng_set_position(0)
    b = d.compute(x, tpos, f0, fft)
    wca.rng_set_position(0)
    assert np.array_equal(a, b)
    port.rng_reset()
    assert np.abs(a - port.d4c(x, fs, tpos, f0, fft)).max() < AP_ABS
    port.rng_reset()


def test_d4c_threshold_golden(wca):
    """D4COption::threshold away from 0.85 (reference include/d4c.hpp) against the real reference's aperiodicity"""
    from conftest import stage_option_cases
    x, fs, tpos, f0, stride, _, d4 = stage_option_cases()
    for name, thr, rows, rowsum in d4:
        wca.rng_set_position(0)
        ap = wca.D4C(fs, threshold=thr).compute(x, tpos, f0, 1024)
        assert np.abs(ap[::stride] - rows).max() < AP_ABS and np.abs(ap.sum(1) - rowsum).max() < AP_ABS * 1024, name


def test_d4c_two_wavefront_kernels_against_the_block_kernels_and_frames_they_leave_out(wca, port, monkeypatch):
    """48 kHz default: two wavefronts per frame (d4c2_*).  Against the workgroup-per-frame kernels on the same input, and on a
    contour with F0 above what their LDS holds (~1.4 kHz), which the block kernel picks up behind them."""
    fs = 48000
    x = make_utterance(fs, 0.5, 99)
    tpos, f0 = port.harvest(x, fs)
    f0 = f0.copy()
    f0[10:20] = 1800.0   # frames the two-wavefront kernel leaves out
    f0[30:34] = 1300.0   # just inside
    fft = wca.cheaptrick_fft_size(fs)
    wca.rng_set_position(0)
    a = wca.D4C(fs).compute(x, tpos, f0, fft)
    monkeypatch.setenv("WC_D4C_IMPL", "block")
    wca.rng_set_position(0)
    b = wca.D4C(fs).compute(x, tpos, f0, fft)
    port.rng_reset()
    ref = port.d4c(x, fs, tpos, f0, fft)
    port.rng_reset()
    assert np.abs(a - b).max() < AP_ABS
    assert np.abs(a - ref).max() < AP_ABS


@pytest.mark.parametrize("fs,hop", [(16000, 5.0), (24000, 1.0), (22050, 5.0)])
def test_d4c_one_transform_kernels_against_the_block_kernels_and_frames_they_leave_out(wca, port, monkeypatch, fs, hop):
    """N = 2048 (16 / 22.05 / 24 kHz) default: one wavefront per frame and per (frame, band) where a real transform is one
    1024-point complex one (d4c1_*).  Against the workgroup-per-frame kernels on the same input, and on a contour with F0 above
    what their LDS holds (d4c1_can: ~930 Hz at 16 kHz), which they list for the block kernel behind them."""
    x = make_utterance(fs, 0.5, 199)
    tpos, f0 = port.harvest(x, fs, frame_period=hop)
    f0 = f0.copy()
    n = len(f0)
    f0[n // 10:n // 10 + 10] = 1700.0   # frames the kernels leave out
    f0[n // 3:n // 3 + 4] = 880.0       # just inside at 16 kHz
    f0[n // 2:n // 2 + 3] = 48.0        # the longest window (4 fs / 47 samples)
    fft = wca.cheaptrick_fft_size(fs)
    wca.rng_set_position(0)
    a = wca.D4C(fs).compute(x, tpos, f0, fft)
    end = wca.rng_get_position()
    monkeypatch.setenv("WC_D4C_IMPL", "block")
    wca.rng_set_position(0)
    b = wca.D4C(fs).compute(x, tpos, f0, fft)
    assert wca.rng_get_position() == end
    port.rng_reset()
    ref = port.d4c(x, fs, tpos, f0, fft)
    port.rng_reset()
    assert np.abs(a - b).max() < AP_ABS
    assert np.abs(a - ref).max() < AP_ABS
