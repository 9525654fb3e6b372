"""Seeded synthetic utterances for tests and bench.

Voiced/unvoiced harmonic-plus-noise signals quantised through int16 exactly like the reference's
WAV reader scales samples (reference tools/audioio.cpp:237-250: value / 2^(nbit-1)).
numpy's default_rng (PCG64) is stable across numpy versions, so seeds pin the waveforms.
"""
import numpy as np


def make_utterance(fs: int, seconds: float, seed: int) -> np.ndarray:
    """Return float64 samples in [-1, 1) of one synthetic utterance."""
    rng = np.random.default_rng(seed)
    n = int(round(fs * seconds))
    t = np.arange(n, dtype=np.float64) / fs
    f_c = rng.uniform(120.0, 300.0)
    phi = rng.uniform(0.0, 2.0 * np.pi)
    psi = rng.uniform(0.0, 2.0 * np.pi)
    f0 = f_c + 40.0 * np.sin(2.0 * np.pi * 0.7 * t + phi)
    phase = 2.0 * np.pi * np.cumsum(f0) / fs
    x = np.zeros(n, dtype=np.float64)
    for h in range(1, 30):
        ph_h = rng.uniform(0.0, 2.0 * np.pi)
        keep = (h * f0) <= 0.45 * fs
        x += keep * np.sin(h * phase + ph_h) / h
    gate = (np.sin(2.0 * np.pi * 1.5 * t + psi) > -0.3).astype(np.float64)
    box = np.ones(200, dtype=np.float64) / 200.0
    gate = np.convolve(gate, box, mode="full")[99:99 + n]  # == mode="same" for n >= 200
    x *= gate
    peak = np.max(np.abs(x))
    if peak > 0:
        x *= 0.5 / peak
    x += rng.normal(0.0, 0.01, n)
    x = np.clip(x, -0.99, 0.99)
    q = np.round(x * 32768.0).astype(np.int64)
    q = np.clip(q, -32768, 32767)
    return q.astype(np.float64) / 32768.0


SIGNAL_KINDS = ("noise", "chirp", "jumps", "impulses", "duet", "square_dc", "quiet", "loud", "gaps", "low_jitter")


def make_signal(fs, seconds, seed):
    """Float64 test signals of kinds make_utterance never produces (not quantised, not clipped): seed % 10 picks the kind
    from SIGNAL_KINDS.  Used by tests/parity_sweep.py --zoo and by the long-utterance F0 fixtures."""
    rng = np.random.default_rng(seed)
    n = int(round(fs * seconds))
    t = np.arange(n) / fs
    kind = SIGNAL_KINDS[seed % len(SIGNAL_KINDS)]

    def voice(f0, nh=25, roll=1.0):
        ph = 2 * np.pi * np.cumsum(f0) / fs
        x = np.zeros(n)
        for h in range(1, nh + 1):
            x += ((h * f0) < 0.45 * fs) * np.sin(h * ph + rng.uniform(0, 2 * np.pi)) / h ** roll
        return x / max(np.abs(x).max(), 1e-9)

    if kind == "noise":
        x = 0.3 * rng.normal(size=n)
    elif kind == "chirp":
        f = 40.0 + (1200.0 - 40.0) * t / seconds
        x = 0.6 * np.sin(2 * np.pi * np.cumsum(f) / fs)
    elif kind == "jumps":
        seg = int(0.15 * fs)
        f0 = np.repeat(rng.uniform(60.0, 700.0, n // seg + 1), seg)[:n]
        x = 0.5 * voice(f0) * np.repeat(rng.uniform(size=n // seg + 1) > 0.25, seg)[:n]
    elif kind == "impulses":
        period = int(rng.integers(fs // 400, fs // 80))
        x = np.zeros(n)
        x[::period] = 0.9
    elif kind == "duet":
        x = 0.35 * voice(np.full(n, rng.uniform(90, 200))) + 0.35 * voice(rng.uniform(210, 500) * (1 + 0.03 * np.sin(2 * np.pi * 5 * t)))
    elif kind == "square_dc":
        x = 0.2 + 0.5 * np.sign(voice(150 + 30 * np.sin(2 * np.pi * 2 * t), nh=1))
    elif kind == "quiet":
        x = 1e-5 * voice(np.full(n, rng.uniform(100, 300))) + 1e-7 * rng.normal(size=n)
    elif kind == "loud":
        x = 3.0 * voice(200 + 50 * np.sin(2 * np.pi * 1.1 * t)) + 0.5 * rng.normal(size=n)
    elif kind == "gaps":
        x = 0.5 * voice(np.full(n, rng.uniform(100, 400)), roll=0.5) * (np.sin(2 * np.pi * 1.3 * t) > 0)  # exact zeros in the gaps
    else:
        f0 = rng.uniform(45, 90) * (1 + 0.02 * rng.normal(size=n))
        x = 0.5 * voice(f0, nh=40)
    return np.ascontiguousarray(x, dtype=np.float64)


SIGNAL_KINDS2 = ("speechlike", "clipped", "stairs", "quantised_tone", "am", "fractional_impulses", "dc", "soprano", "bass", "decay")


def make_signal2(fs, seconds, seed):
    """A second set of float64 test signals (round 5, tests/parity_sweep.py --zoo2): seed % 10 picks the kind from SIGNAL_KINDS2.
    Formant-filtered glottal pulses with jitter, shimmer and fricatives; flat-topped and coarsely quantised waveforms (runs of
    equal samples: first differences that are exactly zero); levels that fall in steps or smoothly by twelve decades (the
    sliding band-pass hands such stretches to direct sums); band-limited impulses with a fractional period; F0 at the edges."""
    rng = np.random.default_rng(seed)
    n = int(round(fs * seconds))
    t = np.arange(n) / fs
    kind = SIGNAL_KINDS2[seed % len(SIGNAL_KINDS2)]

    def voice(f0, nh=25, roll=1.0):
        ph = 2 * np.pi * np.cumsum(f0) / fs
        x = np.zeros(n)
        for h in range(1, nh + 1):
            x += ((h * f0) < 0.45 * fs) * np.sin(h * ph + rng.uniform(0, 2 * np.pi)) / h ** roll
        return x / max(np.abs(x).max(), 1e-9)

    def resonate(x, fc, bw):  # a two-pole resonator, plain recursion (no scipy: the generator must not depend on its version)
        r = np.exp(-np.pi * bw / fs)
        a1, a2 = -2 * r * np.cos(2 * np.pi * fc / fs), r * r
        y = np.zeros(len(x))
        y1 = y2 = 0.0
        for i, v in enumerate(x):
            y0 = v - a1 * y1 - a2 * y2
            y[i] = y0
            y2, y1 = y1, y0
        return y

    if kind == "speechlike":
        f0 = rng.uniform(90, 260) * (1 + 0.15 * np.sin(2 * np.pi * rng.uniform(0.3, 1.2) * t + rng.uniform(0, 6))) * (1 + 0.01 * rng.normal(size=n))
        ph = np.cumsum(f0) / fs
        pulses = np.diff(np.floor(ph), prepend=0.0) * (1 + 0.1 * rng.normal(size=n))  # one sample per period, shimmer
        src = pulses - 0.97 * np.concatenate([[0.0], pulses[:-1]])
        seg = int(0.12 * fs)
        state = rng.integers(0, 4, n // seg + 2)  # 0 silence, 1 fricative, 2-3 voiced
        gate_v = np.repeat(state >= 2, seg)[:n].astype(float)
        gate_f = np.repeat(state == 1, seg)[:n].astype(float)
        box = np.ones(int(0.01 * fs)) / int(0.01 * fs)
        gate_v, gate_f = np.convolve(gate_v, box, "same"), np.convolve(gate_f, box, "same")
        x = src * gate_v
        for fc, bw in ((rng.uniform(300, 800), 80.0), (rng.uniform(900, 2300), 120.0), (rng.uniform(2400, 3400), 200.0)):
            if fc < 0.45 * fs:
                x = resonate(x, fc, bw) * (1 - np.exp(-np.pi * bw / fs))
        x = 0.6 * x / max(np.abs(x).max(), 1e-9) + 0.05 * gate_f * rng.normal(size=n) + 1e-4 * rng.normal(size=n)
        x = np.clip(np.round(x * 32768.0), -32768, 32767) / 32768.0
    elif kind == "clipped":
        x = np.clip(1.5 * voice(rng.uniform(100, 300) * (1 + 0.05 * np.sin(2 * np.pi * 0.9 * t))), -0.5, 0.5)
    elif kind == "stairs":
        levels = np.array([1.0, 1e-3, 1e-6, 1e-9, 1e-12, 1e-8, 1e-7, 1.0, 3e-9, 1.0])
        x = 0.5 * voice(np.full(n, rng.uniform(100, 350))) * np.repeat(levels, n // len(levels) + 1)[:n]
    elif kind == "quantised_tone":
        x = np.round(rng.uniform(8, 40) * np.sin(2 * np.pi * rng.uniform(110, 330) * t)) / 32768.0
    elif kind == "am":
        x = 0.5 * voice(np.full(n, rng.uniform(100, 300))) * (0.5 + 0.5 * np.cos(2 * np.pi * 3.0 * t))
    elif kind == "fractional_impulses":
        period = rng.uniform(fs / 400.0, fs / 80.0)
        x = np.zeros(n)
        k = np.arange(-16, 17)
        for c in np.arange(20.0, n - 20.0, period):
            i0 = int(c)
            x[i0 + k] += 0.8 * np.sinc(i0 + k - c) * np.hanning(len(k) + 2)[1:-1]
    elif kind == "dc":
        x = 0.3 + 1e-3 * voice(np.full(n, rng.uniform(100, 300)))
    elif kind == "soprano":
        x = 0.5 * voice(rng.uniform(600, 760) * (1 + 0.03 * np.sin(2 * np.pi * 5.5 * t)), nh=10) + 1e-3 * rng.normal(size=n)
    elif kind == "bass":
        x = 0.5 * voice(rng.uniform(42, 70) * (1 + 0.02 * np.sin(2 * np.pi * 0.5 * t)), nh=60) + 1e-3 * rng.normal(size=n)
    else:
        x = 0.5 * voice(np.full(n, rng.uniform(100, 300))) * 10.0 ** (-12.0 * t / max(seconds, 1e-9))
    return np.ascontiguousarray(x, dtype=np.float64)


def make_batch(fs: int, seconds: float, n_utt: int, config: int = 0, first: int = 0):
    """List of utterances with seeds 1000*config + u (u = first .. first+n_utt-1)."""
    return [make_utterance(fs, seconds, 1000 * config + u) for u in range(first, first + n_utt)]


def true_f0(fs: int, seconds: float, seed: int, frame_period: float = 5.0):
    """(temporal_positions, f0) of the generator's own contour on the analysis grid (0 where the
    voicing gate is closed).  Used by the stage micro-benchmarks, which need a plausible contour
    without running F0 estimation first."""
    rng = np.random.default_rng(seed)
    n = int(round(fs * seconds))
    f_c = rng.uniform(120.0, 300.0)
    phi = rng.uniform(0.0, 2.0 * np.pi)
    psi = rng.uniform(0.0, 2.0 * np.pi)
    n_frames = int(1000.0 * n / fs / frame_period) + 1
    tpos = np.arange(n_frames) * frame_period / 1000.0
    f0 = f_c + 40.0 * np.sin(2.0 * np.pi * 0.7 * tpos + phi)
    voiced = np.sin(2.0 * np.pi * 1.5 * tpos + psi) > -0.3
    return tpos, np.where(voiced, f0, 0.0)
