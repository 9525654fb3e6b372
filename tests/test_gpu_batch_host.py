"""-m gpu: the host-pointer BATCH entry points of the four stages (include/world_class_c.h: wc_harvest_compute_batch,
wc_cheaptrick_compute_batch, wc_d4c_compute_batch, wc_synthesis_compute_batch) -- separate host arrays per
utterance and one row pointer per frame, as a caller of the reference holds them (reference include/cheaptrick.hpp:30-33,
synthesis.hpp:45-49) -- against the same utterances through the single-utterance calls, and against the CPU oracle."""
import numpy as np
import pytest

from world_class_amd.synth import make_utterance

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wca():
    import world_class_amd as w
    w.lib()
    return w


def test_stage_batches_from_host_arrays_equal_the_single_calls(wca, port):
    fs = 16000
    xs = [make_utterance(fs, sec, 900 + i) for i, sec in enumerate((0.8, 0.35, 1.2, 0.5))]
    hv, ct, d4 = wca.Harvest(fs), wca.CheapTrick(fs), wca.D4C(fs)
    sy = wca.Synthesis(fs, ct.fft_size, 5.0)
    cont = hv.compute_batch(xs)
    for x, (t, f) in zip(xs, cont):
        t1, f1 = hv.compute(x)
        assert np.array_equal(t, t1) and np.array_equal(f, f1)
        to, fo = port.harvest(x, fs)
        assert np.array_equal(f == 0, fo == 0) and np.abs(f - fo).max() < 1e-6
    ts, fs_ = [c[0] for c in cont], [c[1] for c in cont]
    start = [0, 17, 0, 123456]
    sps, pos = ct.compute_batch(xs, ts, fs_, rng_pos=start)
    aps, pos2 = d4.compute_batch(xs, ts, fs_, ct.fft_size, rng_pos=pos)
    ys, pos3 = sy.compute_batch(fs_, sps, aps, rng_pos=pos2)
    for u, x in enumerate(xs):
        wca.rng_set_position(start[u])
        sp = ct.compute(x, ts[u], fs_[u])
        assert wca.rng_get_position() == pos[u]
        ap = d4.compute(x, ts[u], fs_[u], ct.fft_size)
        assert wca.rng_get_position() == pos2[u]
        y = sy.compute(fs_[u], sp, ap)
        assert wca.rng_get_position() == pos3[u]
        assert np.array_equal(sp, sps[u]) and np.array_equal(ap, aps[u])
        assert np.array_equal(y, ys[u])  # (response rows summed in pulse order: the same bits in a batch as alone)
        port.rng_seek(start[u])
        assert (np.abs(sps[u] - port.cheaptrick(x, fs, ts[u], fs_[u])) / sps[u]).max() < 1e-7
        port.rng_reset()
    # rows that do not lie one behind the other (every row an allocation of its own), and NULL positions = fresh process each
    rows = [[np.empty(ct.bins) for _ in range(len(f))] for f in fs_]
    import ctypes as C
    dp = C.POINTER(C.c_double)
    tabs = [(dp * len(r))(*[v.ctypes.data_as(dp) for v in r]) for r in rows]
    tab = (C.c_void_p * len(xs))(*[C.cast(t, C.c_void_p) for t in tabs])
    pa = lambda arrs: (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    xl, fl = (C.c_int * len(xs))(*[len(x) for x in xs]), (C.c_int * len(xs))(*[len(f) for f in fs_])
    assert wca.lib().wc_cheaptrick_compute_batch(ct._h, len(xs), pa(xs), xl, pa(ts), pa(fs_), fl, tab, None) == 0, wca.last_error()
    for u, x in enumerate(xs):
        wca.rng_set_position(0)
        assert np.array_equal(np.stack(rows[u]), ct.compute(x, ts[u], fs_[u]))
    assert wca.lib().wc_cheaptrick_compute_batch(ct._h, len(xs), pa(xs), xl, pa(ts), pa(fs_), fl, None, None) != 0


def test_result_buffers_of_the_python_mirror_are_checked(wca):
    fs = 16000
    x = make_utterance(fs, 0.3, 5)
    t, f = wca.Harvest(fs).compute(x)
    ct = wca.CheapTrick(fs)
    good = np.empty((len(f), ct.bins))
    ct.compute(x, t, f, out=good)
    for bad in (np.empty((len(f), ct.bins), dtype=np.float32), np.empty((len(f) - 1, ct.bins)), np.empty((ct.bins, len(f))).T, [[0.0]]):
        with pytest.raises(ValueError):
            ct.compute(x, t, f, out=bad)
        with pytest.raises(ValueError):
            wca.D4C(fs).compute(x, t, f, ct.fft_size, out=bad)
    sy = wca.Synthesis(fs, ct.fft_size, 5.0)
    ap = wca.D4C(fs).compute(x, t, f, ct.fft_size)
    n = sy.out_length(len(f))
    with pytest.raises(ValueError):
        sy.compute(f, good, ap, out_length=n, out=np.zeros(n - 1))
    with pytest.raises(ValueError):
        sy.compute(f, good, ap, out=np.zeros(n, dtype=np.float32))
    with pytest.raises(ValueError):
        sy.compute(f, good[:-1], ap)
