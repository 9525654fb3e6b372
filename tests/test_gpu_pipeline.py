"""-m gpu: the whole hot path Harvest -> CheapTrick -> D4C -> Synthesis on the device, in the demo order of
reference test/test.cpp:288-384, against the golden outputs of the real reference (noise stream included),
plus size-independent properties at the benchmark's full utterance size."""
import numpy as np
import pytest

from conftest import PIPELINE_CASES, check_headline, headline_case
from world_class_amd.synth import make_utterance, true_f0

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wca():
    import world_class_amd as w
    w.lib()
    return w


def run_pipeline(wca, x, fs, floor=71.0, fp=5.0):
    wca.rng_set_position(0)
    tpos, f0 = wca.Harvest(fs, f0_floor=floor, frame_period=fp).compute(x)
    ct = wca.CheapTrick(fs)
    sp = ct.compute(x, tpos, f0)
    ap = wca.D4C(fs).compute(x, tpos, f0, ct.fft_size)
    y = wca.Synthesis(fs, ct.fft_size, fp).compute(f0, sp, ap)
    return tpos, f0, sp, ap, y


@pytest.mark.parametrize("name", PIPELINE_CASES)
def test_pipeline_golden(golden, wca, name):
    c = golden.case(name)
    tpos, f0, sp, ap, y = run_pipeline(wca, c["x"], c["fs"], c["harvest_floor"], c["frame_period"])
    s = c["stride"]
    assert np.array_equal(f0 == 0, c["f0"] == 0)
    assert np.abs(f0 - c["f0"]).max() < 1e-6
    assert (np.abs(sp[::s] - c["sp_rows"]) / c["sp_rows"]).max() < 1e-7
    assert np.abs(ap[::s] - c["ap_rows"]).max() < 1e-7
    assert np.abs(y - c["y"]).max() < 1e-8


def test_device_batch_equals_single_calls(wca):
    """Packed device API on a ragged batch == the host API utterance by utterance (fresh-process RNG each)."""
    fs = 16000
    xs = [make_utterance(fs, sec, 50 + i) for i, sec in enumerate((0.7, 0.3, 1.1))]
    hv, ct, d4, = wca.Harvest(fs), wca.CheapTrick(fs), wca.D4C(fs)
    sy = wca.Synthesis(fs, ct.fft_size, 5.0)
    tf = hv.compute_batch(xs)
    t, f = [a for a, _ in tf], [b for _, b in tf]
    sps, pos = ct.compute_batch(xs, t, f, rng_pos=[0] * 3)
    aps, pos = d4.compute_batch(xs, t, f, ct.fft_size, rng_pos=pos)
    ys, pos = sy.compute_batch(f, sps, aps, rng_pos=pos)
    for x, fb, spb, apb, yb in zip(xs, f, sps, aps, ys):
        _, f0, sp, ap, y = run_pipeline(wca, x, fs)
        assert np.array_equal(fb, f0) and np.array_equal(spb, sp) and np.array_equal(apb, ap)
        assert np.abs(yb - y).max() < 1e-12  # overlap-add order is the only non-determinism


def test_full_size_properties(wca):
    """48 kHz, 10 s (the benchmark's utterance size): structural checks that need no oracle run."""
    fs = 48000
    x = make_utterance(fs, 10.0, 3000)
    tpos, f0, sp, ap, y = run_pipeline(wca, x, fs)
    assert len(f0) == 2001 and sp.shape == (2001, 1025) and ap.shape == (2001, 1025) and len(y) == 480001
    assert np.isfinite(sp).all() and (sp > 0).all()
    assert ((ap > 0) & (ap <= 1.0)).all()
    assert np.array_equal(ap[f0 == 0], np.full_like(ap[f0 == 0], 1.0 - 1e-12))  # unvoiced rows are the sentinel
    v = f0 > 0
    assert (f0[v] >= 71.0).all() and (f0[v] <= 800.0).all()
    # the estimator recovers the generator's own contour where both call the frame voiced
    tt, tf = true_f0(fs, 10.0, 3000)
    both = v & (tf > 0)
    assert both.mean() > 0.5 and np.median(np.abs(f0[both] - tf[both]) / tf[both]) < 0.01
    # analysis of a time-shifted copy is the shifted analysis (shift = 2 hops) away from the edges
    x2 = np.concatenate([np.zeros(480) + x[0], x])[:len(x)]
    _, f02, sp2, _, _ = run_pipeline(wca, x2, fs)
    inner = slice(300, 1700)
    a, b = f0[inner], f02[302:1702]
    same = (a > 0) & (b > 0)
    assert same.mean() > 0.5 and np.abs(a[same] - b[same]).max() < 0.5
    # resynthesis has the energy of the input (same order of magnitude) and is finite
    assert np.isfinite(y).all() and 0.3 < np.sqrt(np.mean(y ** 2)) / np.sqrt(np.mean(x ** 2)) < 3.0


@pytest.mark.parametrize("name", ["c1_16k_2s_floor71", "m48k_1s"])
def test_fused_pipeline_golden(golden, wca, name):
    """wc_pipeline_run_device (multi-stream, device-chained noise positions) against the reference goldens."""
    c = golden.case(name)
    p = wca.Pipeline(c["fs"], frame_period=c["frame_period"], harvest_f0_floor=c["harvest_floor"])
    (r,), pos = p.run_batch([c["x"]], rng_pos=[0])
    s = c["stride"]
    assert np.array_equal(r["tpos"], c["tpos"])
    assert np.array_equal(r["f0"] == 0, c["f0"] == 0) and np.abs(r["f0"] - c["f0"]).max() < 1e-6
    assert (np.abs(r["sp"][::s] - c["sp_rows"]) / c["sp_rows"]).max() < 1e-7
    assert np.abs(r["ap"][::s] - c["ap_rows"]).max() < 1e-7
    assert np.abs(r["y"] - c["y"]).max() < 1e-8
    assert pos[0] > 0


def test_fused_pipeline_ragged_batch_equals_stage_calls(wca):
    fs = 16000
    xs = [make_utterance(fs, sec, 80 + i) for i, sec in enumerate((0.6, 1.0, 0.25))]
    p = wca.Pipeline(fs)
    outs, pos = p.run_batch(xs, rng_pos=[0, 7, 0])
    for x, r, p0 in zip(xs, outs, [0, 7, 0]):
        wca.rng_set_position(p0)
        tpos, f0 = wca.Harvest(fs).compute(x)
        ct = wca.CheapTrick(fs)
        sp = ct.compute(x, tpos, f0)
        ap = wca.D4C(fs).compute(x, tpos, f0, ct.fft_size)
        y = wca.Synthesis(fs, ct.fft_size, 5.0).compute(f0, sp, ap)
        assert np.array_equal(r["f0"], f0) and np.array_equal(r["sp"], sp) and np.array_equal(r["ap"], ap)
        assert np.abs(r["y"] - y).max() < 1e-12
    assert all(b > a for a, b in zip([0, 7, 0], pos))


def test_host_batch_front_end_pcm16_and_double(wca):
    """wc_pipeline_run_batch_host: ragged batch from host pointers through pinned staging; int16 PCM in (expanded on the
    device like wavread) and out (quantised like wavwrite) gives exactly what the device-resident path gives"""
    from oracle import port_io
    fs = 16000
    xs = [make_utterance(fs, sec, seed) for sec, seed in ((0.5, 31), (0.9, 32), (0.35, 33))]
    pcm = [port_io.pcm16_of(x * 32768.0 / 32767.0) for x in xs]   # any int16 samples will do
    xq = [p.astype(np.float64) / 32768.0 for p in pcm]
    p = wca.Pipeline(fs)
    ref = p.run_batch(xq)
    got = p.run_batch_host(pcm, y_pcm16=True)
    for r, g in zip(ref, got):
        assert np.array_equal(g["tpos"], r["tpos"]) and np.array_equal(g["f0"], r["f0"])
        assert np.array_equal(g["sp"], r["sp"]) and np.array_equal(g["ap"], r["ap"])
        assert g["y"].dtype == np.int16 and np.abs(g["y"].astype(np.int64) - port_io.pcm16_of(r["y"])).max() <= 1
    got = p.run_batch_host(xq, want=("f0", "y"))
    for r, g in zip(ref, got):
        assert set(g) == {"f0", "y"} and np.array_equal(g["f0"], r["f0"])
        assert np.abs(g["y"] - r["y"]).max() < 1e-10   # overlap-add order differs between runs at the 1e-16 level
    # 32-bit float samples: widened on the device, exactly
    x32 = [v.astype(np.float32) for v in xq]
    ref32 = p.run_batch([v.astype(np.float64) for v in x32])
    got = p.run_batch_host(x32, want=("f0", "sp"))
    for r, g in zip(ref32, got):
        assert np.array_equal(g["f0"], r["f0"]) and np.array_equal(g["sp"], r["sp"])
    # the caller's own buffers written again (out=), larger batch so that the half-batch copies overlap the second half
    many = [xq[i % 3] for i in range(10)]
    ref = p.run_batch(many)
    first = p.run_batch_host(many)
    for g in first:
        for v in g.values():
            v.fill(-1.0)
    again = p.run_batch_host(many, out=first)
    for r, g, h in zip(ref, first, again):
        assert all(g[k] is h[k] for k in g)
        assert np.array_equal(g["tpos"], r["tpos"]) and np.array_equal(g["f0"], r["f0"])
        assert np.array_equal(g["sp"], r["sp"]) and np.array_equal(g["ap"], r["ap"])
        assert np.abs(g["y"] - r["y"]).max() < 1e-10


def test_host_batch_front_end_pinned_rows_and_coded_outputs(wca):
    """wc_pipeline_run_batch_host with the caller's rows in page-locked memory (written by the copy engine directly, no staging)
    gives the same rows; wc_pipeline_run_batch_host_coded returns what the reference's codec (src/codec.cpp:211-325) makes of
    those rows: mel-cepstral coefficients and band aperiodicities, checked against the CPU restatement of the codec."""
    from oracle import port_codec as pc
    fs = 48000
    xs = [make_utterance(fs, sec, seed) for sec, seed in ((0.4, 41), (0.7, 42), (0.3, 43), (0.55, 44))]
    p = wca.Pipeline(fs)
    ref = p.run_batch(xs)
    pinned = p.host_buffers([len(x) for x in xs], pinned=True)
    for g in pinned:
        for v in g.values():
            v.fill(-1.0)
    got = p.run_batch_host(xs, out=pinned)
    for r, g, h in zip(ref, pinned, got):
        assert all(g[k] is h[k] for k in g)
        assert np.array_equal(g["tpos"], r["tpos"]) and np.array_equal(g["f0"], r["f0"])
        assert np.array_equal(g["sp"], r["sp"]) and np.array_equal(g["ap"], r["ap"])
        assert np.abs(g["y"] - r["y"]).max() < 1e-10
    nd = 60
    coded = p.run_batch_host_coded(xs, number_of_dimensions=nd)
    for r, c in zip(ref, coded):
        assert np.array_equal(c["f0"], r["f0"]) and np.abs(c["y"] - r["y"]).max() < 1e-10
        assert c["csp"].shape == (len(r["f0"]), nd)
        assert np.abs(c["csp"] - pc.code_spectral_envelope(r["sp"], fs, p.fft_size, nd)).max() < 1e-10
        assert np.abs(c["cap"] - pc.code_aperiodicity(r["ap"], fs, p.fft_size)).max() < 1e-10
    # page-locked utterances are read by the copy engine where they lie, half batch by half batch (no gather into staging);
    # page-locked waveform rows are written per half behind its pulses: the same numbers either way
    xs_pinned = p.host_inputs(xs)
    for g in pinned:
        for v in g.values():
            v.fill(-2.0)
    p.run_batch_host(xs_pinned, out=pinned)
    for r, g in zip(ref, pinned):
        assert np.array_equal(g["tpos"], r["tpos"]) and np.array_equal(g["f0"], r["f0"])
        assert np.array_equal(g["sp"], r["sp"]) and np.array_equal(g["ap"], r["ap"])
        assert np.abs(g["y"] - r["y"]).max() < 1e-10
    cbuf = p.coded_host_buffers([len(x) for x in xs], number_of_dimensions=nd, pinned=True)
    p.run_batch_host_coded(xs_pinned, number_of_dimensions=nd, out=cbuf)
    for c, b in zip(coded, cbuf):
        assert all(np.array_equal(c[k], b[k]) for k in ("f0", "csp", "cap"))
        assert np.abs(c["y"] - b["y"]).max() < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("splits", ["10,13,18,25", "20,30", "6,8,11,15,21", "50"])
def test_host_batch_front_end_in_more_than_two_groups(wca, splits, monkeypatch):
    """wc_pipeline_run_batch_host cuts a batch whose rows leave for the host into up to six groups of growing size (round 4; groups
    beyond the second share the two main streams and the high-priority stream): every grouping returns what the device-resident
    batch gives, with the caller's rows page-locked (DMA into place, per-group uploads) and pageable (staging + host scatter)."""
    fs = 16000
    base = [make_utterance(fs, sec, seed) for sec, seed in ((0.4, 51), (0.65, 52), (0.3, 53), (0.5, 54), (0.8, 55))]
    xs = [base[i % 5] for i in range(22)]
    p = wca.Pipeline(fs)
    ref = p.run_batch(xs)
    p.set_option("host_splits", splits)  # (WC_PIPELINE_HOST_SPLITS is read when a handle is created; a run reads no environment)
    for pinned in (True, False):
        out = p.host_buffers([len(x) for x in xs], pinned=pinned)
        for g in out:
            for v in g.values():
                v.fill(-3.0)
        p.run_batch_host(p.host_inputs(xs) if pinned else xs, out=out)
        for r, g in zip(ref, out):
            assert np.array_equal(g["tpos"], r["tpos"]) and np.array_equal(g["f0"], r["f0"])
            assert np.array_equal(g["sp"], r["sp"]) and np.array_equal(g["ap"], r["ap"])
            assert np.abs(g["y"] - r["y"]).max() < 1e-10


def test_pipeline_at_96_khz_golden(wca):
    """96 kHz: decimation ratio 12, 4096-point CheapTrick / Synthesis, 8192-point D4C and LoveTrain (the unpacking twiddles of
    the 8192-point real transforms lie between the entries of the 4096-entry table and are computed) against the real
    reference (tests/golden/rate96k.npz)"""
    from conftest import rate96k_case
    x, fs, stride, z = rate96k_case()
    wca.rng_set_position(0)
    (r,) = wca.Pipeline(fs).run_batch([x])
    assert np.array_equal(r["tpos"], z["tpos"]) and np.array_equal(r["f0"] == 0, z["f0"] == 0)
    assert np.abs(r["f0"] - z["f0"]).max() < 1e-6
    assert (np.abs(r["sp"][::stride] - z["sp_rows"]) / z["sp_rows"]).max() < 1e-7
    assert (np.abs(r["sp"].sum(axis=1) - z["sp_rowsum"]) / z["sp_rowsum"]).max() < 1e-7
    assert np.abs(r["ap"][::stride] - z["ap_rows"]).max() < 1e-7
    assert np.abs(r["ap"].sum(axis=1) - z["ap_rowsum"]).max() < 1e-7 * r["ap"].shape[1]
    assert np.abs(r["y"] - z["y"]).max() < 1e-8


def test_schedules_agree(wca):
    """the default schedule (two half-batch chains) and WC_PIPELINE_MODE=shared (one set of stage handles, Harvest split
    over streams): same kernels on the same data, so the same parameters bit for bit and the same waveform up to the
    order of the overlap-add"""
    import os
    fs = 16000
    xs = [make_utterance(fs, sec, 90 + i) for i, sec in enumerate((0.6, 1.0, 0.25, 0.8, 0.5))]
    a = wca.Pipeline(fs).run_batch(xs)
    os.environ["WC_PIPELINE_MODE"] = "shared"
    try:
        p = wca.Pipeline(fs)
    finally:
        del os.environ["WC_PIPELINE_MODE"]
    b = p.run_batch(xs)
    for ra, rb in zip(a, b):
        assert np.array_equal(ra["f0"], rb["f0"]) and np.array_equal(ra["sp"], rb["sp"]) and np.array_equal(ra["ap"], rb["ap"])
        assert np.abs(ra["y"] - rb["y"]).max() < 1e-12


def test_resident_batch_side_by_side_and_held_apart_agree(wca, monkeypatch):
    """A device-resident batch of less than 500 s of signal runs its two groups' full-grid kernels side by side, a larger one holds
    them apart by events (round 5: 16 x 10 s 10.2 -> 9.0 ms, wc_pipeline.hip).  Only the order of execution differs: the same bits,
    whichever way a ragged batch is run (options "unchain_below" = 0 / a large number and "schedule" of the handle; the WC_PIPELINE_*
    variables of those names are read once, at creation)."""
    import torch
    fs = 16000
    dev = torch.device("cuda", 0)
    xs = [make_utterance(fs, sec, 120 + i) for i, sec in enumerate((0.9, 0.4, 1.3, 0.7, 1.1, 0.5))]
    p = wca.Pipeline(fs)
    xl = [len(x) for x in xs]
    fl, yl = p.lengths(xl)
    d_x = torch.from_numpy(np.concatenate(xs)).to(dev)
    outs = []
    for knob in ("0", "100000", None, "lanes"):
        p.set_option("schedule", None)
        if knob is None:
            p.set_option("unchain_below", None)
        elif knob == "lanes":  # (held apart, two lanes: the default of batches that fill the chip)
            p.set_option("unchain_below", "0")
            p.set_option("schedule", "lanes")  # (anything but "chains")
        else:
            p.set_option("unchain_below", knob)
            p.set_option("schedule", "chains")  # ("0": held apart as two chains on four streams, rounds 3-5)
        d_t = torch.zeros(sum(fl), dtype=torch.float64, device=dev)
        d_f = torch.zeros_like(d_t)
        d_sp = torch.zeros(sum(fl) * p.bins, dtype=torch.float64, device=dev)
        d_ap = torch.zeros_like(d_sp)
        d_y = torch.zeros(sum(yl), dtype=torch.float64, device=dev)
        p.run_device(d_x.data_ptr(), xl, d_t.data_ptr(), d_f.data_ptr(), d_sp.data_ptr(), d_ap.data_ptr(), d_y.data_ptr(), rng_pos=[0] * len(xs))
        wca.lib().wc_synchronize()
        outs.append([v.cpu().numpy() for v in (d_t, d_f, d_sp, d_ap, d_y)])
    assert (outs[0][1] > 0).mean() > 0.3
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)


def test_user_stream_orders_our_kernels_with_the_callers_work(wca):
    """wc_set_stream (include/world_class_c.h): the calling thread's calls run on the caller's stream -- after the work
    already queued there -- without touching the library's own stream; results equal those on the library stream."""
    import torch
    fs = 16000
    xs = [make_utterance(fs, sec, 80 + i) for i, sec in enumerate((0.8, 0.5, 0.6, 0.4))]
    x_len = [len(x) for x in xs]
    f_len = [wca.get_samples(fs, n, 5.0) for n in x_len]
    y_len = [wca.synthesis_out_length(n, 5.0, fs) for n in f_len]
    pipe = wca.Pipeline(fs)
    bins = pipe.bins
    dev = torch.device("cuda", 0)
    host = torch.from_numpy(np.concatenate(xs)).pin_memory()

    def run(stream):
        d_x = torch.zeros(sum(x_len), dtype=torch.float64, device=dev)
        out = [torch.zeros(n, dtype=torch.float64, device=dev) for n in (sum(f_len), sum(f_len), sum(f_len) * bins, sum(f_len) * bins, sum(y_len))]
        if stream is None:
            d_x.copy_(host)
            torch.cuda.synchronize()
            pipe.run_device(d_x, x_len, *out)
        else:
            torch.cuda.synchronize()
            assert wca.lib().wc_set_stream(stream.cuda_stream) == 0
            try:
                with torch.cuda.stream(stream):
                    # a long-running kernel, then the upload of the samples: our kernels must wait for both
                    junk = torch.randn(4096, 4096, device=dev)
                    for _ in range(20):
                        junk = junk @ junk * 1e-3
                    d_x.copy_(host, non_blocking=True)
                pipe.run_device(d_x, x_len, *out)
                # stages through the same stream: CheapTrick on the freshly written F0
                sp2 = torch.zeros_like(out[2])
                wca.CheapTrick(fs).compute_device(d_x, x_len, out[0], out[1], f_len, sp2, rng_pos=[0] * len(xs))
                assert wca.lib().wc_synchronize() == 0
                assert torch.equal(sp2, out[2])
            finally:
                assert wca.lib().wc_set_stream(None) == 0
        torch.cuda.synchronize()
        return [o.cpu().numpy() for o in out]

    ref = run(None)
    s = torch.cuda.Stream()
    got = run(s)
    for a, b in zip(ref[:4], got[:4]):
        assert np.array_equal(a, b)
    assert np.abs(ref[4] - got[4]).max() < 1e-12  # atomic overlap-add order
    # and back on the library's stream afterwards, with the caller's stream gone
    del s
    again = run(None)
    assert np.array_equal(again[1], ref[1])


def test_headline_workload_against_the_reference(wca):
    """The benchmark's own workload at full size -- 48 kHz x 10 s utterances through the fused pipeline in one batch --
    against what the real reference returns for the same samples: F0 on every frame, every spectrogram / aperiodicity row
    through its sum and every 50th in full, the waveform through block sums and sixteen 4096-sample windows."""
    cases = [headline_case(u) for u in range(8)]  # the bench's eight distinct utterances
    res = wca.Pipeline(48000).run_batch([x for x, _ in cases] * 2)  # both halves of the batch schedule see every utterance
    for (x, g), r in zip(cases * 2, res):
        check_headline(r, g, 1e-6, 1e-7, 1e-7, 1e-8)


def test_config2_workload_against_the_reference(wca):
    """BASELINE config 2 at its full utterance size -- 16 kHz x 10 s through the fused pipeline in one batch (one-wavefront kernels
    at eight points per lane for CheapTrick / Synthesis, one 1024-point transform per real transform for D4C) -- against what the
    real reference returns for the same samples (tests/golden/config2_16k_10s.npz)."""
    cases = [headline_case(u, "config2_16k_10s.npz") for u in range(8)]
    res = wca.Pipeline(16000).run_batch([x for x, _ in cases] * 2)  # both halves of the batch schedule see every utterance
    for (x, g), r in zip(cases * 2, res):
        check_headline(r, g, 1e-6, 1e-7, 1e-7, 1e-8)


def test_c_abi_gather_over_rccl(wca):
    """wc_gather_device (include/world_class_shard.h) on a communicator the caller owns.  One GPU here, so the group has one rank
    -- what this pins is the binding: RCCL found in the process, the grouped broadcasts enqueued on the caller's stream behind
    the pipeline's work, the block landing at its offset."""
    import ctypes as C
    import os
    import torch
    from world_class_amd.shard import SHARD_SIGNATURES, partition_c
    partition_c([3, 2, 1], 2)  # binds the prototypes
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    comm = C.c_void_p()
    assert rccl.ncclCommInitAll(C.byref(comm), 1, (C.c_int * 1)(0)) == 0
    try:
        fs = 16000
        x = make_utterance(fs, 0.6, 77)
        pipe = wca.Pipeline(fs)
        dev = torch.device("cuda", 0)
        f_len, y_len = pipe.lengths([len(x)])
        d_x = torch.from_numpy(x).to(dev)
        out = [torch.zeros(n, dtype=torch.float64, device=dev) for n in (f_len[0], f_len[0], f_len[0] * pipe.bins, f_len[0] * pipe.bins, y_len[0])]
        torch.cuda.synchronize()
        pipe.run_device(d_x, [len(x)], *out)
        d_all = torch.zeros(y_len[0], dtype=torch.float64, device=dev)
        counts = (C.c_longlong * 1)(y_len[0])
        rc = wca.lib().wc_gather_device(comm, 1, 0, out[4].data_ptr(), counts, d_all.data_ptr())
        assert rc == 0, wca.last_error()
        assert wca.lib().wc_synchronize() == 0
        torch.cuda.synchronize()
        assert torch.equal(d_all, out[4]) and float(d_all.abs().sum()) > 0
        assert wca.lib().wc_gather_device(comm, 1, 3, out[4].data_ptr(), counts, d_all.data_ptr()) != 0  # rank outside the group
        # gather to one rank (grouped ncclSend / ncclRecv; a one-rank group: the root's own block, copied on the device)
        d_root = torch.zeros(y_len[0], dtype=torch.float64, device=dev)
        assert wca.lib().wc_gather_to_root_device(comm, 1, 0, 0, out[4].data_ptr(), counts, d_root.data_ptr()) == 0, wca.last_error()
        assert wca.lib().wc_synchronize() == 0
        torch.cuda.synchronize()
        assert torch.equal(d_root, out[4])
        assert wca.lib().wc_gather_to_root_device(comm, 1, 0, 2, out[4].data_ptr(), counts, d_root.data_ptr()) != 0  # root outside the group
    finally:
        rccl.ncclCommDestroy(comm)


def test_only_the_utterances_on_a_tie_are_run_again(wca, monkeypatch):
    """Round 6 (verdict item 3).  A raw candidate within 2e-13 of one of the refinement's integer cuts raises a flag PER UTTERANCE
    (hv_refine_packed_kernel / hv_refine_group_kernel); the flagged utterances -- and only those -- go through the path once more
    with the band-pass as direct FIR sums (hv_exact_twin), into their own slices of the outputs.  Until round 5 one flagged
    utterance re-ran the whole batch (the Harvest stage call) or every stage of every group (the pipeline).  Sixteen utterances:
    number 11 is an impulse train that raises the flag by itself (and whose contour depends on it: 1e-2 Hz), number 3 is
    forced onto it (WC_HARVEST_FORCE_TIE / option "force_tie", test hooks) -- two separate stretches to run again.  The other
    fourteen keep the bits of a run that ignores ties in every output; the flagged ones have the values a FIR band-pass gives
    them -- through the stage call, the two-lane schedule, the chains, the side-by-side schedule, the plain schedule
    (WC_PIPELINE_MODE=shared) and the host front-end."""
    from world_class_amd.synth import SIGNAL_KINDS, make_signal
    fs = 16000
    xs = [make_utterance(fs, 0.5 + 0.07 * (i % 5), 8100 + i) for i in range(16)]
    k, j = 11, 3
    assert SIGNAL_KINDS[1340043 % len(SIGNAL_KINDS)] == "impulses"
    xs[k] = make_signal(fs, 3.0, 1340043)
    # what a flagged utterance must come out as: the FIR band-pass from the start; what the others must: a run that ignores ties
    monkeypatch.setenv("WC_HARVEST_BANDPASS", "fir")
    fir = wca.Pipeline(fs).run_batch(xs, rng_pos=[0] * 16)[0]
    hv_fir = wca.Harvest(fs).compute_batch(xs)
    monkeypatch.delenv("WC_HARVEST_BANDPASS")
    monkeypatch.setenv("WC_HARVEST_TIES", "ignore")
    ign = wca.Pipeline(fs).run_batch(xs, rng_pos=[0] * 16)[0]
    hv_ign = wca.Harvest(fs).compute_batch(xs)
    monkeypatch.delenv("WC_HARVEST_TIES")
    assert 1e-4 < np.abs(ign[k]["f0"] - fir[k]["f0"]).max() < 1.0, "the tie no longer shows on the train: is the test still about it?"

    def same(a, b):
        return all(np.array_equal(a[n], b[n]) for n in ("tpos", "f0", "sp", "ap", "y"))

    # the stage call
    monkeypatch.setenv("WC_HARVEST_FORCE_TIE", str(j))
    hv = wca.Harvest(fs).compute_batch(xs)
    monkeypatch.delenv("WC_HARVEST_FORCE_TIE")
    hv_nat = wca.Harvest(fs).compute_batch(xs)  # (only the train's own flag)
    for u in range(16):
        assert np.array_equal(hv_nat[u][1], (hv_fir if u == k else hv_ign)[u][1]), ("stage call", u)
        assert np.array_equal(hv[u][1], (hv_fir if u in (k, j) else hv_ign)[u][1]), ("stage call, one forced", u)
    # the pipeline's schedules
    for name, opts, env in (("two lanes", {"unchain_below": "0"}, {}), ("chains", {"unchain_below": "0", "schedule": "chains"}, {}),
                            ("side by side", {}, {}), ("plain", {}, {"WC_PIPELINE_MODE": "shared"})):
        for n, v in env.items():
            monkeypatch.setenv(n, v)
        p = wca.Pipeline(fs)
        for n in env:
            monkeypatch.delenv(n)
        for n, v in opts.items():
            p.set_option(n, v)
        p.set_option("force_tie", str(j))
        got = p.run_batch(xs, rng_pos=[0] * 16)[0]
        for u in range(16):
            assert same(got[u], fir[u] if u in (k, j) else ign[u]), (name, u)
    # the host front-end (rows that have left for the host are sent again for the flagged utterances only)
    p = wca.Pipeline(fs)
    p.set_option("force_tie", str(j))
    for pinned in (True, False):
        out = p.host_buffers([len(x) for x in xs], pinned=pinned)
        p.run_batch_host(p.host_inputs(xs) if pinned else xs, out=out, rng_pos=[0] * 16)
        for u in range(16):
            want = fir[u] if u in (k, j) else ign[u]
            assert all(np.array_equal(out[u][n], want[n]) for n in ("tpos", "f0", "sp", "ap")), ("host front-end", pinned, u)
            assert np.abs(out[u]["y"] - want["y"]).max() < 1e-10


def test_two_lane_schedule_retries_after_an_overflow(wca, monkeypatch):
    """Advice of round 5: the two-lane schedule (the default of resident batches of 500 s and more) has its own copy of the retry
    logic; the small-caps and impulse-train tests all ran batches that take the side-by-side or chains path.  Here a batch goes
    through the two lanes (option "unchain_below" = 0) with rate-bounded buffers sized to overflow (WC_DEBUG_SMALL_CAPS=1:
    every group runs again with the hard bounds) and one utterance that raises the tie flag: the chains' bits in every output."""
    from world_class_amd.synth import make_signal
    fs = 16000
    xs = [make_utterance(fs, 0.6 + 0.1 * (i % 4), 8300 + i) for i in range(8)]
    xs[5] = make_signal(fs, 3.0, 1340043)  # an impulse train on a tie
    monkeypatch.setenv("WC_DEBUG_SMALL_CAPS", "1")
    outs = []
    for schedule in ("lanes", "chains"):
        p = wca.Pipeline(fs)
        p.set_option("unchain_below", "0")
        p.set_option("schedule", schedule)
        outs.append(p.run_batch(xs, rng_pos=[0] * len(xs))[0])
    monkeypatch.delenv("WC_DEBUG_SMALL_CAPS")
    plain = wca.Pipeline(fs).run_batch(xs, rng_pos=[0] * len(xs))[0]
    for u in range(len(xs)):
        for n in ("tpos", "f0", "sp", "ap", "y"):
            assert np.array_equal(outs[0][u][n], outs[1][u][n]), (u, n)
            assert np.array_equal(outs[0][u][n], plain[u][n]), (u, n)
