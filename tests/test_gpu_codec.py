"""SURVEY.md section 8(f) N3 on the GPU: the reference's feature codec (include/codec.hpp) as device kernels, through the
reference-named host-pointer functions and through the device-resident variants."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ("c1_16k_2s_floor71", "m24k_1s_1ms", "m48k_1s")


@pytest.fixture(scope="module")
def env():
    import torch
    import world_class_amd as w
    from world_class_amd import codec
    w.lib().wc_set_device(0)
    return w, codec, torch, np.load(os.path.join(ROOT, "tests", "golden", "io", "codec_golden.npz"))


@pytest.mark.parametrize("case", CASES)
def test_codec_golden(env, case):
    w, codec, torch, g = env
    fs, fft = int(g[case + "/fs"]), int(g[case + "/fft"])
    assert codec.number_of_aperiodicities(fs) == int(g[case + "/n_ap"])
    for nd in (25, 60, fft // 4 + 1):
        c = codec.code_spectral_envelope(g[case + "/sp"], fs, fft, nd)
        assert np.abs(c - g[f"{case}/sp_coded_{nd}"]).max() < 1e-11  # values are O(10): log of two math libraries + FFT order
        d = codec.decode_spectral_envelope(g[f"{case}/sp_coded_{nd}"], fs, fft)
        assert np.abs(d / g[f"{case}/sp_decoded_{nd}"] - 1).max() < 1e-11
    assert np.abs(codec.code_aperiodicity(g[case + "/ap"], fs, fft) - g[case + "/ap_coded"]).max() < 1e-11
    assert np.abs(codec.decode_aperiodicity(g[case + "/ap_coded"], fs, fft) - g[case + "/ap_decoded"]).max() < 1e-13


def test_codec_device_batch_vs_oracle(env):
    """a few thousand analysis rows resident on the device against the CPU restatement, and the coded size"""
    w, codec, torch, g = env
    from oracle import port, port_codec as pc
    from world_class_amd.synth import make_utterance
    fs, nd = 48000, 60
    x = make_utterance(fs, 1.0, 99)
    P = port.Port()
    P.rng_reset()
    r = P.pipeline(x, fs)
    P.rng_reset()
    sp, ap = r["sp"], r["ap"]
    reps, n = 16, sp.shape[0]
    d_sp = torch.from_numpy(np.tile(sp, (reps, 1)).ravel()).cuda()
    d_ap = torch.from_numpy(np.tile(ap, (reps, 1)).ravel()).cuda()
    n_ap = codec.number_of_aperiodicities(fs)
    d_csp = torch.empty(reps * n * nd, dtype=torch.float64, device="cuda")
    d_cap = torch.empty(reps * n * n_ap, dtype=torch.float64, device="cuda")
    codec.code_spectral_envelope_device(fs, 2048, reps * n, nd, d_sp, d_csp)
    codec.code_aperiodicity_device(fs, 2048, reps * n, d_ap, d_cap)
    d_sp2, d_ap2 = torch.empty_like(d_sp), torch.empty_like(d_ap)
    codec.decode_spectral_envelope_device(fs, 2048, reps * n, nd, d_csp, d_sp2)
    codec.decode_aperiodicity_device(fs, 2048, reps * n, d_cap, d_ap2)
    w.lib().wc_synchronize()
    csp = d_csp.cpu().numpy().reshape(reps, n, nd)
    cap = d_cap.cpu().numpy().reshape(reps, n, n_ap)
    assert np.array_equal(csp[0], csp[-1]) and np.array_equal(cap[0], cap[-1])
    ref_csp, ref_cap = pc.code_spectral_envelope(sp, fs, 2048, nd), pc.code_aperiodicity(ap, fs, 2048)
    assert np.abs(csp[0] - ref_csp).max() < 1e-11
    assert np.abs(cap[0] - ref_cap).max() < 1e-11
    assert np.abs(d_sp2.cpu().numpy().reshape(reps, n, -1)[3] / pc.decode_spectral_envelope(ref_csp, fs, 2048) - 1).max() < 1e-10
    assert np.abs(d_ap2.cpu().numpy().reshape(reps, n, -1)[5] - pc.decode_aperiodicity(ref_cap, fs, 2048)).max() < 1e-12
    assert (nd + n_ap) * 30 < 2 * 1025  # the point of the codec: 30x fewer doubles per frame than sp + ap


def test_codec_rejects_unsupported_sizes(env):
    w, codec, torch, g = env
    d = torch.zeros(4096, dtype=torch.float64, device="cuda")
    with pytest.raises(w.WorldClassError):
        codec.code_spectral_envelope_device(48000, 2048, 1, 600, d, d)   # more dimensions than the half spectrum has bins
    with pytest.raises(w.WorldClassError):
        codec.code_aperiodicity_device(8000, 1024, 1, d, d)             # no aperiodicity band below 12 kHz sampling
