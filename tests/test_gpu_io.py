"""SURVEY.md section 8(f) N1/N4 on the GPU: int16 PCM expanded / quantised on the device with the reference tools'
arithmetic, and the demo's parameter modification as a kernel between analysis and synthesis."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IO = os.path.join(ROOT, "tests", "golden", "io")


@pytest.fixture(scope="module")
def env():
    import torch
    import world_class_amd as w
    from world_class_amd import io as wio
    w.lib().wc_set_device(0)
    return w, wio, torch


def test_pcm16_round_trip_on_the_device(env):
    w, wio, torch = env
    from oracle import port_io
    g = np.load(os.path.join(IO, "io_golden.npz"))
    pcm, fs = wio.wavread_pcm16(os.path.join(IO, "ref_written_16k.wav"))
    d_pcm = torch.from_numpy(pcm.copy()).cuda()
    d_x = torch.empty(len(pcm), dtype=torch.float64, device="cuda")
    wio.pcm16_to_double_device(d_pcm, len(pcm), d_x)
    w.lib().wc_synchronize()
    assert np.array_equal(d_x.cpu().numpy(), g["wav_read_x"])  # exactly what the reference's wavread returns
    y = np.concatenate([g["wav_x"], [np.nan, np.inf, -np.inf, 1e300, -1e300, 0.0]])
    d_y = torch.from_numpy(y).cuda()
    d_q = torch.empty(len(y), dtype=torch.int16, device="cuda")
    wio.double_to_pcm16_device(d_y, len(y), d_q)
    w.lib().wc_synchronize()
    assert np.array_equal(d_q.cpu().numpy(), port_io.pcm16_of(y))  # wavwrite's quantisation, incl. its out-of-range cases


@pytest.mark.parametrize("tag", ["scale_only", "up", "down", "down_small"])
def test_parameter_modification_golden(env, tag):
    w, wio, torch = env
    g = np.load(os.path.join(IO, "io_golden.npz"))
    fs, fft = int(g["mod_fs"]), int(g["mod_fft"])
    n_args, shift, ratio = g[f"mod_{tag}_args"]
    d_f0 = torch.from_numpy(g["mod_f0"].copy()).cuda()
    d_sp = torch.from_numpy(g["mod_sp"].copy()).cuda()
    wio.modify_parameters_device(fs, fft, len(g["mod_f0"]), d_f0, d_sp, shift, ratio if n_args >= 2 else 0.0)
    w.lib().wc_synchronize()
    assert np.array_equal(d_f0.cpu().numpy(), g[f"mod_{tag}_f0"])
    assert np.abs(d_sp.cpu().numpy() / g[f"mod_{tag}_sp"] - 1).max() < 1e-12  # log/exp of two math libraries


def test_analysis_modification_synthesis_without_host_round_trip(env):
    """the demo's flow (reference test/test.cpp:288-384 with f0 and spec arguments) on device-resident data, against
    the oracle: CPU analysis -> numpy modification -> CPU synthesis"""
    w, wio, torch = env
    from oracle import port, port_io
    from world_class_amd.synth import make_utterance
    fs = 16000
    x = make_utterance(fs, 0.8, 4242)
    P = port.Port()
    P.rng_reset()
    r = P.pipeline(x, fs)
    f0m, spm = port_io.parameter_modification(fs, 1024, r["f0"], r["sp"], 1.3, 0.9)
    # device: analysis stages, modification kernel, synthesis, all on resident buffers
    hv, ct, d4 = w.Harvest(fs), w.CheapTrick(fs), w.D4C(fs)
    sy = w.Synthesis(fs, ct.fft_size, 5.0)
    nf = hv.get_samples(len(x))
    d_x = torch.from_numpy(x).cuda()
    d_t = torch.empty(nf, dtype=torch.float64, device="cuda")
    d_f = torch.empty_like(d_t)
    d_sp = torch.empty(nf * ct.bins, dtype=torch.float64, device="cuda")
    d_ap = torch.empty_like(d_sp)
    hv.compute_device(d_x, [len(x)], d_t, d_f)
    pos = ct.compute_device(d_x, [len(x)], d_t, d_f, [nf], d_sp, rng_pos=[0])
    pos = d4.compute_device(d_x, [len(x)], d_t, d_f, [nf], ct.fft_size, d_ap, rng_pos=pos)
    wio.modify_parameters_device(fs, ct.fft_size, nf, d_f, d_sp, 1.3, 0.9)
    w.lib().wc_synchronize()
    assert np.abs(d_f.cpu().numpy() - f0m).max() < 1e-6
    assert np.abs(d_sp.cpu().numpy().reshape(nf, -1) / spm - 1).max() < 1e-7
    ny = sy.out_length(nf)
    d_y = torch.empty(ny, dtype=torch.float64, device="cuda")
    sy.compute_device(d_f, [nf], d_sp, d_ap, [ny], d_y, rng_pos=pos)
    w.lib().wc_synchronize()
    P.rng_seek(pos[0])
    y_ref = P.synthesis(f0m, spm, r["ap"], fs, 5.0)
    P.rng_reset()
    assert np.abs(d_y.cpu().numpy() - y_ref).max() < 1e-8
