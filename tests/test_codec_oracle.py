"""CPU: the numpy restatement of the reference's codec (oracle/port_codec.py) is pinned to outputs of the real reference
(tests/golden/io/codec_golden.npz, made by oracle/gen_golden_codec.py from rows of the pipeline goldens)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ("c1_16k_2s_floor71", "m24k_1s_1ms", "m48k_1s")


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(ROOT, "tests", "golden", "io", "codec_golden.npz"))


@pytest.mark.parametrize("case", CASES)
def test_restatement_matches_the_reference(g, case):
    from oracle import port_codec as pc
    fs, fft = int(g[case + "/fs"]), int(g[case + "/fft"])
    assert pc.number_of_aperiodicities(fs) == int(g[case + "/n_ap"])
    for nd in (25, 60, fft // 4 + 1):
        c = pc.code_spectral_envelope(g[case + "/sp"], fs, fft, nd)
        assert np.abs(c - g[f"{case}/sp_coded_{nd}"]).max() < 1e-13
        d = pc.decode_spectral_envelope(g[f"{case}/sp_coded_{nd}"], fs, fft)
        assert np.abs(d / g[f"{case}/sp_decoded_{nd}"] - 1).max() < 1e-12
    assert np.abs(pc.code_aperiodicity(g[case + "/ap"], fs, fft) - g[case + "/ap_coded"]).max() < 1e-12
    assert np.abs(pc.decode_aperiodicity(g[case + "/ap_coded"], fs, fft) - g[case + "/ap_decoded"]).max() < 1e-14


def test_number_of_aperiodicities_table(g):
    from oracle import port_codec as pc
    assert [pc.number_of_aperiodicities(int(f)) for f in g["n_ap_table_fs"]] == list(g["n_ap_table"])
