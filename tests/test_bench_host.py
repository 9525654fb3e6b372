"""not gpu: the host-side pieces of bench.py that need no device -- the issue-rate table it prices `issue_frac` on
(profiles/issue_rates.json, written by tools/issue_rate.hip on the GPU box), the kernel tables, and the CPU-baseline leg's
plumbing on a tiny sample (the real reference's OpenMP build through oracle/ref.py, when it has been built)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_issue_rate_table_is_usable():
    r = bench.load_issue_rates()
    assert r is not None, "profiles/issue_rates.json missing or malformed"
    for cls in ("fp64", "int32"):
        assert set(r[cls]) == {1, 2, 3, 4, 8} and all(0.2 < v < 20.0 for v in r[cls].values())
    # more wavefronts per SIMD never make an instruction dearer, and an FP64 instruction is never cheaper than a 32-bit one
    assert all(r["int32"][a] >= r["int32"][b] for a, b in ((1, 2), (2, 3), (3, 4), (4, 8)))
    assert all(r["fp64"][w] >= r["int32"][w] for w in r["fp64"])
    # a single wavefront cannot keep a SIMD busy: its instructions cost at least twice what they cost at full occupancy
    assert r["int32"][1] > 2.0 * r["int32"][8]


def test_kernel_tables_agree():
    assert set(bench.WAVES_PER_SIMD) <= set(bench.KERNEL_STAGE)
    assert set(bench.KERNEL_STAGE.values()) <= set(bench.STAGE_BYTES)
    assert bench.STAGE_BYTES["cheaptrick"] == 24584 and bench.STAGE_BYTES["synthesis"] == 18328
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert "frames/sec" in json.load(f)["metric"]


def test_cpu_baseline_leg_on_a_tiny_sample():
    from oracle import ref
    if not ref.available(omp=True):
        pytest.skip("oracle/_ref not built (needs the reference sources at build time)")
    from world_class_amd.synth import make_utterance
    x = make_utterance(bench.FS, 1.0, 3000)  # (the reference crashes on some very short signals: DESIGN_HISTORY.md section 7)
    frames, t0, t1 = bench._ref_call(([x, x], 2))
    assert frames == 2 * (int(1000.0 * len(x) / bench.FS / 5.0) + 1) and t1 > t0
    assert bench.usable_cores() >= 1
