"""Turn two rocprofv3 PMC passes into profiles/pmc_traffic.json (HBM bytes per bench step of each hot kernel).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o p -f csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o p -f csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS -d gpurun_out/pmc_valu -o p -f csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --valu-dir gpurun_out/pmc_valu --steps 3

Both counters are in KiB.  FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 reports half of a
coalesced read stream); WRITE_SIZE is used as is.  With the default pipeline schedule every kernel is launched
once per half batch, i.e. twice per step: the figure written is the SUM over the launches of one step, which is
what bench.py's `roofline.achieved` is priced on (whole-batch bytes / summed kernel time).
"""
import argparse
import csv
import glob
import json
import os
import sys

KERNELS = {  # substring of the kernel symbol -> name bench.py uses
    "d4c_frames_kernel": "d4c_frames", "d4c_band_kernel": "d4c_bands", "d4c_rows_kernel": "d4c_bands", "ct_frames_kernel": "cheaptrick_frames", "d4c_lovetrain_kernel": "d4c_lovetrain",
    "ct_wave_kernel": "cheaptrick_frames", "d4c2_frames_kernel": "d4c_frames", "d4c2_band_kernel": "d4c_bands", "d4c2_lovetrain_kernel": "d4c_lovetrain",
    "syn_pulse_wave_kernel": "synthesis_pulses", "syn_overlap_add_kernel": "synthesis_pulses",
    "hv_refine_kernel": "harvest_refine", "hv_refine_packed_kernel": "harvest_refine", "hv_refine_group_kernel": "harvest_refine", "hv_bandpass_kernel": "harvest_bandpass", "hv_bandpass_sdft_kernel": "harvest_bandpass",
    "hv_compact_kernel": "harvest_bandpass", "hv_raw_kernel": "harvest_raw", "hv_raw_wave_kernel": "harvest_raw", "hv_rawdesc_kernel": "harvest_raw", "hv_detect_kernel": "harvest_raw",
    "hv_decimate_scan_kernel": "harvest_decimate", "hv_seam_kernel": "harvest_bandpass", "hv_bandpass_quiet_kernel": "harvest_bandpass",
    "hv_contour_kernel": "harvest_contour", "syn_pulse_kernel": "synthesis_pulses", "syn_timebase_kernel": "synthesis_timebase",
}


def total_kib(d, counter):
    out = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                for sub, name in KERNELS.items():
                    if sub in row["Kernel_Name"]:
                        out[name] = out.get(name, 0.0) + float(row["Counter_Value"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--valu-dir", default=None, help="pass with SQ_INSTS_VALU (vector instructions issued, per wave)")
    ap.add_argument("--f64-dir", default=None, help="pass with SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64")
    ap.add_argument("--steps", type=int, required=True, help="warmup + timed steps of the profiled bench run")
    ap.add_argument("--build-hash", default=None, help="source hash of the profiled library (wc_build_hash): bench.py refuses the file for any other build")
    ap.add_argument("--calib", default=None, help="FETCH_SIZE pass over tools/fetch_calibrate.py: measured bytes per counted KiB for 8- and 16-byte loads")
    ap.add_argument("--config3-f64-dir", default=None, help="F64 instruction-counter pass over `tools/microbench.py --stages c --utts 256 --iters 1` (2 launches)")
    ap.add_argument("--config3-valu-dir", default=None, help="SQ_INSTS_VALU pass over the same config-3 command")
    ap.add_argument("-o", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json"))
    a = ap.parse_args()
    fetch, write = total_kib(a.fetch_dir, "FETCH_SIZE"), total_kib(a.write_dir, "WRITE_SIZE")
    if not fetch or not write:
        sys.exit("no counter rows found")
    out = {"_note": "HBM bytes per bench STEP (sum over the two half-batch launches of the default schedule) from rocprofv3 PMC, "
                    "separate passes (--kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE) of `python bench.py --steps 2 --warmup 1 "
                    "--no-cpu-baseline`; counters are KiB; FETCH_SIZE times the factor measured on an 8-byte-per-lane streaming read of known size (`_fetch_calibration`; the guide's 2.0 for 16-byte loads when absent); WRITE_SIZE as is. "
                    "Made by tools/pmc_traffic.py.",
           "_raw_kib_per_step": {}}
    factor = 2.0  # the guide's figure for 16-byte-per-lane streams
    if a.calib:
        cal = {}
        for path in glob.glob(os.path.join(a.calib, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    if row["Counter_Name"] == "FETCH_SIZE" and "hook_stream_read_kernel" in row["Kernel_Name"]:
                        key = "16B" if "<true>" in row["Kernel_Name"] else "8B"
                        cal.setdefault(key, []).append(float(row["Counter_Value"]))
        true_kib = (1 << 28) * 8 / 1024.0
        out["_fetch_calibration"] = {k: {"counted_KiB_per_launch": sum(v) / len(v), "true_KiB_per_launch": true_kib,
                                          "bytes_per_counted_byte": true_kib / (sum(v) / len(v))} for k, v in cal.items() if v}
        if "8B" in out["_fetch_calibration"]:
            factor = out["_fetch_calibration"]["8B"]["bytes_per_counted_byte"]  # these kernels read 8 bytes per lane for the most part
    out["_fetch_factor_used"] = factor
    if a.build_hash:
        out["_build_hash"] = a.build_hash
    for name in sorted(set(fetch) & set(write)):
        f, w_ = fetch[name] / a.steps, write[name] / a.steps
        out["_raw_kib_per_step"][name] = {"FETCH_SIZE": f, "WRITE_SIZE": w_}
        out[name] = int((factor * f + w_) * 1024)
    if a.valu_dir:
        valu = total_kib(a.valu_dir, "SQ_INSTS_VALU")
        out["_valu_note"] = ("SQ_INSTS_VALU per bench step (wave-level vector instructions).  An f64 instruction occupies a SIMD for 4 "
                             "cycles, so the chip issues at most 1024 SIMDs x 2.4 GHz / 4 = 614.4 G of them per second; bench.py divides "
                             "by the live kernel time to report the vector-issue utilisation next to the HBM fraction.")
        out["_valu_insts_per_step"] = {k: v / a.steps for k, v in sorted(valu.items())}
    if a.f64_dir:
        parts = {c: total_kib(a.f64_dir, c) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")}
        names = sorted(set().union(*[set(v) for v in parts.values()]))
        if names:
            out["_fp64_note"] = ("FP64 floating-point operations per bench step from SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 (wave-level "
                                 "instruction counts) x 64 lanes, a fused multiply-add counted as two; an upper bound where lanes are masked off. "
                                 "Peak: 78.6 TFLOP/s (256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz).")
            out["_fp64_insts_per_step"] = {n: {c[14:]: parts[c].get(n, 0.0) / a.steps for c in parts} for n in names}
            out["_fp64_flops_per_step"] = {n: 64.0 * (parts["SQ_INSTS_VALU_ADD_F64"].get(n, 0.0) + parts["SQ_INSTS_VALU_MUL_F64"].get(n, 0.0)
                                                      + 2.0 * parts["SQ_INSTS_VALU_FMA_F64"].get(n, 0.0) + parts["SQ_INSTS_VALU_TRANS_F64"].get(n, 0.0)) / a.steps
                                           for n in names}
    if a.config3_f64_dir:
        parts = {c: total_kib(a.config3_f64_dir, c) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")}
        g = lambda c: parts[c].get("cheaptrick_frames", 0.0)
        launches = 2.0  # microbench: one warm-up and one timed launch
        out["_config3_fp64_flops"] = 64.0 * (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + 2.0 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_TRANS_F64")) / launches
        out["_config3_fp64_insts"] = (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_TRANS_F64")) / launches
    if a.config3_valu_dir:
        out["_config3_valu_insts"] = total_kib(a.config3_valu_dir, "SQ_INSTS_VALU").get("cheaptrick_frames", 0.0) / 2.0  # (two launches)
    with open(a.o, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}))


if __name__ == "__main__":
    main()
