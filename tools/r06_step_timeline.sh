# kernel timeline of one step of the headline workload (the two-lane schedule): full-grid lane, latency lane, idle stretches
export TMPDIR=/tmp
REPO=$PWD
OUT=$PWD/gpurun_out/r06_step_timeline
rm -rf $OUT; mkdir -p $OUT
( cd $REPO && rocprofv3 --kernel-trace -f csv -d "$OUT" -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-serialised > "$OUT/bench.json" 2> "$OUT/bench.err" )
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join("gpurun_out/r06_step_timeline", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wc::" in r["Kernel_Name"] or "syn_prefix" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last-but-one... find starts of steps = hv_decimate_scan<0> launches with a gap before
starts = [i for i, r in enumerate(rows) if "hv_decimate_scan_kernel<0>" in r["Kernel_Name"]]
st = starts[-2]  # first group of the last step
rows = rows[st:]
t0 = int(rows[0]["Start_Timestamp"])
big = ("bandpass_sdft_kernel", "hv_raw_kernel", "hv_refine", "ct_wave", "d4c2_", "syn_pulse_wave", "overlap_add", "hv_decimate", "hv_detect")
busy_end = t0
idle = 0.0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("wc::", "").split("(")[0][:44]
    lane = "F" if any(b in name for b in big) else "s"
    gap = (s - busy_end) / 1e3
    if gap > 0: idle += gap
    print("%s %-46s start %9.1f  dur %8.1f  %s" % (lane, name, (s - t0) / 1e3, (e - s) / 1e3, ("IDLE %.1f us before" % gap) if gap > 2 else ""))
    busy_end = max(busy_end, e)
print("step %.2f ms, nothing running for %.2f ms" % ((busy_end - t0) / 1e6, idle / 1e3))
PY
