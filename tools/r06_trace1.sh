export TMPDIR=/tmp
REPO=$PWD
OUT=$PWD/gpurun_out/r06_trace1
rm -rf $OUT; mkdir -p $OUT
cd /tmp
( cd $REPO && rocprofv3 --kernel-trace --stats -f csv -d "$OUT" -o p -- python tools/microbench.py --utts ${1:-1} --iters 3 > "$OUT/run.txt" 2> "$OUT/run.err" )
cd $REPO
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join("gpurun_out/r06_trace1", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "wc::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last full step: find the last hv_decimate launch
idx = [i for i, r in enumerate(rows) if "hv_decimate" in r["Kernel_Name"] and "<0>" in r["Kernel_Name"]]
st = idx[-1]
t0 = int(rows[st]["Start_Timestamp"])
prev_end = t0
for r in rows[st:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("wc::", "").split("(")[0][:48]
    print("%-50s start %8.1f us  dur %7.1f us  gap %6.1f us  grid %s" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size", "")))
    prev_end = max(prev_end, e)
PY
