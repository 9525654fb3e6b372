#!/bin/bash
# host API calls, kernels and copies of the drop-in caller's four compute() calls (steady state):
#   bash tools/r06_dropin_trace.sh > gpurun_out/dropin_trace.txt 2>&1
export TMPDIR=/tmp
REPO=$PWD
OUT=$PWD/gpurun_out/r06_dropin_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp
( cd $REPO && rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace -f csv -d "$OUT" -o p -- python tools/dropin_probe.py > "$OUT/run.txt" 2> "$OUT/run.err" )
cd $REPO
tail -2 $OUT/run.txt
python - <<'PY'
import csv, glob, os
d = "gpurun_out/r06_dropin_trace"
def load(pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
api = load("*hip_api_trace.csv")
cp = load("*memory_copy_trace.csv")
kr = load("*kernel_trace.csv")
print("api rows", len(api), "copies", len(cp), "kernels", len(kr))
ev = []
for r in api:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api  " + r["Function"]))
for r in cp:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s bytes" % (r.get("Direction", ""), r.get("Bytes", r.get("Size", "")))))
for r in kr:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "KERN " + r["Kernel_Name"].replace("void ", "").replace("wc::", "").split("(")[0][:40]))
ev.sort()
# the last 6 ms before the last event
t_end = max(e[1] for e in ev)
last = [e for e in ev if e[0] > t_end - 7_000_000]
t0 = last[0][0]
for s, e, n in last:
    if e - s > 15_000 or n.startswith("COPY"):
        print("%9.1f us  +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
