"""Per-kernel SQ counter summary from rocprofv3 --pmc passes (development aid).
usage: python tools/pmc_sq.py <dir> [<dir> ...]   (each dir holds a *counter_collection.csv)"""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            k = r["Kernel_Name"]
            if "wc::" not in k:
                continue
            k = k.replace("void ", "").replace("wc::", "").split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
names = sorted({c for k in acc for c in acc[k]})
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    a = {c: acc[k][c] / max(cnt[k][c], 1) for c in acc[k]}
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    print(k)
    for c in names:
        if c in a:
            print("   %-24s %14.0f  %6.1f%% of WAVE_CYCLES" % (c, a[c], 100 * a[c] / wc))
