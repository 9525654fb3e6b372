"""Print the kernel timeline (start/end per kernel, last pipeline step) from a rocprofv3 kernel-trace CSV."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "wc::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 26
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    name = r["Kernel_Name"].replace("void ", "").replace("wc::", "")[:40]
    q = r.get("Queue_Id", "?")
    s = (int(r["Start_Timestamp"]) - t0) / 1e6
    e = (int(r["End_Timestamp"]) - t0) / 1e6
    print("%-40s q=%3s start=%8.2f end=%8.2f dur=%7.2f" % (name, q, s, e, e - s))
