"""Kernels and copies of a host front-end run on one time line (development aid), from
    rocprofv3 --kernel-trace --memory-copy-trace -f csv -d DIR -o p -- python tools/host_frontend_probe.py
    python tools/trace_host_frontend.py DIR
Runs of consecutive copies / blit kernels are merged; times in ms from the start of the last run (the activity behind the last idle stretch)."""
import csv, sys
d = sys.argv[1]
kt = list(csv.DictReader(open(d + '/p_kernel_trace.csv')))
mc = list(csv.DictReader(open(d + '/p_memory_copy_trace.csv')))
ev = []
for r in kt: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K q%s %s' % (r['Queue_Id'], r['Kernel_Name'].replace('void ', '').replace('wc::', '')[:40])))
for r in mc: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C s%s %s' % (r['Stream_Id'], r['Direction'][12:])))
ev.sort()
# the last run = everything behind the last idle stretch of 3 ms or more (the probe's runs are synchronous calls)
t0 = ev[0][0]
busy_until = ev[0][1]
for s_, e_, n_ in ev:
    if s_ - busy_until > 3000000:
        t0 = s_
    busy_until = max(busy_until, e_)
# merge consecutive copies / blit kernels of the same name into runs
runs = []
for s, e, n in ev:
    if s < t0 or e - s < 150000: continue
    key = n if not n.startswith('K') or 'copyBuffer' in n else None
    if key and runs and runs[-1][2] == key and s - runs[-1][1] < 300000:
        runs[-1][1] = max(e, runs[-1][1]); runs[-1][3] += 1
    else:
        runs.append([s, e, n, 1])
for s, e, n, c in runs:
    print("%8.2f -> %8.2f  %6.2f  x%-3d %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, c, n))
