"""Phase timings of syn_pulse_wave_kernel from a WC_SYN_TRACE build (development aid):
    python tools/ab_build.py syntrace -DWC_SYN_TRACE=1
    WC_LIB_PATH=world_class_amd/_variants/syntrace.so WC_SYN_TRACE_FILE=/tmp/s.bin python tools/microbench.py --stages cds --utts 64 --iters 1
    python tools/syn_trace.py /tmp/s.bin"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)[:, :11].astype(np.int64)
a = a[a[:, 10] > 0]
v = a[a[:, 5] > 0]
names = ["tables to LDS", "rows + log spectrum (periodic)", "minimum phase", "delay + inverse transform", "output of the periodic half + dc",
         "noise + its transform", "rows + log spectrum (aperiodic)", "minimum phase", "product + inverse transform", "overlap-add"]
print("%d pulses, %d with a periodic part; shader-clock cycles per phase (mean over voiced pulses):" % (len(a), len(v)))
d = np.diff(v, axis=1)
for i, n in enumerate(names):
    print("  %-40s %9.0f" % (n, d[:, i].mean()))
print("  %-40s %9.0f (voiced) %9.0f (all)" % ("total", (v[:, 10] - v[:, 0]).mean(), (a[:, 10] - a[:, 0]).mean()))
