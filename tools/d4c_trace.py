"""Phase timings of d4c2_frames_kernel from a WC_D4C2_TRACE build (development aid):
    python tools/ab_build.py d4trace -DWC_D4C2_TRACE=1
    WC_LIB_PATH=world_class_amd/_variants/d4trace.so WC_D4C_TRACE=/tmp/t.bin python tools/microbench.py --stages cd --utts 8 --iters 1
    python tools/d4c_trace.py /tmp/t.bin"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)[:, :11].astype(np.int64)
a = a[(a[:, 10] > a[:, 0]) & (a[:, 0] > 0)]
# (windows of at most 2048 samples, the usual case: job-major; the longer ones stamp half-major: "rest of the even half", "odd half")
names = ["set-up, window of job 0, parked", "job 0 even: master read + transform", "job 0 even: weighted read + transform + products + park", "job 0 odd half", "jobs 1 and 2", "reload parked",
         "dc corrections", "power smoothing (order-faithful)", "division + two signed smoothings", "store"]
d = np.diff(a, axis=1)
print("%d gated frames traced; shader-clock cycles per phase (mean / median):" % len(a))
for i, n in enumerate(names):
    print("  %-40s %9.0f %9.0f" % (n, d[:, i].mean(), np.median(d[:, i])))
print("  %-40s %9.0f" % ("total", (a[:, 10] - a[:, 0]).mean()))
