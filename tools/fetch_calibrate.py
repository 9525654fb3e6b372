"""Known byte counts for calibrating rocprofv3's FETCH_SIZE on this GPU (MI355X_MICROARCH.md, HBM section: the counter reports
half of a 16-byte-per-lane streaming read; other widths are to be calibrated):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d OUT -o p -- python tools/fetch_calibrate.py
reads 3 x 2 GiB with 8-byte loads per lane (hook_stream_read_kernel<false>) and 3 x 2 GiB with 16-byte loads (<true>), each buffer
larger than the 256 MB Infinity Cache; tools/pmc_traffic.py --calib OUT turns the counter rows into bytes-per-counted-KiB factors."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import world_class_amd as w  # noqa: E402

L = w.lib()
L.wc_debug_stream_read.restype = C.c_int
L.wc_debug_stream_read.argtypes = [C.c_longlong, C.c_int, C.c_int]
N = 1 << 28  # doubles = 2 GiB
for wide in (0, 1):
    assert L.wc_debug_stream_read(N, wide, 3) == 0, w.last_error()
print("read 3 x %d bytes with 8-byte and with 16-byte loads" % (N * 8))
