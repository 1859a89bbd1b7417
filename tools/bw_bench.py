"""streaming write / copy bandwidth calibration (developer tool)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tigar_amd import device as dev
from tigar_amd import _lib
import ctypes as C
n = 1 << 29   # 4 GiB of doubles
x = dev.DeviceVector(n); y = dev.DeviceVector(n)
L = _lib.lib()
for name, fn, nbytes in (("fill (write 8 B/elt)", lambda: L.tg_vec_fill(x._h, 1.0), 8 * n),
                         ("copy (r+w 16 B/elt)", lambda: L.tg_vec_copy(y._h, x._h), 16 * n),
                         ("axpy (2r+w 24 B/elt)", lambda: L.tg_vec_axpy(y._h, 1.0, x._h), 24 * n)):
    fn(); dev.sync()
    dev.timer_start(0)
    for _ in range(10): fn()
    ms = dev.timer_stop(0) / 10
    print("%-24s %.3f ms -> %.0f GB/s" % (name, ms, nbytes / ms / 1e6))
