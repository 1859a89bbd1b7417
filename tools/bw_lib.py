"""write / copy rates of the library's own vector kernels on buffers from its allocator (developer tool)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tigar_amd import device as dev
from tigar_amd import _lib
L = _lib.lib()
for gib in (4, 16, 32):
    n = gib << 27
    x = dev.DeviceVector(n)
    L.tg_vec_fill(x._h, 1.0); dev.sync()
    dev.timer_start(0)
    for _ in range(5): L.tg_vec_fill(x._h, 1.0)
    ms = dev.timer_stop(0) / 5
    print("tg_vec_fill %2d GiB  %.3f ms -> %.2f TB/s" % (gib, ms, 8 * n / ms / 1e9))
    del x
