"""Host <-> device copy rate through the C ABI (troyhip_copy_h2d / _d2h, pageable numpy memory), and what it would make of the headline if the
operands of every multiply + relinearize came from host memory and the result went back (the boundary itself takes DEVICE pointers: the bench's
`value` never includes this).  usage: python tools/pcie_probe.py [MiB = 1024]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import troy_amd as ta  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ta.KernelProvider.initialize(0)
words = mib * (1 << 20) // 8
host = np.random.default_rng(1).integers(0, 1 << 62, words, dtype=np.uint64)
import ctypes as C  # noqa: E402
from troy_amd import capi  # noqa: E402
dev = ta.DeviceBuffer(words)
lib = dev.lib
back = np.empty(words, dtype=np.uint64)
hp, bp = host.ctypes.data_as(C.c_void_p), back.ctypes.data_as(C.c_void_p)
for name, fn in (("h2d", lambda: capi.check(lib, lib.troyhip_copy_h2d(C.c_void_p(dev.ptr), hp, C.c_size_t(words * 8), None))),
                 ("d2h", lambda: capi.check(lib, lib.troyhip_copy_d2h(bp, C.c_void_p(dev.ptr), C.c_size_t(words * 8), None)))):
    fn()
    ta.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    ta.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {mib} MiB in {dt * 1e3:.1f} ms = {mib / 1024 / dt:.1f} GiB/s")
    if name == "h2d":
        h2d = mib / 1024 / dt
    else:
        d2h = mib / 1024 / dt
N, L = 32768, 14
in_gib = 2 * 2 * L * N * 8 / 2**30   # two size-2 operands
out_gib = 2 * L * N * 8 / 2**30      # the relinearized product
t_op = in_gib / h2d + out_gib / d2h
print(f"BFV N=2^15 L=14 multiply+relinearize with operands from / result to host memory: {in_gib * 1024:.1f} MiB in, {out_gib * 1024:.1f} MiB out "
      f"-> at most {1 / t_op:.0f} ops/s if nothing overlaps, against the device-resident figure of the bench line")
assert np.array_equal(back, host)
