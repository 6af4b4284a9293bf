"""Time the batched NTT (forward and inverse separately) of the key-switch shape with a given build of the library.
usage: python tools/ntt_probe.py <libtroyhip*.so> [batch]     (development tool; variants come from tools/ntt_probe.sh)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import troy_amd as ta  # noqa: E402
from troy_amd import api, capi  # noqa: E402

path = os.path.abspath(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
lib = capi.load(path)
api.KernelProvider.initialize(0, _lib=lib)
N, bits = 32768, [60] + [50] * 13 + [60]
primes = ta.CoeffModulus.Create(N, bits)
ctx = ta.SEALContext(capi.BFV, N, primes, ta.PlainModulus.Batching(N, 20))
K, L = len(primes), len(primes) - 1
rows = B * K * L
D = ta.DeviceBuffer(rows * N)
out_primes = primes[:L] + [primes[K - 1]]
ctx.fill_uniform(D, rows, out_primes, seed=1, inner=L)
pr = np.array(out_primes, dtype=np.uint64)
timer = C.c_void_p()
capi.check(lib, lib.troyhip_timer_create(C.byref(timer)))


def ntt(inv):
    capi.check(lib, lib.troyhip_ntt(ctx.h, C.c_void_p(D.ptr), C.c_uint64(rows), pr.ctypes.data_as(C.c_void_p), len(pr), L, inv, None))


res = []
for inv in (0, 1):
    ntt(inv)
    ta.synchronize()
    capi.check(lib, lib.troyhip_timer_start(timer, None))
    for _ in range(6):
        ntt(inv)
    capi.check(lib, lib.troyhip_timer_stop(timer, None))
    ms = C.c_float()
    capi.check(lib, lib.troyhip_timer_elapsed_ms(timer, C.byref(ms)))
    us = ms.value * 1e3 / 6
    res.append(us)
print(f"{os.path.basename(path):40s} B={B} rows={rows}  fwd {res[0]:8.1f} us ({16.0*N*rows/res[0]/1e3:7.1f} GB/s)   inv {res[1]:8.1f} us ({16.0*N*rows/res[1]/1e3:7.1f} GB/s)")
