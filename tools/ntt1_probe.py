"""Time the plain batched NTT of N = 2^15 (PROBE_N: another size; forward and inverse separately) at the key-switch shape of the benchmark
(primes 60, 58 x 13, 60 bits; rows = B * 15 * 14).  TROYHIP_NTT=twopass selects ntt2.hip, default is ntt1.hip.
usage: python tools/ntt1_probe.py [batch] [reps]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import troy_amd as ta  # noqa: E402
from troy_amd import api, capi  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lib = capi.load(os.environ.get("TROYHIP_LIB")) if os.environ.get("TROYHIP_LIB") else capi.load()
api.KernelProvider.initialize(0, _lib=lib)
N, bits = int(os.environ.get("PROBE_N", "32768")), eval(os.environ.get("PROBE_BITS", "[60] + [58] * 13 + [60]"))
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
    for _ in range(reps):
        ntt(inv)
    capi.check(lib, lib.troyhip_timer_stop(timer, None))
    ms = C.c_float()
    capi.check(lib, lib.troyhip_timer_elapsed_ms(timer, C.byref(ms)))
    res.append(ms.value * 1e3 / reps)
mode = os.environ.get("TROYHIP_NTT", "single-pass")
f = lambda us: 16.0 * N * rows / us / 1e3  # noqa: E731
print(f"{mode:12s} B={B} rows={rows}  fwd {res[0]:8.1f} us ({f(res[0]):7.1f} GB/s, frac {f(res[0])/8000:.3f})   inv {res[1]:8.1f} us ({f(res[1]):7.1f} GB/s, frac {f(res[1])/8000:.3f})"
      f"   mean frac {(f(res[0]) + f(res[1])) / 16000:.3f}   per limb {res[0]/rows*1e3:.1f} / {res[1]/rows*1e3:.1f} ns")
