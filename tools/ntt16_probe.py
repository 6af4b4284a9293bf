"""Per-kernel times of the plain two-pass transform at N = 2^16 (configs[3] shape) with a given build of the library.
usage: python tools/ntt16_probe.py <libtroyhip*.so> [batch] [logn]   (development tool; variants come from tools/ntt_probe.sh)"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import troy_amd as ta  # noqa: E402
from troy_amd import api, capi  # noqa: E402

path = os.path.abspath(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
logn = int(sys.argv[3]) if len(sys.argv) > 3 else 16
lib = capi.load(path)
api.KernelProvider.initialize(0, _lib=lib)
N, bits = 1 << logn, [60] + [50] * 13 + [60]
primes = ta.CoeffModulus.Create(N, bits)
ctx = ta.SEALContext(capi.BGV, N, primes, ta.PlainModulus.Batching(N, 20))
K, L = len(primes), len(primes) - 1
rows = B * K * L
D = ta.DeviceBuffer(rows * N)
out_primes = primes[:L] + [primes[K - 1]]
ctx.fill_uniform(D, rows, out_primes, seed=1, inner=L)
pr = np.array(out_primes, dtype=np.uint64)


def ntt(inv):
    capi.check(lib, lib.troyhip_ntt(ctx.h, C.c_void_p(D.ptr), C.c_uint64(rows), pr.ctypes.data_as(C.c_void_p), len(pr), L, inv, None))


for inv in (0, 1):
    ntt(inv)
ta.synchronize()
capi.check(lib, lib.troyhip_ktime_enable(1))
for _ in range(4):
    ntt(0)
    ntt(1)
ta.synchronize()
buf = C.create_string_buffer(1 << 16)
capi.check(lib, lib.troyhip_ktime_report(buf, C.c_size_t(len(buf))))
capi.check(lib, lib.troyhip_ktime_enable(0))
ks = json.loads(buf.value.decode())
gb = 16.0 * N * rows / 1e3
print(f"{os.path.basename(path):36s} N=2^{logn} B={B} rows={rows} " + "  ".join(f"{k['name'].replace('ntt2_kernel', '').strip()} {k['total_us'] / k['calls']:.0f}us ({gb * k['calls'] / k['total_us']:.0f} GB/s)" for k in ks))
