"""Time BFV multiply (the BEHZ kernels' only caller) with a given build of the library.
usage: python tools/behz_probe.py <libtroyhip*.so> [batch]     (development tool; variants come from tools/behz_probe.sh)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import troy_amd as ta  # noqa: E402
from troy_amd import api, capi  # noqa: E402

path = os.path.abspath(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lib = capi.load(path)  # TROYHIP_BEHZ_FOLD=0 in the environment keeps the unfolded epilogue
api.KernelProvider.initialize(0, _lib=lib)
N, bits = 32768, [60] + [58] * 13 + [60]
primes = ta.CoeffModulus.Create(N, bits)
ctx = ta.SEALContext(capi.BFV, N, primes, ta.PlainModulus.Batching(N, 20))
L = len(primes) - 1
a = api.Ciphertext(ctx, B, 2, L)
b = api.Ciphertext(ctx, B, 2, L)
for i, ct in enumerate((a, b)):
    ctx.fill_uniform(ct.buf, B * 2 * L, primes[:L], seed=11 + i)
ev = api.Evaluator(ctx)
out = api.Ciphertext(ctx, B, 3, L, capacity=3)
timer = C.c_void_p()
capi.check(lib, lib.troyhip_timer_create(C.byref(timer)))
ev.multiply(a, b, out)
ta.synchronize()
reps = 8
capi.check(lib, lib.troyhip_timer_start(timer, None))
for _ in range(reps):
    ev.multiply(a, b, out)
capi.check(lib, lib.troyhip_timer_stop(timer, None))
ms = C.c_float()
capi.check(lib, lib.troyhip_timer_elapsed_ms(timer, C.byref(ms)))
print(f"{os.path.basename(path):40s} B={B}  multiply {ms.value * 1e3 / reps:9.1f} us per batch  ({ms.value * 1e3 / reps / B:7.2f} us per ciphertext pair)")
