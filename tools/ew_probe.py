"""Element-wise op rates (add, negate over 128 CKKS ciphertexts of N = 2^15, 14 limbs): TB/s of compulsory traffic; TROYHIP_LIB=<probe library> compares builds."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import troy_amd as ta
from troy_amd import capi
lib = capi.load()
ta.KernelProvider.initialize(0)
N, bits, B = 32768, [60] + [40] * 13 + [60], 128
primes = ta.CoeffModulus.Create(N, bits)
ctx = ta.SEALContext(2, N, primes, 0)
L = len(primes) - 1
ev = ta.Evaluator(ctx)
a = ta.Ciphertext(ctx, B, 2, L, True, 2.0 ** 40, 1); ctx.fill_uniform(a.buf, B * 2 * L, primes[:L], seed=1)
b = ta.Ciphertext(ctx, B, 2, L, True, 2.0 ** 40, 1); ctx.fill_uniform(b.buf, B * 2 * L, primes[:L], seed=2)
def timeit(f, reps=20):
    f(); ta.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    ta.synchronize()
    return (time.perf_counter() - t0) / reps
sa, sb = a.struct(), b.struct()
t = timeit(lambda: capi.check(lib, lib.troyhip_add(ctx.h, C.byref(sa), C.byref(sb), C.c_uint64(B), None)))
byt = 3 * B * 2 * L * N * 8
print("add        %8.1f us  %6.2f TB/s (%.3f of 8)" % (t * 1e6, byt / t / 1e12, byt / t / 8e12))
t = timeit(lambda: capi.check(lib, lib.troyhip_negate(ctx.h, C.byref(sa), C.c_uint64(B), None)))
byt = 2 * B * 2 * L * N * 8
print("negate     %8.1f us  %6.2f TB/s (%.3f of 8)" % (t * 1e6, byt / t / 1e12, byt / t / 8e12))
