"""Per-operation timings of the evaluator on one GPU (development tool): ops/s for a batch of B ciphertexts.
usage: python tools/op_bench.py [B]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402,F401
import troy_amd as ta  # noqa: E402
from troy_amd import capi  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ta.KernelProvider.initialize(0)


def timeit(f, reps=5):
    f()
    ta.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    ta.synchronize()
    return (time.perf_counter() - t0) / reps


for name, scheme, N, bits, tb in (("BFV  N=2^15 K=15", capi.BFV, 32768, [60] + [58] * 13 + [60], 20), ("CKKS N=2^15 K=15", capi.CKKS, 32768, [60] + [40] * 13 + [60], 0),
                                  ("BGV  N=2^16 K=15", capi.BGV, 65536, [60] + [50] * 13 + [60], 20), ("BFV  N=2^13 K=5 ", capi.BFV, 8192, [40, 36, 36, 36, 40], 20)):
    primes = ta.CoeffModulus.Create(N, bits)
    t = ta.PlainModulus.Batching(N, tb) if tb else 0
    ctx = ta.SEALContext(scheme, N, primes, t)
    K, L = len(primes), len(primes) - 1
    ev = ta.Evaluator(ctx)
    ntt = scheme == capi.CKKS
    a = ta.Ciphertext(ctx, B, 2, L)
    b = ta.Ciphertext(ctx, B, 2, L)
    ctx.fill_uniform(a.buf, B * 2 * L, primes[:L], seed=1)
    ctx.fill_uniform(b.buf, B * 2 * L, primes[:L], seed=2)
    a.is_ntt_form = b.is_ntt_form = ntt
    rlk, gk = ta.RelinKeys(ctx), ta.GaloisKeys(ctx)
    key = ta.DeviceBuffer((K - 1) * 2 * K * N)
    ctx.fill_uniform(key, (K - 1) * 2 * K, primes, seed=3)
    rlk.keys[0] = key
    g = ctx.galois_elt_from_step(1)
    gk.keys[ta.GaloisKeys.getIndex(g)] = key
    res = {}
    m = ev.multiply(a, b)
    res["multiply"] = timeit(lambda: ev.multiply(a, b))
    res["relinearize"] = timeit(lambda: ev.relinearize(m, rlk))
    r = ev.relinearize(m, rlk)
    rot = lambda: (ev.rotateVectorInplace if ntt else ev.rotateRowsInplace)(r, 1, gk)  # noqa: E731
    res["rotate(1)"] = timeit(rot)
    res["add"] = timeit(lambda: ev.addInplace(r, r))
    if scheme == capi.CKKS:
        res["rescale"] = timeit(lambda: ev.rescaleToNext(r))
    else:
        res["mod_switch"] = timeit(lambda: ev.modSwitchToNext(r))
    print(name, f"B={B}", "  ".join(f"{k} {B / v:9.0f}/s ({v * 1e3:7.3f} ms)" for k, v in res.items()), flush=True)
