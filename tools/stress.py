"""Determinism soak (development tool): the same batch through multiply + relinearize + rotate many times; every repetition must
give bit-identical limbs (a data race in a fused kernel, an LDS-DMA overwrite or an uninitialised scratch read would show up
as a differing hash).  usage: python tools/stress.py [reps] [batch]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import troy_amd as ta  # noqa: E402
from troy_amd import capi  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ta.KernelProvider.initialize(0)
for name, scheme, N, bits, tb in (("BFV", capi.BFV, 32768, [60] + [58] * 13 + [60], 20), ("CKKS", capi.CKKS, 32768, [60] + [40] * 13 + [60], 0), ("BGV", capi.BGV, 16384, [50, 45, 45, 50], 20)):
    primes = ta.CoeffModulus.Create(N, bits)
    t = ta.PlainModulus.Batching(N, tb) if tb else 0
    ctx = ta.SEALContext(scheme, N, primes, t)
    K, L = len(primes), len(primes) - 1
    ev = ta.Evaluator(ctx)
    ntt = scheme == capi.CKKS
    a, b = ta.Ciphertext(ctx, B, 2, L), ta.Ciphertext(ctx, B, 2, L)
    ctx.fill_uniform(a.buf, B * 2 * L, primes[:L], seed=1)
    ctx.fill_uniform(b.buf, B * 2 * L, primes[:L], seed=2)
    a.is_ntt_form = b.is_ntt_form = ntt
    rlk, gk = ta.RelinKeys(ctx), ta.GaloisKeys(ctx)
    key = ta.DeviceBuffer((K - 1) * 2 * K * N)
    ctx.fill_uniform(key, (K - 1) * 2 * K, primes, seed=3)
    rlk.keys[0] = key
    g = ctx.galois_elt_from_step(1)
    gk.keys[ta.GaloisKeys.getIndex(g)] = key
    first = None
    for i in range(reps):
        r = ev.multiply(a, b)
        ev.relinearizeInplace(r, rlk)
        (ev.rotateVectorInplace if ntt else ev.rotateRowsInplace)(r, 1, gk)
        h = hashlib.sha256(r.buf.to_numpy().tobytes()).hexdigest()
        if first is None:
            first = h
        assert h == first, f"{name}: repetition {i} differs"
    print(f"{name}: {reps} repetitions of multiply+relinearize+rotate on {B} ciphertexts, identical ({first[:16]})", flush=True)
