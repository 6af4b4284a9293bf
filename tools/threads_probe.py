"""Host threads: T threads, each with its OWN context, evaluator and HIP stream, run multiply + relinearize + rotate concurrently (ctypes drops the
GIL inside the library) -- every thread's result must equal what the same inputs give on one thread.  The library's shared state is the caching
device pool, the error slot and the per-launch timing; contexts are single-owner.  usage: python tools/threads_probe.py [T = 4] [reps = 12]"""
import ctypes as C
import hashlib
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import troy_amd as ta  # noqa: E402
from troy_amd import api, capi, synth  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ta.KernelProvider.initialize(0)
lib = capi.load()
N, bits, B = 8192, [50, 40, 40, 50], 8
primes = ta.CoeffModulus.Create(N, bits)
t = ta.PlainModulus.Batching(N, 20)
L, K = len(primes) - 1, len(primes)
rk, gk = synth.uniform_kswitch_key(1, primes, N), synth.uniform_kswitch_key(2, primes, N)


def work(scheme, seed, stream, out, idx):
    try:
        ctx = ta.SEALContext(scheme, N, primes, t if scheme != capi.CKKS else 0)
        ev = ta.Evaluator(ctx, stream=stream)
        rlk, gks = ta.RelinKeys(ctx), ta.GaloisKeys(ctx)
        rlk.set(0, rk)
        elt = ctx.galois_elt_from_step(1)
        gks.set_elt(elt, gk)
        ntt = scheme == capi.CKKS
        xa, xb = synth.uniform_ct(seed, primes[:L], 2, N, B), synth.uniform_ct(seed + 1, primes[:L], 2, N, B)
        h = hashlib.sha256()
        for _ in range(reps):
            a = api.Ciphertext.from_numpy(ctx, xa, ntt, capacity=3)
            b = api.Ciphertext.from_numpy(ctx, xb, ntt, capacity=3)
            m = ev.multiply(a, b)
            ev.relinearizeInplace(m, rlk)
            (ev.rotateVectorInplace if ntt else ev.rotateRowsInplace)(m, 1, gks)
            ta.synchronize(stream)
            h.update(m.cpu().tobytes())
        out[idx] = h.hexdigest()
    except Exception as e:  # noqa: BLE001
        out[idx] = "ERROR " + type(e).__name__ + " " + str(e)[:200]


jobs = [((capi.BFV, capi.CKKS, capi.BGV)[i % 3], 10 * i) for i in range(T)]
serial = [None] * T
for i, (scheme, seed) in enumerate(jobs):
    work(scheme, seed, None, serial, i)
streams = []
for _ in range(T):
    s = C.c_void_p()
    capi.check(lib, lib.troyhip_stream_create(C.byref(s)))
    streams.append(s)
par = [None] * T
th = [threading.Thread(target=work, args=(jobs[i][0], jobs[i][1], streams[i], par, i)) for i in range(T)]
for x in th:
    x.start()
for x in th:
    x.join()
bad = sum(1 for i in range(T) if par[i] != serial[i] or str(par[i]).startswith("ERROR"))
for i in range(T):
    print(f"thread {i} scheme {jobs[i][0]}: {'same' if par[i] == serial[i] else 'DIFFERENT'} {str(par[i])[:60]}")
print(f"{T} threads x {reps} repetitions: {bad} failures")
sys.exit(1 if bad else 0)
