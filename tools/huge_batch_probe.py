"""A very large batch of very small ciphertexts (grid dimensions past 65535 in every launcher): multiply + relinearize + rotate of `batch` distinct
ciphertexts at N = 64 / 256; the first, the last and a few in between against the CPU oracle.  usage: python tools/huge_batch_probe.py [batch = 70001]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import troy_amd as ta  # noqa: E402
import cases  # noqa: E402
from oracle import ref as R  # noqa: E402
from troy_amd import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 70001
ta.KernelProvider.initialize(0)
bad = 0
SHAPES = ((cases.BFV, 64, [40, 40, 40]), (cases.CKKS, 256, [50, 40, 50]), (cases.BGV, 64, [36, 36, 40]))
if len(sys.argv) > 2:  # python tools/huge_batch_probe.py <batch> <N>: one more ring degree, all three schemes
    n_ = int(sys.argv[2])
    SHAPES = ((cases.BFV, n_, [40, 40, 40]), (cases.CKKS, n_, [50, 40, 50]), (cases.BGV, n_, [36, 36, 40]))
for scheme, N, bits in SHAPES:
    cfg = dict(scheme=scheme, N=N, bits=bits, tbits=14 if N <= 256 else 20)
    t0 = time.time()
    be = cases.GpuBackend(cfg, batch=B)
    orc = cases.oracle_backend(cfg)
    primes = be.primes
    L = len(primes) - 1
    rk, gk = synth.uniform_kswitch_key(7, primes, N), synth.uniform_kswitch_key(8, primes, N)
    e1 = be.elt_from_step(1)
    for side in (be, orc):
        side.set_relin_key(rk)
        side.set_galois_key(e1, gk)
    ntt = scheme == cases.CKKS
    xa, xb = synth.uniform_ct(11, primes[:L], 2, N, B), synth.uniform_ct(12, primes[:L], 2, N, B)
    a = be.api.Ciphertext.from_numpy(be.ctx, xa, ntt, 1.0, 1, capacity=3)
    b = be.api.Ciphertext.from_numpy(be.ctx, xb, ntt, 1.0, 1, capacity=3)
    m = be.ev.multiply(a, b)
    be.ev.relinearizeInplace(m, be.rlk)
    be.rotate(m, 1)
    got = m.cpu()
    op = R.OP_ROTATE_VECTOR if ntt else R.OP_ROTATE_ROWS
    for i in sorted({0, 1, 65534, 65535, 65536, B // 2, B - 2, B - 1} & set(range(B))):
        exp = orc.impl.eval(op, orc.impl.eval(R.OP_RELIN, orc.impl.eval(R.OP_MULTIPLY, R.Ct(xa[i], ntt), R.Ct(xb[i], ntt))), iarg=1)
        if not np.array_equal(got[i][:2], exp.data):
            bad += 1
            print("MISMATCH scheme", scheme, "item", i)
    print(f"scheme {scheme} N={N} batch={B}: multiply + relinearize + rotate, 8 items against the oracle, {time.time() - t0:.1f} s")
print("failures:", bad)
sys.exit(1 if bad else 0)
