"""Every rotation step of a small ring through the NAF path (evaluator_cuda.cu:2119-2176): keys for the power-of-two steps +-1, +-2, .. only, as
GaloisTool::getEltsAll provides; every step in (-N/2, N/2) and the column / conjugate rotation, product vs CPU oracle.  usage: python tools/rotate_sweep.py [N = 64]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import troy_amd as ta  # noqa: E402
import cases  # noqa: E402
from troy_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ta.KernelProvider.initialize(0)
bad = total = 0
for scheme, bits in ((cases.BFV, [40, 36, 40]), (cases.CKKS, [50, 40, 50]), (cases.BGV, [36, 36, 40])):
    cfg = dict(scheme=scheme, N=N, bits=bits, tbits=14 if N <= 256 else 20)
    be, orc = cases.GpuBackend(cfg, batch=2), cases.oracle_backend(cfg)
    primes = be.primes
    steps, s = [], 1
    while s < N // 2:
        steps += [s, -s]
        s *= 2
    for i, st in enumerate(steps):
        key = synth.uniform_kswitch_key(100 + i, primes, N)
        for side in (be, orc):
            side.set_galois_key(side.elt_from_step(st), key)
    key = synth.uniform_kswitch_key(99, primes, N)
    for side in (be, orc):
        side.set_galois_key(2 * N - 1, key)
    ntt = scheme == cases.CKKS
    x = synth.uniform_ct(5, primes[:-1], 2, N)[0]
    for st in list(range(-(N // 2) + 1, N // 2)):
        got, exp = be.export(be.rotate(be.ct(x, ntt), st)), orc.export(orc.rotate(orc.ct(x, ntt), st))
        total += 1
        if cases.compare({"r": got}, {"r": exp}):
            bad += 1
            print("MISMATCH scheme", scheme, "step", st)
    got, exp = be.export(be.conjugate(be.ct(x, ntt))), orc.export(orc.conjugate(orc.ct(x, ntt)))
    total += 1
    bad += bool(cases.compare({"c": got}, {"c": exp}))
print(f"N = {N}: {total} rotations in three schemes against the oracle, {bad} failures")
sys.exit(1 if bad else 0)
