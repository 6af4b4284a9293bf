"""Many primes (the reference allows up to 64: SEAL_COEFF_MOD_COUNT_MAX, src/utils/defines.h): the light scenario (multiply, relinearize, rotate /
rescale at the first level) at K = 20, 33, 48, 64 primes, product vs CPU oracle limb for limb.  usage: python tools/max_limbs_probe.py [N = 1024] [K list, e.g. 48,64]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import troy_amd as ta  # noqa: E402
import cases  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ta.KernelProvider.initialize(0)
bad = 0
for scheme in (cases.BFV, cases.CKKS, cases.BGV):
    for K in ([int(k) for k in sys.argv[2].split(',')] if len(sys.argv) > 2 else (20, 33, 48, 64)):
        cfg = dict(scheme=scheme, N=N, bits=[40] * (K - 1) + [45], tbits=16 if N <= 1024 else 20)
        t0 = time.time()
        try:
            got = cases.scenario(cases.GpuBackend(cfg, batch=2), cfg, light=True)
            exp = cases.scenario(cases.oracle_backend(cfg), cfg, light=True)
            diff = cases.compare(got, exp)
            assert not diff, diff[:4]
            print(f"scheme {scheme} K={K}: {len(got)} results equal, {time.time() - t0:.1f} s", flush=True)
        except Exception as e:
            bad += 1
            print(f"FAILED scheme {scheme} K={K}:", type(e).__name__, str(e)[:300], flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
