"""Parity soak (development tool): random parameter sets beyond the seeds of tests/test_gpu_parity.py, product vs CPU oracle limb for limb.
usage: python tools/random_soak.py <first seed> <count> [large] [narrow]
(large: N = 4096 .. 65536, light scenario; narrow: every prime of 34 .. 49 bits, the range of the FP64 butterflies and the FP64 BEHZ kernels)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import troy_amd as ta  # noqa: E402
import cases  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
large = "large" in sys.argv[3:]
pool = [34, 36, 40, 45, 49] if "narrow" in sys.argv[3:] else None
ta.KernelProvider.initialize(0)
bad = 0
ran = ops = 0
t0 = time.time()
for seed in range(first, first + count):
    try:
        if large:
            cfg, n = cases.check_random_config(seed, sizes=(4096, 8192, 16384, 32768, 65536), batch=1, light=True, pool=pool)
        else:
            cfg, n = cases.check_random_config(seed, pool=pool)
        if n is not None:
            ran += 1
            ops += n
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, str(e)[:400], flush=True)
    except Exception as e:
        bad += 1
        print("ERROR seed", seed, type(e).__name__, str(e)[:300], flush=True)
print(f"{count} seeds from {first}{' (large)' if large else ''}{' (narrow primes)' if pool else ''}: {ran} parameter sets accepted, {ops} results compared, {bad} failures, {time.time() - t0:.0f} s")
