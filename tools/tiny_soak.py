"""Parity soak at TINY ring degrees (N = 2 .. 128: every tile of every kernel is larger than a polynomial): random parameter sets, product vs CPU
oracle limb for limb.  usage: python tools/tiny_soak.py <first seed> <count>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import troy_amd as ta  # noqa: E402
import cases  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
ta.KernelProvider.initialize(0)
bad = ran = ops = 0
t0 = time.time()
for seed in range(first, first + count):
    try:
        cfg, n = cases.check_random_config(seed, sizes=(2, 4, 8, 16, 32, 64, 128), batch=3)
        if n is not None:
            ran += 1
            ops += n
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, str(e)[:400], flush=True)
    except Exception as e:
        bad += 1
        print("ERROR seed", seed, type(e).__name__, str(e)[:300], flush=True)
print(f"{count} seeds from {first} (N = 2 .. 128): {ran} parameter sets accepted, {ops} results compared, {bad} failures, {time.time() - t0:.0f} s")
