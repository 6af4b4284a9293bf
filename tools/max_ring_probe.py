"""The largest ring the reference allows (N = 2^17 = 131072, src/utils/defines.h) through multiply / relinearize / rotate / rescale at the first
level on random prime sets, product vs CPU oracle limb for limb.  usage: python tools/max_ring_probe.py [first seed = 1] [count = 6]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import troy_amd as ta  # noqa: E402
import cases  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ta.KernelProvider.initialize(0)
bad = 0
for seed in range(first, first + count):
    t0 = time.time()
    try:
        cfg, n = cases.check_random_config(seed, sizes=(131072,), batch=1, light=True)
        print(f"seed {seed}: scheme {cfg['scheme']} bits {cfg['bits']}: {n} results equal, {time.time() - t0:.1f} s", flush=True)
    except Exception as e:
        bad += 1
        print("FAILED seed", seed, type(e).__name__, str(e)[:400], flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
