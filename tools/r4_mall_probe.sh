#!/bin/bash
# does the digit-expanding pass run faster when its output stays in the 256 MB Infinity Cache?  Its time per ciphertext at batch 2 .. 128 on one stream
# (bench.py per_kernel): D is 50-55 MB per ciphertext at N = 2^15.
for wl in ckks_n32768_chain bfv_n32768_l14; do
for b in 2 4 8 16 32 128; do
  python - $wl $b <<'PY'
import json, subprocess, sys
wl, b = sys.argv[1], sys.argv[2]
r = subprocess.run([sys.executable, "bench.py", "--workload", wl, "--batch", b, "--streams", "1", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--ntt-reps", "2", "--no-verify"], capture_output=True, text=True)
d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
for k in d["roofline"]["per_kernel"]:
    if "0, 1, 6, 5, 0, 1, 0" in k["name"] or "0, 1, 6, 5, 0, 2, 0" in k["name"] or "0, 0, 9, 0, 1, 0, 1" in k["name"] or "0, 0, 9, 0, 1, 0, 3" in k["name"]:
        print(f"{wl:20s} B={int(b):4d}  {k['name']:40s} {k['us']:9.1f} us  = {k['us'] / int(b):8.2f} us per ciphertext")
PY
done; done
