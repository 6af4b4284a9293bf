#!/bin/bash
# One probe: does a launch small enough for its pass-1 output to stay in the 256 MiB Infinity Cache run the two-pass kernels faster per
# ciphertext?  bench.py per_kernel at batch 2 / 4 / 8 / 32 / 128 on one stream, every kernel of the step in us per ciphertext.   usage: tools/r6_ic_probe.sh [workload ..]
for wl in ${@:-bgv_n65536_relin_rot}; do
for b in 2 4 8 32 128; do
  python - $wl $b <<'PY'
import json, subprocess, sys
wl, b = sys.argv[1], sys.argv[2]
r = subprocess.run([sys.executable, "bench.py", "--workload", wl, "--batch", b, "--streams", "1", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--ntt-reps", "2", "--no-verify"], capture_output=True, text=True)
d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
print(f"{wl} B={int(b):4d}  {d['value']:9.1f} {d['unit']}")
for k in d["roofline"]["per_kernel"]:
    print(f"    {k['name']:44s} {k['us']:9.1f} us  = {k['us'] / int(b):8.2f} us per ciphertext")
PY
done; done
