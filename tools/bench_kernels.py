"""one bench line reduced to: workload, value, verified, and the per-step time of the kernels whose name contains one of the given substrings
(same-box A/B of a kernel inside its workload: tools/ab.sh <tag> cur <variant> -- python tools/bench_kernels.py <workload> <substring> ...)
usage: python tools/bench_kernels.py <workload> [--no-verify] <substring> [<substring> ..]"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
wl = args.pop(0)
extra = []
if args and args[0] == "--no-verify":
    extra.append(args.pop(0))
r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", wl, "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--ntt-reps", "2"] + extra, capture_output=True, text=True)
lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
if not lines:
    print(r.stdout[-1500:], r.stderr[-1500:])
    sys.exit(1)
d = json.loads(lines[-1])
print(wl, d["value"], d.get("verified"))
for k in d["roofline"].get("per_kernel", []):
    if any(sub in k["name"] for sub in args):
        print(f"  {k['name']:54s} {k['us']:8.1f} us")
