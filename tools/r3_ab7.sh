#!/bin/bash
# same-box A/B: previous build (tools/probe_libs/libtroyhip_prev.so) against the current one -- plain single-pass transform (58- and 40-bit primes) and
# the workloads that use the single-pass forward kernels
mkdir -p gpurun_out/r3
for a in prev cur prev cur; do
  if [ $a = prev ]; then export TROYHIP_LIB=$PWD/tools/probe_libs/libtroyhip_prev.so; else unset TROYHIP_LIB; fi
  echo "== $a"; python tools/ntt1_probe.py 128 6 2>&1 | tail -1
  PROBE_BITS="[60] + [40] * 13 + [60]" python tools/ntt1_probe.py 128 6 2>&1 | tail -1
  for wl in ckks_n32768_chain bfv_n32768_l14; do
    python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3/ab7_${a}_${wl}.json 2> gpurun_out/r3/ab7_${a}_${wl}.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r3/ab7_${a}_${wl}.json").read().strip().split("\n")[-1])
    print("${a} ${wl}", d["value"], d["unit"], "verified", d["verified"], "frac", d["roofline"]["frac"])
    for k in d["roofline"]["per_kernel"]:
        if "ntt1_fwd" in k["name"]: print("    %-44s x%-3d %9.1f us  frac %s" % (k["name"], k["calls"], k["us"], k.get("frac")))
except Exception as e:
    print("${a} ${wl} FAILED", e); print(open("gpurun_out/r3/ab7_${a}_${wl}.err").read()[-1500:])
PY
  done
done
