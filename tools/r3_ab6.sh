#!/bin/bash
# same-box A/B of a previous build (tools/probe_libs/libtroyhip_prev.so via TROYHIP_LIB) against the current one over the FP64 workloads
mkdir -p gpurun_out/r3
for a in prev cur prev cur; do
  for wl in ckks_n32768_chain bgv_n65536_relin_rot bfv_n8192_l4; do
    if [ $a = prev ]; then export TROYHIP_LIB=$PWD/tools/probe_libs/libtroyhip_prev.so; else unset TROYHIP_LIB; fi
    python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3/ab6_${a}_${wl}.json 2> gpurun_out/r3/ab6_${a}_${wl}.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r3/ab6_${a}_${wl}.json").read().strip().split("\n")[-1])
    print("${a} ${wl}", d["value"], d["unit"], "verified", d["verified"], "frac", d["roofline"]["frac"], "build", d.get("build_id"))
except Exception as e:
    print("${a} ${wl} FAILED", e); print(open("gpurun_out/r3/ab6_${a}_${wl}.err").read()[-1500:])
PY
  done
done
