#!/bin/bash
# GPU suite + same-box A/B of the FP64 instances (TROYHIP_FP64=off / default) over the workloads with narrow primes
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu > gpurun_out/r3/gpu_suite.log 2>&1; tail -4 gpurun_out/r3/gpu_suite.log
for wl in bfv_n8192_l4 ckks_n32768_chain bgv_n65536_relin_rot bfv_n32768_l14; do
  for fp in off on; do
    TROYHIP_FP64=$fp python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3/ab3_${wl}_fp${fp}.json 2> gpurun_out/r3/ab3_${wl}_fp${fp}.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r3/ab3_${wl}_fp${fp}.json").read().strip().split("\n")[-1])
    print("${wl} fp=${fp}", d["value"], d["unit"], "ms/step", d["ms_per_step"], "verified", d["verified"], "roofline frac", d["roofline"]["frac"])
    for k in d.get("roofline", {}).get("per_kernel", []):
        print("    %-44s x%-3d %9.1f us  frac %s" % (k["name"], k["calls"], k["us"], k.get("frac")))
except Exception as e:
    print("${wl} fp=${fp} FAILED", e); print(open("gpurun_out/r3/ab3_${wl}_fp${fp}.err").read()[-1500:])
PY
  done
done
