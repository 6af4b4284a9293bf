#!/bin/bash
# same-box A/B of the FP64 key-switch instances: parity suite, then the workloads with TROYHIP_FP64=off / default
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3/gpu_parity.log 2>&1; tail -3 gpurun_out/r3/gpu_parity.log
for wl in ckks_n32768_chain bgv_n65536_relin_rot bfv_n8192_l4; do
  for fp in off on; do
    TROYHIP_FP64=$fp python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3/ab_${wl}_fp${fp}.json 2> gpurun_out/r3/ab_${wl}_fp${fp}.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r3/ab_${wl}_fp${fp}.json").read().strip().split("\n")[-1])
    print("${wl} fp=${fp}", d["value"], d["unit"], "ms/step", d["ms_per_step"])
    for k in d.get("roofline", {}).get("per_kernel", [])[:16]:
        print("    %-44s x%-3d %9.1f us  frac %s" % (k["name"], k["calls"], k["us"], k.get("frac")))
except Exception as e:
    print("${wl} fp=${fp} FAILED", e); print(open("gpurun_out/r3/ab_${wl}_fp${fp}.err").read()[-1500:])
PY
  done
done
