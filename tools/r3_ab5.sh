#!/bin/bash
# same-box A/B of the auxiliary base (TROYHIP_AUX_BASE=reference / default) over the BFV workloads
mkdir -p gpurun_out/r3
for a in reference auto; do
  for wl in bfv_n32768_l14 bfv_n8192_l4; do
    TROYHIP_AUX_BASE=$a python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3/aux_${a}_${wl}.json 2> gpurun_out/r3/aux_${a}_${wl}.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r3/aux_${a}_${wl}.json").read().strip().split("\n")[-1])
    print("aux=${a} ${wl}", d["value"], d["unit"], "verified", d["verified"], "frac", d["roofline"]["frac"])
    for k in d["roofline"]["per_kernel"]:
        print("    %-44s x%-3d %9.1f us  frac %s" % (k["name"], k["calls"], k["us"], k.get("frac")))
except Exception as e:
    print("aux=${a} ${wl} FAILED", e); print(open("gpurun_out/r3/aux_${a}_${wl}.err").read()[-2500:])
PY
  done
done
