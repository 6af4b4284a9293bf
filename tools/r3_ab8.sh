#!/bin/bash
# BEHZ kernel forms at configs[1] (BFV N = 8192, L = 4): the matrix-core form against the VALU kernels
mkdir -p gpurun_out/r3
for a in default valu default valu; do
  if [ $a = valu ]; then export TROYHIP_BEHZ=valu; else unset TROYHIP_BEHZ; fi
  python bench.py --workload bfv_n8192_l4 --steps 50 --warmup 5 --no-cpu-baseline $([ "$a" = valu ] && echo --no-per-kernel) > gpurun_out/r3/ab8_${a}.json 2> gpurun_out/r3/ab8_${a}.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r3/ab8_${a}.json").read().strip().split("\n")[-1])
    print("${a}", d["value"], d["unit"], "verified", d["verified"])
    for k in d["roofline"].get("per_kernel", []):
        if "behz" in k["name"]: print("    %-44s x%-3d %9.1f us  frac %s" % (k["name"], k["calls"], k["us"], k.get("frac")))
except Exception as e:
    print("${a} FAILED", e); print(open("gpurun_out/r3/ab8_${a}.err").read()[-1500:])
PY
done
