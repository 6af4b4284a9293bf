#!/bin/bash
# same-box A/B of probe builds (tools/probe_libs/): headline bench per build + the plain NTT probe
mkdir -p gpurun_out/r3
for so in tools/probe_libs/*.so; do
  n=$(basename $so .so)
  TROYHIP_LIB=$PWD/$so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify > gpurun_out/r3/ab2_$n.json 2> gpurun_out/r3/ab2_$n.err
  TROYHIP_LIB=$PWD/$so python tools/ntt1_probe.py 128 6 > gpurun_out/r3/ab2_$n.ntt1 2>&1
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r3/ab2_$n.json").read().strip().split("\n")[-1])
    ks = {k["name"]: k["us"] for k in d["roofline"]["per_kernel"]}
    print("$n", d["value"], "ops/s | tensor %.0f  ks1 %.0f ks2 %.0f inv<t,f> %.0f inv<f,f> %.0f inv<t,t> %.0f" % (ks.get("ntt2_kernel<0, 0, 9, 0, 1, 0, 2>", 0), ks.get("ntt2_kernel<0, 1, 6, 5, 0, 1, 0>", 0), ks.get("ntt2_kernel<0, 0, 9, 0, 1, 0, 1>", 0), ks.get("ntt1_inv_kernel<true, false>", 0), ks.get("ntt1_inv_kernel<false, false>", 0), ks.get("ntt1_inv_kernel<true, true>", 0)), "| roofline frac", d["roofline"]["frac"])
except Exception as e:
    print("$n FAILED", e); print(open("gpurun_out/r3/ab2_$n.err").read()[-800:])
PY
  tail -1 gpurun_out/r3/ab2_$n.ntt1
done
