#!/bin/bash
# removal probes of the two-pass kernels inside the CKKS chain and the BGV workload (tools/ntt_probe.sh builds: exp1 = no HBM traffic,
# exp2 = no LDS exchange, exp4 = no butterflies): per-kernel time of the key-switch passes; results are garbage by construction (--no-verify)
for v in base exp1 exp2 exp4; do
  export TROYHIP_LIB=$PWD/tools/probe_libs/libtroyhip_$v.so
  for wl in ckks_n32768_chain bgv_n65536_relin_rot; do
    timeout 600 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-verify --ntt-reps 2 2>/tmp/pk.err | tail -1 > /tmp/pk.json
    python - "$v" "$wl" <<'PY' || tail -3 /tmp/pk.err
import json, sys
d = json.loads(open("/tmp/pk.json").read())
print(sys.argv[1], sys.argv[2], d["value"])
for k in d["roofline"]["per_kernel"]:
    if "ntt2" in k["name"] and k["us"] > 300: print("    %-44s x%-3d %9.1f us" % (k["name"], k["calls"], k["us"]))
PY
  done
done
