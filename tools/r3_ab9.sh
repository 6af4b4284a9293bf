#!/bin/bash
# same-box A/B of probe builds (tools/probe_libs/libtroyhip_<v>.so) over the workloads with a digit-expanding first pass
for v in ${VARIANTS:-base pffp base pffp}; do
  export TROYHIP_LIB=$PWD/tools/probe_libs/libtroyhip_$v.so
  for wl in ckks_n32768_chain bgv_n65536_relin_rot bfv_n32768_l14 bfv_n8192_l4; do
    timeout 600 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --ntt-reps 2 2>/tmp/ab9.err | tail -1 > /tmp/ab9.json
    python - "$v" "$wl" <<'PY' || tail -3 /tmp/ab9.err
import json, sys
d = json.loads(open("/tmp/ab9.json").read())
print(sys.argv[1], sys.argv[2], d["value"], "verified", d["verified"])
for k in d["roofline"]["per_kernel"]:
    n = k["name"]
    if ("_kernel<0, 1," in n) and (n.endswith("1, 0>") or n.endswith("2, 0>")): print("    %-44s x%-3d %9.1f us" % (n, k["calls"], k["us"]))
PY
  done
done
