#!/bin/bash
# bench.py over its flag space (workload x batch x streams): every line must verify and account for every kernel
for wl in bfv_n32768_l14 bfv_n32768_l14_p49 bfv_n8192_l4 ckks_n32768_chain bgv_n65536_relin_rot ckks_matmul_128; do
  for b in 0 1 8 48; do
    for s in 1 2 3; do
      out=$(timeout 600 python bench.py --workload $wl --batch $b --streams $s --steps 2 --warmup 1 --no-cpu-baseline --ntt-reps 2 2>/tmp/bm.err | tail -1)
      python - "$wl" "$b" "$s" <<PY || { echo "FAILED $wl batch $b streams $s"; tail -3 /tmp/bm.err; }
import json, sys
d = json.loads('''$out''')
miss = [k["name"] for k in d["roofline"].get("per_kernel", []) if not k.get("frac")]
print("%-22s batch %-5s streams %s -> %10.1f %s verified %s%s" % (sys.argv[1], d["config"]["batch_per_gpu"], d["config"]["streams_per_gpu"], d["value"], d["unit"], d["verified"], (" UNACCOUNTED " + str(miss)) if miss else ""))
assert d["verified"] is True
PY
    done
  done
done
