#!/bin/bash
# HBM traffic per kernel from PMC counters (MI355X_MICROARCH.md section HBM): FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (TCC slots), kernel-trace only, over the DEFAULT bench command at the given batch.  Writes profiles/r03_traffic.json.
# usage (on the GPU box): tools/measure_traffic.sh [batch]
B=${1:-128}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/traffic_r03_$C -o p -- python3 $R/bench.py --steps 1 --warmup 0 --batch $B --streams 1 --no-roofline --no-cpu-baseline --no-per-kernel --no-verify > $R/gpurun_out/traffic_r03_$C.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/traffic_r03sp_$C -o p -- python3 $R/bench.py --roofline-only --batch $B --ntt-reps 2 --no-cpu-baseline > $R/gpurun_out/traffic_r03sp_$C.log 2>&1
  TROYHIP_NTT=twopass timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/traffic_r03tp_$C -o p -- python3 $R/bench.py --roofline-only --batch $B --ntt-reps 2 --no-cpu-baseline > $R/gpurun_out/traffic_r03tp_$C.log 2>&1
done
cd $R
B=$B python3 - <<'PY'
import csv, json, collections, glob, re, os
B = int(os.environ['B'])
def load(tag):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"gpurun_out/traffic_{tag}_{C}/**/p_counter_collection.csv", recursive=True) + glob.glob(f"gpurun_out/traffic_{tag}_{C}/p_counter_collection.csv")
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] != C: continue
            name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].replace("troyhip::", "")
            out[name][C].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return out
N, L, K, nb = 32768, 14, 15, 15
P = 8.0 * N
res = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/measure_traffic.sh) over `bench.py --batch %d --streams 1 --steps 1`; "
                 "FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md section HBM (gfx950 tallies 128-B requests as 64 B); counter unit KiB" % B,
       "source": "profiles/r03_traffic.json (tools/measure_traffic.sh %d, this build)" % B, "N": N, "batch": B, "per_kernel": {}, "hbm_bytes_per_limb_transform": {}}
op = load("r03")
for name, v in op.items():
    if not v.get("FETCH_SIZE") or not v.get("WRITE_SIZE"): continue
    f = sum(x[1] for x in v["FETCH_SIZE"]) * 1024 * 2
    w = sum(x[1] for x in v["WRITE_SIZE"]) * 1024
    res["per_kernel"][name] = {"calls": len(v["FETCH_SIZE"]), "fetch_bytes_corrected": f, "write_bytes": w, "hbm_bytes": f + w}
# the roofline launches: rows = B * (L+1) * L per transform; ntt1 = 1 launch (lean + guarded forward kernels together) per transform
rows = min(B, 128) * (L + 1) * L  # bench.py:ntt_roofline caps the launch at the rows of 128 ciphertexts
def per_row(d, names, launches_each):
    tot = 0.0
    for n in names:
        v = d.get(n)
        if not v: return None
        # take the calls whose grid matches the roofline shape: the LAST launches_each calls of each counter (the roofline runs after the step)
        f = sum(x[1] for x in v["FETCH_SIZE"][-launches_each:]) * 1024 * 2 / launches_each
        w = sum(x[1] for x in v["WRITE_SIZE"][-launches_each:]) * 1024 / launches_each
        tot += f + w
    return tot / rows
sp = load("r03sp")
n1 = per_row(sp, ["ntt1_inv_kernel<true, false>", "ntt1_inv_kernel<false, false>"], 3)
n1f = None
if "ntt1_fwd_kernel<true, false>" in sp and "ntt1_fwd_kernel<false, false>" in sp:
    n1f = per_row(sp, ["ntt1_fwd_kernel<true, false>", "ntt1_fwd_kernel<false, false>"], 3)
if n1 and n1f:
    res["hbm_bytes_per_limb_transform"]["ntt1"] = (n1 + n1f) / 2
    res["hbm_bytes_per_limb_transform"]["ntt1_forward"] = n1f
    res["hbm_bytes_per_limb_transform"]["ntt1_inverse"] = n1
tp = load("r03tp")
names = [k for k in tp if k.startswith("ntt2_kernel")]
if len(names) == 4:
    tot = 0.0
    for n in names:
        v = tp[n]
        tot += (sum(x[1] for x in v["FETCH_SIZE"]) * 2048 + sum(x[1] for x in v["WRITE_SIZE"]) * 1024) / len(v["FETCH_SIZE"])
    res["hbm_bytes_per_limb_transform"]["ntt2"] = tot / 2 / rows
res["algorithmic_bytes_per_limb_transform"] = 16 * N
import sys
sys.path.insert(0, ".")
from troy_amd import capi
res["build_id"] = capi.build_id()  # bench.py ignores this file when the loaded library is another build
json.dump(res, open("gpurun_out/r03_traffic.json", "w"), indent=1)
for k, v in sorted(res["per_kernel"].items(), key=lambda kv: -kv[1]["hbm_bytes"])[:14]:
    print(f"{k[:60]:60s} calls={v['calls']:3d} fetch={v['fetch_bytes_corrected']/1e9:8.3f} GB write={v['write_bytes']/1e9:8.3f} GB")
print(res["hbm_bytes_per_limb_transform"], "algorithmic", 16 * N)
PY
