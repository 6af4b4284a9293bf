#!/bin/bash
# HBM traffic per kernel from PMC counters (MI355X_MICROARCH.md section HBM): FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), kernel-trace
# only, over ONE step of a workload's bench command (per_kernel[].traffic) and over its roofline launches (roofline.traffic).
# Writes gpurun_out/${ROUND}_traffic_<workload>.json, stamped with the library's build id; copy it to profiles/ to have bench.py use it.
# usage (on the GPU box): tools/measure_traffic.sh <workload> [batch]
ROUND=${ROUND:-r06}
WL=${1:-bfv_n32768_l14}
B=${2:-0}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/traffic_${WL}_op_$C $R/gpurun_out/traffic_${WL}_rf_$C
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/traffic_${WL}_op_$C -o p -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --batch $B --streams 1 --no-roofline --no-cpu-baseline --no-per-kernel --no-verify > $R/gpurun_out/traffic_${WL}_op_$C.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/traffic_${WL}_rf_$C -o p -- python3 $R/bench.py --workload $WL --roofline-only --batch $B --ntt-reps 2 --no-cpu-baseline > $R/gpurun_out/traffic_${WL}_rf_$C.log 2>&1
done
cd $R
WL=$WL ROUND=$ROUND python3 - <<'PY'
import csv, json, collections, glob, re, os, sys
WL = os.environ["WL"]
ROUND = os.environ.get("ROUND", "r06")
def load(tag):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"gpurun_out/traffic_{WL}_{tag}_{C}/**/p_counter_collection.csv", recursive=True) + glob.glob(f"gpurun_out/traffic_{WL}_{tag}_{C}/p_counter_collection.csv")
        if not f: continue
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] != C: continue
            name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].replace("troyhip::", "")
            out[name][C].append(float(r["Counter_Value"]))
    return out
def line_of(tag):  # the JSON line bench.py printed in that pass (either counter's log)
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            for ln in reversed(open(f"gpurun_out/traffic_{WL}_{tag}_{C}.log").read().splitlines()):
                if ln.startswith("{") and '"metric"' in ln: return json.loads(ln)
        except Exception: pass
    return None
op_line, rf_line = line_of("op"), line_of("rf")
cfg = (op_line or rf_line or {}).get("config", {})
res = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/measure_traffic.sh) over `bench.py --workload %s --streams 1 --steps 1`; "
                 "FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md section HBM (gfx950 tallies 128-B requests as 64 B); counter unit KiB" % WL,
       "source": "profiles/%s_traffic_%s.json (tools/measure_traffic.sh %s, this build)" % (ROUND, WL, WL), "workload": WL, "N": cfg.get("N"), "batch": cfg.get("batch_per_gpu"),
       "per_kernel": {}, "hbm_bytes_per_limb_transform": {}}
for name, v in load("op").items():
    if not v.get("FETCH_SIZE") or not v.get("WRITE_SIZE"): continue
    f = sum(v["FETCH_SIZE"]) * 1024 * 2
    w = sum(v["WRITE_SIZE"]) * 1024
    res["per_kernel"][name] = {"calls": len(v["FETCH_SIZE"]), "fetch_bytes_corrected": f, "write_bytes": w, "hbm_bytes": f + w}
# the roofline launches (bench.py:ntt_roofline): 3 forward + 3 inverse transforms of `rows` limb rows each (warm-up pair, instrumented pair, 2 timed);
# every kernel of the run whose name says it is a transform kernel counts
if rf_line and rf_line.get("roofline") and rf_line["roofline"].get("limb_transforms_per_launch"):
    rows = rf_line["roofline"]["limb_transforms_per_launch"]
    tot = {"fwd": 0.0, "inv": 0.0}
    for name, v in load("rf").items():
        if not (name.startswith("ntt1") or name.startswith("ntt2") or name.startswith("ntt_")): continue
        b = sum(v.get("FETCH_SIZE", [])) * 2048 + sum(v.get("WRITE_SIZE", [])) * 1024
        inv = "_inv_" in name or re.match(r"ntt2(_fp)?_kernel<1,", name)
        tot["inv" if inv else "fwd"] += b
    single = any(n.startswith("ntt1") for n in load("rf"))
    key = "ntt1" if single else "ntt2"
    res["hbm_bytes_per_limb_transform"][key] = (tot["fwd"] + tot["inv"]) / (6.0 * rows)
    res["hbm_bytes_per_limb_transform"][key + "_forward"] = tot["fwd"] / (3.0 * rows)
    res["hbm_bytes_per_limb_transform"][key + "_inverse"] = tot["inv"] / (3.0 * rows)
    res["roofline_rows"] = rows
res["algorithmic_bytes_per_limb_transform"] = 16 * (cfg.get("N") or 0)
sys.path.insert(0, ".")
from troy_amd import capi
res["build_id"] = capi.build_id()  # bench.py ignores this file when the loaded library is another build
json.dump(res, open(f"gpurun_out/{ROUND}_traffic_{WL}.json", "w"), indent=1)
for k, v in sorted(res["per_kernel"].items(), key=lambda kv: -kv[1]["hbm_bytes"])[:14]:
    print(f"{k[:64]:64s} calls={v['calls']:3d} fetch={v['fetch_bytes_corrected']/1e9:8.3f} GB write={v['write_bytes']/1e9:8.3f} GB")
print(res["hbm_bytes_per_limb_transform"], "algorithmic", res["algorithmic_bytes_per_limb_transform"])
PY
