#!/bin/bash
# HBM traffic of the NTT kernels from PMC counters (MI355X_MICROARCH.md section HBM): FETCH_SIZE and WRITE_SIZE in
# SEPARATE passes (TCC slots), kernel-trace only.  usage: tools/measure_traffic.sh <tag> [batch]
TAG=${1:-x}; B=${2:-16}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/traffic_${TAG}_$C -o p -- python3 $R/bench.py --steps 1 --warmup 0 --batch $B --ntt-reps 4 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python3 - <<PY
import csv, json, collections
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"gpurun_out/traffic_${TAG}_{C}/p_counter_collection.csv")))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] != C: continue
        agg[(r["Kernel_Name"].split("(")[0][-48:], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(f"{k[0]} grid={k[1]}", {})[C] = {"calls": len(v), "avg_kb": sum(v) / len(v)}
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", {}).get("avg_kb", 0))[:14]:
    print(k, {c: round(x["avg_kb"]) for c, x in v.items()}, "calls", {c: x["calls"] for c, x in v.items()})
json.dump(out, open("gpurun_out/traffic_${TAG}.json", "w"), indent=1)
PY
