#!/bin/bash
# deeper SQ breakdown of selected kernels (two counter sets). usage: tools/gpu_pmc2.sh <tag> [kernel-name-substring ...]
TAG=${1:-x}; shift; export PMC_FILTER="${*:-ntt2}"; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcA_$TAG -o p -- python3 $R/bench.py --steps 1 --warmup 0 --batch 16 --ntt-reps 2 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/pmcB_$TAG -o p -- python3 $R/bench.py --steps 1 --warmup 0 --batch 16 --ntt-reps 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, collections, os
flt = os.environ['PMC_FILTER'].split()
for S in ("A","B"):
    rows = list(csv.DictReader(open(f"gpurun_out/pmc{S}_$TAG/p_counter_collection.csv")))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur=collections.defaultdict(float); seen=set()
    for r in rows:
        if not any(f in r["Kernel_Name"] for f in flt): continue
        k = r["Kernel_Name"].split("(")[0][-34:] + "/g" + r["Grid_Size"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur[k] += (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    for k,v in agg.items():
        print(S, k, "us=%.0f"%dur[k], " ".join(f"{c.replace('SQ_','')}={x:.3g}" for c,x in sorted(v.items())))
PY
