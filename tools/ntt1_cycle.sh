#!/bin/bash
# one GPU iteration on the single-pass NTT: timing (both paths), kernel stats, SQ counters.  usage: tools/ntt1_cycle.sh <tag> [batch]
TAG=${1:-x}; B=${2:-64}
mkdir -p gpurun_out
python tools/ntt1_probe.py $B | tee gpurun_out/ntt1_$TAG.txt
TROYHIP_NTT=twopass python tools/ntt1_probe.py $B | tee -a gpurun_out/ntt1_$TAG.txt
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ntt1prof_$TAG -o p -- python3 $R/tools/ntt1_probe.py $B 3 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/ntt1pmc_$TAG -o p -- python3 $R/tools/ntt1_probe.py 16 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/ntt1pmc2_$TAG -o p -- python3 $R/tools/ntt1_probe.py 16 1 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/ntt1prof_$TAG/**/p_kernel_stats.csv", recursive=True) + glob.glob("gpurun_out/ntt1prof_$TAG/p_kernel_stats.csv"):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:8]:
        print(f"{r['Name'][:80]:80s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
    break
PY
for d in ntt1pmc_$TAG ntt1pmc2_$TAG; do f=$(find gpurun_out/$d -name 'p_counter_collection.csv' | head -1); [ -n "$f" ] && python3 tools/pmc_summary.py $f | grep -i ntt | cut -c1-600; done
