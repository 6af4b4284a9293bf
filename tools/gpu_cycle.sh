#!/bin/bash
# one GPU iteration: parity tests, bench, kernel stats, SQ counters.  usage: tools/gpu_cycle.sh <tag> [batch]
TAG=${1:-x}; B=${2:-16}
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest=$?; tail -2 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1; cut -c1-1400 gpurun_out/bench_$TAG.log
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o p -- python3 $R/bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_$TAG -o p -- python3 $R/bench.py --steps 1 --warmup 0 --batch 8 --ntt-reps 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_$TAG/p_kernel_stats.csv")))
for r in rows[:14]:
    print(f"{r['Name'][:72]:72s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} max_us={float(r['MaxNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
python3 tools/pmc_summary.py gpurun_out/pmc_$TAG/p_counter_collection.csv | cut -c1-400 | head -12
