#!/bin/bash
# copy the judged summaries of tools/r3_profiles.sh from gpurun_out/final (scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
F=gpurun_out/final
cp $F/r03_bench_*.json $F/r03_*_kernel_stats.csv $F/r03_pmc_sq_*.txt $F/r03_roofline_trace_steady.txt profiles/ 2>/dev/null
[ -f gpurun_out/r03_traffic.json ] && cp gpurun_out/r03_traffic.json profiles/r03_traffic.json
ls -la profiles/ | grep r03
