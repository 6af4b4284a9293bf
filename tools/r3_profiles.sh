#!/bin/bash
# round-3 artefacts for profiles/ (README there): bench lines of every workload, rocprofv3 kernel-stat summaries of every workload and of the
# roofline launches, SQ counters of the default step, PMC HBM traffic stamped with the build id.  usage (on the GPU box): tools/r3_profiles.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
# PMC traffic FIRST: the bench lines below then carry it (the file is tied to this build by its build id)
bash tools/measure_traffic.sh 256 > $O/traffic.log 2>&1; tail -16 $O/traffic.log; cp gpurun_out/r03_traffic.json $O/ 2>/dev/null; cp gpurun_out/r03_traffic.json profiles/r03_traffic.json 2>/dev/null
for W in bfv_n32768_l14 bfv_n8192_l4 ckks_n32768_chain bgv_n65536_relin_rot ckks_matmul_128; do
  timeout 900 python3 bench.py --workload $W 2> $O/bench_$W.err | tail -1 > $O/r03_bench_$W.json
  cut -c1-200 $O/r03_bench_$W.json
done
cd /tmp; export TMPDIR=/tmp
for W in bfv_n32768_l14 bfv_n8192_l4 ckks_n32768_chain bgv_n65536_relin_rot ckks_matmul_128; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -o p -- python3 $R/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-verify > $O/prof_$W.log 2>&1
  f=$(find $O/prof_$W -name 'p_kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/r03_${W}_kernel_stats.csv && echo "== $W" && python3 $R/tools/kstats.py $f 14
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roofline -o p -- python3 $R/bench.py --roofline-only --no-cpu-baseline > $O/prof_roofline.log 2>&1
f=$(find $O/prof_roofline -name 'p_kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r03_roofline_kernel_stats.csv && python3 $R/tools/kstats.py $f 8
t=$(find $O/prof_roofline -name 'p_kernel_trace.csv' | head -1); [ -n "$t" ] && python3 $R/tools/trace_steady.py $t ntt1_ 3 80 | tee $O/r03_roofline_trace_steady.txt
tail -1 $O/prof_roofline.log | cut -c1-700
# SQ counters (own passes, kernel trace only): the headline step and the CKKS chain step (FP64 instances)
for W in bfv_n32768_l14 ckks_n32768_chain; do
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_sq_$W -o p -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --batch 32 --streams 1 --ntt-reps 2 --no-cpu-baseline --no-per-kernel --no-verify > $O/pmc_sq_$W.log 2>&1
  f=$(find $O/pmc_sq_$W -name 'p_counter_collection.csv' | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f | grep -v rocclr | cut -c1-520 > $O/r03_pmc_sq_$W.txt
  cut -c1-260 $O/r03_pmc_sq_$W.txt | head -16
done
cd $R
