#!/bin/bash
# round-5 artefacts for profiles/ (profiles/README.md): PMC traffic of every workload (first: the bench lines then carry per_kernel[].traffic), the bench
# line of every workload (verified, CPU baseline on the headline and configs[1]), rocprofv3 --kernel-trace --stats summaries of every workload's command
# and of the roofline launches, the small-batch table through troyn.hpp, the reference's timetest on GPU and reference CPU, two ranks on one GPU.
# usage (GPU box): tools/r5_profiles.sh        -> gpurun_out/final/  (copy what is judged into profiles/ with: cp gpurun_out/final/r05_* profiles/)
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
WLS="bfv_n32768_l14 bfv_n8192_l4 ckks_n32768_chain bgv_n65536_relin_rot ckks_matmul_128 bfv_n32768_l14_p49"
for wl in $WLS; do
  tools/measure_traffic.sh $wl > $O/traffic_$wl.log 2>&1
  cp gpurun_out/r05_traffic_$wl.json $O/ 2>/dev/null && cp gpurun_out/r05_traffic_$wl.json profiles/
done
for wl in $WLS; do
  extra="--no-cpu-baseline"; case $wl in bfv_n32768_l14|bfv_n8192_l4) extra="";; esac
  python bench.py --workload $wl $extra > $O/r05_bench_$wl.json 2> $O/bench_$wl.err
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o p -- python3 $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-verify > $O/prof_$wl.log 2>&1)
  f=$(find $O/prof_$wl -name "p_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r05_${wl}_kernel_stats.csv
  rm -rf $O/prof_$wl
done
for wl in bfv_n32768_l14 bfv_n32768_l14_p49; do
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rf_$wl -o p -- python3 $R/bench.py --workload $wl --roofline-only --no-cpu-baseline > $O/prof_rf_$wl.log 2>&1)
  f=$(find $O/prof_rf_$wl -name "p_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r05_roofline_${wl}_kernel_stats.csv
  rm -rf $O/prof_rf_$wl
done
# SQ counters (own passes, kernel trace only) over the roofline launches of the headline and of its 49-bit twin, and over one step of configs[1]
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; SQ2="SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
for wl in bfv_n32768_l14 bfv_n32768_l14_p49; do
  : > $O/r05_pmc_sq_roofline_$wl.txt
  for set in "$SQ1" "$SQ2"; do
    rm -rf $O/pmc_$wl
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$wl -o p -- python3 $R/bench.py --workload $wl --roofline-only --batch 32 --ntt-reps 4 --no-cpu-baseline > $O/pmc_$wl.log 2>&1)
    f=$(find $O/pmc_$wl -name "p_counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f | grep -v "fill_uniform\|copyBuffer" >> $O/r05_pmc_sq_roofline_$wl.txt
  done
  rm -rf $O/pmc_$wl
done
: > $O/r05_pmc_sq_bfv_n8192_l4.txt
for set in "$SQ1" "$SQ2"; do
  rm -rf $O/pmc_c1
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_c1 -o p -- python3 $R/bench.py --workload bfv_n8192_l4 --steps 1 --warmup 0 --batch 256 --streams 1 --no-roofline --no-cpu-baseline --no-per-kernel --no-verify > $O/pmc_c1.log 2>&1)
  f=$(find $O/pmc_c1 -name "p_counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f | grep -v "fill_uniform\|copyBuffer" >> $O/r05_pmc_sq_bfv_n8192_l4.txt
done
rm -rf $O/pmc_c1
# kernel timeline of ONE ciphertext (and of eight) at the headline and configs[1]
{ for a in "bfv_n32768_l14 1" "bfv_n32768_l14 8" "bfv_n8192_l4 1" "bfv_n8192_l4 8"; do echo "## $a"; tools/r4_b1_timeline.sh $a final 2>/dev/null | grep -v simple_timer; done; } > $O/r05_b1_timeline.txt
rm -rf gpurun_out/b1tl_*_final
tools/r4_small_batch.sh final > /dev/null 2>&1; cp $O/small_batch.txt $O/r05_small_batch.txt
tools/r5_timetest.sh final > /dev/null 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/dist_two_ranks.py bfv_n32768_l14 64 2>$O/dist.err | tail -1 > $O/r05_dist_two_ranks.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 tools/dist_two_ranks.py bfv_n8192_l4 256 2>>$O/dist.err | tail -1 >> $O/r05_dist_two_ranks.txt
# VALU issue rate of every kernel of one lane's step (headline and its 49-bit twin)
: > $O/r05_valu_rate.txt
for wl in bfv_n32768_l14 bfv_n32768_l14_p49 ckks_n32768_chain; do echo "== $wl" >> $O/r05_valu_rate.txt; tools/r4_valu_rate.sh $wl 128 2>/dev/null | grep -v "^==" >> $O/r05_valu_rate.txt; done
# the reference's own GPU acceptance tests and apps, unchanged, on the device
{ for b in troytest timetest linear linear_ckks; do echo "=== $b"; (cd oracle/_ref/dropin && ./${b}_gpu 2>&1 | tail -25); done; } > $O/r05_dropin_gpu.txt
# two lanes: what overlap buys (profiles/r05_overlap.txt carries the interpretation)
bash tools/r5_overlap.sh > /dev/null 2>&1; cp gpurun_out/r05_overlap.txt $O/r05_overlap_raw.txt
ls -la $O | head -80
