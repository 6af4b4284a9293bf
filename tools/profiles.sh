#!/bin/bash
# ONE parameterised script for everything under profiles/ (it replaces the per-round tools/r4_*.sh / r5_*.sh).  Runs on the GPU box:
#   tools/profiles.sh <tag> [section ..]        e.g.  tools/profiles.sh r06 traffic bench stats      -> gpurun_out/<tag>/<tag>_*   (copy what is judged into profiles/)
# sections (default: traffic bench stats roofline):
#   traffic   PMC HBM bytes per kernel of every workload (tools/measure_traffic.sh; first, so that the bench lines carry per_kernel[].traffic and roofline.step)
#   bench     the bench line of every workload (verified; CPU baseline on the headline and configs[1])            -> <tag>_bench_<workload>.json
#   stats     rocprofv3 --kernel-trace --stats of every workload's bench command                                  -> <tag>_<workload>_kernel_stats.csv
#   roofline  ... of the roofline launches alone (headline and its 49-bit twin)                                   -> <tag>_roofline_<workload>_kernel_stats.csv
#   sq        SQ counters (own passes, kernel trace only) over the roofline launches and one configs[1] step      -> <tag>_pmc_sq_*.txt
#   valu      VALU issue rate of every kernel of one lane's step (tools/valu_rate.py)                              -> <tag>_valu_rate.txt
#   timeline  kernel timeline of one ciphertext / eight (tools/trace_timeline.py)                                 -> <tag>_b1_timeline.txt
#   small     multiply + relinearize through troyn.hpp at B = 1 / 8 / 128 (/ 1024) next to the C-ABI line          -> <tag>_small_batch.txt
#   timetest  the reference's own test/timetest.cu on the GPU (synchronised and asynchronous) and on the reference CPU path -> <tag>_timetest.txt
#   dropin    the reference's own GPU tests and apps, unchanged, on the device                                     -> <tag>_dropin_gpu.txt
#   dist      two ranks sharing GPU 0: scatter -> compute -> gather (tools/dist_two_ranks.py)                     -> <tag>_dist_two_ranks.txt
#   overlap   one lane / two lanes half a step apart / in phase, and the kernel trace of the two-lane run         -> <tag>_overlap_raw.txt
#   soak      parity soak: random parameter sets vs the oracle under every library switch (SOAK_SCALE=4: the long form) -> <tag>_random_soak.txt
# WLS="..." restricts the workloads.
TAG=${1:?usage: tools/profiles.sh <tag> [section ..]}; shift
SECTIONS=${@:-traffic bench stats roofline}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
WLS=${WLS:-bfv_n32768_l14 bfv_n8192_l4 ckks_n32768_chain bgv_n65536_relin_rot ckks_matmul_128 bfv_n32768_l14_p49}
prof() { (cd /tmp && export TMPDIR=/tmp && rocprofv3 "$@"); } # rocprofv3 wants a writable cwd and TMPDIR; the program itself follows `--`
stats_of() { # <out csv> <bench args ..>
  local out=$1; shift; rm -rf $O/prof_tmp
  prof --kernel-trace --stats --output-format csv -d $O/prof_tmp -o p -- python3 $R/bench.py "$@" > $O/prof_tmp.log 2>&1
  local f=$(find $O/prof_tmp -name "p_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out; rm -rf $O/prof_tmp
}
pmc_of() { # <out txt (appended)> <counter set> <bench args ..>
  local out=$1 set=$2; shift 2; rm -rf $O/pmc_tmp
  prof --kernel-trace --pmc $set --output-format csv -d $O/pmc_tmp -o p -- python3 $R/bench.py "$@" > $O/pmc_tmp.log 2>&1
  local f=$(find $O/pmc_tmp -name "p_counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f | grep -v "fill_uniform\|copyBuffer" >> $out; rm -rf $O/pmc_tmp
}
for section in $SECTIONS; do case $section in
traffic)
  for wl in $WLS; do ROUND=$TAG tools/measure_traffic.sh $wl > $O/traffic_$wl.log 2>&1; cp gpurun_out/${TAG}_traffic_$wl.json $O/ 2>/dev/null && cp gpurun_out/${TAG}_traffic_$wl.json profiles/; done ;;
bench)
  for wl in $WLS; do
    extra="--no-cpu-baseline"; case $wl in bfv_n32768_l14|bfv_n8192_l4) extra="";; esac
    python bench.py --workload $wl $extra > $O/${TAG}_bench_$wl.json 2> $O/bench_$wl.err
  done ;;
stats)
  for wl in $WLS; do stats_of $O/${TAG}_${wl}_kernel_stats.csv --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-verify; done ;;
roofline)
  for wl in bfv_n32768_l14 bfv_n32768_l14_p49; do stats_of $O/${TAG}_roofline_${wl}_kernel_stats.csv --workload $wl --roofline-only --no-cpu-baseline; done ;;
sq)
  SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; SQ2="SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
  for wl in bfv_n32768_l14 bfv_n32768_l14_p49; do
    : > $O/${TAG}_pmc_sq_roofline_$wl.txt
    for set in "$SQ1" "$SQ2"; do pmc_of $O/${TAG}_pmc_sq_roofline_$wl.txt "$set" --workload $wl --roofline-only --batch 32 --ntt-reps 4 --no-cpu-baseline; done
  done
  : > $O/${TAG}_pmc_sq_bfv_n8192_l4.txt
  for set in "$SQ1" "$SQ2"; do pmc_of $O/${TAG}_pmc_sq_bfv_n8192_l4.txt "$set" --workload bfv_n8192_l4 --steps 1 --warmup 0 --batch 256 --streams 1 --no-roofline --no-cpu-baseline --no-per-kernel --no-verify; done ;;
valu)
  : > $O/${TAG}_valu_rate.txt
  for wl in bfv_n32768_l14 bfv_n32768_l14_p49 ckks_n32768_chain bgv_n65536_relin_rot; do
    echo "== $wl" >> $O/${TAG}_valu_rate.txt; rm -rf $O/valu_tmp; mkdir -p $O/valu_tmp
    prof --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/valu_tmp -o p -- python3 $R/bench.py --workload $wl --steps 1 --warmup 0 --batch 128 --streams 1 --no-roofline --no-cpu-baseline --no-per-kernel --no-verify > $O/valu_tmp.log 2>&1
    python tools/valu_rate.py $O/valu_tmp >> $O/${TAG}_valu_rate.txt; rm -rf $O/valu_tmp
  done ;;
timeline)
  { for a in "bfv_n32768_l14 1" "bfv_n32768_l14 8" "bfv_n8192_l4 1" "bfv_n8192_l4 8"; do
      set -- $a; echo "## $a"; d=$O/tl_tmp; rm -rf $d; mkdir -p $d
      prof --kernel-trace --output-format csv -d $d -o t -- python3 $R/bench.py --workload $1 --batch $2 --streams 1 --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-per-kernel --no-verify > $d/bench.log 2>&1
      f=$(find $d -name '*kernel_trace.csv' | head -1)
      for per in 7 8 9 10 11 12 13 14 15 16 17 18; do if python tools/trace_timeline.py $f $per 20 > $d/tl.txt 2>/dev/null; then echo "kernels per step: $per"; cat $d/tl.txt; break; fi; done
      rm -rf $d
    done; } > $O/${TAG}_b1_timeline.txt ;;
small)
  g++ -std=c++17 -O2 -Iinclude tests/cpp/bench_troyn.cpp -o /tmp/bench_troyn troy_amd/libtroyhip.so -Wl,-rpath,$R/troy_amd -Wl,-rpath-link,/opt/rocm/lib || continue
  { echo "# build $(python -c 'from troy_amd import capi; print(capi.build_id())' 2>/dev/null)  $(date -u +%FT%TZ)"
    echo "## through troyn.hpp (tests/cpp/bench_troyn.cpp)"
    /tmp/bench_troyn bfv_n32768_l14 20 1 8 128; /tmp/bench_troyn bfv_n8192_l4 50 1 8 128 1024
    echo "## through the C ABI (bench.py --streams 1 --batch B)"
    for a in "bfv_n32768_l14 20 1 8 128" "bfv_n8192_l4 50 1 8 128 1024"; do
      set -- $a; wl=$1; steps=$2; shift 2
      for b in "$@"; do
        python bench.py --workload $wl --batch $b --streams 1 --steps $steps --warmup 3 --no-cpu-baseline --no-roofline --no-per-kernel 2>/dev/null | tail -1 |
          python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({"workload": d["config"]["workload"], "api": "C ABI (bench.py)", "batch": d["config"]["batch_per_gpu"], "ms_per_step": d["ms_per_step"], "ops_per_s": d["value"], "verified": d["verified"]}))'
      done
    done; } > $O/${TAG}_small_batch.txt 2>&1 ;;
timetest)
  { echo "# $(date -u +%FT%TZ)  host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)"
    echo "=== GPU (MI355X): the reference's test/timetest.cu, unchanged, compiled against include/troy_cuda.cuh, linked with libtroyhip.so -- TROYHIP_SYNC=1: every library call"
    echo "=== returns with the device idle, so the file's host timers (which never synchronise) measure the work itself"
    (cd oracle/_ref/dropin && TROYHIP_SYNC=1 ./timetest_gpu)
    echo "=== GPU, default (asynchronous) execution: the same timers now read ENQUEUE times wherever an operation does not synchronise by itself"
    (cd oracle/_ref/dropin && ./timetest_gpu)
    echo "=== CPU (one core): the same file compiled against the reference's src/troy_cpu.h, linked with the reference's CPU half (oracle/_ref/ref_timetest)"
    if [ -x oracle/_ref/ref_timetest ]; then timeout 1500 oracle/_ref/ref_timetest; else echo "oracle/_ref/ref_timetest not prebuilt"; fi; } > $O/${TAG}_timetest.txt 2>&1 ;;
dropin)
  { for b in troytest timetest linear linear_ckks; do echo "=== $b"; (cd oracle/_ref/dropin && ./${b}_gpu 2>&1 | tail -25); done; } > $O/${TAG}_dropin_gpu.txt ;;
dist)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/dist_two_ranks.py bfv_n32768_l14 64 2>$O/dist.err | tail -1 > $O/${TAG}_dist_two_ranks.txt
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 tools/dist_two_ranks.py bfv_n8192_l4 256 2>>$O/dist.err | tail -1 >> $O/${TAG}_dist_two_ranks.txt ;;
overlap)
  OUT=$O/${TAG}_overlap_raw.txt; : > $OUT
  line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %9.1f ops/s  %7.3f ms/step' % ('$1', d['value'], d['ms_per_step']))"; }
  for rep in 1 2; do
    python3 bench.py --streams 1 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-verify 2>/dev/null | line "one lane (B = 256)" >> $OUT
    python3 bench.py --streams 2 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-verify 2>/dev/null | line "two lanes, half a step apart" >> $OUT
    BENCH_LANE_PHASE=same python3 bench.py --streams 2 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-verify 2>/dev/null | line "two lanes, in phase" >> $OUT
  done
  for ph in half same; do
    d=$O/overlap_tmp; rm -rf $d
    if [ $ph = same ]; then export BENCH_LANE_PHASE=same; else unset BENCH_LANE_PHASE; fi
    prof --kernel-trace --output-format csv -d $d -o t -- python3 $R/bench.py --streams 2 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-per-kernel --no-verify > $d.log 2>&1
    echo "---- kernel trace, two lanes, phase = $ph" >> $OUT
    python3 tools/overlap_analysis.py $(find $d -name '*kernel_trace.csv' | head -1) 0.4 >> $OUT 2>&1; rm -rf $d
  done; unset BENCH_LANE_PHASE ;;
soak)
  out=$O/${TAG}_random_soak.txt; S=${SOAK_SCALE:-1}; seed=$((7000000 + 1000000 * S))
  python -c "from troy_amd import capi; print('libtroyhip.so build', capi.build_id(), '(' + capi.load().troyhip_build_info().decode() + ')')" > $out
  run() { echo "\$ $*" >> $out; env "$@" >> $out 2>&1; }
  run python tools/random_soak.py $((seed)) $((3000 * S))
  run python tools/random_soak.py $((seed + 10000)) $((1500 * S)) narrow
  run python tools/random_soak.py $((seed + 20000)) $((400 * S)) large
  run python tools/random_soak.py $((seed + 30000)) $((400 * S)) large narrow
  run TROYHIP_NTT=single python tools/random_soak.py $((seed + 40000)) $((400 * S)) large
  run TROYHIP_NTT=single python tools/random_soak.py $((seed + 50000)) $((400 * S)) large narrow
  run TROYHIP_NTT=twopass python tools/random_soak.py $((seed + 60000)) $((200 * S)) large narrow
  run TROYHIP_SMALL=merged python tools/random_soak.py $((seed + 70000)) $((300 * S)) large
  run TROYHIP_SMALL=split python tools/random_soak.py $((seed + 80000)) $((300 * S)) large
  run TROYHIP_FP64=off python tools/random_soak.py $((seed + 90000)) $((300 * S)) large narrow
  run TROYHIP_AUX_BASE=reference python tools/random_soak.py $((seed + 100000)) $((750 * S))
  run python tools/tiny_soak.py $((seed + 110000)) $((900 * S))
  run python tools/abi_fuzz.py ;;
*) echo "unknown section $section" ;;
esac; done
ls -la $O | grep "${TAG}_" | head -80
