#!/bin/bash
# SQ counters of one step of configs[1] (BFV N = 8192): the small-base BEHZ kernels and the FP64 two-pass kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_cfgb; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $O -o p -- python3 $R/bench.py --workload bfv_n8192_l4 --steps 1 --warmup 0 --batch 256 --streams 1 --ntt-reps 2 --no-cpu-baseline --no-per-kernel --no-verify > $O/log.txt 2>&1
f=$(find $O -name 'p_counter_collection.csv' | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f | grep -v rocclr | cut -c1-420
