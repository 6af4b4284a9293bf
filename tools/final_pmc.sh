#!/bin/bash
# SQ counters of every kernel of the default step (final build of the round), kernel trace only + --pmc in its own pass.
# usage (on the GPU box): tools/final_pmc.sh      -> gpurun_out/final/r02_final_pmc_sq.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_sq -o p -- python3 $R/bench.py --steps 1 --warmup 0 --batch 32 --streams 1 --ntt-reps 2 --no-cpu-baseline --no-per-kernel > $O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc_mfma -o p -- python3 $R/bench.py --steps 1 --warmup 0 --batch 32 --streams 1 --no-roofline --no-cpu-baseline --no-per-kernel > $O/pmc_mfma.log 2>&1
cd $R
for d in pmc_sq pmc_mfma; do f=$(find $O/$d -name 'p_counter_collection.csv' | head -1); [ -n "$f" ] && python3 tools/pmc_summary.py $f | grep -v rocclr | cut -c1-520; done > $O/r02_final_pmc_sq.txt
cat $O/r02_final_pmc_sq.txt | cut -c1-300
