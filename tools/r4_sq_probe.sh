#!/bin/bash
# SQ counters (four passes of four) over the roofline launches of the single-pass kernels: WLS="bfv_n32768_l14 bfv_n32768_l14_p49" tools/r4_sq_probe.sh -> gpurun_out/sqp/sq_<workload>.txt
R=$PWD; O=$R/gpurun_out/sqp; mkdir -p $O
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; SQ2="SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; SQ3="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; SQ4="SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_WR"
for wl in ${WLS:-bfv_n32768_l14_p49}; do
  : > $O/sq_$wl.txt
  for set in "$SQ1" "$SQ2" "$SQ3" "$SQ4"; do
    rm -rf $O/pmc_$wl
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$wl -o p -- python3 $R/bench.py --workload $wl --roofline-only --batch 32 --ntt-reps 4 --no-cpu-baseline > $O/pmc_$wl.log 2>&1)
    f=$(find $O/pmc_$wl -name "p_counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f | grep -v "fill_uniform\|copyBuffer" >> $O/sq_$wl.txt
  done
  cat $O/sq_$wl.txt
done
