#!/bin/bash
# VALU issue rate of every kernel of one step: tools/r4_valu_rate.sh <workload> [batch]  -> gpurun_out/valu_<workload>.txt
wl=${1:-bfv_n32768_l14}; b=${2:-128}
R=$PWD; O=$R/gpurun_out/valu_$wl; rm -rf $O; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O -o p -- python3 $R/bench.py --workload $wl --steps 1 --warmup 0 --batch $b --streams 1 --no-roofline --no-cpu-baseline --no-per-kernel --no-verify > $O/log 2>&1)
python tools/valu_rate.py $O | tee gpurun_out/valu_$wl.txt
