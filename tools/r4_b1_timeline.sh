#!/bin/bash
# kernel timeline of one multiply+relinearize at B = 1 (or $2) : tools/r4_b1_timeline.sh <workload> [batch] [tag] -> gpurun_out/b1_<workload>_<batch>_<tag>.txt
wl=${1:-bfv_n32768_l14}; b=${2:-1}; tag=${3:-cur}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
d=gpurun_out/b1tl_${wl}_${b}_$tag; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 bench.py --workload $wl --batch $b --streams 1 --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-per-kernel --no-verify > $d/bench.log 2>&1
f=$(find $d -name '*kernel_trace.csv' | head -1)
for per in 7 8 9 10 11 12 13 14 15 16 17 18; do
  if python tools/trace_timeline.py $f $per 20 > $d/tl.txt 2>/dev/null; then echo "kernels per step: $per"; cat $d/tl.txt; break; fi
done | tee gpurun_out/b1_${wl}_${b}_$tag.txt
tail -1 $d/bench.log | cut -c1-200
