#!/bin/bash
# profiles/r05_overlap.txt: what running the two lanes of the headline on two HIP streams buys, and why (round-4 verdict, item 4).
#   A/B on one box: one lane, two lanes half a step out of phase (the default), two lanes in phase; then a rocprofv3 kernel trace of the
#   two-lane run: how long two kernels are really in flight together, which pairs, and what a launch costs alone vs overlapped.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
OUT=$R/gpurun_out/r05_overlap.txt; : > $OUT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %9.1f ops/s  %7.3f ms/step' % ('$1', d['value'], d['ms_per_step']))"; }
cd $R
for rep in 1 2; do
python3 bench.py --streams 1 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-verify 2>/dev/null | line "one lane (B = 256)" >> $OUT
python3 bench.py --streams 2 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-verify 2>/dev/null | line "two lanes, half a step apart" >> $OUT
BENCH_LANE_PHASE=same python3 bench.py --streams 2 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-verify 2>/dev/null | line "two lanes, in phase" >> $OUT
done
cd /tmp; export TMPDIR=/tmp
for ph in half same; do
  d=$R/gpurun_out/overlap_trace_$ph; rm -rf $d
  if [ $ph = same ]; then export BENCH_LANE_PHASE=same; else unset BENCH_LANE_PHASE; fi
  rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 $R/bench.py --streams 2 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-per-kernel --no-verify > $d.log 2>&1
  f=$(find $d -name '*kernel_trace.csv' | head -1)
  echo "---- kernel trace, two lanes, phase = $ph" >> $OUT
  python3 $R/tools/overlap_analysis.py $f 0.4 >> $OUT 2>&1
done
cat $OUT
