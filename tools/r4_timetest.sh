#!/bin/bash
# round 4, item 2: the reference's own benchmark (test/timetest.cu: BFV N = 16384 {60,40,40,40,40,60}, t of 59 bits :468-481; the CKKS / BGV twins
# :209-466) timed through include/troyn.hpp on the GPU and, from the SAME source, on the reference's CPU half (oracle/_ref/ref_timetest) on this
# box's host cores (one thread).  usage (GPU box): tools/r4_timetest.sh [tag] [gpu divisor] [cpu divisor]  -> gpurun_out/<tag>/timetest.txt
tag=${1:-r4}; gd=${2:-1}; cd_=${3:-20}
out=gpurun_out/$tag
mkdir -p $out
g++ -std=c++17 -O2 -Iinclude tests/cpp/test_troyn_timetest.cpp -o /tmp/timetest troy_amd/libtroyhip.so -Wl,-rpath,$PWD/troy_amd -Wl,-rpath-link,/opt/rocm/lib || exit 1
{
  echo "# $(date -u +%FT%TZ)  host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)  ms per call; device synchronised before every reading"
  for s in bfv ckks bgv; do
    tb=20; [ $s = bfv ] && tb=59
    echo "=== GPU through troyn.hpp: $s N=16384 tbits=$tb repetitions = reference / $gd"
    /tmp/timetest 16384 --time $gd --scheme $s --tbits $tb | grep -v '^ok'
    if [ -x oracle/_ref/ref_timetest ]; then
      echo "=== reference CPU path (src/troy_cpu.h, 1 thread): $s N=16384 tbits=$tb repetitions = reference / $cd_"
      oracle/_ref/ref_timetest 16384 --time $cd_ --scheme $s --tbits $tb | grep -v '^ok'
    fi
  done
} 2>&1 | tee $out/timetest.txt
