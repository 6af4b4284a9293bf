#!/bin/bash
# same-box A/B of single-pass NTT builds at N = 2^12 .. 2^14 (FP64-class and integer primes).  usage: tools/r4_ntt1s_ab.sh <tag> <name> [<name> ..]
tag=$1; shift
mkdir -p gpurun_out/$tag
for rep in 1 2; do
for a in "$@"; do
  if [ $a = cur ]; then unset TROYHIP_LIB; else export TROYHIP_LIB=$PWD/tools/probe_libs/libtroyhip_$a.so; fi
  for n in 4096 8192 16384; do
    for bits in "[49] * 5" "[60, 58, 58, 58, 60]"; do
      b=$((128 * 32768 / n)); echo -n "$a N=$n $bits  "; PROBE_N=$n PROBE_BITS="$bits" python tools/ntt1_probe.py $b 6 2>&1 | tail -1
    done
  done
done
done 2>&1 | tee gpurun_out/$tag/ntt1s_ab.txt
