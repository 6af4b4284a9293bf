#!/bin/bash
# same-box A/B of single-pass NTT builds (tools/probe_libs/libtroyhip_<name>.so via TROYHIP_LIB; "cur" = the tree's library) at the headline's key-switch
# shape, for the three prime classes.  usage: tools/r4_ntt1_ab.sh <tag> <name> [<name> ..]
tag=$1; shift
mkdir -p gpurun_out/$tag
for rep in 1 2; do
for a in "$@"; do
  if [ $a = cur ]; then unset TROYHIP_LIB; else export TROYHIP_LIB=$PWD/tools/probe_libs/libtroyhip_$a.so; fi
  for bits in "[60] + [58] * 13 + [60]" "[49] * 15" "[60] + [40] * 13 + [60]"; do
    echo -n "$a  $bits  "; PROBE_BITS="$bits" python tools/ntt1_probe.py 128 6 2>&1 | tail -1
  done
done
done 2>&1 | tee gpurun_out/$tag/ntt1_ab.txt
