#!/bin/bash
# Same-box A/B: ONE command under several variants of the library, alternating, two passes (the pool's boxes differ by +-2 %: only numbers of one
# box compare).  usage (GPU box):  tools/ab.sh <tag> <variant> [<variant> ..] -- <command ...>        output: gpurun_out/<tag>/ab.txt
#   variant:  cur                       the tree's library, no switch
#             <name>                    TROYHIP_LIB=tools/probe_libs/libtroyhip_<name>.so   (tools/ntt_probe.sh <name>:-D<macro> builds one;
#                                       `make -C troy_amd/csrc probes` builds "probes", the build that reads the development switches)
#             VAR=value[,VAR=value..]   environment switches (the shipped four, or the development ones together with the probes library:
#                                       TROYHIP_LIB=tools/probe_libs/libtroyhip_probes.so,TROYHIP_KS=split)
# examples:   tools/ab.sh ntt cur TROYHIP_NTT=twopass -- python tools/ntt1_probe.py 128 6
#             tools/ab.sh small cur TROYHIP_SMALL=split -- /tmp/bench_troyn bfv_n32768_l14 50 1 2 8
#             PROBE_N=8192 PROBE_BITS="[49] * 5" tools/ab.sh n13 cur prev -- python tools/ntt1_probe.py 512 6
tag=$1; shift
variants=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do variants+=("$1"); shift; done
shift
mkdir -p gpurun_out/$tag
for pass in 1 2; do
  for v in "${variants[@]}"; do
    echo "== $v (pass $pass)"
    case $v in
      cur) "$@" 2>&1 | tail -${AB_TAIL:-4} ;;
      *=*) env $(echo "$v" | tr ',' ' ') "$@" 2>&1 | tail -${AB_TAIL:-4} ;;
      *)   TROYHIP_LIB=$PWD/tools/probe_libs/libtroyhip_$v.so "$@" 2>&1 | tail -${AB_TAIL:-4} ;;
    esac
  done
done 2>&1 | tee gpurun_out/$tag/ab.txt
