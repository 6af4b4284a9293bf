#!/bin/bash
# build throw-away variants of the library with BEHZ_EXP probe bits (tools/probe_libs/, git-ignored) here; run them on the GPU box with
#   gpurun -- 'tools/behz_probe.sh run'
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "run" ]; then
    mkdir -p gpurun_out
    # per variant: the multiply time (HIP events) and, from a rocprofv3 kernel trace of the same run, the two BEHZ kernels' averages
    export TMPDIR=/tmp
    for so in tools/probe_libs/libtroyhip_behz*.so; do
        n=$(basename $so .so)
        rm -rf gpurun_out/bp_$n
        rocprofv3 --kernel-trace --stats -d gpurun_out/bp_$n -o t --output-format csv -- python3 tools/behz_probe.py $so ${2:-32} 2>/dev/null | grep multiply
        f=$(find gpurun_out/bp_$n -name '*kernel_stats.csv' | head -1)
        python tools/kstats.py $f 30 | grep behz | sed 's/^/      /'
        rm -rf gpurun_out/bp_$n
    done | tee gpurun_out/behz_probe.log
    exit 0
fi
mkdir -p tools/probe_libs
for v in "$@"; do # name:-Dflags  or a bare number n -> -DBEHZ_EXP=n
    case $v in
        *:*) name=${v%%:*}; flags=${v#*:} ;;
        *) name=exp$v; flags=-DBEHZ_EXP=$v ;;
    esac
    make -s -j8 -C troy_amd/csrc OBJDIR=$PWD/troy_amd/csrc/build/probe_behz_$name OUT=$PWD/tools/probe_libs/libtroyhip_behz_$name.so EXTRA_HIPFLAGS="$flags"
done
