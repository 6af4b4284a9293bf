#!/bin/bash
# build throw-away variants of the library (tools/probe_libs/, git-ignored) here; run them on the GPU box with
#   gpurun -- 'tools/ntt_probe.sh run'
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "run" ]; then
    mkdir -p gpurun_out
    for so in tools/probe_libs/*.so; do python tools/ntt_probe.py $so ${2:-16} 2>&1 | tail -1; done | tee gpurun_out/ntt_probe.log
    exit 0
fi
mkdir -p tools/probe_libs
build() { # name flags...
    name=$1; shift
    make -s -j8 -C troy_amd/csrc OBJDIR=$PWD/troy_amd/csrc/build/probe_$name OUT=$PWD/tools/probe_libs/libtroyhip_$name.so EXTRA_HIPFLAGS="$*"
}
for v in "$@"; do
    case $v in
        base) build base ;;
        exp*) build $v -DN2_EXP=${v#exp} ;;
        w*) build $v -DN2_MIN_WAVES=${v#w} ;;
        *:*) build ${v%%:*} ${v#*:} ;;   # name:-Dflag -Dflag
    esac
done
