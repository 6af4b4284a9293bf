#!/bin/bash
# Round-5 parity soak (same counts as round 4) on the GPU: random parameter sets, product vs CPU oracle limb for limb, under every value of the library switches that picks
# a different kernel family (single-pass NTT of N = 2^12 .. 2^15, FP64 BEHZ of small bases, merged / split small launches).  Output: gpurun_out/r05_random_soak.txt
set -u
out=gpurun_out/r05_random_soak.txt
mkdir -p gpurun_out
python - > $out <<'PY'
from troy_amd import capi
print("libtroyhip.so build", capi.build_id(), "(" + capi.load().troyhip_build_info().decode() + ")")
PY
run() { echo "\$ $*" >> $out; env "$@" >> $out 2>&1; }
run python tools/random_soak.py 500000 3000
run python tools/random_soak.py 510000 1500 narrow
run python tools/random_soak.py 520000 400 large
run python tools/random_soak.py 530000 400 large narrow
run TROYHIP_NTT=single python tools/random_soak.py 540000 400 large
run TROYHIP_NTT=single python tools/random_soak.py 550000 400 large narrow
run TROYHIP_NTT=twopass python tools/random_soak.py 560000 200 large narrow
run TROYHIP_SMALL=merged python tools/random_soak.py 570000 300 large
run TROYHIP_SMALL=split python tools/random_soak.py 580000 300 large
run TROYHIP_SMALL=merged python tools/random_soak.py 590000 1000
run TROYHIP_FP64=off python tools/random_soak.py 600000 300 large narrow
run python tools/tiny_soak.py 610000 900
run python tools/abi_fuzz.py
cat $out
