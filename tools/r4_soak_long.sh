#!/bin/bash
# A longer parity soak (fresh seeds, four times the counts of tools/r4_soak.sh) -> gpurun_out/r04_random_soak_long.txt
set -u
out=gpurun_out/r04_random_soak_long.txt
mkdir -p gpurun_out
python - > $out <<'PY'
from troy_amd import capi
print("libtroyhip.so build", capi.build_id(), "(" + capi.load().troyhip_build_info().decode() + ")")
PY
run() { echo "\$ $*" >> $out; env "$@" >> $out 2>&1; }
run python tools/random_soak.py 1000000 12000
run python tools/random_soak.py 1100000 6000 narrow
run python tools/random_soak.py 1200000 1600 large
run python tools/random_soak.py 1300000 1600 large narrow
run TROYHIP_NTT=single python tools/random_soak.py 1400000 1600 large
run TROYHIP_NTT=single python tools/random_soak.py 1500000 1600 large narrow
run TROYHIP_NTT=twopass python tools/random_soak.py 1600000 800 large
run TROYHIP_SMALL=merged python tools/random_soak.py 1700000 1200 large
run TROYHIP_SMALL=split python tools/random_soak.py 1800000 1200 large
run TROYHIP_FP64=off python tools/random_soak.py 1900000 1200 large
run TROYHIP_AUX_BASE=reference python tools/random_soak.py 2000000 3000
run python tools/tiny_soak.py 2100000 3600
cat $out
