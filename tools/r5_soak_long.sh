#!/bin/bash
# A longer parity soak (fresh seeds, four times the counts of tools/r5_soak.sh) -> gpurun_out/r05_random_soak_long.txt
set -u
out=gpurun_out/r05_random_soak_long.txt
mkdir -p gpurun_out
python - > $out <<'PY'
from troy_amd import capi
print("libtroyhip.so build", capi.build_id(), "(" + capi.load().troyhip_build_info().decode() + ")")
PY
run() { echo "\$ $*" >> $out; env "$@" >> $out 2>&1; }
run python tools/random_soak.py 3000000 12000
run python tools/random_soak.py 3100000 6000 narrow
run python tools/random_soak.py 3200000 1600 large
run python tools/random_soak.py 3300000 1600 large narrow
run TROYHIP_NTT=single python tools/random_soak.py 3400000 1600 large
run TROYHIP_NTT=single python tools/random_soak.py 3500000 1600 large narrow
run TROYHIP_NTT=twopass python tools/random_soak.py 3600000 800 large
run TROYHIP_SMALL=merged python tools/random_soak.py 3700000 1200 large
run TROYHIP_SMALL=split python tools/random_soak.py 3800000 1200 large
run TROYHIP_FP64=off python tools/random_soak.py 3900000 1200 large
run TROYHIP_AUX_BASE=reference python tools/random_soak.py 4000000 3000
run python tools/tiny_soak.py 4100000 3600
cat $out
