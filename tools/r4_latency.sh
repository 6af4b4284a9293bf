#!/bin/bash
# small-batch latency A/B on one box: bench_troyn at B = 1 / 8 under the switches that matter for one ciphertext.  usage: tools/r4_latency.sh [tag]
tag=${1:-r4}
out=gpurun_out/$tag
mkdir -p $out
g++ -std=c++17 -O2 -Iinclude tests/cpp/bench_troyn.cpp -o /tmp/bench_troyn troy_amd/libtroyhip.so -Wl,-rpath,$PWD/troy_amd -Wl,-rpath-link,/opt/rocm/lib || exit 1
{
  for env in "" "TROYHIP_SMALL=split" "TROYHIP_SMALL=merged"; do
    echo "## env: ${env:-default}"
    env $env /tmp/bench_troyn bfv_n32768_l14 50 1 2 8 | grep -v "ALL OK"
    env $env /tmp/bench_troyn bfv_n8192_l4 100 1 8 32 | grep -v "ALL OK"
  done
} 2>&1 | tee $out/latency.txt
# the merged forms at the LARGE batch (would one path do?): bench.py on the same box
for wl in bfv_n32768_l14 bfv_n8192_l4; do
  for env in "" "TROYHIP_SMALL=merged"; do
    echo "## $wl env: ${env:-default}" | tee -a $out/latency.txt
    env $env python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-per-kernel 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], "verified", d["verified"])' | tee -a $out/latency.txt
  done
done
