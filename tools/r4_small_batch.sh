#!/bin/bash
# round 4, item 1: what a caller of the drop-in C++ surface gets.  multiply + relinearize through include/troyn.hpp (tests/cpp/bench_troyn.cpp:
# single calls, a loop of single calls, the slab-batched forms) at B = 1, 8, 128 (headline) and 1, 8, 128, 1024 (configs[1]), next to the C-ABI
# line of bench.py at the same batch on the same box.  usage (GPU box): tools/r4_small_batch.sh [tag]   -> gpurun_out/<tag>/small_batch.txt
tag=${1:-r4}
out=gpurun_out/$tag
mkdir -p $out
g++ -std=c++17 -O2 -Iinclude tests/cpp/bench_troyn.cpp -o /tmp/bench_troyn troy_amd/libtroyhip.so -Wl,-rpath,$PWD/troy_amd -Wl,-rpath-link,/opt/rocm/lib || exit 1
{
  echo "# build $(python -c 'from troy_amd import capi; print(capi.load().troyhip_build_id().decode())' 2>/dev/null)  $(date -u +%FT%TZ)"
  echo "## through troyn.hpp (tests/cpp/bench_troyn.cpp)"
  /tmp/bench_troyn bfv_n32768_l14 20 1 8 128
  /tmp/bench_troyn bfv_n8192_l4 50 1 8 128 1024
  echo "## through the C ABI (bench.py --streams 1 --batch B)"
  for b in 1 8 128; do
    python bench.py --workload bfv_n32768_l14 --batch $b --streams 1 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-per-kernel 2>/dev/null | tail -1 |
      python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({"workload": d["config"]["workload"], "api": "C ABI (bench.py)", "batch": d["config"]["batch_per_gpu"], "ms_per_step": d["ms_per_step"], "ops_per_s": d["value"], "verified": d["verified"]}))'
  done
  for b in 1 8 128 1024; do
    python bench.py --workload bfv_n8192_l4 --batch $b --streams 1 --steps 50 --warmup 3 --no-cpu-baseline --no-roofline --no-per-kernel 2>/dev/null | tail -1 |
      python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({"workload": d["config"]["workload"], "api": "C ABI (bench.py)", "batch": d["config"]["batch_per_gpu"], "ms_per_step": d["ms_per_step"], "ops_per_s": d["value"], "verified": d["verified"]}))'
  done
} 2>&1 | tee $out/small_batch.txt
