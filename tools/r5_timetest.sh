#!/bin/bash
# profiles/r05_timetest.txt: the reference's OWN benchmark file, test/timetest.cu, unchanged -- once against include/ and libtroyhip.so on the GPU
# (oracle/_ref/dropin/timetest_gpu) and once against the reference's own CPU half on one core of this box (oracle/_ref/ref_timetest, oracle/ref_shim).
# Both binaries are prebuilt in the build container by `make -C oracle ref dropin` (they travel with the snapshot); BFV N = 16384 {60,40,40,40,40,60},
# t of 59 bits, the file's own repetition counts (1000 / 100) and timers (ms per op).
out=gpurun_out/${1:-final}; mkdir -p $out
{
  echo "# $(date -u +%FT%TZ)  host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)"
  echo "=== GPU (MI355X): test/timetest.cu compiled against include/troy_cuda.cuh, linked with libtroyhip.so -- TROYHIP_SYNC=1: every library call returns with the"
  echo "=== device idle, so the file's host timers (which never synchronise) measure the work itself"
  (cd oracle/_ref/dropin && TROYHIP_SYNC=1 ./timetest_gpu)
  echo "=== GPU, default (asynchronous) execution: the same timers now read ENQUEUE times wherever an operation does not synchronise by itself"
  (cd oracle/_ref/dropin && ./timetest_gpu)
  echo "=== CPU (one core): the same file compiled against the reference's src/troy_cpu.h, linked with the reference's CPU half"
  if [ -x oracle/_ref/ref_timetest ]; then timeout 1500 oracle/_ref/ref_timetest; else echo "oracle/_ref/ref_timetest not prebuilt"; fi
} 2>&1 | tee $out/r05_timetest.txt
