#!/bin/bash
# phase stamps of one workgroup of the single-pass forward kernel (-DN1_TIMING build): integer guard-free against FP64 instance
export TROYHIP_LIB=$PWD/tools/probe_libs/libtroyhip_timing.so
echo "== 58-bit primes (integer guard-free)"; python tools/ntt1_probe.py 128 2 2>&1 | tail -12
echo "== 40-bit primes (FP64)"; PROBE_BITS="[60] + [40] * 13 + [60]" python tools/ntt1_probe.py 128 2 2>&1 | tail -12
