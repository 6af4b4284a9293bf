#!/bin/bash
# same-box A/B of how the key-switch sums are stored: cur (through the wave's exchange area, contiguous KiB per instruction) against
# mac0 (from a thread's eight consecutive coefficients); probe build: tools/ntt_probe.sh mac0:-DN2_MAC_STORE_LINEAR=0
for wl in bfv_n32768_l14 ckks_n32768_chain bgv_n65536_relin_rot bfv_n8192_l4; do
  AB_TAIL=4 tools/ab.sh maclin_$wl cur mac0 -- python tools/bench_kernels.py $wl "0, 0, 9, 0, 1, 0, 1" "0, 0, 9, 0, 1, 0, 3"
done
