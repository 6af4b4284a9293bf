#!/bin/bash
# same-box A/B of the round: the round-5 library (tools/probe_libs/libtroyhip_r5base.so, built from the round's first commit) against the tree's, all large workloads, alternating, two passes
for wl in bfv_n32768_l14 bfv_n32768_l14_p49 ckks_n32768_chain bgv_n65536_relin_rot bfv_n8192_l4; do
  AB_TAIL=1 tools/ab.sh r6_round_$wl r5base cur -- python tools/bench_kernels.py $wl __none__
done
