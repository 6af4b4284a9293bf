for wl in bfv_n32768_l14 ckks_n32768_chain bgv_n65536_relin_rot bfv_n32768_l14_p49 bfv_n8192_l4; do
  AB_TAIL=4 tools/ab.sh maclin_$wl cur macold -- python tools/bench_kernels.py $wl "0, 0, 9, 0, 1, 0"
done
