#!/bin/bash
# the artefacts profiles/README.md lists for the round's final build.  usage (on the GPU box): tools/final_profiles.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
for W in bfv_n32768_l14 bfv_n8192_l4 ckks_n32768_chain bgv_n65536_relin_rot ckks_matmul_128; do
  timeout 900 python3 bench.py --workload $W 2> $O/bench_$W.err | tail -1 > $O/r02_bench_$W.json
  cut -c1-260 $O/r02_bench_$W.json
done
python3 tools/behz_cmp.py 32 > $O/r02_behz_forms.txt 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o p -- python3 $R/bench.py --no-cpu-baseline > $O/prof_default.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roofline -o p -- python3 $R/bench.py --roofline-only --no-cpu-baseline > $O/prof_roofline.log 2>&1
cd $R
for T in default roofline; do
  f=$(find $O/prof_$T -name 'p_kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/r02_${T}_kernel_stats.csv && head -12 $f | cut -c1-200
done
tail -1 $O/prof_roofline.log | cut -c1-900
