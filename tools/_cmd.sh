python -m pytest tests -q -x -m gpu 2>&1 | tail -2
for v in cur mac4; do
[ $v = mac4 ] && cp tools/probe_libs/libtroyhip_mac4.so troy_amd/libtroyhip.so
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$v -o p -- python3 $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
echo $v; python tools/kstats.py gpurun_out/prof_$v/p_kernel_stats.csv 12 | grep ks_mac
done
