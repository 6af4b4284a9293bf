cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_x -o p -- python3 $GRAFT_REPO_ROOT/bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/kstats.py gpurun_out/prof_x/p_kernel_stats.csv 16
