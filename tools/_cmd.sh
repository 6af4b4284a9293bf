python -m pytest tests -q -x -m gpu 2>&1 | tail -2
python bench.py --no-cpu-baseline --batch 32 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B32 fused', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
TROYHIP_TENSOR=split python bench.py --no-cpu-baseline --batch 32 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B32 split', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B64', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_x -o p -- python3 $GRAFT_REPO_ROOT/bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/kstats.py gpurun_out/prof_x/p_kernel_stats.csv 12
