python -m pytest tests -q -x -m gpu 2>&1 | tail -2
TROYHIP_KS=split python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "scenario" 2>&1 | tail -2
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B32 fused', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
TROYHIP_KS=split python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B32 split', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
python bench.py --no-cpu-baseline --batch 64 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B64 fused', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
