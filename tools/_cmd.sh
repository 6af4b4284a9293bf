python -m pytest tests -q -x -m gpu 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B64', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
