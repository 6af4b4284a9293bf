python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -2
python tools/ntt_probe.py tools/probe_libs/libtroyhip_nocs.so 16 | tail -1
python tools/ntt_probe.py troy_amd/libtroyhip.so 16 | tail -1
python tools/ntt_probe.py tools/probe_libs/libtroyhip_exp6.so 16 | tail -1
python tools/ntt_probe.py tools/probe_libs/libtroyhip_nocs.so 32 | tail -1
python tools/ntt_probe.py troy_amd/libtroyhip.so 32 | tail -1
python bench.py --batch 32 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B32', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
