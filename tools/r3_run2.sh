#!/bin/bash
# round-3 GPU pass 2: whole GPU suite, then every workload's bench line (verification on), short
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu > gpurun_out/r3/gpu_suite.log 2>&1; tail -5 gpurun_out/r3/gpu_suite.log
python bench.py --steps 20 --warmup 3 > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err; tail -c 600 gpurun_out/r3/bench_default.err
for wl in bfv_n8192_l4 ckks_n32768_chain bgv_n65536_relin_rot ckks_matmul_128; do
  python bench.py --workload $wl --warmup 3 --no-cpu-baseline > gpurun_out/r3/bench_$wl.json 2> gpurun_out/r3/bench_$wl.err; tail -c 400 gpurun_out/r3/bench_$wl.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(f.split("/")[-1], d["value"], d["unit"], "ms/step", d["ms_per_step"], "steps", d["steps"], "verified", d["verified"], "|", d["verified_what"])
        print("   roofline frac", d["roofline"].get("frac"), "launch_us", d["roofline"].get("launch_us"), "traffic", d["roofline"].get("traffic"), "devices", d.get("rank_devices"))
        if d.get("cpu_baseline"): print("   cpu", json.dumps(d["cpu_baseline"])[:600])
    except Exception as e:
        print(f, "FAILED", e)
PY
