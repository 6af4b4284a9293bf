#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the single-pass inverse kernels inside one headline step, per library variant (cur = the tree's library, or a probe library of
# tools/probe_libs/), then a same-box A/B of the headline bench over the same variants.  usage (GPU box): tools/r5_nt_traffic.sh <variant> ..
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
out=$R/gpurun_out/r05_nt_traffic_${WL:-bfv_n32768_l14}.txt
mkdir -p $R/gpurun_out; : > $out
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = cur ]; then unset TROYHIP_LIB; else export TROYHIP_LIB=$R/tools/probe_libs/libtroyhip_$v.so; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/nt_${v}_$C
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/nt_${v}_$C -o p -- python3 $R/bench.py --workload ${WL:-bfv_n32768_l14} --steps 1 --warmup 0 --streams 1 --no-roofline --no-cpu-baseline --no-per-kernel --no-verify > $R/gpurun_out/nt_${v}_$C.log 2>&1
  done
  V=$v python3 - >> $out <<'PY'
import csv, glob, os, re, collections
v = os.environ["V"]; R = os.environ.get("GRAFT_REPO_ROOT") or "."
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for i, C in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    for f in glob.glob(f"{R}/gpurun_out/nt_{v}_{C}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != C: continue
            name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].replace("troyhip::", "")
            tot[name][i] += float(r["Counter_Value"]) * 1024 * (2 if i == 0 else 1)
            if i == 0: tot[name][2] += 1
for name, (f, w, n) in sorted(tot.items()):
    if name.startswith("ntt"): print("%-8s %-36s calls %2d  fetch %7.3f GB  write %7.3f GB" % (v, name, n, f / 1e9, w / 1e9))
PY
done
unset TROYHIP_LIB
cd $R
AB_TAIL=1 bash tools/ab.sh r05_nt "$@" -- python bench.py --workload ${WL:-bfv_n32768_l14} --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --ntt-reps 2 > /dev/null 2>&1
python3 - >> $out <<'PY'
import json
for ln in open("gpurun_out/r05_nt/ab.txt"):
    ln = ln.strip()
    if ln.startswith("=="): print(ln, end="  ")
    elif ln.startswith("{"):
        d = json.loads(ln); print(d["value"], d["unit"], d["ms_per_step"], "ms", " ".join("%s %.0f" % (k["name"].replace("_kernel", "").replace(" ", ""), k["us"]) for k in d["roofline"]["per_kernel"] if "ntt" in k["name"]) if d.get("roofline") and d["roofline"].get("per_kernel") else "")
PY
cat $out
