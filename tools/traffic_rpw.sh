#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for RPW in 0 3 7 14; do
  export TROYHIP_NTT1_RPW=$RPW; [ $RPW = 0 ] && unset TROYHIP_NTT1_RPW
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/trpw_${RPW}_$C -o p -- python3 $R/bench.py --roofline-only --batch 128 --ntt-reps 2 --no-cpu-baseline > /dev/null 2>&1
  done
  python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(float); n=collections.Counter(); dur=collections.defaultdict(float)
for C,mul in (("FETCH_SIZE",2048),("WRITE_SIZE",1024)):
    f=glob.glob("$R/gpurun_out/trpw_${RPW}_%s/**/p_counter_collection.csv"%C, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]!=C or "ntt1" not in r["Kernel_Name"]: continue
        k=r["Kernel_Name"].split("(")[0][-30:]
        tot[(k,C)]+=float(r["Counter_Value"])*mul
        if C=="FETCH_SIZE": n[k]+=1; dur[k]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
for k in n:
    rows=26880*(13/15 if "true" in k else 2/15)
    print("rpw=$RPW", k, "calls",n[k], "fetch/row %.0f write/row %.0f avg_us %.0f"%(tot[(k,"FETCH_SIZE")]/n[k]/rows, tot[(k,"WRITE_SIZE")]/n[k]/rows, dur[k]/n[k]))
PY
done
