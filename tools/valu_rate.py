"""VALU issue rate per kernel from one rocprofv3 pass: --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU.
Per kernel: wave-instructions per microsecond of kernel time (chip-wide) and as a fraction of 1024 SIMDs x 2.4 GHz / 4 cycles -- a kernel far below its
neighbours with the same instruction mix is stalled on something other than issue (how the scattered stores of the forward single-pass kernel were found).
usage: valu_rate.py <dir with p_counter_collection.csv and p_kernel_trace.csv>"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
cc = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/p_kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"].split("(")[0].replace("void troyhip::", "").replace("troyhip::", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen[k]:
        seen[k].add(r["Dispatch_Id"])
        agg[k]["us"] += dur.get(r["Dispatch_Id"], 0.0)
peak = 1024 * 2400.0 / 4.0  # wave-instructions per microsecond at four cycles each
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
    if v["us"] < 1:
        continue
    rate = v["SQ_INSTS_VALU"] / v["us"]
    print(f"{k[:60]:60s} calls {len(seen[k]):3d}  {v['us']:9.1f} us  VALU {v['SQ_INSTS_VALU'] / 1e6:9.1f} M wave-instr  {rate / 1e3:7.1f} k/us = {rate / peak:5.2f} of a 4-cycle issue peak")
