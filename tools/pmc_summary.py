#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv: per kernel, counters summed over dispatches."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
meta, calls = {}, collections.Counter()
seen = set()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-70:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"])
    if (r["Dispatch_Id"]) not in seen:
        seen.add(r["Dispatch_Id"]); calls[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    print(f"{k} calls={calls[k]} vgpr/sgpr/lds={meta[k]} " + " ".join(f"{c}={int(x)}" for c, x in sorted(v.items())))
