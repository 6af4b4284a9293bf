"""print the top rows of a rocprofv3 *_kernel_stats.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f"{r['Name'][:84]:84s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f}us {float(r['TotalDurationNs'])/tot*100:5.1f}%")
