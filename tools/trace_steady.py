"""rocprofv3 *_kernel_trace.csv -> per kernel: first (cold) call, mean of the calls after the first `skip`, and the span of the last
`last` dispatches against the sum of their durations (back-to-back launches of one stream: the two agree when nothing overlaps or idles).
usage: trace_steady.py p_kernel_trace.csv [substring] [skip=3] [last=40]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
last = int(sys.argv[4]) if len(sys.argv) > 4 else 40
rows = [r for r in rows if sub in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in by.items():
    steady = d[skip:] or d
    print(f"{k[:72]:72s} calls {len(d):4d}  first {d[0]:9.1f} us  all {sum(d)/len(d):9.1f} us  after the first {skip}: {sum(steady)/len(steady):9.1f} us")
tail = rows[-last:]
if tail:
    span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e3
    dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail) / 1e3
    print(f"last {len(tail)} dispatches: span {span:.1f} us, sum of durations {dur:.1f} us")
