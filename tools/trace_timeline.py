"""rocprofv3 *_kernel_trace.csv -> the kernel sequence of ONE steady step (the last complete period of the dispatch stream): per launch its
duration and the idle gap before it, plus totals -- where the time of a small-batch op goes (kernel chain vs launch gaps).
usage: trace_timeline.py p_kernel_trace.csv <kernels per step> [steps to average = 10]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-per * steps:]
assert len(tail) == per * steps, "trace shorter than the requested window"
names = [r["Kernel_Name"] for r in tail[:per]]
for s in range(steps):
    assert [r["Kernel_Name"] for r in tail[s * per:(s + 1) * per]] == names, "the window is not periodic with this many kernels per step"
dur = [0.0] * per
gap = [0.0] * per
for s in range(steps):
    for i in range(per):
        r = tail[s * per + i]
        dur[i] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 / steps
        k = s * per + i
        if k > 0:
            gap[i] += (int(r["Start_Timestamp"]) - int(tail[k - 1]["End_Timestamp"])) / 1e3 / (steps if i else steps - 1)
for i in range(per):
    r = tail[i]
    print(f"{i:3d} {names[i][:64]:64s} wgs {(int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1):6d} x {r['Workgroup_Size_X']:>4s}  gap {gap[i]:6.1f} us  run {dur[i]:7.1f} us")
print(f"per step: kernels {sum(dur):.1f} us + gaps {sum(gap):.1f} us = {sum(dur) + sum(gap):.1f} us")
