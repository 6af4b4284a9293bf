"""rocprofv3 kernel trace of a TWO-LANE run (bench.py --streams 2) -> how much the kernels of the two lanes really run at the same time, and what it buys.

For every kernel name: launches, mean duration, and the mean duration split by what the OTHER lane was doing meanwhile (alone / overlapped: the share of
the launch during which a kernel of the other queue was resident).  Pair matrix: seconds during which kernel X of one lane and kernel Y of the other were
both in flight.  Totals: wall time of the window, sum of durations, time with 0 / 1 / 2 kernels in flight.
usage: overlap_analysis.py <p_kernel_trace.csv> [skip_fraction = 0.3]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * skip):]  # steady state only
short = lambda n: re.sub(r"^void ", "", n).split("(")[0].replace("troyhip::", "")  # noqa: E731
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", r.get("Stream_Id", "0"))) for r in rows]
queues = sorted({e[3] for e in ev})
print("queues:", queues, " launches:", len(ev))
t0, t1 = min(e[0] for e in ev), max(e[1] for e in ev)
# sweep: number of kernels in flight over time
pts = sorted([(s, 1) for s, _, _, _ in ev] + [(e, -1) for _, e, _, _ in ev])
infl, last, hist = 0, t0, collections.Counter()
for t, d in pts:
    hist[infl] += t - last
    last, infl = t, infl + d
wall = t1 - t0
print("window %.1f ms; sum of kernel durations %.1f ms; in flight: " % (wall / 1e6, sum(e[1] - e[0] for e in ev) / 1e6) + ", ".join("%d kernels %.1f %%" % (k, 100.0 * v / wall) for k, v in sorted(hist.items())))
# per launch: overlapped share with launches of OTHER queues
by_q = collections.defaultdict(list)
for e in ev:
    by_q[e[3]].append(e)
stat = collections.defaultdict(lambda: [0, 0.0, 0.0])  # name -> [n, total dur, total overlapped]
pair = collections.Counter()
for q in queues:
    others = sorted([e for p in queues if p != q for e in by_q[p]])
    j0 = 0
    for s, e, n, _ in by_q[q]:
        while j0 < len(others) and others[j0][1] <= s:
            j0 += 1
        ov, j = 0, j0
        while j < len(others) and others[j][0] < e:
            o = min(e, others[j][1]) - max(s, others[j][0])
            if o > 0:
                ov += o
                pair[(n, others[j][2])] += o
            j += 1
        st = stat[n]
        st[0] += 1
        st[1] += e - s
        st[2] += min(ov, e - s)
print("%-44s %6s %10s %10s" % ("kernel", "calls", "mean us", "overlapped"))
for n, (c, d, o) in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    print("%-44s %6d %10.1f %9.0f %%" % (n[:44], c, d / c / 1e3, 100.0 * o / d))
print("pairs in flight together (ms, top 12; each pair counted from both sides):")
for (a, b), o in pair.most_common(12):
    print("  %-40s x %-40s %8.1f" % (a[:40], b[:40], o / 1e6))
