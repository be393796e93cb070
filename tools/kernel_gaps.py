#!/usr/bin/env python3
"""Idle time between kernels of a rocprofv3 --kernel-trace CSV, all queues on one timeline: per kernel name the
launches, mean duration, mean idle time BEFORE it (start - latest end of anything earlier; host pauses > 0.2 ms
are not counted) and mean OVERLAP with what ran before it (latest earlier end - start, where the kernel started before its
predecessor's completion was recorded: dispatch-to-completion times of back-to-back launches overlap).  python tools/kernel_gaps.py <kernel_trace.csv>"""
import csv, sys, collections, re
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name.split("(")[0][:80]
stat = collections.OrderedDict()
latest_end = None
busy = idle = overlap = 0
queues = set()
for s, e, name, q in rows:
    queues.add(q)
    st = stat.setdefault(short(name), [0, 0, 0, 0])
    st[0] += 1
    st[1] += e - s
    busy += e - s
    if latest_end is not None:
        gap = s - latest_end
        if 0 < gap < 200000:
            st[2] += gap
            idle += gap
        elif gap < 0:
            st[3] += min(-gap, e - s)
            overlap += min(-gap, e - s)
    latest_end = e if latest_end is None else max(latest_end, e)
print("| kernel | launches | mean us | mean idle before, us | mean overlap with the kernels before it, us |")
print("|---|---|---|---|---|")
for name, (n, dur, gap, ov) in sorted(stat.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("| `%s` | %d | %.1f | %.1f | %.1f |" % (name, n, dur / n / 1e3, gap / n / 1e3, ov / n / 1e3))
print("\n%d kernels on %d queue(s): sum of durations %.2f ms, of which overlapping an earlier kernel %.2f ms; idle between kernels (gaps < 0.2 ms) %.2f ms" % (len(rows), len(queues), busy / 1e6, overlap / 1e6, idle / 1e6))
