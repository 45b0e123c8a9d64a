#!/usr/bin/env python3
"""Timeline of the last iteration in a rocprofv3 --kernel-trace CSV: start/end (us, relative), queue, kernel.  Iterations are
delimited by elbo_update_kernel launches.  usage: timeline.py <*_kernel_trace.csv>"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "elbo_update_kernel" in r[3]]
a, b = marks[-2], marks[-1]
it = rows[a + 1:b + 1]
t0 = it[0][0]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:52]
busy = {}
for s, e, q, n in it:
    print("%8.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, short(n)))
    busy[q] = busy.get(q, 0) + (e - s)
print("span %.1f us; per-queue busy:" % ((it[-1][1] - t0) / 1e3), {q: round(v / 1e3, 1) for q, v in busy.items()})
