#!/usr/bin/env python3
"""Per-iteration timeline from a rocprofv3 --kernel-trace CSV: span, sum of kernel durations, idle gaps (by the kernel that
precedes them).  Iterations are delimited by the adam_kernel / elbo_update_kernel launches.  Usage: trace_gaps.py <*_kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2] or "elbo_update" in r[2]]
print("kernels", len(rows), "adam launches", len(adam))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
for a, b in list(zip(adam[:-1], adam[1:]))[-3:]:
    it = rows[a + 1:b + 1]
    span = it[-1][1] - it[0][0]
    busy = sum(e - s for s, e, _ in it)
    gaps = defaultdict(lambda: [0, 0])
    for (s0, e0, n0), (s1, e1, n1) in zip(it[:-1], it[1:]):
        g = s1 - e0
        if g > 0:
            gaps[short(n0) + " -> " + short(n1)][0] += g; gaps[short(n0) + " -> " + short(n1)][1] += 1
    tot_gap = sum(v[0] for v in gaps.values())
    print("iteration: %d kernels, span %.3f ms, busy %.3f ms, idle %.3f ms" % (len(it), span / 1e6, busy / 1e6, tot_gap / 1e6))
    per = defaultdict(lambda: [0, 0])
    for s, e, n in it:
        per[short(n)][0] += e - s; per[short(n)][1] += 1
for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])[:40]:
    print("  %-62s %4d  %8.1f us" % (k, v[1], v[0] / 1e3))
print("largest idle gaps (last iteration):")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:15]:
    print("  %-100s %3d  %7.1f us" % (k, v[1], v[0] / 1e3))
