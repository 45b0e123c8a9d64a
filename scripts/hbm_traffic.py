#!/usr/bin/env python3
"""HBM bytes per launch of every kernel from two rocprofv3 passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace only),
joined with the median launch duration of a third --kernel-trace run.  FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950
tallies 128-byte requests at 64 bytes: MI355X_MICROARCH.md, HBM section).
usage: hbm_traffic.py <fetch_dir> <write_dir> <trace_dir> [skip_first_n_launches_per_kernel]"""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+)(<[^(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:70]


def counters(d, name):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg


def durations(d):
    f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        g = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
        agg[(short(r["Kernel_Name"]), g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return agg


fetch, write, dur = counters(sys.argv[1], "FETCH_SIZE"), counters(sys.argv[2], "WRITE_SIZE"), durations(sys.argv[3])
med = lambda v: sorted(v)[len(v) // 2]
rows = []
for key in fetch:
    rd = 2 * med(fetch[key]) * 1024 / 1e6
    wr = med(write.get(key, [0.0])) * 1024 / 1e6
    us = med(dur[key]) if key in dur else float("nan")
    rows.append((len(fetch[key]) * us, key, rd, wr, us, len(fetch[key])))
print("# kernel | grid threads | HBM read MB per launch (2 x FETCH_SIZE) | write MB (WRITE_SIZE) | median us | launches | (read+write)/time")
for _, key, rd, wr, us, n in sorted(rows, key=lambda r: -r[0] if r[0] == r[0] else 0):
    print("%-58s %9d  rd %8.1f  wr %8.1f  %8.1f us  n=%3d  %6.0f GB/s" % (key[0][:58], key[1], rd, wr, us, n, (rd + wr) / us * 1e3 if us == us and us > 0 else 0))
