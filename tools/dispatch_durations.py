#!/usr/bin/env python3
"""Per-dispatch durations of one kernel from a rocprofv3 --kernel-trace csv (the back-to-back behaviour of a kernel:
does it hold its first launches' time for a second?).  usage: dispatch_durations.py <dir> <kernel-name-substring>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if not rows:
    sys.exit("no dispatch of %r" % sys.argv[2])
d = [(e - s) / 1e3 for s, e, _ in rows]
print(rows[0][2][:160])
print("dispatches %d  span %.3f s  mean %.1f us  min %.1f  max %.1f" % (len(d), (rows[-1][1] - rows[0][0]) / 1e9, sum(d) / len(d), min(d), max(d)))
for lo in range(0, len(d), 500):
    part = d[lo:lo + 500]
    print("dispatches %5d..%5d: mean %.1f us  min %.1f  max %.1f" % (lo, lo + len(part) - 1, sum(part) / len(part), min(part), max(part)))
print("first 60 (us):", " ".join("%.0f" % v for v in d[:60]))
print("last 60 (us):", " ".join("%.0f" % v for v in d[-60:]))
