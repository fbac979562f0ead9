#!/usr/bin/env python3
"""Reduce a rocprofv3 --kernel-trace --stats output dir to the rows of this library's kernels.
usage: tools/prof_summary.py <rocprof_dir> <out.csv> [note]"""
import csv, glob, os, sys
d, out = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
keep = [r for r in rows if "cry::" in r["Name"]]
with open(out, "w") as o:
    if note: o.write("# %s\n" % note)
    o.write("# source: rocprofv3 --kernel-trace --stats (%s); only cry:: kernels kept\n" % os.path.basename(f))
    w = csv.writer(o)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
    for r in keep:
        w.writerow([r["Name"].split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
print(open(out).read())
