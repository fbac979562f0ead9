#!/usr/bin/env python3
"""Average every counter per kernel over the dispatches of the rocprofv3 --pmc passes under <dir>: tools/pmc_table.py <dir>"""
import collections, csv, glob, json, os, sys
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
    for (disp, k, c), v in per.items():
        acc[k][c].append(v)
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if "cry::" in k}
json.dump(out, open(os.path.join(d, "pmc_table.json"), "w"), indent=1)
for k, cs in out.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-40s %16.1f" % (c, v))
