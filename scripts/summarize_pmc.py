#!/usr/bin/env python
"""Per-kernel average of one rocprofv3 PMC counter (counter_collection CSV)."""
import csv, glob, os, sys, collections
d, cnt = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: [0.0, 0])
for f in files:
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") != cnt:
            continue
        name = row["Kernel_Name"].split("(")[0]
        a = acc[name]; a[0] += float(row["Counter_Value"]); a[1] += 1
print("counter", cnt, "files", len(files))
for name, (tot, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("%-90s launches %5d  avg %14.1f" % (name[:90], n, tot / n))
