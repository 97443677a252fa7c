#!/usr/bin/env python3
"""Prints per-kernel averages of every counter in a rocprofv3 counter_collection.csv."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if n.startswith("at::") or "elementwise" in n:
            continue
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[n]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, cs in acc.items():
    print(n)
    for c, v in sorted(cs.items()):
        v = v[len(v) // 3:] if len(v) > 3 else v
        print(f"    {c:34s} {sum(v)/len(v):16.1f}  (n={len(v)})")
