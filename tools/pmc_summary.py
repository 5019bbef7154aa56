#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter, mean value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0][-60:]
            acc[k][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    print("#", path)
    for k, cs in acc.items():
        for c, vals in cs.items():
            per = defaultdict(float)
            for d, v in vals:
                per[d] += v
            vs = list(per.values())
            print(f"{k:60s} {c:24s} dispatches={len(vs)} mean={sum(vs) / len(vs):.6g}")
