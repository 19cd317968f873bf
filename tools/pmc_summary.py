#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes: per kernel, mean counter value per
dispatch.  usage: pmc_summary.py <dir containing pmc_*/...counter_collection.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            k = k.split("(")[0].replace("void ", "").replace("gkoc::", "").replace("(anonymous namespace)::", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k[:110])
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:40s} mean/dispatch = {sum(v)/len(v):18.1f}   (n={len(v)})")
