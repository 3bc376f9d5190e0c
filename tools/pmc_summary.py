#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel name."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"][:96]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, d in acc.items():
    print(name)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
