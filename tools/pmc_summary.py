#!/usr/bin/env python3
"""Condenses rocprofv3 --pmc CSV output (counter_collection.csv files under a directory) into per-kernel,
per-counter averages per launch.  usage: pmc_summary.py DIR [DIR ...] > summary.csv"""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: [0.0, 0])
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                name = name.split("(")[0].split("<")[0]
                key = (name, row["Counter_Name"])
                acc[key][0] += float(row["Counter_Value"])
                acc[key][1] += 1
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "avg_per_launch", "launches"])
for (k, c), (s, n) in sorted(acc.items()):
    w.writerow([k, c, f"{s / n:.6g}", n])
