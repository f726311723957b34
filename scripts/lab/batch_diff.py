#!/usr/bin/env python
"""Lab: the kernels of exactly ONE timed batch = (trace of bench.py --steps 2) - (trace of --steps 1), per kernel name.
usage: batch_diff.py trace_steps1.csv trace_steps2.csv"""
import collections
import csv
import re
import sys


def load(path):
    per = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(.*", "", r.get("Kernel_Name") or r.get("kernel_name"))[:90]
            per[name][0] += 1
            per[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return per


a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in set(a) | set(b):
    n = b[k][0] - a[k][0]
    us = b[k][1] - a[k][1]
    if n:
        rows.append((us, n, k))
tot = sum(r[0] for r in rows)
print(f"one batch: {tot / 1e3:.2f} ms of kernels in {sum(r[1] for r in rows)} launches")
for us, n, k in sorted(rows, reverse=True)[:60]:
    print(f"{us / 1e3:9.3f} ms {100 * us / tot:5.1f}%  n={n:5d}  avg {us / n:8.1f} us  {k}")
