#!/usr/bin/env python
"""Lab: idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (one stream): start[i+1] - end[i]."""
import collections
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*", "", r.get("Kernel_Name") or r.get("kernel_name"))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
# the steady part: the last third of the trace
rows = rows[len(rows) // 3:]
busy = sum(e - s for s, e, _ in rows) / 1e3
span = (rows[-1][1] - rows[0][0]) / 1e3
gaps = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1)]
small = [g for g in gaps if g < 50]
print(f"{len(rows)} launches: span {span/1e3:.3f} ms, kernels busy {busy/1e3:.3f} ms, gaps {sum(gaps)/1e3:.3f} ms "
      f"({sum(small)/1e3:.3f} ms in {len(small)} gaps < 50 us: mean {sum(small)/len(small):.2f} us, median {sorted(small)[len(small)//2]:.2f} us)")
by = collections.defaultdict(list)
for i, g in enumerate(gaps):
    if g < 50:
        by[rows[i][2][:60] + " -> " + rows[i + 1][2][:40]].append(g)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"  {sum(v):8.1f} us  n={len(v):4d}  mean {sum(v)/len(v):6.2f}  {k}")
