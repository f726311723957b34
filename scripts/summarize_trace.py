#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV: total / count / avg per (kernel, grid) and per kernel."""
import csv, sys, collections, re
path = sys.argv[1]
per = collections.defaultdict(lambda: [0, 0.0])
perk = collections.defaultdict(lambda: [0, 0.0])
with open(path) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("kernel_name")
        name = re.sub(r"\(.*", "", name)
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        grid = (r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        key = (name, grid)
        per[key][0] += 1; per[key][1] += dur
        perk[name][0] += 1; perk[name][1] += dur
tot = sum(v[1] for v in perk.values())
print(f"total kernel time {tot/1e3:.2f} ms over {sum(v[0] for v in perk.values())} launches")
print("== per kernel ==")
for k, v in sorted(perk.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1]/1e3:10.2f} ms {100*v[1]/tot:5.1f}% n={v[0]:6d} avg={v[1]/v[0]:9.1f} us  {k[:110]}")
print("== per (kernel, grid) ==")
for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{v[1]/1e3:10.2f} ms {100*v[1]/tot:5.1f}% n={v[0]:6d} avg={v[1]/v[0]:9.1f} us grid={k[1]} {k[0][:90]}")
