"""Effective shader clock per kernel from a rocprofv3 `--pmc GRBM_GUI_ACTIVE --kernel-trace` pass:
clock = GRBM_GUI_ACTIVE / 8 (rocprofv3 reports the sum over the 8 XCDs) / dispatch wall time (MI355X_MICROARCH.md 'DVFS give-back').
The quotient reads HIGH on dispatches much shorter than 0.3 ms (the counter runs from before the first wave to after the last), so the
table lists the mean dispatch duration beside it.   usage: grbm_clock.py <dir with *counter_collection.csv> [min_us=20]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
acc = defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    a = acc[r["Kernel_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9; a[2] += 1
rows = sorted(((v[1] / v[2], k, v) for k, v in acc.items()), reverse=True)
print("# effective clock = GRBM_GUI_ACTIVE / 8 / wall (GHz); dispatches shorter than ~0.3 ms read high")
print(f"{'kernel':90s} calls  avg_us  GHz")
for avg, k, v in rows:
    if avg * 1e6 < min_us:
        continue
    name = k.replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{name[:88]:90s} {v[2]:5d} {avg * 1e6:7.1f} {v[0] / 8 / v[1] / 1e9:5.2f}")
