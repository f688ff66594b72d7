#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> average / minimum duration per (kernel, grid size): separates the train-size and the
sampling-size launches of one kernel template.  usage: kernel_by_grid.py <dir or *_kernel_trace.csv> [title]"""
import csv, glob, os, sys
from collections import defaultdict

src = sys.argv[1]
if os.path.isdir(src):
    cands = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    src = max(cands, key=os.path.getmtime)
acc = defaultdict(list)
for r in csv.DictReader(open(src)):
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    acc[(r["Kernel_Name"], grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("# " + (sys.argv[2] if len(sys.argv) > 2 else "rocprofv3 --kernel-trace: average duration per (kernel, grid size)"))
for (name, grid), d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if name.startswith("__amd") or "at::native" in name:
        continue
    print(f"{name[:100]:100s} grid {grid:8d} calls {len(d):4d} avg {sum(d) / len(d) / 1e3:7.2f} us  min {min(d) / 1e3:7.2f}")
