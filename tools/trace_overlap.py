#!/usr/bin/env python3
"""How much of a kernel trace runs concurrently?  rocprofv3 --kernel-trace CSV (start / end timestamps per dispatch, the HW queue of
each): over the last `frac` of the trace (default: the second half - warm-up is in the first) the span, the union of the kernels'
intervals (the time at least one kernel is in flight), the time at least two are, the sum of durations, per queue the busy time.
    python3 tools/trace_overlap.py <dir with *_kernel_trace.csv> [frac] ["title"]"""
import csv, glob, os, sys
d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
title = sys.argv[3] if len(sys.argv) > 3 else d
files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = []
with open(files[-1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1 - frac)
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
ev = []
for s, e, q, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth = 0; last = ev[0][0]; busy1 = busy2 = 0
for t, dlt in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += dlt; last = t
total = sum(e - s for s, e, _, _ in rows)
print(f"# {title}")
print(f"dispatches {len(rows)}, span {span / 1e6:.3f} ms: at least one kernel in flight {100 * busy1 / span:.1f} % of it, at least two {100 * busy2 / span:.1f} %; "
      f"sum of kernel durations {total / 1e6:.3f} ms = {total / span:.2f} x the span")
by = {}
for s, e, q, n in rows:
    b = by.setdefault(q, [0, 0]); b[0] += e - s; b[1] += 1
for q, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print(f"  queue {q}: {n} dispatches, {t / 1e6:.3f} ms of kernel time ({100 * t / span:.1f} % of the span)")
