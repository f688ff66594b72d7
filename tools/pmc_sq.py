#!/usr/bin/env python3
"""Per-kernel SQ counters from a rocprofv3 --pmc run (csv): sums per kernel name, ratios to SQ_WAVE_CYCLES."""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sdrm::" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    calls = max(n[(k, "SQ_WAVE_CYCLES")], 1)
    print(f"{k[:100]:100s} calls {calls:4d} " + " ".join(f"{name[3:]}={v / wc:.3f}" for name, v in sorted(c.items()) if name != "SQ_WAVE_CYCLES")
          + f" WAVE_CYCLES/call={wc / calls:.3e} MFMA_BUSY/call={c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / calls:.3e} BUSY_CYCLES/call={c.get('SQ_BUSY_CYCLES', 0) / calls:.3e}")
