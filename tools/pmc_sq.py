#!/usr/bin/env python3
"""Per-kernel SQ counters from a `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT` run (csv output):

    python tools/pmc_sq.py <rocprof output dir> [out.json]

Per kernel and grid size: the wave-cycle fractions (parked on s_waitcnt/s_barrier, stalled at issue, issuing), LDS
bank-conflict cycles, and the matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), the
kernel cycles taken from SQ_BUSY_CYCLES / 32 (the counter sums the busy cycles of the 32 shader engines' SQs...) - when
that normalisation is in doubt, compare kernels with each other rather than against 1.0."""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import _build  # noqa: E402  (stamp: hash of the kernel sources the counters belong to)

acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sdrm::" not in k:
            continue
        key = (k, int(r["Grid_Size"]))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(key, r["Counter_Name"])] += 1
out = {}
for key, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    k, grid = key
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    calls = max(n[(key, "SQ_WAVE_CYCLES")], 1)
    row = {"grid_size": grid, "launches": calls,
           "wave_cycles_per_launch": wc / calls,
           "mfma_busy_cycles_per_launch": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / calls,
           "mfma_busy_per_simd_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / calls / 1024.0,
           "frac_wave_parked": c.get("SQ_WAIT_ANY", 0.0) / wc,
           "frac_wave_issue_stalled": c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
           "frac_wave_issuing": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
           "frac_wave_lds_issue_stalled": c.get("SQ_WAIT_INST_LDS", 0.0) / wc,
           "lds_bank_conflict_per_wave_cycle": c.get("SQ_LDS_BANK_CONFLICT", 0.0) / wc}
    out.setdefault(k, []).append(row)
    print(f"{k[:96]:96s} grid {grid:8d} x{calls:4d}  mfma busy/SIMD {row['mfma_busy_per_simd_cycles']:9.0f} cyc  parked {row['frac_wave_parked']:.2f} "
          f"issue-stalled {row['frac_wave_issue_stalled']:.2f} issuing {row['frac_wave_issuing']:.2f} lds-conflict {row['lds_bank_conflict_per_wave_cycle']:.3f}")
if len(sys.argv) > 2:
    json.dump({"note": "mfma_busy_per_simd_cycles / (kernel duration x shader clock) = matrix-pipe utilisation",
               "source_hash": _build.source_hash(), "git_head": os.environ.get("GIT_HEAD"), "kernels": out},
              open(sys.argv[2], "w"), indent=1)
