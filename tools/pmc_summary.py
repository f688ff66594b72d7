#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes of `python bench.py` (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE;
the TCC slots do not fit both, MI355X_MICROARCH.md "rocprofv3 PMC slots") into per-kernel HBM bytes per launch.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Units and correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of wide (16 B/lane) coalesced reads, so bytes = (2*FETCH + WRITE) * 1024.
Calibration on this workload's own access pattern: the forward GEMM at 24576x352x352 must fetch its
activation matrix once (34.6 MB) + weights (0.5 MB): 2*FETCH gives 38 MB; its WRITE is exactly 24576*352*4."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import _build  # noqa: E402  (stamp: hash of the kernel sources the counters belong to)


def per_kernel(d):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            out[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return out


def main():
    fetch, write, dst = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), sys.argv[3]
    res = {}
    for key in sorted(set(fetch) | set(write)):
        name, grid = key
        if "sdrm::" not in name:
            continue
        f = sum(fetch.get(key, [0])) / max(len(fetch.get(key, [0])), 1)
        w = sum(write.get(key, [0])) / max(len(write.get(key, [0])), 1)
        res.setdefault(name, []).append({"grid_size": grid, "launches": len(fetch.get(key, [])), "FETCH_SIZE_KiB": round(f, 1),
                                         "WRITE_SIZE_KiB": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024)})
    # launch-weighted mean per kernel name (what `rocprofv3 --stats` averages over)
    summary = {}
    for name, rows in res.items():
        n = sum(r["launches"] for r in rows) or 1
        summary[name] = {"hbm_bytes_per_launch_mean": int(sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows) / n), "by_grid": rows}
    json.dump({"unit": "bytes", "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024  [gfx950 correction]",
               "source_hash": _build.source_hash(), "git_head": os.environ.get("GIT_HEAD"), "kernels": summary},
              open(dst, "w"), indent=1)
    for name, v in summary.items():
        print(f"{v['hbm_bytes_per_launch_mean'] / 1e6:10.2f} MB  {name[:110]}")


if __name__ == "__main__":
    main()
