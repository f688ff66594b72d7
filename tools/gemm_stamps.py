#!/usr/bin/env python3
"""Diagnostic (stamped build, never benchmarked): where a GEMM work-group spends its cycles, and the clock the chip
holds while the launch runs (s_memtime / s_memrealtime around every work-group after ~2 s of back-to-back launches).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSDRM_STAMPS -o tools/libsdrm_stamps.so sdrm_amd/csrc/sdrm_hip.hip -ldl
    CFGS=0,4 python tools/gemm_stamps.py"""
import ctypes as C, sys, numpy as np
import os
lib = C.CDLL(os.environ.get("STAMPLIB", "tools/libsdrm_stamps.so"))
lib.sdrm_debug_gemm_stamps.restype = C.c_int
WARM_S = float(os.environ.get("WARM_S", "2.0"))
for cfg in [int(c) for c in os.environ.get("CFGS", "0,1").split(",")]:
    for (v, M, N, K, us) in [(0, 24576, 352, 352, 59.0), (0, 24576, 352, 448, 71.0), (2, 352, 352, 24576, 400.0), (0, 5440, 352, 352, 16.0), (0, 1344, 352, 352, 8.0)]:
        mb = 8192
        buf = (C.c_ulonglong * (8 * mb))()
        nb = lib.sdrm_debug_gemm_stamps(v, cfg, M, N, K, buf, mb, int(WARM_S * 1e6 / us))
        if nb <= 0:
            print(f"cfg{cfg} v{v} {M}x{N}x{K}: n/a ({nb})")
            continue
        a = np.frombuffer(buf, dtype=np.uint64).reshape(mb, 8)[:nb].astype(np.int64)
        a = a[a[:, 3] > 0]
        pro, loop, epi = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
        life, real = a[:, 3] - a[:, 0], a[:, 5] - a[:, 4]
        ok = real > 0
        clk = np.median(life[ok] / real[ok]) * 0.1          # GHz: shader cycles per 10 ns tick of the 100 MHz counter
        wall_us = (a[:, 5].max() - a[:, 4].min()) / 100.0    # the 100 MHz counter is common to the chip
        print(f"cfg{cfg} v{v} {M}x{N}x{K}: {len(a)} stamped work-groups; launch wall {wall_us:.1f} us; clock held {clk:.3f} GHz; "
              f"lifetime med {np.median(life):.0f} cyc = prologue {np.median(pro):.0f} (p90 {np.percentile(pro, 90):.0f}) + loop {np.median(loop):.0f} "
              f"(p90 {np.percentile(loop, 90):.0f}) + epilogue {np.median(epi):.0f}; sum of lifetimes / (wall x clock x 256 CUs) = "
              f"{life.sum() / (wall_us * 1e3 * clk * 256):.2f} resident work-groups per CU", flush=True)
        if os.environ.get("FINE") and v == 0:   # -DSDRM_STAMPS=2 build: slots 6 / 7 = first loads issued / first K-step landed in LDS
            print(f"      prologue split: entry -> first loads issued {np.median(a[:, 6] - a[:, 0]):.0f} (p90 {np.percentile(a[:, 6] - a[:, 0], 90):.0f}) cyc, "
                  f"-> first K-step landed {np.median(a[:, 7] - a[:, 6]):.0f} (p90 {np.percentile(a[:, 7] - a[:, 6], 90):.0f}), "
                  f"-> fragments read, both barriers {np.median(a[:, 1] - a[:, 7]):.0f} (p90 {np.percentile(a[:, 1] - a[:, 7], 90):.0f})")
        if os.environ.get("TIMELINE") and M >= 5000 and v == 0:
            t0 = a[:, 4].min()
            st, en = (a[:, 4] - t0) / 100.0, (a[:, 5] - t0) / 100.0
            pro_end = st + pro / (clk * 1e3)
            loop_end = pro_end + loop / (clk * 1e3)
            edges = np.arange(0, wall_us + 2.0, 2.0)
            print("   t[us]  alive  in-prologue  in-loop  in-epilogue  started  (work-groups on the whole chip, 2-us bins, sampled at the bin centre)")
            for lo in edges[:-1]:
                c = lo + 1.0
                alive = ((st <= c) & (en > c)).sum()
                inpro = ((st <= c) & (pro_end > c)).sum()
                inloop = ((pro_end <= c) & (loop_end > c)).sum()
                print(f"   {c:5.1f}  {alive:5d}  {inpro:11d}  {inloop:7d}  {alive - inpro - inloop:11d}  {((st >= lo) & (st < lo + 2.0)).sum():7d}")
