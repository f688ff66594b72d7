#!/usr/bin/env python3
"""Diagnostic (stamped build, never benchmarked): where a GEMM work-group spends its cycles."""
import ctypes as C, sys, numpy as np
import os
lib = C.CDLL(os.environ.get("STAMPLIB", "tools/libsdrm_stamps.so"))
lib.sdrm_debug_gemm_stamps.restype = C.c_int
for cfg in [int(c) for c in os.environ.get("CFGS", "0,1").split(",")]:
    for (v, M, N, K) in [(0, 24576, 352, 352), (0, 5440, 352, 352), (0, 2720, 352, 352), (0, 1344, 352, 352)]:
        mb = 8192
        buf = (C.c_ulonglong * (4 * mb))()
        nb = lib.sdrm_debug_gemm_stamps(v, cfg, M, N, K, buf, mb)
        a = np.frombuffer(buf, dtype=np.uint64).reshape(mb, 4)[:nb].astype(np.int64)
        pro, loop, epi = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
        x0 = a[0::8]                                   # blocks of XCD 0 (every XCD has its own counter base)
        t0 = x0[:, 0].min()
        span = x0[:, 3].max() - t0
        print(f"cfg{cfg} {M}x{N}x{K}: blocks {nb} XCD0 span {span} cyc; prologue med {np.median(pro):.0f} (p90 {np.percentile(pro,90):.0f}) "
              f"loop med {np.median(loop):.0f} (p90 {np.percentile(loop,90):.0f}) epilogue med {np.median(epi):.0f}; "
              f"XCD0 block starts p10/50/90/100 {np.percentile(x0[:,0]-t0,[10,50,90,100]).astype(int)} ends p10/50/90 {np.percentile(x0[:,3]-t0,[10,50,90]).astype(int)}")
