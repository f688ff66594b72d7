#!/usr/bin/env python3
"""Diagnostic (stamped build, never benchmarked): where a GEMM work-group spends its cycles."""
import ctypes as C, sys, numpy as np
lib = C.CDLL("tools/libsdrm_stamps.so")
lib.sdrm_debug_gemm_stamps.restype = C.c_int
for cfg in (0, 2):
    lib.sdrm_debug_set_tile(cfg)
    for (v, M, N, K) in [(0, 24576, 352, 352), (0, 5504, 352, 352), (0, 512, 352, 352)]:
        mb = 8192
        buf = (C.c_ulonglong * (4 * mb))()
        nb = lib.sdrm_debug_gemm_stamps(v, M, N, K, buf, mb)
        a = np.frombuffer(buf, dtype=np.uint64).reshape(mb, 4)[:nb].astype(np.int64)
        t0 = a[:, 0].min()
        pro, loop, epi = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
        span = a[:, 3].max() - t0
        print(f"cfg{cfg} {M}x{N}x{K}: blocks {nb} kernel span {span} cyc; prologue med {np.median(pro):.0f} (p90 {np.percentile(pro,90):.0f}) "
              f"loop med {np.median(loop):.0f} (p90 {np.percentile(loop,90):.0f}) epilogue med {np.median(epi):.0f}; "
              f"start spread {np.percentile(a[:,0]-t0,[50,90,100])}")
