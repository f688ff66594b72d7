#!/usr/bin/env python3
"""Diagnostic (stamped build): cycles per K-step of the GEMM main loops against the number of work-groups resident on a CU.
256 x r equal work-groups (one 64x64 tile each, 140 K-steps of 16), all started together: r = 1..6 per CU.
    STAMPLIB=tools/libsdrm_stamps.so python tools/residency_probe.py"""
import ctypes as C, os, numpy as np
lib = C.CDLL(os.environ.get("STAMPLIB", "tools/libsdrm_stamps.so"))
lib.sdrm_debug_gemm_stamps.restype = C.c_int
K = int(os.environ.get('K', 2240))
RS = [int(v) for v in os.environ.get('RS', '1,2,3,4,5,6,7,8').split(',')]
WARM = int(os.environ.get('WARM', 200))
for v, name in ((2, "wgrad loop (k-major fragments, ds_read_b32)"), (0, "NT loop (k-minor fragments, ds_read_b128)")):
    print(name)
    for r in RS:
        M, N = 1024, 1024 * r
        if os.environ.get('MN'):   # e.g. MN=16384,64: every work-group streams its own A columns (no sharing through L2)
            M, N = (int(v) * (r if i == 0 else 1) for i, v in enumerate(os.environ['MN'].split(',')))
        mb = 4096
        buf = (C.c_ulonglong * (8 * mb))()
        nb = lib.sdrm_debug_gemm_stamps(v, 0, M, N, K, buf, mb, WARM)
        if nb <= 0:
            print(f"  r={r}: n/a ({nb})"); continue
        a = np.frombuffer(buf, dtype=np.uint64).reshape(mb, 8)[:nb].astype(np.int64)
        a = a[a[:, 3] > 0]
        loop = a[:, 2] - a[:, 1]
        cu = (a[:, 6] >> 8) & 0xF | ((a[:, 6] >> 12) & 1) << 4 | ((a[:, 6] >> 13) & 7) << 5 | (a[:, 7] & 0xF) << 8
        per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
        wall = (a[:, 5].max() - a[:, 4].min()) / 100.0
        ks = K // 16
        print(f"  {len(a):5d} work-groups, per CU min {per_cu.min()} med {int(np.median(per_cu))} max {per_cu.max()}: loop {np.median(loop) / ks:7.0f} cycles per K-step "
              f"(p10 {np.percentile(loop, 10) / ks:.0f}, p90 {np.percentile(loop, 90) / ks:.0f}); matrix pipe needs {512 * int(np.median(per_cu))}; launch wall {wall:.1f} us", flush=True)
