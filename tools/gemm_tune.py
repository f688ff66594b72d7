#!/usr/bin/env python3
"""Times the engine's MFMA GEMM kernel (through sdrm_debug_gemm) on the shapes of the ML-1M / ML-100k
train and sample steps, per tile configuration.  Run on the GPU box:  python tools/gemm_tune.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import _lib  # noqa: E402

lib = _lib.load()
SHAPES = [  # (variant, M, N, K, label)    variant 0: A[M,K]*B[N,K]^T   1: A[M,K]*B[K,N]   2: A[K,M]^T*B[K,N]
    (0, 24576, 352, 448, "train fwd L0  B=8192"),
    (0, 24576, 352, 352, "train fwd hid B=8192"),
    (1, 24576, 352, 352, "train dgrad   B=8192"),
    (2, 352, 352, 24576, "train wgrad   B=8192 (no split)"),
    (0, 5440, 352, 352, "sample fwd    n=5429"),   # the step's padded extent (85 row tiles of 64: 510 work-groups, ONE round); 5504 rows - what
                                                   # rounds 1-4 timed here - is 516 work-groups, a second round (profiles/r05_tile_rows_cliff.txt)
    (0, 3072, 352, 352, "3072 rows (8-GPU train shard)"),
    (1, 3072, 352, 352, "3072 rows dgrad"),
    (0, 6144, 352, 352, "6144 rows (4-GPU train shard)"),
    (0, 1408, 352, 352, "1358 rows (4-GPU sample shard)"),
    (0, 2752, 352, 352, "2715 rows (2-GPU sample shard)"),
    (0, 1024, 352, 352, "1024 rows"),
    (0, 512, 352, 352, "train fwd     B=160"),
    (0, 1664, 832, 928, "ml100k fwd L0 B=550"),
    (0, 1664, 832, 832, "ml100k fwd    B=550"),
    (0, 2560, 64, 160, "adm fwd L0    B=850"),
]


def run(variant, M, N, K, cfg, reps=50):
    us = C.c_float()
    rc = lib.sdrm_debug_gemm_time(variant, cfg, M, N, K, reps, C.byref(us), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return us.value, 2.0 * M * N * K / us.value / 1e6


if __name__ == "__main__":
    ncfg = int(os.environ.get("NCFG", "2"))
    for variant, M, N, K, label in SHAPES:
        row = [f"{label[:22]:22s} v{variant} {M:6d}x{N:4d}x{K:6d}"]
        for cfg in [int(c) for c in os.environ.get("CFGS", ",".join(str(i) for i in range(ncfg))).split(",")]:
            try:
                us, tf = run(variant, M, N, K, cfg)
                row.append(f"c{cfg}:{us:7.1f}us{tf:6.1f}TF")
            except AssertionError as exc:
                row.append(f"cfg{cfg}: n/a({exc})")
        print("  ".join(row), flush=True)
