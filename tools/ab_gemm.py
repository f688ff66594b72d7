#!/usr/bin/env python3
"""A/B of two builds of libsdrm_hip.so on the bare GEMM launches, interleaved rounds in ONE process on one device
(cdna_hip_programming.md rule 24):  python tools/ab_gemm.py tools/libsdrm_prev.so sdrm_amd/libsdrm_hip.so"""
import ctypes as C
import os
import sys

import numpy as np
import torch

SHAPES = [(0, 24576, 352, 448, "train fwd L0"), (0, 24576, 352, 352, "train fwd hid"), (0, 5504, 352, 352, "sample n=5429"),
          (0, 3072, 352, 352, "3072 rows"), (0, 1664, 832, 832, "ml100k"), (2, 352, 352, 24576, "wgrad unsplit")]


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    lib.sdrm_debug_gemm_time.restype = C.c_int
    lib.sdrm_debug_gemm_time.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_float), C.c_void_p]
    return lib


def main():
    torch.zeros(1, device="cuda")
    libs = [load(p) for p in sys.argv[1:3]]
    cfgs = [int(c) for c in os.environ.get("CFGS", "0").split(",")]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for v, M, N, K, label in SHAPES:
        for cfg in cfgs:
            res = [[], []]
            for r in range(7):
                for i, lib in enumerate(libs):
                    us = C.c_float()
                    rc = lib.sdrm_debug_gemm_time(v, cfg, M, N, K, 30, C.byref(us), st)
                    res[i].append(us.value if rc == 0 else float("nan"))
            a, b = np.median(res[0]), np.median(res[1])
            print(f"{label:16s} cfg{cfg} {M}x{N}x{K}:  A {a:7.2f} us (min {min(res[0]):7.2f})   B {b:7.2f} us (min {min(res[1]):7.2f})   B/A {b / a:.3f}", flush=True)


if __name__ == "__main__":
    main()
