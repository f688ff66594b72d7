#!/usr/bin/env python3
"""torch.matmul (rocBLAS / hipBLASLt fp32) on the GEMM shapes of the ML-1M train and sample steps, beside the engine's
kernel (tools/gemm_tune.py) - no epilogues, no PReLU-on-load, operands hot: a like-for-like check of the bare GEMM."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_tune as g
torch.backends.cuda.matmul.allow_tf32 = False
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / n * 1e6
for label, variant, M, N, K in [("train fwd hidden  (NT)", 0, 24576, 352, 352), ("train fwd layer 0 (NT)", 0, 24576, 352, 448),
                                ("train dgrad       (NN)", 1, 24576, 352, 352), ("train wgrad       (TN)", 2, 352, 352, 24576),
                                ("sample fwd        (NT)", 0, 5440, 352, 352), ("8-GPU shard fwd   (NT)", 0, 3072, 352, 352),
                                ("ml100k fwd        (NT)", 0, 1664, 832, 832)]:
    if variant == 0:
        A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); fn = lambda: torch.matmul(A, B.t())
    elif variant == 1:
        A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); fn = lambda: torch.matmul(A, B)
    else:
        A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda"); fn = lambda: torch.matmul(A.t(), B)
    us_lib = t(fn)
    us_eng, _ = g.run(variant, M, N, K, -1, reps=50)
    fl = 2.0 * M * N * K
    print(f"{label}  {M:6d}x{N:4d}x{K:6d}   torch.matmul {us_lib:7.1f} us {fl / us_lib / 1e6:6.1f} TF   engine kernel {us_eng:7.1f} us {fl / us_eng / 1e6:6.1f} TF"
          + ("   (engine: one launch, no split-K)" if variant == 2 else ""), flush=True)
