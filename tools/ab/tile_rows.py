import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdrm_amd import _lib
lib = _lib.load()
def run(M, cfg, reps=200):
    us = C.c_float()
    rc = lib.sdrm_debug_gemm_time(0, cfg, M, 352, 352, reps, C.byref(us), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    return us.value
for M in (5376, 5440, 5504, 5568):
    print(f"M = {M} ({M // 64} row tiles of 64 x 6 = {M // 64 * 6} work-groups; {M // 32 * 11} of 32x32): 64x64x16 {run(M, 0):.2f} us   32x32x32 {run(M, 4):.2f} us", flush=True)
