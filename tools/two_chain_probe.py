#!/usr/bin/env python3
"""Do two half-size launch chains on two streams beat one full-size chain?  (a train-size NT launch loses
about 27 % to its head and tail; two concurrent chains could fill each other's.)  The same kernel, same total work:
  one stream : reps launches of M x N x K
  two streams: reps launches of M/2 x N x K on each, from two host threads at once
Prints microseconds per full-size-equivalent launch.  usage: two_chain_probe.py [M N K reps]"""
import ctypes, sys, threading, time
import torch
from sdrm_amd import _lib

M, N, K, reps = (int(v) for v in (sys.argv[1:5] + [24576, 352, 352, 400][len(sys.argv) - 1:]))
lib = _lib.load()
fn = lib.sdrm_debug_gemm_time
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p]


def run(m, stream, out, i, cfg=0, variant=0):
    us = ctypes.c_float()
    rc = fn(variant, cfg, m, N, K, reps, ctypes.byref(us), ctypes.c_void_p(stream.cuda_stream))
    assert rc == 0, rc
    out[i] = us.value


def together(ms, cfg=0):
    streams = [torch.cuda.Stream() for _ in ms]
    out = [0.0] * len(ms)
    th = [threading.Thread(target=run, args=(m, s, out, i, cfg)) for i, (m, s) in enumerate(zip(ms, streams))]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    wall = (time.perf_counter() - t0) * 1e6 / reps
    return out, wall


for cfg in (0,):
    for _ in range(2):   # second round: warm clocks
        one, w1 = together([M], cfg)
        half, wh = together([M // 2], cfg)
        two, w2 = together([M // 2, M // 2], cfg)
        three, w3 = together([M // 3 // 64 * 64] * 3, cfg)
    print(f"tile {cfg}  {M}x{N}x{K}, {reps} launches per chain")
    print(f"  one chain, full rows      : {one[0]:7.2f} us per launch")
    print(f"  one chain, half rows      : {half[0]:7.2f} us per launch  (x2 = {2 * half[0]:.2f})")
    print(f"  two chains of half rows   : {two[0]:7.2f} / {two[1]:7.2f} us per launch pair  (host wall {w2:.2f})")
    print(f"  three chains of third rows: {max(three):7.2f} us per launch triple (host wall {w3:.2f})")
