#!/usr/bin/env python3
"""Persistent skinny sampler (ADM shape L=W=40, T=93, H=5): time per reverse step against the number of rows.
One work-group owns 16 rows for the whole loop, so n = 4096 is one work-group per CU: if the step time does not
fall with n the kernel is bound by the serial chain of a row's layers, not by throughput."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H = 40, 40, 93, 5
NS = [int(v) for v in os.environ.get("NS", "1024,4096,8192,9558,12288,16384,32768").split(",")]
e = Engine(L, W, T, H, max(NS))
e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
for n in NS:
    for multires in (False, True):
        for _ in range(3): e.sample(n, multires=multires, seed=1)
        torch.cuda.synchronize(); t = time.perf_counter()
        R = 10
        for k in range(R): e.sample(n, multires=multires, seed=1, call_id=k)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t) / R / T * 1e6
        print(f"n={n:6d} multires={int(multires)}  {us:7.2f} us/step  {1e6 / us:9.0f} steps/s  {n / us:8.1f} rows/us")
