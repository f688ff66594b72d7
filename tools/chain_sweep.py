#!/usr/bin/env python3
"""Sampling throughput vs number of row chains (sdrm_debug_set_chains) at the row counts one rank of a
1/2/4/8-GPU run sees.  Two drivers: one reverse step per C-ABI call (what bench.py does) and the whole
reverse loop in one call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H = 340, 340, 78, 1
for n in [int(v) for v in os.environ.get("ROWS", "5429,2715,1358,679").split(",")]:
    e = Engine(L, W, T, H, n)
    e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
    for C in (1, 2, 3, 4):
        e.debug_set(chains=C)
        res = []
        for per_call in (1, T):
            def run(loops):
                for _ in range(loops):
                    e.sample_begin(n, seed=2, call_id=1)
                    while e.sample_steps(per_call) > 0:
                        pass
                    e.sample_end()
            run(1); torch.cuda.synchronize(); t = time.perf_counter(); run(3); torch.cuda.synchronize()
            res.append((time.perf_counter() - t) / (3 * T) * 1e6)
        print(f"n={n:5d} chains={C}: {res[0]:6.1f} us/step (1 step per call)  {res[1]:6.1f} us/step (whole loop per call)", flush=True)
    e.close()
