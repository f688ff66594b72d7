#!/usr/bin/env python3
"""Reverse-sampling step time of the ML-1M net (340, 340, 78, 1) by row count: the per-layer path (three launches per step, the
reverse update fused by the size rule) against the persistent sampler (csrc/sample_persist.h), driven one step per call
(what bench.py does) and one call for the whole loop (what sample_ddpm does).  PHILOX mode, HIP events.

    python3 tools/sample_persist_probe.py [rows,rows,..]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import synth  # noqa: E402
from sdrm_amd.engine import Engine  # noqa: E402

L, W, T, H = 340, 340, 78, 1
rows = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [339, 679, 1024, 1358, 2048, 2715]
init = synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H)
for n in rows:
    line = []
    for name, mode in (("per-layer", 0), ("persistent", 2)):
        e = Engine(L, W, T, H, max_rows=n).debug_set(sample_persist=mode)
        e.set_params(init)
        for chunk in (1, T):
            for rep in range(2):
                e.sample_begin(n, seed=2, call_id=rep)
                while e.sample_steps(chunk) != 0:
                    pass
                e.sample_end()
            torch.cuda.synchronize()
            l0 = e.launch_count()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for rep in range(6):
                e.sample_begin(n, seed=2, call_id=10 + rep)
                while e.sample_steps(chunk) != 0:
                    pass
                e.sample_end()
            t1.record()
            torch.cuda.synchronize()
            line.append(f"{name} x{chunk}: {1e3 * t0.elapsed_time(t1) / 6 / T:6.2f} us/step ({(e.launch_count() - l0) / 6 / T:.2f} launches)")
        e.close()
    print(f"n = {n:5d}: " + " | ".join(line), flush=True)
