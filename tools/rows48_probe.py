#!/usr/bin/env python3
"""Train-step time of the ML-1M net (340, 340, 78, 1) by batch size and forward family: per-layer tiles, 96-row work-groups
(csrc/rowchain.h), 48-row work-groups (csrc/rows48.h).  PHILOX mode, one engine per case, HIP events over R steps.

    python3 tools/rows48_probe.py [R]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import synth  # noqa: E402
from sdrm_amd.engine import Engine  # noqa: E402

L, W, T, H = 340, 340, 78, 1
R = int(sys.argv[1]) if len(sys.argv) > 1 else 60
MODES = {"tiles": dict(rowchain=0, rows48=0), "rows96": dict(rowchain=2, rows48=0), "rows48": dict(rowchain=0, rows48=2, rows48_split=0),
         "x2": dict(rowchain=0, rows48=2, rows48_split=2), "x2+tile-wgrad": dict(rowchain=0, rows48=2, rows48_split=2, wgrad_strips=0),
         "x4": dict(rowchain=0, rows48=2, rows48_split=4), "x4+tile-wgrad": dict(rowchain=0, rows48=2, rows48_split=4, wgrad_strips=0),
         "auto": dict()}
if len(sys.argv) > 2:
    MODES = {k: v for k, v in MODES.items() if k in sys.argv[2].split(",")}
init = synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H)
for B in [int(b) for b in sys.argv[3].split(',')] if len(sys.argv) > 3 else (512, 768, 1024, 1536, 2048, 2816, 3072, 4096, 5120, 6144, 8192):
    x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()
    line = []
    for name, kw in MODES.items():
        e = Engine(L, W, T, H, max_rows=B).debug_set(**kw)
        e.set_params(init)
        for k in range(8):
            e.train_step(x0, 1e-5, seed=1, step=k)
        torch.cuda.synchronize()
        l0 = e.launch_count()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for k in range(R):
            e.train_step(x0, 1e-5, seed=1, step=8 + k)
        t1.record()
        torch.cuda.synchronize()
        line.append(f"{name} {1e3 * t0.elapsed_time(t1) / R:7.1f} us ({(e.launch_count() - l0) / R:.0f} launches)")
        e.close()
    print(f"B = {B:5d}: " + " | ".join(line), flush=True)
