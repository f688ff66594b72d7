#!/usr/bin/env python3
"""Train-step time of one config for env sweeps (SDRM_WGRAD_SLICES, SDRM_WGRAD_BLOCKS, SDRM_TILE ...):
    CFG=ml1m|ml100k|b160|adm python tools/train_step_probe.py   -> microseconds per train step (median of 5 x 100 steps)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
CFGS = {"ml1m": (340, 340, 78, 1, 8192), "shard8": (340, 340, 78, 1, 1024), "shard4": (340, 340, 78, 1, 2048), "ml100k": (830, 830, 83, 2, 550), "b160": (340, 340, 78, 1, 160), "adm": (40, 40, 93, 5, 850)}
L, W, T, H, B = CFGS[os.environ.get("CFG", "ml1m")]
e = Engine(L, W, T, H, B)
e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()
for k in range(60): e.train_step(x0, 1e-5, seed=1, step=k)
ts = []
for r in range(5):
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(100): e.train_step(x0, 1e-5, seed=1, step=k)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 100 * 1e6)
print(f"{np.median(ts):.1f}")
