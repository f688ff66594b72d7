#!/usr/bin/env python3
"""ADM-shaped train steps (L=W=40, T=93, H=5, B=850) for `rocprofv3 --kernel-trace --stats`: where a skinny net's step goes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H, B = 40, 40, 93, 5, 850
e = Engine(L, W, T, H, B)
e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()
for _ in range(20): e.train_step(x0, 1e-5, seed=1, step=0)
torch.cuda.synchronize(); t = time.perf_counter()
for k in range(200): e.train_step(x0, 1e-5, seed=1, step=k)
torch.cuda.synchronize()
print(f"ADM train step {(time.perf_counter() - t) / 200 * 1e6:.1f} us")
