#!/usr/bin/env python3
"""ML-100k-shaped steps (L=W=830, T=83, H=2, B=550, n=843) for `rocprofv3 --kernel-trace --stats`."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H, B, n = 830, 830, 83, 2, 550, 843
e = Engine(L, W, T, H, max(B, n))
e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()
for _ in range(10): e.train_step(x0, 1e-5, seed=1, step=0)
torch.cuda.synchronize(); t = time.perf_counter()
for k in range(100): e.train_step(x0, 1e-5, seed=1, step=k)
torch.cuda.synchronize()
print(f"ML-100k train step {(time.perf_counter() - t) / 100 * 1e6:.1f} us")
for _ in range(2): e.sample(n, seed=1)
torch.cuda.synchronize(); t = time.perf_counter()
for k in range(5): e.sample(n, seed=1, call_id=k)
torch.cuda.synchronize()
print(f"ML-100k sample step {(time.perf_counter() - t) / 5 / T * 1e6:.1f} us")
