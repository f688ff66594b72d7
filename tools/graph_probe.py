#!/usr/bin/env python3
"""Does a captured graph shorten the gaps between the dependent launches of a step?  ML-1M shape, N=1.
Captures (a) the 78 sampling steps of one call, (b) one train step, replays them and times both ways."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H, B, n = 340, 340, 78, 1, 8192, 5429
e = Engine(L, W, T, H, max(B, n))
e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()

def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6

def sample_stream():
    e.sample_begin(n, seed=1, call_id=0)
    e.sample_steps(T)
us = timeit(sample_stream, 20)
print(f"sampling call, stream launches: {us / T:.2f} us/step")
g = torch.cuda.CUDAGraph()
e.sample_begin(n, seed=1, call_id=0)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    e.sample_steps(T)
us = timeit(g.replay, 20)
print(f"sampling call, graph replay:    {us / T:.2f} us/step")

def train_stream():
    e.train_step(x0, 1e-5, seed=1, step=0)
us = timeit(train_stream, 50)
print(f"train step, stream launches: {us:.1f} us")
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    e.train_step(x0, 1e-5, seed=1, step=0)
us = timeit(g2.replay, 50)
print(f"train step, graph replay:    {us:.1f} us")
