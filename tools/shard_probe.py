#!/usr/bin/env python3
"""What one rank of an N-GPU run sees: train step at B/N rows and sample step at n/N rows on one GPU
(no collectives), with the per-kernel breakdown from the engine's event profiler."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H = 340, 340, 78, 1
for N in [int(v) for v in os.environ.get("SHARDS", "1,2,4,8").split(",")]:
    B, n = 8192 // N, (5429 + N - 1) // N
    e = Engine(L, W, T, H, max(B, n))
    e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
    x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()
    sums = torch.zeros(8, dtype=torch.float64, device="cuda"); grad = torch.zeros(e.P, device="cuda")
    def train():
        e.train_forward(x0, seed=1, step=3, sums=sums); e.train_backward(sums=sums, grad=grad); e.adam_step(1e-5, grad=grad)
    cpu_us = {}
    def timeit(fn, reps, tag=None):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): fn()
        t_enq = time.perf_counter() - t          # host time to enqueue (the GPU may still be busy)
        torch.cuda.synchronize()
        if tag: cpu_us[tag] = t_enq / reps * 1e6
        return (time.perf_counter() - t) / reps * 1e6
    tt = timeit(train, 100, 'train')
    e.sample_begin(n, seed=2, call_id=1)
    def samp():
        if e.sample_steps(1) == 0:
            e.sample_end(); e.sample_begin(n, seed=2, call_id=1)
    ts = timeit(samp, 156, 'sample')
    print(f"N={N}: rows/rank {B:5d} train step {tt:7.1f} us   n/rank {n:5d} sample step {ts:6.1f} us   "
          f"cycle(15:78) {15*tt+78*ts:8.0f} us  (compute only, no all-reduce); host enqueue {cpu_us['train']:.0f} / {cpu_us['sample']:.0f} us per step", flush=True)
    e.close()
