#!/usr/bin/env python3
"""Captured graph vs stream launches for the latency-bound configurations (ADM, ML-1M at B = 160, an 8-GPU shard of the
headline job): is the step bound by the host's launch rate or by the GPU's dependent-launch chain?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine


def timeit(fn, reps):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    t_enq = time.perf_counter() - t
    torch.cuda.synchronize()
    return t_enq / reps * 1e6, (time.perf_counter() - t) / reps * 1e6


for name, (L, W, T, H, B, n) in {"ADM": (40, 40, 93, 5, 850, 9558), "ML-1M B=160": (340, 340, 78, 1, 160, 5429),
                                "ML-1M 8-GPU shard (B=1024, n=679)": (340, 340, 78, 1, 1024, 679), "ML-100k": (830, 830, 83, 2, 550, 843)}.items():
    e = Engine(L, W, T, H, max(B, n))
    e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
    x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()
    enq, wall = timeit(lambda: e.train_step(x0, 1e-5, seed=1, step=0), 200)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        e.train_step(x0, 1e-5, seed=1, step=0)
    _, gwall = timeit(g.replay, 200)
    line = f"{name}: train step: host enqueue {enq:.1f} us, stream wall {wall:.1f} us, graph replay {gwall:.1f} us"
    if not (L <= 64 and W <= 64):
        def sample_stream():
            e.sample_begin(n, seed=1, call_id=0)
            e.sample_steps(T)
        enq, wall = timeit(sample_stream, 10)
        g2 = torch.cuda.CUDAGraph()
        e.sample_begin(n, seed=1, call_id=0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g2):
            e.sample_steps(T)
        _, gwall = timeit(g2.replay, 10)
        line += f"; reverse step: host enqueue {enq / T:.1f} us, stream wall {wall / T:.1f} us, graph replay {gwall / T:.1f} us"
    print(line, flush=True)
    e.close()
