#!/usr/bin/env python3
"""sdrm_train_step_sharded over a ONE-rank RCCL communicator at the row counts of 1/2/4/8-GPU shards (every collective really
issued; what it measures is the launch / hand-off side of the exchange, not link time):  SDRM_AR_BUCKETS=1|2 python tools/exchange_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H = 340, 340, 78, 1
for N in (1, 2, 4, 8):
    B = 8192 // N
    e = Engine(L, W, T, H, B)
    e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
    e.comm_init_rank(1, 0, Engine.comm_unique_id())
    x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()
    def run(fn):
        for k in range(40): fn(k)
        ts = []
        for r in range(5):
            torch.cuda.synchronize(); t = time.perf_counter()
            for k in range(100): fn(k)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 100 * 1e6)
        return np.median(ts)
    plain = run(lambda k: e.train_step(x0, 1e-5, seed=1, step=k))
    shard = run(lambda k: e.train_step_sharded(x0, 1e-5, seed=1, step=k))
    print(f"rows/rank {B:5d}: sdrm_train_step {plain:7.1f} us   sdrm_train_step_sharded (one-rank communicator) {shard:7.1f} us   exchange side +{shard - plain:5.1f} us", flush=True)
    e.close()
