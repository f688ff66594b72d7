#!/usr/bin/env python3
"""Train-step time against the split-K plan of the batched weight-gradient launch (SDRM_WGRAD_BLOCKS = work-groups a
problem's launch aims for; the slice count per problem follows, see pick_splits), one engine per setting, rounds
interleaved in one process.  env B, L, T, H."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import synth  # noqa: E402
from sdrm_amd.engine import Engine  # noqa: E402

B, L, T, H = (int(os.environ.get(k, d)) for k, d in (("B", 8192), ("L", 340), ("T", 78), ("H", 1)))
VAR = os.environ.get("VAR", "SDRM_WGRAD_BLOCKS")      # or SDRM_WGRAD_SLICES: one slice count for every problem of the batch
targets = [int(v) for v in os.environ.get("TARGETS", "1024,300,450,600,800,1500,2300").split(",")]
x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()
engines = {}
for tgt in targets:
    os.environ[VAR] = str(tgt)
    e = Engine(L, L, T, H, B)
    e.set_params(synth.flatten_params(synth.init_params(L, L, T, H, seed=1), H))
    engines[tgt] = e
res = {t: [] for t in targets}
for r in range(6):
    for tgt, e in engines.items():
        for _ in range(5):
            e.train_step(x0, 1e-4, seed=1, step=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(40):
            e.train_step(x0, 1e-4, seed=1, step=k)
        torch.cuda.synchronize()
        res[tgt].append((time.perf_counter() - t0) / 40 * 1e6)
for tgt in targets:
    e = engines[tgt]
    e.profile_begin(64)
    for k in range(3):
        e.train_step(x0, 1e-4, seed=1, step=k)
    prof = e.profile_end()
    wg = [v for k, v in prof.items() if "wgrad" in k]
    print(f"{VAR}={tgt:5d}: train step med {np.median(res[tgt][1:]):7.1f} us (min {min(res[tgt]):7.1f}); "
          f"batched wgrad launch {wg[0][0] / wg[0][1] * 1e3 if wg else float('nan'):6.1f} us", flush=True)
