#!/usr/bin/env python3
"""Equal-sparsity binarisation (csrc/select.h) on the BASELINE shapes: device time per call, the
algorithmic HBM rate (17 B per element: three select sweeps + binarise read + 1-byte write) against 8 TB/s, and
numpy's np.quantile + compare on the host cores beside it."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine

e = Engine(8, 8, 4, 0, 16)
res = {}
for name, (users, items, q) in {"ML-100k": (843, 1008, 0.937), "ML-1M": (5429, 3125, 0.9553), "ADM": (9558, 8582, 0.9877)}.items():
    M = synth.synth_scores(users, items, seed=3)
    x = torch.from_numpy(M).cuda()
    for _ in range(3):
        e.equal_sparsity(x, q)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    ev0.record()
    for _ in range(reps):
        out = e.equal_sparsity(x, q)
    ev1.record(); torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) / reps * 1e3
    t = time.perf_counter()
    want = (M >= np.quantile(M.flatten(), q))
    cpu_s = time.perf_counter() - t
    assert np.array_equal(out.cpu().numpy().astype(bool), want)
    n = users * items
    res[name] = {"elements": n, "device_us": round(us, 1), "algorithmic_GBps": round(17.0 * n / us / 1e3, 1),
                 "frac_of_8TBps": round(17.0 * n / us / 1e3 / 8000.0, 3), "numpy_cpu_ms": round(cpu_s * 1e3, 1),
                 "speedup": round(cpu_s * 1e6 / us, 1)}
    print(name, res[name], flush=True)
print(json.dumps(res))
