#!/usr/bin/env python3
"""The SURVEY §8f rows built so far, on the BASELINE shapes, with the host (numpy) form timed beside them:
  * equal-sparsity binarisation (csrc/select.h): device time per call and the algorithmic HBM rate (17 B per element:
    three select sweeps + binarise read + 1-byte write) against 8 TB/s;
  * Recall/NDCG@k (csrc/rank.h): device time for k = 1,3,5,10,20,50 in one call, against sdrm_amd/metrics.py
    (the numpy restatement of utilities.py) called once per k as the reference does."""
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
from sdrm_amd import metrics
for name, (users, items) in {"ML-100k": (843, 1008), "ML-1M": (5429, 3125), "ADM": (9558, 8582)}.items():
    scores = synth.synth_scores(users, items, seed=4)
    train, held = synth.synth_interactions(users, items, seed=5, p_train=0.04, p_held=0.01)
    x = torch.from_numpy(scores).cuda()
    ks = (1, 3, 5, 10, 20, 50)
    for _ in range(2):
        e.rank_metrics(x, held, train, ks)
    hp, hi = (torch.from_numpy(held.indptr.astype(np.int64)).cuda(), torch.from_numpy(held.indices.astype(np.int32)).cuda())
    torch.cuda.synchronize()
    t = time.perf_counter()
    r, n = e.rank_metrics(x, held, train, ks)
    torch.cuda.synchronize()
    dev_ms = (time.perf_counter() - t) * 1e3          # includes the CSR index upload of the Python wrapper
    t = time.perf_counter()
    masked = metrics.mask_training_examples(train, scores.copy())
    with np.errstate(all="ignore"):
        want = [(metrics.recall_at_k_batch(masked.copy(), held, k=k), metrics.NDCG_binary_at_k_batch(masked.copy(), held, k=k)) for k in ks]
    cpu_ms = (time.perf_counter() - t) * 1e3
    ok = all(np.array_equal(np.nan_to_num(r[q].cpu().numpy(), nan=-1), np.nan_to_num(want[q][0], nan=-1)) for q in range(len(ks)))
    res["rank_" + name] = {"users": users, "items": items, "held": int(held.nnz), "device_ms_incl_index_upload": round(dev_ms, 2),
                           "numpy_cpu_ms": round(cpu_ms, 1), "recall_identical": bool(ok)}
    print("rank", name, res["rank_" + name], flush=True)
print(json.dumps(res))
