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
# ---- VAE decode on the engine, and the fused decode + equal-sparsity chain, against the PyTorch hook + the stand-alone select
from sdrm_amd.train_SDRM import VAE


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, (latent, hidden, items, users, q) in {"ML-100k": (830, 930, 1008, 843, 0.937), "ML-1M": (340, 600, 3125, 5429, 0.9553),
                                               "ADM": (40, 200, 8582, 9558, 0.9877)}.items():
    tensors = [torch.from_numpy(t).cuda() for t in synth.synth_vae_decoder(latent, hidden, items, seed=5)]
    z = torch.from_numpy(synth.synth_latents(users, latent, seed=6)).cuda()
    vae = VAE(items, hidden, latent).cuda().eval()
    with torch.no_grad():
        for p_, t_ in zip((vae.decoder[0].weight, vae.decoder[0].bias, vae.decoder[2].weight, vae.decoder[2].bias), tensors):
            p_.copy_(t_)
        us_torch = timed(lambda: vae.decode(z))
        us_torch_chain = timed(lambda: e.equal_sparsity(vae.decode(z), q))
    us_dec = timed(lambda: e.vae_decode(z, *tensors))
    us_chain = timed(lambda: e.equal_sparsity(e.vae_decode(z, *tensors), q))
    flops = 2.0 * users * (latent * hidden + hidden * items)
    res["decode_" + name] = {"users": users, "items": items, "latent": latent, "hidden": hidden,
                             "engine_decode_us": round(us_dec, 1), "engine_decode_TF": round(flops / us_dec / 1e6, 1),
                             "torch_decode_us": round(us_torch, 1),
                             "engine_decode+equal_sparsity_us": round(us_chain, 1), "torch_decode+sdrm_equal_sparsity_us": round(us_torch_chain, 1),
                             "out_MB": round(users * items * 4 / 1e6, 1)}
    print("decode", name, res["decode_" + name], flush=True)
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
