#!/usr/bin/env python3
"""End-to-end quality report: ML-100k / SVD downstream, README hyper-parameters, N seeds (argv[1], default 5) on the MI355X
engine, next to the reference's own runs (tests/golden/e2e_ml100k_svd.npz, ten seeds run on CPU in the build container).
Writes gpurun_out/e2e_report.json with the difference of the means and its standard error (Welch)."""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import pipeline
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
HP = dict(epochs=265, batch=550, lr=2.1e-5, T=83, nd=1.0, H=2, vae_batch=780, vae_hidden=930, latent=830, vae_lr=6e-4)
ref = np.load(os.path.join(GOLD, "e2e_ml100k_svd.npz"))
split = pipeline.load_split(os.path.join(GOLD, "ml100k.npz"))
out = {"hyper": HP, "k": [1, 3, 5, 10, 20, 50], "engine": {}, "reference": {}, "seconds_per_run": []}
runs = {"M": [], "F": [], "V": []}
NSEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for seed in range(NSEEDS):
    t0 = time.time()
    res = pipeline.run_experiment(split, HP, seed, tempfile.mkdtemp())
    out["seconds_per_run"].append(round(time.time() - t0, 1))
    for tag in runs:
        runs[tag].append(res[tag][0].tolist())
    print("seed", seed, {t: runs[t][-1][3] for t in runs}, flush=True)
for tag in runs:
    a = np.asarray(runs[tag])
    out["engine"][tag] = {"recall_runs": runs[tag], "recall@10_mean": float(a[:, 3].mean()), "recall@10_std": float(a[:, 3].std(ddof=1))}
    r = ref[tag + "_recall"]
    out["reference"][tag] = {"recall@10_runs": r[:, 3].tolist(), "recall@10_mean": float(r[:, 3].mean()), "recall@10_std": float(r[:, 3].std(ddof=1))}
out["difference"] = {}
for tag in runs:
    e, r = np.asarray(runs[tag])[:, 3], ref[tag + "_recall"][:, 3]
    se = float(np.sqrt(e.var(ddof=1) / len(e) + r.var(ddof=1) / len(r)))
    out["difference"][tag] = {"engine_minus_reference": float(e.mean() - r.mean()), "standard_error": se, "z": float((e.mean() - r.mean()) / se),
                              "engine_runs": len(e), "reference_runs": len(r)}
# the diffusion output against the SAME pipeline's VAE baseline, paired by seed
for side, src in (("engine", {t: np.asarray(runs[t])[:, 3] for t in runs}), ("reference", {t: ref[t + "_recall"][:, 3] for t in runs})):
    for tag in ("M", "F"):
        d = src[tag] - src["V"]
        out[side][tag]["minus_vae_mean"] = float(d.mean()); out[side][tag]["minus_vae_se"] = float(d.std(ddof=1) / np.sqrt(len(d)))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/e2e_report.json", "w"), indent=1)
print(json.dumps(out["difference"]))
print(json.dumps({t: (out["engine"][t]["recall@10_mean"], out["reference"][t]["recall@10_mean"]) for t in runs}))
