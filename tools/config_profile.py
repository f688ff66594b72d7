#!/usr/bin/env python3
"""One BASELINE.json configuration's hot path for the profiler: R train steps, then one full-resolution and one
multi-resolution sampling call, PHILOX mode, synthetic latents (what bench.py's other_configs leg times).

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/config_profile.py adm
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/config_profile.py adm   (and WRITE_SIZE)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import synth  # noqa: E402
from sdrm_amd.engine import Engine  # noqa: E402

CONFIGS = {"ml100k": dict(L=830, W=830, T=83, H=2, B=550, n=843), "ml1m_b160": dict(L=340, W=340, T=78, H=1, B=160, n=5429),
           "adm": dict(L=40, W=40, T=93, H=5, B=850, n=9558), "ml1m": dict(L=340, W=340, T=78, H=1, B=8192, n=5429),
           "ml1m_shard8": dict(L=340, W=340, T=78, H=1, B=1024, n=679), "ml1m_b512": dict(L=340, W=340, T=78, H=1, B=512, n=5429),
           "ml1m_b2048": dict(L=340, W=340, T=78, H=1, B=2048, n=5429),
           "ml1m_n5120": dict(L=340, W=340, T=78, H=1, B=8192, n=5120),
           "ml1m_shard2": dict(L=340, W=340, T=78, H=1, B=4096, n=2715), "ml1m_shard4": dict(L=340, W=340, T=78, H=1, B=2048, n=1358)}
c = CONFIGS[sys.argv[1]]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 50
e = Engine(c["L"], c["W"], c["T"], c["H"], max_rows=max(c["B"], c["n"]))
e.set_params(synth.flatten_params(synth.init_params(c["L"], c["W"], c["T"], c["H"], seed=1), c["H"]))
x0 = torch.from_numpy(synth.synth_latents(c["B"], c["L"], seed=0)).cuda()
for k in range(5):
    e.train_step(x0, 1e-5, seed=1, step=k)
torch.cuda.synchronize()
l0 = e.launch_count()
for k in range(R):
    e.train_step(x0, 1e-5, seed=1, step=k)
torch.cuda.synchronize()
l1 = e.launch_count()
e.sample(c["n"], seed=2, call_id=1)
torch.cuda.synchronize()
l2 = e.launch_count()
e.sample(c["n"], seed=2, call_id=2, multires=True)
torch.cuda.synchronize()
l3 = e.launch_count()
print(f"{sys.argv[1]}: {R} train steps: {(l1 - l0) / R:.2f} launches per step; full-resolution sampling of {c['n']} rows: {l2 - l1} launches for "
      f"{c['T']} steps ({(l2 - l1) / c['T']:.2f} per step); multi-resolution: {l3 - l2} launches ({(l3 - l2) / c['T']:.2f} per step)")
e.close()
