import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H = 830, 830, 83, 2
for n in (843, 1686):
    line = []
    for chains in (1, 2, 3):
        for mr in (False, True):
            e = Engine(L, W, T, H, max_rows=n).debug_set(chains=chains)
            e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
            for rep in range(3): e.sample(n, seed=2, call_id=rep, multires=mr)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for rep in range(8): e.sample(n, seed=2, call_id=10 + rep, multires=mr)
            torch.cuda.synchronize()
            line.append(f"{chains}ch{'m' if mr else ''} {(time.perf_counter() - t0) / 8 / T * 1e6:.2f}")
            e.close()
    print(f"ML-100k n = {n}: " + " | ".join(line), flush=True)
