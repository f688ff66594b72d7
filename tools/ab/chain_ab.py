import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H = 340, 340, 78, 1
rows = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [679, 1358, 2715, 4096, 5429, 8192]
for n in rows:
    line = []
    for chains in (1, 2, 3, 4):
        for fused in (1, 2):
            e = Engine(L, W, T, H, max_rows=n).debug_set(chains=chains, fused_reverse=fused, sample_persist=0)
            e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
            for rep in range(3):
                e.sample(n, seed=2, call_id=rep)
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for rep in range(8):
                e.sample(n, seed=2, call_id=10 + rep)
            t1.record(); torch.cuda.synchronize()
            line.append(f"{chains}ch{'f' if fused == 2 else ''} {1e3 * t0.elapsed_time(t1) / 8 / T:.2f}")
            e.close()
    print(f"n = {n}: " + " | ".join(line), flush=True)
