"""Two sampler row chains: which tile and which form of the reverse update per chain?  us per reverse step, whole calls."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H = 340, 340, 78, 1
rows = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2715, 5429]
for n in rows:
    for chains in (1, 2):
        line = []
        for tile in (0, 4):
            for fused in (0, 2):
                e = Engine(L, W, T, H, max_rows=n).debug_set(chains=chains, fused_reverse=fused, sample_persist=0, tile=tile)
                e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
                for rep in range(3):
                    e.sample(n, seed=2, call_id=rep)
                torch.cuda.synchronize()
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                for rep in range(8):
                    e.sample(n, seed=2, call_id=10 + rep)
                t1.record(); torch.cuda.synchronize()
                line.append(f"tile {tile} fused {fused}: {1e3 * t0.elapsed_time(t1) / 8 / T:.2f}")
                e.close()
        print(f"n = {n} chains {chains}: " + " | ".join(line), flush=True)
