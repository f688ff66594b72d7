import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H = 340, 340, 78, 1
for n in (5429, 5120, 4608):
    for name, kw in (("64x64 (default)", dict(nt32_rows=4096)), ("32x32", dict(nt32_rows=8192)), ("32x32 fused", dict(nt32_rows=8192, fused_reverse=2)),
                     ("64x64 2 chains", dict(nt32_rows=4096, chains=2)), ("32x32 2 chains", dict(nt32_rows=8192, chains=2))):
        e = Engine(L, W, T, H, max_rows=n).debug_set(**kw)
        e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
        for rep in range(3):
            e.sample(n, seed=2, call_id=rep)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for rep in range(10):
            e.sample(n, seed=2, call_id=10 + rep)
        t1.record(); torch.cuda.synchronize()
        print(f"n = {n} {name}: {1e3 * t0.elapsed_time(t1) / 10 / T:.2f} us per reverse step", flush=True)
        e.close()
