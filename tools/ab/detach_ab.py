"""The one chain of a small sampling call on an auxiliary stream (SDRM_DETACH=1, default) against the caller's stream (=0): us per reverse
step of whole calls and of one step per call, no train steps in between."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
cases = [("ML-1M n=679", 340, 340, 78, 1, 679), ("ML-1M n=1358", 340, 340, 78, 1, 1358), ("ML-100k n=843", 830, 830, 83, 2, 843)]
for name, L, W, T, H, n in cases:
    e = Engine(L, W, T, H, max_rows=n)
    e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
    for rep in range(3): e.sample(n, seed=2, call_id=rep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(10): e.sample(n, seed=2, call_id=10 + rep)
    torch.cuda.synchronize(); whole = (time.perf_counter() - t0) / 10 / T * 1e6
    e.sample_begin(n, seed=2, call_id=99)
    per = []
    for w in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(15): e.sample_steps(1)
        torch.cuda.synchronize(); per.append((time.perf_counter() - t0) / 15 * 1e6)
    print(f"{name} SDRM_DETACH={os.environ.get('SDRM_DETACH', '1')}: whole calls {whole:.2f} us/step, 15-step windows {sorted(per)[1]:.2f} us/step", flush=True)
    e.close()
