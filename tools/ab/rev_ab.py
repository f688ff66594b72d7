import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H, n = 340, 340, 78, 1, int(sys.argv[1]) if len(sys.argv) > 1 else 5429
for fused in (0, 2):
    e = Engine(L, W, T, H, max_rows=n).debug_set(fused_reverse=fused)
    e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
    for rep in range(3):
        e.sample(n, seed=2, call_id=rep)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for rep in range(10):
        e.sample(n, seed=2, call_id=10 + rep)
    t1.record(); torch.cuda.synchronize()
    print(f"lib {os.environ.get('SDRM_LIB', 'in-tree')}: n = {n} fused_reverse = {fused}: {1e3 * t0.elapsed_time(t1) / 10 / T:.2f} us per reverse step", flush=True)
    e.close()
