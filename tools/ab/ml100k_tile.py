import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H, B = 830, 830, 83, 2, 550
init = synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H)
x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()
for name, kw in (("default (32x32 NT)", dict()), ("64x64 NT", dict(nt32_rows_train=0)), ("tile 0 forced", dict(tile=0)), ("tile 1 (64x64x32)", dict(tile=1)),
                 ("tile 2 (64x128)", dict(tile=2)), ("tile 3 (128x128)", dict(tile=3))):
    e = Engine(L, W, T, H, max_rows=B).debug_set(**kw)
    e.set_params(init)
    for k in range(6):
        e.train_step(x0, 1e-5, seed=1, step=k)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for k in range(60):
        e.train_step(x0, 1e-5, seed=1, step=8 + k)
    t1.record(); torch.cuda.synchronize()
    print(f"ML-100k train step, {name}: {1e3 * t0.elapsed_time(t1) / 60:.1f} us", flush=True)
    e.close()
