import sys, time, torch
sys.path.insert(0, "/root/repo")
from sdrm_amd import synth
from sdrm_amd.engine import Engine
for name, (L, W, T, H, B) in {"ADM": (40, 40, 93, 5, 850), "ML-1M B=160": (340, 340, 78, 1, 160), "ML-100k": (830, 830, 83, 2, 550)}.items():
    e = Engine(L, W, T, H, B)
    e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
    x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()
    for _ in range(20): e.train_step(x0, 1e-5, seed=1, step=0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(300): e.train_step(x0, 1e-5, seed=1, step=k)
    t_enq = time.perf_counter() - t
    torch.cuda.synchronize(); t_all = time.perf_counter() - t
    print(f"{name}: host enqueue {t_enq/300*1e6:.1f} us/step, wall {t_all/300*1e6:.1f} us/step")
    e.close()
