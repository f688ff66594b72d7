"""Per-layer backward with every dgrad launched together with the weight gradient that waits for the same incoming gradient
(SDRM_PAIR_BWD=1 by size / 2 always / 0 never): us per train step, one process per setting."""
import os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, os.environ["REPO"])
from sdrm_amd import synth
from sdrm_amd.engine import Engine
out = []
for name, L, W, T, H, B in (("ML-1M B=160", 340, 340, 78, 1, 160), ("B=512", 340, 340, 78, 1, 512), ("B=1024", 340, 340, 78, 1, 1024),
                            ("B=1280", 340, 340, 78, 1, 1280), ("ML-100k B=550", 830, 830, 83, 2, 550)):
    e = Engine(L, W, T, H, max_rows=B)
    e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
    x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()
    for k in range(8): e.train_step(x0, 1e-5, seed=1, step=k)
    ts = []
    for w in range(5):
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        l0 = e.launch_count(); t0.record()
        for k in range(60): e.train_step(x0, 1e-5, seed=1, step=10 + k)
        t1.record(); torch.cuda.synchronize()
        ts.append(1e3 * t0.elapsed_time(t1) / 60)
    out.append(f"{name}: {sorted(ts)[2]:.1f} us ({(e.launch_count() - l0) / 60:.0f} launches)")
    e.close()
print(" | ".join(out), flush=True)
'''
repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for v in ("0", "1", "2", "0", "1"):
    env = dict(os.environ, SDRM_PAIR_BWD=v, REPO=repo)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(f"SDRM_PAIR_BWD={v}: {r.stdout.strip()} {r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)
