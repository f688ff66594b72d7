"""Is bench.py's walk bound by the host at shard sizes?  One rank of eight (1024 users, 679 sampled rows): wall time of the enqueue loop
alone against the loop + device sync, per job cycle, with the sampling call on the caller's stream (SDRM_DETACH=0) or detached (=1)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H = 340, 340, 78, 1
B, n = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (1024, 679)))
e = Engine(L, W, T, H, max_rows=max(B, n))
e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()
n_train, cycle = 15, 93
def is_train(k): j = k % cycle; return ((j + 1) * n_train) // cycle > (j * n_train) // cycle
def walk(cycles):
    k = 0; sampling = False; tc = 0
    for _ in range(cycles * cycle):
        if is_train(k):
            e.train_step(x0, 1e-5, seed=1, step=tc); tc += 1
        else:
            if not sampling:
                e.sample_begin(n, seed=2, call_id=k); sampling = True
            if e.sample_steps(1) == 0:
                e.sample_end(); sampling = False
        k += 1
walk(2); torch.cuda.synchronize()
for rep in range(3):
    l0 = e.launch_count()
    t0 = time.perf_counter(); walk(4); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B} n={n} chains={e.sampler_chains} launches per cycle {(e.launch_count() - l0) / 4:.0f}: enqueue {1e3 * (t1 - t0) / 4:.3f} ms per cycle, with sync {1e3 * (t2 - t0) / 4:.3f} ms per cycle -> {93 * 4 / (t2 - t0):.0f} steps/s", flush=True)
