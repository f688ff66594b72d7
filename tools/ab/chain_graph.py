"""Sampler row chains: is the host's launch rate what bounds three and four chains?  Per n and chain count: us per reverse step of
whole calls from stream launches (GPU time by events, and the host's enqueue time alone), and the same call replayed from a captured
graph (chains fork and join inside the capture)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H = 340, 340, 78, 1
rows = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2715, 5429]
for n in rows:
    for chains in (1, 2, 3, 4, 6, 8):
        e = Engine(L, W, T, H, max_rows=n).debug_set(chains=min(chains, 4), fused_reverse=1, sample_persist=0)
        if chains > 4:
            e.close(); continue
        e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
        for rep in range(3):
            e.sample(n, seed=2, call_id=rep)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(); h0 = time.perf_counter()
        for rep in range(8):
            e.sample(n, seed=2, call_id=10 + rep)
        h1 = time.perf_counter()
        t1.record(); torch.cuda.synchronize()
        gpu = 1e3 * t0.elapsed_time(t1) / 8 / T
        host = 1e6 * (h1 - h0) / 8 / T
        # graph
        gus = float("nan")
        try:
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=side):
                    out = e.sample(n, seed=2, call_id=99)
                torch.cuda.synchronize()
                for rep in range(3): g.replay()
                torch.cuda.synchronize()
                t0.record(side)
                for rep in range(8): g.replay()
                t1.record(side); torch.cuda.synchronize()
                gus = 1e3 * t0.elapsed_time(t1) / 8 / T
        except Exception as ex:  # noqa
            print("graph capture failed:", repr(ex)[:300], flush=True)
        print(f"n = {n} chains {chains}: stream {gpu:.2f} us/step (host enqueue {host:.2f} us/step) | graph replay {gus:.2f} us/step", flush=True)
        e.close()
