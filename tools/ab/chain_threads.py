"""What would row chains give if the host's launch rate did not bound them?  k engines, each sampling n/k rows with one chain on a
stream of its own, driven by k host threads (the ctypes call releases the GIL; a call is 78 steps): wall-clock us per reverse step of the
n rows together."""
import os, sys, time, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdrm_amd import synth
from sdrm_amd.engine import Engine
L, W, T, H = 340, 340, 78, 1
rows = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2715, 5429]
REPS = 8
for n in rows:
    for k in (1, 2, 3, 4, 6, 8):
        nk = (n + k - 1) // k
        nk = (nk + 63) // 64 * 64 if k > 1 else n
        sizes = [min(nk, n - i * nk) for i in range(k)]
        sizes = [s for s in sizes if s > 0]
        engs = []
        flat = synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H)
        for s in sizes:
            e = Engine(L, W, T, H, max_rows=s).debug_set(chains=1, fused_reverse=1, sample_persist=0)
            e.set_params(flat)
            engs.append(e)
        streams = [torch.cuda.Stream() for _ in sizes]
        bar = threading.Barrier(len(sizes) + 1)

        def work(i):
            with torch.cuda.stream(streams[i]):
                for rep in range(3):
                    engs[i].sample(sizes[i], seed=2, call_id=rep)
                streams[i].synchronize()
                bar.wait()
                for rep in range(REPS):
                    engs[i].sample(sizes[i], seed=2, call_id=10 + rep)
                streams[i].synchronize()
                bar.wait()

        th = [threading.Thread(target=work, args=(i,)) for i in range(len(sizes))]
        for t in th: t.start()
        bar.wait(); t0 = time.perf_counter()
        bar.wait(); t1 = time.perf_counter()
        for t in th: t.join()
        print(f"n = {n}: {len(sizes)} threads x {sizes[0]} rows: {1e6 * (t1 - t0) / REPS / T:.2f} us/step", flush=True)
        for e in engs: e.close()
