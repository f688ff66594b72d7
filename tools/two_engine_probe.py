#!/usr/bin/env python3
"""Upper bound for splitting a train step into concurrent half-batch chains: two ENGINES with B/2 users each, stepping at
once on two streams from two host threads (no coupling at all), against one engine with B users.  If a pair of half steps
does not beat one full step, no in-engine two-chain design will.  usage: two_engine_probe.py [B steps]"""
import sys, threading, time
import torch
from sdrm_amd import synth
from sdrm_amd.engine import Engine

B, steps = (int(v) for v in (sys.argv[1:3] + [8192, 300][len(sys.argv) - 1:]))
L = W = 340; T = 78; H = 1


def make(b):
    e = Engine(L, W, T, H, max_rows=3 * b)
    e.set_params(synth.flatten_params(synth.init_params(L, W, T, H, seed=1), H))
    x0 = torch.from_numpy(synth.synth_latents(b, L, seed=0)).cuda()
    return e, x0


def loop(e, x0, stream, out, i):
    with torch.cuda.stream(stream):
        for k in range(20):
            e.train_step(x0, 1e-3, seed=1, step=k)
        stream.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            e.train_step(x0, 1e-3, seed=1, step=20 + k)
        stream.synchronize()
        out[i] = (time.perf_counter() - t0) * 1e6 / steps


def together(engs):
    out = [0.0] * len(engs)
    th = [threading.Thread(target=loop, args=(e, x, torch.cuda.Stream(), out, i)) for i, (e, x) in enumerate(engs)]
    for t in th: t.start()
    for t in th: t.join()
    return out


full = make(B)
halves = [make(B // 2) for _ in range(2)]
thirds = [make(B // 3 // 64 * 64) for _ in range(3)]
for rnd in range(2):
    a = together([full])
    h = together([halves[0]])
    b = together(halves)
    c = together(thirds)
print(f"B = {B}, {steps} train steps per engine (ML-1M net)")
print(f"  one engine, B users         : {a[0]:7.1f} us per step")
print(f"  one engine, B/2 users       : {h[0]:7.1f} us per step (x2 = {2 * h[0]:.1f})")
print(f"  two engines at once, B/2    : {max(b):7.1f} us per pair of steps")
print(f"  three engines at once, ~B/3 : {max(c):7.1f} us per triple of steps ({3 * (B // 3 // 64 * 64)} users)")
