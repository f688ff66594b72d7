#!/usr/bin/env python3
"""Diagnostic (stamped build): work-group timeline of the engine's own batched weight-gradient launch inside a train step.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSDRM_STAMPS -DSDRM_SOURCE_HASH=... -o tools/libsdrm_stamps.so ...
    python tools/wgrad_stamps.py            (env B, L, T, H; SDRM_WGRAD_BLOCKS to try other split plans)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrm_amd import _build, _lib, synth  # noqa: E402

_build.LIB_PATH = os.path.abspath(os.environ.get("STAMPLIB", "tools/libsdrm_stamps.so"))
_build.is_stale = lambda: False
_build.source_hash = lambda: _build.binary_hash() or "unhashed"
from sdrm_amd.engine import Engine  # noqa: E402

B, L, T, H = (int(os.environ.get(k, d)) for k, d in (("B", 8192), ("L", 340), ("T", 78), ("H", 1)))
lib = _lib.load()
e = Engine(L, L, T, H, B)
e.set_params(synth.flatten_params(synth.init_params(L, L, T, H, seed=1), H))
x0 = torch.from_numpy(synth.synth_latents(B, L, seed=0)).cuda()
for k in range(int(os.environ.get("WARM", 300))):
    e.train_step(x0, 1e-4, seed=1, step=k)
torch.cuda.synchronize()
mb = 16384
CLS = int(os.environ.get("CLS", "-1"))     # -1: the batched wgrad launch; 0..5: an NT launch class of the train step (0 fwd layer 0,
lib.sdrm_debug_stamp_class(CLS)            # 1 fwd hidden, 2 fwd out, 3 dgrad: the LAST launch of that class in the step is kept)
assert lib.sdrm_debug_wgrad_stamps_begin(mb) == 0
for k in range(3):
    e.train_step(x0, 1e-4, seed=1, step=k)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * mb))()
lib.sdrm_debug_wgrad_stamps_read.restype = C.c_int
nb = lib.sdrm_debug_wgrad_stamps_read(buf, mb)
a = np.frombuffer(buf, dtype=np.uint64).reshape(mb, 8)[:nb].astype(np.int64)
a = a[a[:, 3] > 0]       # work-groups beyond a problem's slice count return at once and leave no stamp
pro, loop, epi = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
life, real = a[:, 3] - a[:, 0], a[:, 5] - a[:, 4]
clk = np.median(life[real > 0] / real[real > 0]) * 0.1
t0 = a[:, 4].min()
wall_us = (a[:, 5].max() - t0) / 100.0
print(f"class {CLS}: B={B} L={L} T={T} H={H}: grid {nb}, {len(a)} working work-groups; launch wall {wall_us:.1f} us; clock {clk:.3f} GHz; lifetime med "
      f"{np.median(life):.0f} cyc = prologue {np.median(pro):.0f} + loop {np.median(loop):.0f} (p10 {np.percentile(loop, 10):.0f}, p90 {np.percentile(loop, 90):.0f}) "
      f"+ epilogue {np.median(epi):.0f}; mean residency {life.sum() / (wall_us * 1e3 * clk * 256):.2f} per CU")
hw, xcc = a[:, 6], a[:, 7] & 0xF
cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF))     # (xcc, se, sh, cu)
ids, counts = np.unique(cu, return_counts=True)
print(f"   distinct CUs used {len(ids)}; work-groups per CU: min {counts.min()} med {np.median(counts):.0f} max {counts.max()}; "
      f"histogram {dict(zip(*np.unique(counts, return_counts=True)))}")
if CLS >= 0:
    # NT launch: which CUs got the half-empty last column tiles (352 = 5.5 x 64)?  bid -> logical tile as xcd_remap does
    bid = np.nonzero(np.frombuffer(buf, dtype=np.uint64).reshape(mb, 8)[:nb, 3] > 0)[0]
    n = nb
    q, r = n >> 3, n & 7
    x, sl = bid & 7, bid >> 3
    logical = np.where(x < r, x * (q + 1), r * (q + 1) + (x - r) * q) + sl
    tiles_n = -(-(-(-L // 32) * 32) // 64)
    tn = logical % tiles_n
    if os.environ.get("ROTN"):            # the -DSDRM_ROTN experiment rotates the N-tile order by the row tile
        tn = (tn + (logical // tiles_n)) % tiles_n
    ragged = tn == tiles_n - 1
    per_cu_r = np.array([ragged[cu == c].sum() for c in ids])
    per_cu_end = np.array([a[cu == c, 5].max() for c in ids])
    print(f"   half-empty tiles per CU: histogram {dict(zip(*np.unique(per_cu_r, return_counts=True)))}; a CU's last work-group ends at "
          + ", ".join(f"{k} ragged: {np.median((per_cu_end[per_cu_r == k] - a[:, 4].min()) / 100.0):.1f} us" for k in np.unique(per_cu_r)))
st, en = (a[:, 4] - t0) / 100.0, (a[:, 5] - t0) / 100.0
print("   t[us]  alive  started  ended   (whole chip, 10-us bins)")
for lo in np.arange(0, wall_us + 10.0, 10.0):
    c = lo + 5.0
    print(f"   {c:5.1f}  {((st <= c) & (en > c)).sum():5d}  {((st >= lo) & (st < lo + 10)).sum():7d}  {((en >= lo) & (en < lo + 10)).sum():5d}")
e.close()
