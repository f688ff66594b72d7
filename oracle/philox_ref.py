"""numpy restatement of the engine's counter-based RNG (sdrm_amd/csrc/philox.h) — TEST INFRASTRUCTURE.

Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
reference constants), counter = (global row, column quad (pair for the forward-only mask), purpose | sub-step << 8, step or call id),
key = the 64-bit seed.  This is NOT a restatement of anything in the reference (which uses torch's
default generators); it exists so that PHILOX-mode runs of the HIP engine can be
replayed through the oracle with explicit randoms.  Integer outputs (t, keep masks, Tj) must match the
device bit for bit; normals match to ~1e-6 (the device uses hardware log2/sin/cos)."""
from __future__ import annotations

import numpy as np

PURPOSE_TRAIN_ELEM, PURPOSE_TRAIN_T, PURPOSE_SAMPLE_XT, PURPOSE_SAMPLE_STEP, PURPOSE_SAMPLE_TJ, PURPOSE_FORWARD = 1, 2, 3, 4, 5, 6
M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, seed):
    c0, c1, c2, c3 = (np.asarray(np.broadcast_arrays(c0, c1, c2, c3)[i], dtype=np.uint64) & MASK for i in range(4))
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def box_muller(a, b):
    u = ((a >> np.uint64(8)) + np.uint64(1)).astype(np.float64) * 2.0 ** -24
    v = (b >> np.uint64(8)).astype(np.float64) * 2.0 ** -24
    r = np.sqrt(-2.0 * np.log(u))
    return (r * np.cos(2 * np.pi * v)).astype(np.float32), (r * np.sin(2 * np.pi * v)).astype(np.float32)


def bounded(w, n):
    return ((w * np.uint64(n)) >> np.uint64(32)).astype(np.int64)


def _grid(row0, n, L):
    rows = (np.arange(n, dtype=np.uint64) + np.uint64(row0))[:, None]
    pairs = np.arange((L + 1) // 2, dtype=np.uint64)[None, :]
    return rows, pairs


def _interleave(a0, a1, L):
    out = np.empty((a0.shape[0], 2 * a0.shape[1]), dtype=a0.dtype)
    out[:, 0::2], out[:, 1::2] = a0, a1
    return out[:, :L]


def train_randoms(seed, step, row0, B, L, T, nd):
    """eps [B,L] f32 (already * nd), t [B] i64, keep [3,B,L] u8 — what k_prep_train draws: one Philox call per group of
    four columns (counter = column quad): (x, y) and (z, w) are two Box-Muller pairs, the low byte of word j holds the
    three keep bits (passes P, S, Q) of column j."""
    rows = (np.arange(B, dtype=np.uint64) + np.uint64(row0))[:, None]
    quads = np.arange((L + 3) // 4, dtype=np.uint64)[None, :]
    words = philox4x32_10(rows, quads, PURPOSE_TRAIN_ELEM, step, seed)
    n0, n1 = box_muller(words[0], words[1])
    n2, n3 = box_muller(words[2], words[3])
    eps = np.empty((B, 4 * quads.shape[1]), dtype=np.float32)
    eps[:, 0::4], eps[:, 1::4], eps[:, 2::4], eps[:, 3::4] = n0, n1, n2, n3
    eps = eps[:, :L] * np.float32(nd)
    keep = np.empty((3, B, 4 * quads.shape[1]), dtype=np.uint8)
    for k in range(3):
        for j in range(4):
            keep[k, :, j::4] = ((words[j] >> np.uint64(k)) & np.uint64(1)).astype(np.uint8)
    keep = np.ascontiguousarray(keep[:, :, :L])
    tx, _, _, _ = philox4x32_10(rows[:, 0], 0, PURPOSE_TRAIN_T, step, seed)
    t = 1 + bounded(tx, T)
    return eps.astype(np.float32), t, keep


def forward_keep(seed, step, row0, n, L):
    rows, pairs = _grid(row0, n, L)
    _, _, z, _ = philox4x32_10(rows, pairs, PURPOSE_FORWARD, step, seed)
    return _interleave((z & np.uint64(1)).astype(np.uint8), ((z >> np.uint64(8)) & np.uint64(1)).astype(np.uint8), L)


def _quads(row0, n, L):
    rows = (np.arange(n, dtype=np.uint64) + np.uint64(row0))[:, None]
    quads = np.arange((L + 3) // 4, dtype=np.uint64)[None, :]
    return rows, quads


def _quad_normals(words, L):
    n0, n1 = box_muller(words[0], words[1])
    n2, n3 = box_muller(words[2], words[3])
    out = np.empty((n0.shape[0], 4 * n0.shape[1]), dtype=np.float32)
    out[:, 0::4], out[:, 1::4], out[:, 2::4], out[:, 3::4] = n0, n1, n2, n3
    return out[:, :L]


def sample_randoms(seed, call_id, row0, n, L, T, nd, multires):
    """xT [n,L], z [T+1,n,L] (already * nd, z[0]=z[1]=0), keep [T+1,n,L], Tj [n] (or None) — what
    k_sample_init / k_reverse_update / the fused reverse epilogue / k_skinny_sample draw: one Philox call per group of
    four columns (counter = column quad): (x, y) and (z, w) are two Box-Muller pairs (bits 8..31 of each word), bit 0
    of word j is the keep bit of column j.  keep[i] rides on the call of sub-step i+1."""
    rows, quads = _quads(row0, n, L)
    xT = _quad_normals(philox4x32_10(rows, quads, PURPOSE_SAMPLE_XT, call_id, seed), L)
    z = np.zeros((T + 1, n, L), np.float32)
    keep = np.zeros((T + 1, n, L), np.uint8)
    for sub in range(2, T + 2):
        words = philox4x32_10(rows, quads, PURPOSE_SAMPLE_STEP | (sub << 8), call_id, seed)
        if sub <= T:
            z[sub] = _quad_normals(words, L) * np.float32(nd)
        kq = np.empty((n, 4 * quads.shape[1]), dtype=np.uint8)
        for j in range(4):
            kq[:, j::4] = (words[j] & np.uint64(1)).astype(np.uint8)
        keep[sub - 1] = kq[:, :L]
    Tj = None
    if multires:
        tx, _, _, _ = philox4x32_10(rows[:, 0], 0, PURPOSE_SAMPLE_TJ, call_id, seed)
        Tj = 1 + bounded(tx, max(T - 1, 1))
    return xT, z, keep, Tj
