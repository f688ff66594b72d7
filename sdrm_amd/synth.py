"""Deterministic synthetic inputs for the SDRM denoising path.

Everything here is numpy `RandomState` (the frozen legacy generator), so the
same seed gives the same bytes in the survey container, on the GPU box and in
any later round.  Used by `bench.py`, by the parity tests and by
`tests/golden/make_golden.py` (which feeds these inputs to the reference and
stores only the reference's *outputs* next to the seeds).

Parameter names and order follow the reference's `SDRM.named_parameters()`
(`/root/reference/train_SDRM.py:86-95`):

    emb_layer.weight [T,T]   emb_layer.bias [T]
    dnn.0.weight [W,L+T]     dnn.0.bias [W]      dnn.1.weight [1]
    (H>=1) dnn.2.weight [W,W]  dnn.2.bias [W]    dnn.3.weight [1]
    dnn.{2+2H}.weight [L,W]  dnn.{2+2H}.bias [L]
"""
from __future__ import annotations

import numpy as np

PRELU_INIT = 0.25  # torch.nn.PReLU default slope


def param_names(H: int):
    """Distinct parameter tensors in reference order (shared hidden layer listed once)."""
    names = ["emb_layer.weight", "emb_layer.bias", "dnn.0.weight", "dnn.0.bias", "dnn.1.weight"]
    if H >= 1:
        names += ["dnn.2.weight", "dnn.2.bias", "dnn.3.weight"]
    last = 2 + 2 * H
    names += [f"dnn.{last}.weight", f"dnn.{last}.bias"]
    return names


def param_shapes(L: int, W: int, T: int, H: int):
    shapes = {
        "emb_layer.weight": (T, T),
        "emb_layer.bias": (T,),
        "dnn.0.weight": (W, L + T),
        "dnn.0.bias": (W,),
        "dnn.1.weight": (1,),
    }
    if H >= 1:
        shapes["dnn.2.weight"] = (W, W)
        shapes["dnn.2.bias"] = (W,)
        shapes["dnn.3.weight"] = (1,)
    last = 2 + 2 * H
    shapes[f"dnn.{last}.weight"] = (L, W)
    shapes[f"dnn.{last}.bias"] = (L,)
    return shapes


def param_count(L: int, W: int, T: int, H: int) -> int:
    return int(sum(int(np.prod(s)) for s in param_shapes(L, W, T, H).values()))


def alias_keys(H: int):
    """state_dict() keys that alias the shared hidden layer (Q1): dnn.4.* == dnn.2.* ..."""
    out = {}
    for rep in range(1, H):
        out[f"dnn.{2 + 2 * rep}.weight"] = "dnn.2.weight"
        out[f"dnn.{2 + 2 * rep}.bias"] = "dnn.2.bias"
        out[f"dnn.{3 + 2 * rep}.weight"] = "dnn.3.weight"
    return out


def init_params(L: int, W: int, T: int, H: int, seed: int = 1):
    """Default-`nn.Linear`-shaped init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for
    weights and biases, PReLU slope 0.25 (distribution of `train_SDRM.py:89-95`;
    not torch's bit stream)."""
    rs = np.random.RandomState(seed)
    out = {}
    shapes = param_shapes(L, W, T, H)
    pending_bound = None
    for name in param_names(H):
        shp = shapes[name]
        if name.endswith(".weight") and len(shp) == 2:
            pending_bound = 1.0 / np.sqrt(shp[1])
            out[name] = rs.uniform(-pending_bound, pending_bound, size=shp).astype(np.float32)
        elif name.endswith(".bias"):
            out[name] = rs.uniform(-pending_bound, pending_bound, size=shp).astype(np.float32)
        else:  # PReLU slope
            out[name] = np.full(shp, PRELU_INIT, dtype=np.float32)
    return out


def flatten_params(params: dict, H: int) -> np.ndarray:
    return np.concatenate([np.asarray(params[n], dtype=np.float32).ravel() for n in param_names(H)])


def unflatten_params(flat: np.ndarray, L: int, W: int, T: int, H: int) -> dict:
    shapes = param_shapes(L, W, T, H)
    out, off = {}, 0
    for n in param_names(H):
        k = int(np.prod(shapes[n]))
        out[n] = np.asarray(flat[off:off + k], dtype=np.float32).reshape(shapes[n]).copy()
        off += k
    assert off == flat.size
    return out


def synth_latents(n_rows: int, L: int, seed: int = 0) -> np.ndarray:
    """x0 ~ N(0,1): trained-VAE latents are ~N(0,1) under the KL prior."""
    return np.random.RandomState(seed).standard_normal((n_rows, L)).astype(np.float32)


def synth_scores(n_users: int, n_items: int, seed: int = 0, kind: str = "normal") -> np.ndarray:
    """Decoder-output-shaped scores [n_users, n_items] for the equal-sparsity / ranking rows:
    "normal": N(-2, 1.5) logits; "ties": the same rounded to 1/8 (many exact duplicates, also at the threshold);
    "narrow": all values inside one binade (stresses the first radix digit)."""
    rs = np.random.RandomState(seed)
    x = (rs.standard_normal((n_users, n_items)) * 1.5 - 2.0).astype(np.float32)
    if kind == "ties":
        x = (np.round(x * 8.0) / 8.0).astype(np.float32)
    elif kind == "narrow":
        x = (1.0 + 0.5 * rs.random_sample((n_users, n_items))).astype(np.float32)
    elif kind != "normal":
        raise ValueError(kind)
    return x


def synth_interactions(n_users: int, n_items: int, seed: int = 0, p_train: float = 0.05, p_held: float = 0.02):
    """(train, heldout) disjoint binary interaction matrices as scipy CSR [n_users, n_items]: every third user of the
    first nine has nothing held out / very many held out / nothing seen, to exercise the metric edge cases."""
    from scipy.sparse import csr_matrix
    rs = np.random.RandomState(seed)
    u = rs.random_sample((n_users, n_items))
    train = u < p_train
    held = (u >= p_train) & (u < p_train + p_held)
    for r in range(min(n_users, 9)):
        if r % 3 == 0:
            held[r, :] = False
        elif r % 3 == 1:
            held[r, :] = (u[r] >= p_train) & (u[r] < p_train + 0.5)
        else:
            train[r, :] = False
    return csr_matrix(train.astype(np.float64)), csr_matrix(held.astype(np.float64))


def synth_vae_decoder(latent: int, hidden: int, n_items: int, seed: int = 0):
    """Tensors of a `Linear(latent, hidden) -> Tanh -> Linear(hidden, n_items)` decoder (train_SDRM.py:212-214), drawn like
    the reference initialises them (:224-228: xavier-uniform weights, N(0, 0.001) biases) but from numpy's RandomState:
    (w1 [hidden, latent], b1 [hidden], w2 [n_items, hidden], b2 [n_items]), float32."""
    rs = np.random.RandomState(seed)
    a1, a2 = np.sqrt(6.0 / (latent + hidden)), np.sqrt(6.0 / (hidden + n_items))
    w1 = rs.uniform(-a1, a1, size=(hidden, latent)).astype(np.float32)
    b1 = (rs.standard_normal(hidden) * 0.001).astype(np.float32)
    w2 = rs.uniform(-a2, a2, size=(n_items, hidden)).astype(np.float32)
    b2 = (rs.standard_normal(n_items) * 0.001).astype(np.float32)
    return w1, b1, w2, b2


def synth_train_randoms(B: int, L: int, T: int, nd: float, seed: int):
    """One train step's explicit randoms: eps=nd*N(0,1) [B,L], t~U{1..T} [B] i64,
    three Bernoulli(0.5) keep-masks [3,B,L] u8 (pass order P,S,Q)."""
    rs = np.random.RandomState(seed)
    eps = (rs.standard_normal((B, L)) * nd).astype(np.float32)
    t = rs.randint(1, T + 1, size=(B,)).astype(np.int64)
    masks = (rs.random_sample((3, B, L)) < 0.5).astype(np.uint8)
    return eps, t, masks


def synth_sample_randoms(n: int, L: int, T: int, nd: float, seed: int, multires: bool = False):
    """Explicit randoms for one reverse-sampling call.

    x_T [n,L]; z [T+1,n,L] (z[i] used at step i, z[1] ignored = 0, already
    scaled by nd); masks [T+1,n,L] u8; T_j [n] i64 in [1,T-1] for multi-res
    (`np.random.randint(1, n_timesteps)`, `train_SDRM.py:42`) else all T.
    """
    rs = np.random.RandomState(seed)
    xT = rs.standard_normal((n, L)).astype(np.float32)
    z = (rs.standard_normal((T + 1, n, L)) * nd).astype(np.float32)
    z[0] = 0
    z[1] = 0
    masks = (rs.random_sample((T + 1, n, L)) < 0.5).astype(np.uint8)
    if multires:
        Tj = rs.randint(1, max(T, 2), size=(n,)).astype(np.int64)
    else:
        Tj = np.full((n,), T, dtype=np.int64)
    return xT, z, masks, Tj


def stats(a: np.ndarray, n_samples: int = 16):
    """Size-independent checksum triple used by the full-size fixtures: sum and
    L2 norm in float64, plus `n_samples` evenly strided elements."""
    flat = np.asarray(a, dtype=np.float32).ravel()
    idx = np.linspace(0, flat.size - 1, n_samples).astype(np.int64)
    return (np.float64(flat.astype(np.float64).sum()),
            np.float64(np.sqrt((flat.astype(np.float64) ** 2).sum())),
            flat[idx].copy())
