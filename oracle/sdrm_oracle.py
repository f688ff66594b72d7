"""CPU oracle for the SDRM denoising hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

A from-scratch restatement (PyTorch CPU fp32 tensor ops, explicit forward and
explicit hand-derived backward — no autograd, no nn.Module) of the one hot path
of the reference, `/root/reference/train_SDRM.py`:

    schedule            train_SDRM.py:275-276,300-303   
    q_sample            train_SDRM.py:202-203,326-328   (A.2, Q3)
    time embedding      train_SDRM.py:105-112           (A.3, Q5)
    SDRM.forward        train_SDRM.py:86-103            (A.4, Q1 shared hidden layer, Q2 dropout always on)
    score-matching loss train_SDRM.py:191-199           (A.5, Q6)
    backward            train_SDRM.py:336 (autograd)    (A.5 closed-form seeds; Q7 latent dgrad skipped)
    Adam + lr decay     train_SDRM.py:309,316,337       (A.6, Q8 coupled L2)
    reverse step        train_SDRM.py:20-25             (A.7, Q9)
    sample_ddpm         train_SDRM.py:27-63             (A.7, Q11 multi-res as an active-row mask)

Pinned by: `tests/golden/*.npz`, produced by importing the reference itself in
the build container (`tests/golden/make_golden.py`); `tests/test_oracle_golden.py`
checks every function here against those vectors.  The reference has no tests
or golden vectors of its own, so these are the only pin.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.  The product (`sdrm_amd/`) never does; it fails loudly if
the HIP library is missing.
"""
from __future__ import annotations

import math

import numpy as np
import torch

MU = 0.1
ADAM_B1, ADAM_B2, ADAM_EPS, ADAM_WD = 0.9, 0.999, 1e-8, 1e-4
BETA1, BETA2 = 1e-4, 0.02


def _t(a, dtype=torch.float32):
    if isinstance(a, torch.Tensor):
        return a.to(dtype)
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def schedule(T: int, beta1: float = BETA1, beta2: float = BETA2):
    """beta, alpha, alpha-bar, each [T+1] fp32 (train_SDRM.py:300-303)."""
    beta = (beta2 - beta1) * torch.linspace(0, 1, T + 1, dtype=torch.float32) + beta1
    alpha = 1 - beta
    alphabar = torch.cumsum(alpha.log(), dim=0).exp()
    alphabar[0] = 1
    return beta, alpha, alphabar


def timestep_table(T: int) -> torch.Tensor:
    """Row t = sinusoidal embedding of timestep t, width T (train_SDRM.py:105-112)."""
    half = T // 2
    freqs = torch.exp(-math.log(10_000) * torch.arange(0, half, dtype=torch.float32) / half)
    args = torch.arange(0, T + 1, dtype=torch.float32)[:, None] * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if T % 2:
        emb = torch.cat([emb, torch.zeros(T + 1, 1)], dim=-1)
    return emb


def q_sample(x0, t, eps, alphabar):
    """sqrt(abar[t]) * x0 + (1 - abar[t]) * eps  — note (1-abar), not its sqrt (Q3)."""
    return alphabar.sqrt()[t, None] * x0 + (1 - alphabar[t, None]) * eps


def reverse_update(x, eps_hat, z, i, beta, alpha, alphabar):
    """One DDPM reverse update (train_SDRM.py:20-25). `i` int or int64 tensor [n]."""
    if isinstance(i, torch.Tensor):
        b, a, ab = beta[i, None], alpha[i, None], alphabar[i, None]
    else:
        b, a, ab = beta[i], alpha[i], alphabar[i]
    mean = (x - eps_hat * ((1 - a) / (1 - ab).sqrt())) / a.sqrt()
    return mean + b.sqrt() * z


def prelu(v, a):
    return torch.where(v >= 0, v, a * v)


class Oracle:
    """Parameters live in a dict keyed by the reference's parameter names."""

    def __init__(self, L, W, T, H, params, beta1=BETA1, beta2=BETA2):
        self.L, self.W, self.T, self.H = L, W, T, H
        self.last = 2 + 2 * H
        self.p = {k: _t(v).clone() for k, v in params.items()}
        self.beta, self.alpha, self.alphabar = schedule(T, beta1, beta2)
        self.temb = timestep_table(T)
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.adam_t = 0

    # ------------------------------------------------------------------ forward
    def forward(self, x, t, keep, cache=None):
        """eps_hat = f(x, t) with the given keep-mask (1 = survive, scaled by 2)."""
        p = self.p
        e = self.temb[t] @ p["emb_layer.weight"].T + p["emb_layer.bias"]
        xd = x * keep * 2.0
        u = torch.cat([xd, e], dim=-1)
        pre = [u @ p["dnn.0.weight"].T + p["dnn.0.bias"]]
        h = [prelu(pre[0], p["dnn.1.weight"])]
        for _ in range(self.H):
            pre.append(h[-1] @ p["dnn.2.weight"].T + p["dnn.2.bias"])
            h.append(prelu(pre[-1], p["dnn.3.weight"]))
        y = torch.tanh(h[-1] @ p[f"dnn.{self.last}.weight"].T + p[f"dnn.{self.last}.bias"])
        if cache is not None:
            cache.update(t=t, u=u, pre=pre, h=h, y=y)
        return y

    def backward(self, cache, gy, grads, neg_override=None):
        """Accumulate parameter gradients of one pass given dL/dy (no latent dgrad, Q7).

        `neg_override` (list of H+1 bool tensors, True = take the slope branch) replaces the sign
        test on the pre-activations.  PReLU'(v) jumps at v = 0, so two fp32 implementations whose
        pre-activations differ in the last bit can legitimately pick different branches for an element
        that is zero within rounding; parity tests pass the HIP engine's own branch choice here after
        checking that it differs from the oracle's only at such elements ("kink flips")."""
        p, L = self.p, self.L
        y, pre, h, u, t = cache["y"], cache["pre"], cache["h"], cache["u"], cache["t"]
        negs = neg_override if neg_override is not None else [q <= 0 for q in pre]
        wl, bl = f"dnn.{self.last}.weight", f"dnn.{self.last}.bias"
        d = gy * (1 - y * y)
        grads[wl] += d.T @ h[-1]
        grads[bl] += d.sum(0)
        dh = d @ p[wl]
        for k in range(self.H, 0, -1):
            neg = negs[k]  # torch prelu backward: input > 0 ? g : a*g
            grads["dnn.3.weight"] += (dh * torch.where(neg, pre[k], torch.zeros(()))).sum().reshape(1)
            d = dh * torch.where(neg, p["dnn.3.weight"], torch.ones(()))
            grads["dnn.2.weight"] += d.T @ h[k - 1]
            grads["dnn.2.bias"] += d.sum(0)
            dh = d @ p["dnn.2.weight"]
        neg = negs[0]
        grads["dnn.1.weight"] += (dh * torch.where(neg, pre[0], torch.zeros(()))).sum().reshape(1)
        d = dh * torch.where(neg, p["dnn.1.weight"], torch.ones(()))
        grads["dnn.0.weight"] += d.T @ u
        grads["dnn.0.bias"] += d.sum(0)
        de = d @ p["dnn.0.weight"][:, L:]
        grads["emb_layer.weight"] += de.T @ self.temb[t]
        grads["emb_layer.bias"] += de.sum(0)

    # ------------------------------------------------------------------ loss
    def loss_and_grads(self, x0, eps, t, keeps, neg_override=None, caches=None):
        """Three forwards + score-matching loss + all parameter gradients.
        Returns (loss, grads, (P,S,Q), x_pert).  `eps` is already scaled by nd.
        neg_override: optional [3][H+1] branch masks (see `backward`); caches: optional list that
        receives the three per-pass activation caches."""
        x0, eps = _t(x0), _t(eps)
        t = _t(t, torch.int64)
        keeps = [_t(k) for k in keeps]
        xp = q_sample(x0, t, eps, self.alphabar)
        cP, cS, cQ = {}, {}, {}
        P = self.forward(xp, t, keeps[0], cP)
        S = self.forward(x0, t, keeps[1], cS)
        Q = self.forward(x0 + MU * eps, t, keeps[2], cQ)
        mu2 = MU ** 2
        R = P - x0
        D = (Q - S) / mu2 - R
        N = R.numel()
        A = (D * D).mean()
        C = ((R - S) ** 2).mean()
        Rbar = R.mean()
        V = ((R - Rbar) ** 2).sum() / (N - 1)
        den = 1e-8 + V
        loss = 0.5 * (A + C) / den
        k = 0.5 / den
        gD = k * 2.0 * D / N
        gC = k * 2.0 * (R - S) / N
        gV = -(0.5 * (A + C) / (den * den)) * 2.0 * (R - Rbar) / (N - 1)
        gP = -gD + gC + gV
        gQ = gD / mu2
        gS = -gD / mu2 - gC
        grads = {n: torch.zeros_like(v) for n, v in self.p.items()}
        ov = neg_override if neg_override is not None else [None, None, None]
        self.backward(cP, gP, grads, ov[0])
        self.backward(cS, gS, grads, ov[1])
        self.backward(cQ, gQ, grads, ov[2])
        if caches is not None:
            caches.extend([cP, cS, cQ])
        return loss, grads, (P, S, Q), xp

    # ------------------------------------------------------------------ optimiser
    def adam_step(self, grads, lr):
        """torch.optim.Adam(lr, weight_decay=1e-4, eps=1e-8): coupled L2 (Q8)."""
        self.adam_t += 1
        k = self.adam_t
        bc1 = 1 - ADAM_B1 ** k
        bc2 = 1 - ADAM_B2 ** k
        step_size = lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        for n, w in self.p.items():
            g = grads[n] + ADAM_WD * w
            self.m[n].mul_(ADAM_B1).add_(g, alpha=1 - ADAM_B1)
            self.v[n].mul_(ADAM_B2).addcmul_(g, g, value=1 - ADAM_B2)
            denom = (self.v[n].sqrt() / bc2_sqrt).add_(ADAM_EPS)
            w.addcdiv_(self.m[n], denom, value=-step_size)

    def train_step(self, x0, eps, t, keeps, lr):
        loss, grads, outs, _ = self.loss_and_grads(x0, eps, t, keeps)
        self.adam_step(grads, lr)
        return float(loss), grads, outs

    @staticmethod
    def epoch_lr(base_lr, ep, epochs):
        """Linear per-epoch decay (train_SDRM.py:316)."""
        return base_lr * (1 - ep / epochs)

    # ------------------------------------------------------------------ sampling
    def sample(self, xT, z, keeps, Tj=None):
        """Reverse loop.  z [T+1,n,L] (already scaled by nd; z[1] is ignored: no
        noise at i==1), keeps [T+1,n,L].  Tj None = full resolution (all rows run
        i=T..1); otherwise row j runs i=Tj[j]..1 — identical to the reference's
        per-user batch-1 loop because rows are independent (Q11)."""
        x = _t(xT).clone()
        z, keeps = _t(z), _t(keeps)
        n = x.shape[0]
        Tj = torch.full((n,), self.T, dtype=torch.int64) if Tj is None else _t(Tj, torch.int64)
        for i in range(int(Tj.max()), 0, -1):
            act = Tj >= i
            if not bool(act.any()):
                continue
            xa = x[act]
            tt = torch.full((xa.shape[0],), i, dtype=torch.int64)
            eps_hat = self.forward(xa, tt, keeps[i][act])
            zi = z[i][act] if i > 1 else torch.zeros_like(xa)
            x[act] = reverse_update(xa, eps_hat, zi, i, self.beta, self.alpha, self.alphabar)
        return x

    # ------------------------------------------------------------------ helpers
    def flat(self, names):
        return np.concatenate([self.p[n].numpy().ravel() for n in names])


def flat_of(d, names):
    return np.concatenate([np.asarray(d[n]).ravel() for n in names])
