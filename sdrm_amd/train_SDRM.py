"""Drop-in Python surface of the reference's `train_SDRM.py`, backed by libsdrm_hip.so.

Same public names, argument order and return values as /root/reference/train_SDRM.py, so that
`from train_SDRM import train_SDRM, sample_ddpm` in main.py:148-175 /
hyperparameter_search.py:147,386,691 can be pointed here:

    train_SDRM(dl, N_ITEMS, VAE_HIDDEN, VAE_LATENT, ..., verbose=False) -> (DIFF, variational_ae)   (:271-340)
    sample_ddpm(n_sample, diff_net, vae_net, diff_latent_dim, noise_divider, timesteps, n_timesteps) (:27-63)
    SDRM (:86-112)  VAE (:206-268)  perturb_input (:202)  denoise_add_noise (:20)  score_matching_loss (:191)
    train_variational_autoencoder (:115-188)  checkpoint (:75)  resume (:66)  DEVICE (:18)

What runs where: the eps-net forward/backward/Adam and the whole reverse-sampling loop run in the HIP
engine (no torch ops); the VAE's pre-training and its encode hook stay in PyTorch on the same device, as
the north-star asks; the decode hook at the end of `sample_ddpm` runs on the engine's GEMM when the
decoder is the reference's two-layer MLP and as the PyTorch module otherwise.  The engine draws its randomness with on-device Philox keyed
by `torch.initial_seed()`-derived seeds; explicit randoms can be injected for parity runs.

Differences a caller can observe (all documented in DESIGN.md): the latents of the frozen eval-mode VAE
are identical to the reference's (z = mu(x)); `loss.item()` is not called per step (Q14), the per-step
loss stays on the device in `DIFF.last_loss`; `SDRM.parameters()` returns read-only views of the engine's live
parameters (`EngineParameter`: in-place writes raise; `state_dict()/load_state_dict()` move them in and out)."""
from __future__ import annotations

import math
import time
import warnings
from collections import OrderedDict

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import synth
from .engine import Engine, SdrmError
from .vae_hooks import VAE, checkpoint, resume, train_variational_autoencoder  # noqa: F401  (part of the reference's module surface)

warnings.filterwarnings("ignore")

DEVICE = "cuda" if torch.cuda.is_available() else "cpu"

# schedule of the most recent train_SDRM() call, kept as module state like the reference (Q10);
# sample_ddpm() prefers the schedule stored on the net it is given.
b_t = a_t = ab_t = None


# --------------------------------------------------------------------------------------------------
class EngineParameter(torch.Tensor):
    """A view of the engine's live parameter vector.  Reads behave like any tensor; in-place writes raise: the engine's kernels
    read padded / transposed / fragment-packed copies that only `load_state_dict` and the train step refresh, so an optimiser
    stepping these tensors would change nothing the net computes with (the reference's callers never do: they only call
    `train_SDRM`, `diff_net.eval()` and `diff_net.forward`, hyperparameter_search.py:53,67,78)."""

    _HARMLESS = frozenset({"requires_grad_", "retain_grad", "share_memory_", "register_hook"})   # end in "_" / mutate nothing of the data

    @staticmethod
    def _writes(func, name):
        """True when `func` writes through one of its tensor arguments: by its schema where it has one (aten overloads carry alias
        info), else by torch's naming rule for in-place methods and the in-place operator dunders."""
        if name in EngineParameter._HARMLESS:
            return False
        schema = getattr(func, "_schema", None)
        if schema is not None:
            return bool(schema.is_mutable)
        return (name.endswith("_") and not name.endswith("__")) or name in ("__setitem__", "__iadd__", "__isub__", "__imul__", "__itruediv__",
                                                                            "__ifloordiv__", "__imod__", "__ipow__", "__iand__", "__ior__",
                                                                            "__ixor__", "__ilshift__", "__irshift__", "__imatmul__")

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", "")
        outs = (kwargs or {}).get("out")
        outs = outs if isinstance(outs, (tuple, list)) else (outs,)
        if cls._writes(func, name) or any(isinstance(o, EngineParameter) for o in outs):
            raise SdrmError(f"in-place {name} on a parameter of an engine-backed SDRM: its parameters are updated by the engine's train "
                            "step (train_SDRM / Engine.train_step) or replaced with load_state_dict()")
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **(kwargs or {}))
        # a result that shares the parameter's storage (`.data`, `.detach()`, a view, a slice) is a parameter view too: it keeps the
        # guard, so `p.data.add_()` / `p.detach().mul_()` / `p.view(-1).zero_()` raise like `p.add_()` does
        if name != "as_subclass" and isinstance(out, torch.Tensor) and not isinstance(out, EngineParameter):   # (as_subclass: the caller's explicit way out)
            src = next((a for a in args if isinstance(a, EngineParameter)), None)
            try:
                if src is not None and out.device == src.device and out.untyped_storage().data_ptr() == src.untyped_storage().data_ptr():
                    out = out.as_subclass(EngineParameter)
            except RuntimeError:
                pass   # (a result without storage, e.g. a meta / sparse tensor: not a view of the parameters)
        return out


# --------------------------------------------------------------------------------------------------
class SDRM:
    """eps-predictor with the reference's constructor (:87): `SDRM(N_ITEMS, EMB_DIM, LATENT_DIM, n_hidden_layers)`.
    Parameters are initialised like `nn.Linear` / `nn.PReLU` defaults (from torch's CPU generator) and live
    in the HIP engine.  The H hidden layers share one weight/bias/slope (Q1)."""

    def __init__(self, N_ITEMS, EMB_DIM, LATENT_DIM=200, n_hidden_layers=4, max_rows=1024, device=None):
        self.L, self.T, self.W, self.H = int(N_ITEMS), int(EMB_DIM), int(LATENT_DIM), int(n_hidden_layers)
        self.EMB_DIM, self.n_hidden_layers = self.T, self.H
        self.training = True
        self._device = device
        self._engine = None
        self._max_rows = int(max_rows)
        self._seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFF
        self._calls = 0
        self.last_loss = None
        shapes = synth.param_shapes(self.L, self.W, self.T, self.H)
        init, bound = OrderedDict(), None
        for name in synth.param_names(self.H):
            shp = shapes[name]
            if name.endswith(".weight") and len(shp) == 2:
                bound = 1.0 / math.sqrt(shp[1])
                init[name] = torch.empty(shp).uniform_(-bound, bound)
            elif name.endswith(".bias"):
                init[name] = torch.empty(shp).uniform_(-bound, bound)
            else:
                init[name] = torch.full(shp, synth.PRELU_INIT)
        self._pending = torch.cat([v.reshape(-1) for v in init.values()])

    # -- engine management ---------------------------------------------------------------------
    def engine(self, rows=1) -> Engine:
        """The live engine, (re)built with room for `rows` rows; state survives a rebuild."""
        if self._engine is None or rows > self._engine.max_rows:
            cap = max(self._max_rows, int(rows))
            state = None
            if self._engine is not None:
                m, v, step = self._engine.get_adam_state()
                state = (self._engine.get_params(), m, v, step)
                self._engine.close()
            self._engine = Engine(self.L, self.W, self.T, self.H, cap, device=self._device)
            if state is not None:
                self._engine.set_params(state[0])
                self._engine.set_adam_state(state[1], state[2], state[3])
            else:
                self._engine.set_params(self._pending)
            self._max_rows = cap
        return self._engine

    # -- nn.Module-like surface ----------------------------------------------------------------
    def to(self, device=None, **_):
        if device is not None and str(device) != "cpu":
            self._device = torch.device(device).index
        return self

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)  # dropout stays on regardless (F.dropout default, Q2)

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def cpu(self):
        raise SdrmError("this SDRM lives in the HIP engine on a ROCm device (no CPU path); use state_dict() to take its parameters to the host")

    def zero_grad(self, set_to_none=True):
        return None   # gradients live in the engine and are overwritten by every backward

    def modules(self):
        return iter([self])

    def _flat(self):
        return self._engine.params_view() if self._engine is not None else self._pending

    def named_parameters(self):
        """The reference's `named_parameters()` order (a13).  With a live engine the tensors ALIAS the engine's parameter vector (they
        follow every train step, like an nn.Module's parameters); they are `EngineParameter`s: reading is free, an in-place write raises,
        and torch.optim refuses them when it is built (they are views, not leaves) instead of silently training a copy."""
        flat, off = self._flat(), 0
        live = self._engine is not None
        out = []
        # views of a leaf that requires grad: torch.optim refuses them at construction ("can't optimize a non-leaf Tensor") - built
        # under enable_grad(), so that holds for a caller inside torch.no_grad() as well (evaluation code usually is)
        with torch.enable_grad():
            if live:
                flat = flat.requires_grad_(True)
            for name, shp in ((n, synth.param_shapes(self.L, self.W, self.T, self.H)[n]) for n in synth.param_names(self.H)):
                k = int(np.prod(shp))
                v = flat[off:off + k].reshape(shp)
                out.append((name, (v.as_subclass(EngineParameter) if live else v)))
                off += k
        return iter(out)

    def parameters(self):
        return [p for _, p in self.named_parameters()]

    def state_dict(self):
        """Reference key layout, including the aliased keys of the shared hidden layer (dnn.4.* ...)."""
        sd = OrderedDict((n, p.as_subclass(torch.Tensor).detach().clone()) for n, p in self.named_parameters())
        out = OrderedDict()
        last = 2 + 2 * self.H
        aliases = synth.alias_keys(self.H)
        order = ["emb_layer.weight", "emb_layer.bias"] + [f"dnn.{i}.{k}" for i in range(last + 1)
                                                          for k in (("weight", "bias") if i % 2 == 0 else ("weight",))]
        for key in order:
            out[key] = sd[aliases.get(key, key)]
        return out

    def load_state_dict(self, sd, strict=True):
        names = synth.param_names(self.H)
        missing = [n for n in names if n not in sd]
        if missing and strict:
            raise KeyError(f"missing keys: {missing}")
        flat = torch.cat([torch.as_tensor(sd[n], dtype=torch.float32).reshape(-1).cpu() for n in names])
        if self._engine is not None:
            self._engine.set_params(flat)
        else:
            self._pending = flat
        return self

    def forward(self, x, t):
        """eps_hat = f(x, t) (:97-103) on torch tensors; a fresh dropout mask per call (Q2)."""
        x = torch.as_tensor(x, dtype=torch.float32)
        t = torch.as_tensor(t).reshape(-1)
        eng = self.engine(x.shape[0])
        self._calls += 1
        return eng.forward(x, t.expand(x.shape[0]) if t.numel() == 1 else t, seed=self._seed, step=self._calls)

    __call__ = forward

    def timestep_embedding(self, timesteps, dim):
        """Sinusoidal embedding (:105-112) as a torch tensor (the engine uses its own table)."""
        half = dim // 2
        freqs = torch.exp(-math.log(10_000) * torch.arange(0, half, dtype=torch.float32) / half).to(timesteps.device)
        args = timesteps[:, None].float() * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if dim % 2:
            emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
        return emb


# --------------------------------------------------------------------------------------------------
def perturb_input(x, t, noise):
    """sqrt(abar[t]) x + (1 - abar[t]) noise (:202-203), torch ops on the module-level schedule."""
    return torch.as_tensor(ab_t.sqrt()[t, None] * x + (1 - ab_t[t, None]) * noise, dtype=torch.float)


def denoise_add_noise(x, t, pred_noise, z=None):
    """One reverse update (:20-25), torch ops on the module-level schedule."""
    if z is None:
        z = torch.randn_like(x)
    noise = b_t.sqrt()[t] * z
    mean = (x - pred_noise * ((1 - a_t[t]) / (1 - ab_t[t]).sqrt())) / a_t[t].sqrt()
    return mean + noise


def score_matching_loss(model, XT, t, epsilon_theta, epsilon, mu):
    """Loss value of :191-199 from three forwards of `model` (value only: the engine's fused train step
    computes the same quantity together with its gradients; this helper exists for callers that log it)."""
    score_x = model(XT, t)
    perturbed_score_x = model(XT + mu * epsilon, t)
    score_diff = (perturbed_score_x - score_x) / (mu ** 2)
    residual = epsilon_theta - XT
    return 0.5 * (F.mse_loss(score_diff, residual) + F.mse_loss(residual, score_x)) / (1e-8 + residual.var())


def _set_schedule_globals(eng):
    global b_t, a_t, ab_t
    b, a, ab = eng.get_schedule()
    b_t, a_t, ab_t = (torch.from_numpy(v).to(DEVICE) for v in (b, a, ab))


def train_SDRM(dl, N_ITEMS, VAE_HIDDEN, VAE_LATENT, VAE_BATCH_SIZE, VAE_LR, DIFF_LATENT, N_HIDDEN_MLP_LAYERS, DIFF_LR,
               DIFF_TRAINING_EPOCHS, TIMESTEPS, noise_divider, VAE_DIR_PATH, TRAIN_PARTIAL_VALID_DATA, VALID_DATA,
               OPTIMIZATION_OBJECTIVE, verbose=False, variational_ae=None, cache_latents=False):
    """(:271-340) Train the VAE (PyTorch), freeze it, then train the eps-net on its latents in the HIP
    engine.  `dl` yields `(x, _)` with x a sparse/dense [b, N_ITEMS] tensor.  Extras (keyword-only in
    spirit): `variational_ae` = an already trained VAE to reuse; `cache_latents` = encode the feed once
    per call instead of once per batch per epoch (identical latents since the frozen eval-mode encoder
    is deterministic, Q13; only the order of torch RNG consumption changes)."""
    if not torch.cuda.is_available():
        raise SdrmError("train_SDRM needs a ROCm device (no CPU fallback)")
    if variational_ae is None:
        variational_ae = VAE(input_dim=N_ITEMS, hidden_dim=VAE_HIDDEN, latent_dim=VAE_LATENT).to(DEVICE)
        train_variational_autoencoder(variational_ae, train_data=TRAIN_PARTIAL_VALID_DATA, test_data=VALID_DATA,
                                      epochs=500, batch_size=VAE_BATCH_SIZE, lr=VAE_LR,
                                      early_stop_metric=OPTIMIZATION_OBJECTIVE, VAE_DIR_PATH=VAE_DIR_PATH, verbose=verbose)
    assert variational_ae.model_is_trained
    for p in variational_ae.parameters():
        p.requires_grad = False
    variational_ae.eval()

    DIFF = SDRM(N_ITEMS=VAE_LATENT, EMB_DIM=TIMESTEPS, LATENT_DIM=DIFF_LATENT, n_hidden_layers=N_HIDDEN_MLP_LAYERS)
    DIFF.to(DEVICE).train()
    eng = DIFF.engine(1)
    _set_schedule_globals(eng)

    def latents(x):
        with torch.no_grad():
            x = x.to_dense() if x.layout != torch.strided else x
            z, _ = variational_ae.encode(x.to(DEVICE))
        return z.float().contiguous()

    cached = [latents(x) for x, _ in iter(dl)] if cache_latents else None
    start, step = time.time(), 0
    for ep in range(DIFF_TRAINING_EPOCHS):
        if verbose:
            print(f"SDRM Epoch: {ep + 1}/{DIFF_TRAINING_EPOCHS}", end="\r")
        lr = DIFF_LR * (1 - ep / DIFF_TRAINING_EPOCHS)                     # linear decay (:316)
        feed = cached if cached is not None else (latents(x) for x, _ in iter(dl))
        for z in feed:
            eng = DIFF.engine(z.shape[0])
            DIFF.last_loss = eng.train_step(z, lr, seed=DIFF._seed, step=step, nd=noise_divider)   # (:326-337)
            step += 1
    if verbose:
        torch.cuda.synchronize()
        print(f"SDRM training complete, Training took {np.round((time.time() - start) / 60, 2)} minutes")
    return DIFF, variational_ae


def decoder_tensors(vae_net):
    """The four tensors of a `Linear -> Tanh -> Linear` decoder (the reference's, :212-214) if `vae_net` has one on a ROCm
    device in float32, else None (any other decode hook is called as the module it is)."""
    dec = getattr(vae_net, "decoder", None)
    if not (isinstance(dec, nn.Sequential) and len(dec) == 3 and isinstance(dec[0], nn.Linear) and isinstance(dec[1], nn.Tanh)
            and isinstance(dec[2], nn.Linear) and dec[0].bias is not None and dec[2].bias is not None):
        return None
    ts = (dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias)
    if any(t.dtype != torch.float32 or not t.is_cuda for t in ts):
        return None
    return ts


def _decode(eng, vae_net, latents):
    """vae_net.decode(latents) (:49 / :61).  The reference's own decoder (this module's `VAE`, or any module that sets
    `sdrm_engine_decode = True` and keeps the `Linear -> Tanh -> Linear` decoder) runs on the engine's MFMA GEMM
    (`sdrm_vae_decode`); any other decode hook is called as the module it is."""
    ts = decoder_tensors(vae_net) if (isinstance(vae_net, VAE) or getattr(vae_net, "sdrm_engine_decode", False)) else None
    if ts is None or not engine_decode_pays(latents.shape[0], ts[0].shape[0], ts[2].shape[0]):
        return vae_net.decode(latents)
    return eng.vae_decode(latents, *ts)


def engine_decode_pays(n_users: int, hidden: int, n_items: int) -> bool:
    """Where the engine's decode beats the PyTorch module it replaces: since round 4 (the five staging launches became one) at
    every BASELINE shape - ML-100k (843 x 1008) 50 us against 62, ML-1M (5429 x 3125) 224 against 294, ADM (9558 x 8582) 361
    against 381 (profiles/r04_next_rows_bench.txt) - so the hook is always routed to the engine; kept as the one place to change
    that."""
    return True


@torch.no_grad()
def sample_ddpm(n_sample, diff_net, vae_net, diff_latent_dim, noise_divider=1.0, timesteps: str = None,
                n_timesteps=None, verbose=False):
    """(:27-63) Reverse sampling from pure noise, then `vae_net.decode`.  `timesteps='random'` = the
    multi-resolution branch (each user starts at its own T_j ~ U{1..n_timesteps-1}); run as one batched
    loop with inactive rows masked, which equals the reference's per-user batch-1 loop (Q11)."""
    diff_net.eval()
    vae_net.eval()
    start = time.time()
    if n_timesteps is not None and int(n_timesteps) != diff_net.T:
        raise SdrmError(f"n_timesteps={n_timesteps} does not match the trained schedule length {diff_net.T}")
    if int(diff_latent_dim) != diff_net.L:
        raise SdrmError("diff_latent_dim does not match the eps-net input width")
    eng = diff_net.engine(n_sample)
    diff_net._calls += 1
    latents = eng.sample(n_sample, nd=noise_divider, multires=(timesteps == "random"), seed=diff_net._seed,
                         call_id=diff_net._calls)
    samples = _decode(eng, vae_net, latents)
    if verbose:
        print(f"Sampling {n_sample}/{n_sample}, Sampling took {np.round((time.time() - start) / 60, 2)} minutes")
    return samples
