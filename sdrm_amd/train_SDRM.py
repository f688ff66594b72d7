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
loss stays on the device in `DIFF.last_loss`; `SDRM.parameters()` returns snapshots (the live
parameters sit in the engine, `state_dict()/load_state_dict()` move them in and out)."""
from __future__ import annotations

import math
import os
import time
import warnings
from collections import OrderedDict

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import metrics as utilities
from . import synth
from .engine import Engine, SdrmError, utility_engine

warnings.filterwarnings("ignore")

DEVICE = "cuda" if torch.cuda.is_available() else "cpu"

# schedule of the most recent train_SDRM() call, kept as module state like the reference (Q10);
# sample_ddpm() prefers the schedule stored on the net it is given.
b_t = a_t = ab_t = None


def _pruned(msg):
    try:  # the reference signals VAE checkpoint IO failures to Optuna (:72,:83)
        import optuna  # type: ignore
        return optuna.TrialPruned(msg)
    except Exception:
        return RuntimeError(msg)


def checkpoint(model, filename, VAE_DIR_PATH):
    """Save model parameters to file (:75-83)."""
    try:
        torch.save(model.state_dict(), os.path.normpath(os.path.join(VAE_DIR_PATH, filename)))
    except Exception:
        print("Failed to save model parameters to %s" % filename)
        raise _pruned("checkpoint failed")


def resume(model, filename, VAE_DIR_PATH):
    """Load model parameters from file (:66-72)."""
    try:
        model.load_state_dict(torch.load(os.path.normpath(os.path.join(VAE_DIR_PATH, filename))))
    except Exception:
        print("Failed to load model parameters from %s" % filename)
        raise _pruned("resume failed")


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

    def _flat(self):
        return self._engine.get_params() if self._engine is not None else self._pending

    def named_parameters(self):
        flat, off = self._flat(), 0
        for name, shp in ((n, synth.param_shapes(self.L, self.W, self.T, self.H)[n]) for n in synth.param_names(self.H)):
            k = int(np.prod(shp))
            yield name, flat[off:off + k].reshape(shp)
            off += k

    def parameters(self):
        return [p for _, p in self.named_parameters()]

    def state_dict(self):
        """Reference key layout, including the aliased keys of the shared hidden layer (dnn.4.* ...)."""
        sd = OrderedDict((n, p.detach().clone()) for n, p in self.named_parameters())
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
class VAE(nn.Module):
    """MultiVAE++ (:206-268), PyTorch: the encode/decode hooks the denoising engine sits between."""

    def __init__(self, input_dim, hidden_dim, latent_dim, p_drop=0.5):
        super().__init__()
        self.latent_dim = latent_dim
        self.encoder = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.Tanh(), nn.Linear(hidden_dim, 2 * latent_dim))
        self.decoder = nn.Sequential(nn.Linear(latent_dim, hidden_dim), nn.Tanh(), nn.Linear(hidden_dim, input_dim))
        self.dropout = nn.Dropout(p=p_drop)
        self.model_is_trained = False
        self.is_training = 0
        self.weight_decay = 0
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight.data)
                m.bias.data.normal_(0.0, 0.001)

    def encode(self, x):
        h = self.encoder(self.dropout(F.normalize(x, p=2, dim=1)))
        mu, logvar = torch.chunk(h, chunks=2, dim=1)
        kl = -0.5 * torch.mean(torch.sum(1 + logvar - mu.pow(2) - logvar.exp(), dim=1))
        eps = torch.randn_like(mu)  # consumed even in eval, like the reference (Q12)
        return mu + self.is_training * eps * torch.exp(0.5 * logvar), kl

    def decode(self, z):
        return self.decoder(z)

    def forward(self, x):
        z, kl = self.encode(x)
        return self.decode(z), kl

    def get_l2_reg(self):
        if self.weight_decay <= 0:
            return torch.zeros((), device=next(self.parameters()).device)
        return self.weight_decay * sum(torch.norm(p, p=2) ** 2 for n, p in self.named_parameters() if n.endswith(".weight"))

    def sample(self, n_samples):
        z = torch.randn(n_samples, self.latent_dim, device=next(self.parameters()).device)
        return self.decode(z).cpu().detach().numpy()


def train_variational_autoencoder(model, train_data, test_data, epochs, batch_size, lr, early_stop_metric="NDCG@50",
                                  VAE_DIR_PATH="./", verbose=False):
    """VAE pre-stage (:115-188): multinomial NLL + annealed KL, early stopping on Recall/NDCG@k of a
    per-user hold-out of `test_data`, best epoch restored.  Plain PyTorch (not part of the hot path)."""
    os.makedirs(os.path.normpath(VAE_DIR_PATH), exist_ok=True)
    dev = next(model.parameters()).device
    anneal_cap, anneal_count = 0.2, 0.0
    best_metric, best_epoch, stale = -np.inf, 0, 0
    optimizer = torch.optim.Adam(model.parameters(), lr=lr)
    k = int(early_stop_metric.split("@")[1])
    start = time.time()
    for epoch in range(epochs):
        losses = []
        model.train()
        model.is_training = 1
        train_data = train_data[np.random.permutation(train_data.shape[0])]
        for lo in range(0, train_data.shape[0], batch_size):
            hi = min(lo + batch_size, train_data.shape[0])
            anneal = min(anneal_cap, 1.0 * anneal_count / 20_000)
            X = torch.tensor(train_data[lo:hi].toarray(), dtype=torch.float32, device=dev)
            optimizer.zero_grad()
            out, kl = model(X)
            neg_ll = -torch.mean(torch.sum(F.log_softmax(out, dim=1) * X, dim=1))
            loss = neg_ll + anneal * kl + model.get_l2_reg()
            losses.append(loss.item())
            loss.backward()
            optimizer.step()
            anneal_count += 1
        model.eval()
        model.is_training = 0
        scores = []
        valid_train, valid_test = utilities.split_train_test_proportion_from_csr_matrix(test_data, batch_size=1000)
        with torch.no_grad():
            for lo in range(0, valid_train.shape[0], 500):
                hi = min(lo + 500, valid_train.shape[0])
                X = valid_train[lo:hi]
                pred, _ = model(torch.tensor(X.toarray(), dtype=torch.float32, device=dev))
                if dev.type == "cuda":
                    # utilities.py:116-171 on the device (sdrm_rank_metrics): the [500, N_ITEMS] scores stay in HBM
                    rec, ndcg = utility_engine(dev).rank_metrics(pred, valid_test[lo:hi], train=X, ks=(k,))
                    scores.append((rec if "Recall" in early_stop_metric else ndcg)[0].cpu().numpy())
                else:
                    pred = utilities.mask_training_examples(X, pred.cpu().numpy())
                    fn = utilities.recall_at_k_batch if "Recall" in early_stop_metric else utilities.NDCG_binary_at_k_batch
                    scores.append(fn(pred, valid_test[lo:hi], k=k))
        avg = np.nanmean(np.concatenate(scores))
        if verbose:
            print(f"Epoch: {epoch}, Loss: {np.round(np.mean(losses), 4)}, {early_stop_metric}: {np.round(avg, 4)}", end="\r")
        if avg > best_metric:
            best_metric, best_epoch, stale = avg, epoch, 0
            checkpoint(model, f"epoch-{epoch}.pth", VAE_DIR_PATH)
        else:
            stale += 1
            if stale > 20:
                if verbose:
                    print(f"MultiVAE++ training complete. Early stopping at epoch {epoch}, "
                          f"Training took {np.round((time.time() - start) / 60, 2)} minutes")
                break
    resume(model, f"epoch-{best_epoch}.pth", VAE_DIR_PATH)
    model.model_is_trained = True
    model.is_training = 0


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
    """Where the engine's decode beats the PyTorch module it replaces (measured, profiles/r02_next_rows_bench.txt and
    r03_next_rows_bench.txt): ML-1M (5429 x 3125, hidden 600) 246 us against 301; ML-100k (843 x 1008) 60 against 53 - five
    staging launches on a 0.8 GFLOP problem - and ADM (9558 x 8582, hidden 200: a 200-deep contraction) 414 against 403 lose.
    So: a hidden width that fills the K loop, and enough work to amortise the staging.  `sdrm_vae_decode` itself is correct at
    every shape (tests/test_vae_decode.py); this only routes the hook."""
    return hidden >= 256 and float(n_users) * float(n_items) * float(hidden) >= 4e9


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
