#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the upstream reference.

Runs ONLY where `/root/reference` exists (the build container, CPU).  Nothing
from the reference is copied: this script *imports* `train_SDRM.py`, drives its
own functions (`train_SDRM`, `sample_ddpm`, `SDRM.forward`, `perturb_input`,
`denoise_add_noise`, `SDRM.timestep_embedding`) with injected randomness and
stores inputs + the reference's outputs as small `.npz` files.  The GPU box
never sees the reference; it sees these vectors.

Recipe: stub the two absent third-party modules (`optuna`,
`bottleneck`), import with PYTHONDONTWRITEBYTECODE=1, and replace the VAE
pre-stage with an identity "encoder" so that the batches fed through `dl` are
the latents themselves.  Randomness is injected by temporarily replacing
`torch.randn`, `torch.randn_like`, `torch.randint`, `numpy.random.randint` and
`torch.nn.functional.dropout` with recorders that draw from a seeded numpy
`RandomState`; every draw is logged into the fixture.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
from __future__ import annotations

import contextlib
import os
import sys
import types

import numpy as np

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

import pandas  # noqa: F401,E402  (import before the bottleneck stub)
import torch  # noqa: E402

from sdrm_amd import synth  # noqa: E402

REFERENCE = os.environ.get("SDRM_REFERENCE", "/root/reference")


def load_reference():
    optuna = types.ModuleType("optuna")

    class TrialPruned(Exception):
        pass

    optuna.TrialPruned = TrialPruned
    sys.modules.setdefault("optuna", optuna)
    bn = types.ModuleType("bottleneck")
    bn.argpartition = np.argpartition
    sys.modules.setdefault("bottleneck", bn)
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    import train_SDRM as ref  # type: ignore

    assert ref.DEVICE == "cpu"
    return ref


class Recorder:
    """Seeded source for every random draw the reference makes; logs each draw."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.log = []  # (kind, array)

    def normal(self, shape):
        a = self.rs.standard_normal(tuple(shape)).astype(np.float32)
        self.log.append(("normal", a))
        return torch.from_numpy(a.copy())

    def integers(self, lo, hi, shape):
        a = self.rs.randint(lo, hi, size=tuple(shape)).astype(np.int64)
        self.log.append(("int", a))
        return torch.from_numpy(a.copy())

    def np_int(self, lo, hi):
        v = int(self.rs.randint(lo, hi))
        self.log.append(("npint", np.asarray(v, dtype=np.int64)))
        return v

    def keep_mask(self, shape):
        a = (self.rs.random_sample(tuple(shape)) < 0.5).astype(np.uint8)
        self.log.append(("mask", a))
        return torch.from_numpy(a.astype(np.float32))

    def take(self, kind):
        return [a for k, a in self.log if k == kind]


@contextlib.contextmanager
def injected_randomness(ref, rec: Recorder):
    saved = (torch.randn, torch.randn_like, torch.randint, ref.np.random.randint, ref.F.dropout)

    def randn(*size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        return rec.normal(size)

    def randn_like(x, **kw):
        return rec.normal(x.shape)

    def randint(lo, hi, size, **kw):
        return rec.integers(lo, hi, size)

    def np_randint(lo, hi=None, size=None):
        assert size is None
        return rec.np_int(lo, hi)

    def dropout(x, p=0.5, training=True, inplace=False):
        assert p == 0.5
        return x * rec.keep_mask(x.shape) / (1.0 - p)

    torch.randn, torch.randn_like, torch.randint = randn, randn_like, randint
    ref.np.random.randint = np_randint
    ref.F.dropout = dropout
    try:
        yield
    finally:
        torch.randn, torch.randn_like, torch.randint, ref.np.random.randint, ref.F.dropout = saved


class IdentityVAE(torch.nn.Module):
    """Stands in for the VAE pre-stage: encode(x) = (x, 0), decode(z) = z."""

    def __init__(self, input_dim=None, hidden_dim=None, latent_dim=None):
        super().__init__()
        self.model_is_trained = False
        self.latent_dim = latent_dim

    def encode(self, x):
        return x, torch.zeros(())

    def decode(self, z):
        return z


def run_reference_training(ref, L, W, T, H, batches, epochs, lr, nd, init, seed):
    """Call the reference's real `train_SDRM()` (`train_SDRM.py:271-340`) with the
    VAE stage stubbed; return the trained net and a per-step trace."""
    rec = Recorder(seed)
    trace = {"P": [], "grads": [], "params_after": [], "loss": []}

    # NB: `ref.SDRM` itself cannot be swapped (its __init__ resolves the global
    # name through super(SDRM, self)), so weights are injected when the optimiser
    # is built (`train_SDRM.py:309`, right after the net).  Outputs are captured by
    # wrapping the class's `forward` (the train loop calls `.forward` directly, so
    # module hooks would miss the first of the three passes).
    orig_forward = ref.SDRM.forward

    def recording_forward(self, x, t):
        y = orig_forward(self, x, t)
        trace["P"].append(y.detach().numpy().copy())
        return y

    ref.SDRM.forward = recording_forward

    holder = {}

    class SpyAdam(torch.optim.Adam):
        def __init__(self, params, **kw):
            params = list(params)
            with torch.no_grad():
                for p, n in zip(params, synth.param_names(H)):
                    assert tuple(p.shape) == init[n].shape, (n, p.shape)
                    p.copy_(torch.from_numpy(init[n]))
            holder["params"] = params
            holder["kw"] = kw
            super().__init__(params, **kw)
            holder["opt"] = self

        def step(self, closure=None):
            trace["grads"].append([p.grad.detach().numpy().copy() for p in holder["params"]])
            out = super().step(closure)
            trace["params_after"].append([p.detach().numpy().copy() for p in holder["params"]])
            return out

    def fake_vae_training(model, **kw):
        model.model_is_trained = True

    saved = (ref.VAE, ref.train_variational_autoencoder, torch.optim.Adam)
    ref.VAE, ref.train_variational_autoencoder = IdentityVAE, fake_vae_training
    torch.optim.Adam = SpyAdam
    dl = [(torch.from_numpy(b.copy()).to_sparse(), None) for b in batches]
    try:
        with injected_randomness(ref, rec):
            net, vae = ref.train_SDRM(
                dl, N_ITEMS=L, VAE_HIDDEN=8, VAE_LATENT=L, VAE_BATCH_SIZE=8, VAE_LR=1e-3,
                DIFF_LATENT=W, N_HIDDEN_MLP_LAYERS=H, DIFF_LR=lr, DIFF_TRAINING_EPOCHS=epochs,
                TIMESTEPS=T, noise_divider=nd, VAE_DIR_PATH="/tmp/unused_vae_dir",
                TRAIN_PARTIAL_VALID_DATA=None, VALID_DATA=None, OPTIMIZATION_OBJECTIVE="Recall@10")
    finally:
        ref.VAE, ref.train_variational_autoencoder, torch.optim.Adam = saved
        ref.SDRM.forward = orig_forward
    opt = holder.get("opt")
    names = [n for n, _ in net.named_parameters()]
    assert names == synth.param_names(H), (names, synth.param_names(H))
    trace["names"] = names
    trace["adam_kw"] = holder.get("kw")
    if opt is not None and len(opt.state):
        trace["exp_avg"] = [opt.state[p]["exp_avg"].numpy().copy() for p in holder["params"]]
        trace["exp_avg_sq"] = [opt.state[p]["exp_avg_sq"].numpy().copy() for p in holder["params"]]
        trace["adam_step"] = int(opt.state[holder["params"][0]]["step"])
    trace["normal"] = rec.take("normal")
    trace["int"] = rec.take("int")
    trace["mask"] = rec.take("mask")
    trace["schedule"] = (ref.b_t.numpy().copy(), ref.a_t.numpy().copy(), ref.ab_t.numpy().copy())
    return net, vae, trace


# ----------------------------------------------------------------------------
def fixture_schedule(ref):
    out = {}
    for T in (3, 8, 78, 83, 93, 198):
        init = synth.init_params(4, 4, T, 0, seed=1)
        _, _, tr = run_reference_training(ref, 4, 4, T, 0, batches=[], epochs=0, lr=1e-5, nd=1.0, init=init, seed=0)
        b, a, ab = tr["schedule"]
        out[f"beta_{T}"], out[f"alpha_{T}"], out[f"alphabar_{T}"] = b, a, ab
    np.savez_compressed(os.path.join(HERE, "schedule.npz"), **out)


def fixture_timestep_embedding(ref):
    out = {}
    for T in (3, 7, 8, 78, 83, 93, 198):
        net = ref.SDRM(4, T, 4, 0)
        t = torch.arange(0, T + 1, dtype=torch.long)
        out[f"temb_{T}"] = net.timestep_embedding(t, T).numpy().copy()
    np.savez_compressed(os.path.join(HERE, "timestep_embedding.npz"), **out)


def fixture_forward(ref):
    """`SDRM.forward` (`train_SDRM.py:97-103`) with injected weights and mask."""
    out = {}
    cases = [(24, 24, 7, 0), (24, 24, 8, 1), (20, 20, 9, 2), (40, 40, 93, 5), (136, 136, 12, 3), (24, 40, 6, 2)]
    for ci, (L, W, T, H) in enumerate(cases):
        init = synth.init_params(L, W, T, H, seed=10 + ci)
        net = ref.SDRM(L, T, W, H)
        with torch.no_grad():
            for n, p in net.named_parameters():
                p.copy_(torch.from_numpy(init[n]))
        net.eval()  # dropout stays on regardless (Q2)
        for B in (1, 5):
            rs = np.random.RandomState(100 * ci + B)
            x = rs.standard_normal((B, L)).astype(np.float32)
            t = rs.randint(1, T + 1, size=(B,)).astype(np.int64)
            rec = Recorder(1000 * ci + B)
            with injected_randomness(ref, rec), torch.no_grad():
                y = net.forward(torch.from_numpy(x), torch.from_numpy(t))
            key = f"c{ci}_B{B}"
            out[key + "_x"], out[key + "_t"] = x, t
            out[key + "_mask"] = rec.take("mask")[0]
            out[key + "_y"] = y.numpy().copy()
        out[f"c{ci}_dims"] = np.asarray([L, W, T, H], dtype=np.int64)
        out[f"c{ci}_flat"] = synth.flatten_params(init, H)
    out["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(os.path.join(HERE, "forward.npz"), **out)


def pack_training_trace(prefix, out, tr, batches, epochs, lr, nd, dims, init):
    L, W, T, H = dims
    out[prefix + "dims"] = np.asarray([L, W, T, H], dtype=np.int64)
    out[prefix + "hyper"] = np.asarray([lr, nd, epochs, len(batches)], dtype=np.float64)
    out[prefix + "init_flat"] = synth.flatten_params(init, H)
    n_steps = epochs * len(batches)
    assert len(tr["grads"]) == n_steps and len(tr["P"]) == 3 * n_steps
    for s in range(n_steps):
        x0 = batches[s % len(batches)]
        out[prefix + f"s{s}_x0"] = x0
        out[prefix + f"s{s}_raw_noise"] = tr["normal"][s]
        out[prefix + f"s{s}_t"] = tr["int"][s]
        out[prefix + f"s{s}_masks"] = np.stack(tr["mask"][3 * s:3 * s + 3])
        out[prefix + f"s{s}_P"] = tr["P"][3 * s]
        out[prefix + f"s{s}_S"] = tr["P"][3 * s + 1]
        out[prefix + f"s{s}_Q"] = tr["P"][3 * s + 2]
        out[prefix + f"s{s}_grad_flat"] = np.concatenate([g.ravel() for g in tr["grads"][s]])
        out[prefix + f"s{s}_param_flat"] = np.concatenate([p.ravel() for p in tr["params_after"][s]])
    out[prefix + "exp_avg_flat"] = np.concatenate([m.ravel() for m in tr["exp_avg"]])
    out[prefix + "exp_avg_sq_flat"] = np.concatenate([m.ravel() for m in tr["exp_avg_sq"]])
    out[prefix + "adam_step"] = np.asarray(tr["adam_step"])


def reference_loss(ref, net_init, dims, x0, raw_noise, nd, t, masks, schedule):
    """Loss value of one step, recomputed through the reference's own
    `perturb_input` + `score_matching_loss` (`train_SDRM.py:191-203`) because
    `train_SDRM()` discards it."""
    L, W, T, H = dims
    net = ref.SDRM(L, T, W, H)
    with torch.no_grad():
        for n, p in net.named_parameters():
            p.copy_(torch.from_numpy(net_init[n]))
    ref.b_t, ref.a_t, ref.ab_t = (torch.from_numpy(s.copy()) for s in schedule)
    q = [torch.from_numpy(m.astype(np.float32)) for m in masks]
    saved = ref.F.dropout
    ref.F.dropout = lambda x, p=0.5, training=True, inplace=False: x * q.pop(0) / 0.5
    try:
        enc = torch.from_numpy(x0.copy())
        noise = torch.from_numpy(raw_noise.copy()) * nd
        tt = torch.from_numpy(t.copy())
        xp = ref.perturb_input(enc, tt, noise)
        pred = net.forward(xp, tt)
        loss = ref.score_matching_loss(net, XT=enc, t=tt, epsilon_theta=pred, epsilon=noise, mu=.1)
    finally:
        ref.F.dropout = saved
    return float(loss.detach()), xp.detach().numpy().copy()


TRAIN_CASES = [
    # L,  W,  T, H, batch rows, lr,    nd
    (24, 24, 9, 2, (5, 3), 2.1e-3, 1.0),
    (20, 20, 8, 0, (4, 4), 1.0e-3, 0.7),
    (24, 40, 6, 1, (6, 2), 5.0e-4, 1.0),
    (40, 40, 93, 5, (7, 5), 1.3e-3, 1.0),
]
# a net inside the envelope of the row-owned train forward (csrc/rowchain.h: L == W, padded width 128..352); 37 rows are one
# full group of 32 users and a partial one, 21 rows a lone partial group
TRAIN_WIDE_CASES = [(100, 100, 7, 1, (37, 21), 5.0e-4, 0.9)]


def fixture_train(ref, cases=TRAIN_CASES, name="train", seed0=0):
    """Whole `train_SDRM()` runs: 2 epochs x 2 batches (short last batch) so the
    per-epoch lr decay (`train_SDRM.py:316`), the shared hidden layer (Q1) and the
    global Adam step counter are all exercised."""
    out = {}
    for ci, (L, W, T, H, rows, lr, nd) in enumerate(cases):
        init = synth.init_params(L, W, T, H, seed=20 + ci + seed0)
        rs = np.random.RandomState(200 + ci + seed0)
        batches = [rs.standard_normal((r, L)).astype(np.float32) for r in rows]
        epochs = 2
        _, _, tr = run_reference_training(ref, L, W, T, H, batches, epochs, lr, nd, init, seed=300 + ci + seed0)
        prefix = f"c{ci}_"
        pack_training_trace(prefix, out, tr, batches, epochs, lr, nd, (L, W, T, H), init)
        loss0, xpert0 = reference_loss(ref, init, (L, W, T, H), batches[0], tr["normal"][0], nd,
                                       tr["int"][0], tr["mask"][0:3], tr["schedule"])
        out[prefix + "s0_loss"] = np.asarray(loss0, dtype=np.float64)
        out[prefix + "s0_xpert"] = xpert0
        out[prefix + "adam_kw"] = np.asarray([tr["adam_kw"]["lr"], tr["adam_kw"]["weight_decay"], tr["adam_kw"]["eps"]])
    out["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def fixture_train_wide(ref):
    fixture_train(ref, cases=TRAIN_WIDE_CASES, name="train_wide", seed0=50)


def fixture_elementwise(ref):
    """`perturb_input` (`:202-203`) and `denoise_add_noise` (`:20-25`), tensor-t and
    int-t forms, with and without injected noise (the i==1 case passes z=0)."""
    out = {}
    T, L, n = 11, 12, 6
    init = synth.init_params(4, 4, T, 0, seed=1)
    run_reference_training(ref, 4, 4, T, 0, batches=[], epochs=0, lr=1e-5, nd=1.0, init=init, seed=0)  # sets globals
    rs = np.random.RandomState(5)
    x = rs.standard_normal((n, L)).astype(np.float32)
    noise = rs.standard_normal((n, L)).astype(np.float32)
    t = rs.randint(1, T + 1, size=(n,)).astype(np.int64)
    out["T"] = np.asarray(T)
    out["x"], out["noise"], out["t"] = x, noise, t
    out["perturbed"] = ref.perturb_input(torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(noise)).numpy().copy()
    eps = rs.standard_normal((n, L)).astype(np.float32)
    z = rs.standard_normal((n, L)).astype(np.float32)
    out["eps"], out["z"] = eps, z
    for i in (T, 5, 2, 1):
        out[f"rev_int_{i}"] = ref.denoise_add_noise(torch.from_numpy(x), i, torch.from_numpy(eps), torch.from_numpy(z)).numpy().copy()
    out["rev_int_1_nonoise"] = ref.denoise_add_noise(torch.from_numpy(x), 1, torch.from_numpy(eps), 0).numpy().copy()
    # tensor-t form as used by the multi-resolution branch (batch 1, t = tensor([i]))
    out["rev_tensor_4"] = ref.denoise_add_noise(torch.from_numpy(x[:1]), torch.as_tensor([4]), torch.from_numpy(eps[:1]),
                                                torch.from_numpy(z[0])).numpy().copy()
    np.savez_compressed(os.path.join(HERE, "elementwise.npz"), **out)


def fixture_sampling(ref):
    """`sample_ddpm` (`train_SDRM.py:27-63`), both branches, after a short real
    training run (the schedule globals come from that run, Q10)."""
    out = {}
    cases = [(24, 24, 9, 2, 1.0), (20, 20, 8, 0, 0.6), (40, 40, 13, 3, 1.0)]
    for ci, (L, W, T, H, nd) in enumerate(cases):
        init = synth.init_params(L, W, T, H, seed=40 + ci)
        rs = np.random.RandomState(400 + ci)
        batches = [rs.standard_normal((6, L)).astype(np.float32)]
        net, vae, tr = run_reference_training(ref, L, W, T, H, batches, 1, 1e-3, nd, init, seed=500 + ci)
        prefix = f"c{ci}_"
        out[prefix + "dims"] = np.asarray([L, W, T, H], dtype=np.int64)
        out[prefix + "nd"] = np.asarray(nd)
        out[prefix + "flat"] = np.concatenate([p.detach().numpy().ravel() for _, p in net.named_parameters()])
        n = 5
        # full resolution
        rec = Recorder(600 + ci)
        with injected_randomness(ref, rec):
            full = ref.sample_ddpm(n, net, vae, L, noise_divider=nd, n_timesteps=T)
        normals, masks = rec.take("normal"), rec.take("mask")
        assert len(normals) == 1 + (T - 1) and len(masks) == T
        z = np.zeros((T + 1, n, L), np.float32)
        mk = np.zeros((T + 1, n, L), np.uint8)
        for k, i in enumerate(range(T, 0, -1)):
            if i > 1:
                z[i] = normals[1 + k]  # raw N(0,1); consumer multiplies by nd
            mk[i] = masks[k]
        out[prefix + "full_xT"], out[prefix + "full_rawz"], out[prefix + "full_masks"] = normals[0], z, mk
        out[prefix + "full_out"] = full.numpy().copy()
        # multi resolution ('random')
        rec = Recorder(700 + ci)
        with injected_randomness(ref, rec):
            multi = ref.sample_ddpm(n, net, vae, L, noise_divider=nd, timesteps='random', n_timesteps=T)
        Tj = np.asarray([int(a) for a in rec.take("npint")], dtype=np.int64)
        normals, masks = rec.take("normal"), rec.take("mask")
        z = np.zeros((T + 1, n, L), np.float32)
        mk = np.zeros((T + 1, n, L), np.uint8)
        ni, mi = 1, 0
        for j in range(n):
            for i in range(int(Tj[j]), 0, -1):
                if i > 1:
                    z[i, j] = normals[ni]
                    ni += 1
                mk[i, j] = masks[mi][0]
                mi += 1
        assert ni == len(normals) and mi == len(masks)
        out[prefix + "multi_xT"], out[prefix + "multi_rawz"], out[prefix + "multi_masks"] = normals[0], z, mk
        out[prefix + "multi_Tj"] = Tj
        out[prefix + "multi_out"] = multi.numpy().copy()
    out["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(os.path.join(HERE, "sampling.npz"), **out)


def fixture_fullsize(ref):
    """Checksums only (sum, L2, 16 strided samples) of one full-size train step and
    one full-size forward for the BASELINE shapes; inputs come from
    `sdrm_amd.synth` seeds so nothing large is stored."""
    out = {}
    shapes = {
        "ml100k": (830, 830, 83, 2, 550, 2.1e-5, 265),
        "ml1m": (340, 340, 78, 1, 160, 9.8e-5, 15),
        "ml1m_big": (340, 340, 78, 1, 2048, 9.8e-5, 15),
        "adm": (40, 40, 93, 5, 850, 1.3e-5, 185),
    }
    for name, (L, W, T, H, B, lr, epochs) in shapes.items():
        init = synth.init_params(L, W, T, H, seed=1)
        x0 = synth.synth_latents(B, L, seed=0)
        eps, t, masks = synth.synth_train_randoms(B, L, T, 1.0, seed=2)
        net = ref.SDRM(L, T, W, H)
        with torch.no_grad():
            for n, p in net.named_parameters():
                p.copy_(torch.from_numpy(init[n]))
        b = (0.02 - 1e-4) * torch.linspace(0, 1, T + 1) + 1e-4
        # schedule through the reference's own code path
        run_reference_training(ref, 4, 4, T, 0, batches=[], epochs=0, lr=1e-5, nd=1.0,
                               init=synth.init_params(4, 4, T, 0, seed=1), seed=0)
        assert torch.equal(ref.b_t, b)
        q = [torch.from_numpy(m.astype(np.float32)) for m in masks]
        saved = ref.F.dropout
        ref.F.dropout = lambda x, p=0.5, training=True, inplace=False: x * q.pop(0) / 0.5
        outs = []
        hook = net.register_forward_hook(lambda m, i, o: outs.append(o.detach().numpy().copy()))
        try:
            opt = torch.optim.Adam(net.parameters(), lr=lr, weight_decay=0.0001, eps=1e-8)
            enc = torch.from_numpy(x0.copy())
            noise = torch.from_numpy(eps.copy())
            tt = torch.from_numpy(t.copy())
            xp = ref.perturb_input(enc, tt, noise)
            pred = net(xp, tt)  # via __call__ so the hook sees P as well
            loss = ref.score_matching_loss(net, XT=enc, t=tt, epsilon_theta=pred, epsilon=noise, mu=.1)
            loss.backward()
            grads = [p.grad.detach().numpy().copy() for _, p in net.named_parameters()]
            opt.step()
        finally:
            ref.F.dropout = saved
            hook.remove()
        pfx = name + "_"
        out[pfx + "dims"] = np.asarray([L, W, T, H, B], dtype=np.int64)
        out[pfx + "lr"] = np.asarray(lr)
        out[pfx + "loss"] = np.asarray(float(loss.detach()), dtype=np.float64)
        for tag, arr in (("P", outs[0]), ("S", outs[1]), ("Q", outs[2])):
            s, l2, smp = synth.stats(arr)
            out[pfx + tag + "_sum"], out[pfx + tag + "_l2"], out[pfx + tag + "_smp"] = s, l2, smp
        out[pfx + "grad_l2"] = np.asarray([np.sqrt((g.astype(np.float64) ** 2).sum()) for g in grads])
        out[pfx + "grad_absmax"] = np.asarray([np.abs(g).max() for g in grads])
        out[pfx + "grad_sum"] = np.asarray([g.astype(np.float64).sum() for g in grads])
        gs = [synth.stats(g)[2] for g in grads if g.size >= 16]
        out[pfx + "grad_smp"] = np.stack(gs)
        newp = np.concatenate([p.detach().numpy().ravel() for _, p in net.named_parameters()])
        s, l2, smp = synth.stats(newp, 64)
        out[pfx + "param_sum"], out[pfx + "param_l2"], out[pfx + "param_smp"] = s, l2, smp
        dl2 = np.sqrt(((newp.astype(np.float64) - synth.flatten_params(init, H).astype(np.float64)) ** 2).sum())
        out[pfx + "update_l2"] = np.asarray(dl2)
    np.savez_compressed(os.path.join(HERE, "fullsize.npz"), **out)


def fixture_host(ref):
    """Host-side contracts: state_dict key layout of SDRM for several H (aliased shared layer, Q1/a13),
    and the metric / split helpers of utilities.py on small random data."""
    import utilities as refutil  # the reference's own module (bottleneck stubbed by np.argpartition)
    from scipy.sparse import csr_matrix
    out = {}
    for H in (0, 1, 2, 5):
        net = ref.SDRM(6, 4, 6, H)
        out[f"state_keys_H{H}"] = np.asarray(list(net.state_dict().keys()))
        out[f"param_keys_H{H}"] = np.asarray([n for n, _ in net.named_parameters()])
    rs = np.random.RandomState(9)
    pred = rs.standard_normal((12, 40)).astype(np.float32)
    held = csr_matrix((rs.random_sample((12, 40)) < 0.15).astype(np.float64))
    dense_held = held.toarray()
    dense_held[0, :] = 0          # a user with nothing held out (0/0 -> nan in both metrics)
    held = csr_matrix(dense_held)
    train = csr_matrix((rs.random_sample((12, 40)) < 0.2).astype(np.float64))
    out["pred"], out["held"], out["train"] = pred, held.toarray(), train.toarray()
    for k in (1, 5, 10):
        with np.errstate(all="ignore"):
            out[f"recall_{k}"] = refutil.recall_at_k_batch(pred.copy(), held, k=k)
            out[f"ndcg_{k}"] = refutil.NDCG_binary_at_k_batch(pred.copy(), held, k=k)
    out["masked"] = refutil.mask_training_examples(train, pred.copy())
    data = csr_matrix((rs.random_sample((30, 25)) < 0.3).astype(np.float64))
    tr, te = refutil.split_train_test_proportion_from_csr_matrix(data, batch_size=7, random_seed=123, test_prop=0.2)
    out["split_in"], out["split_train"], out["split_test"] = data.toarray(), tr.toarray(), te.toarray()
    np.savez_compressed(os.path.join(HERE, "host.npz"), **out)


def fixture_ml100k(ref):
    """The reference's ML-100k split (data/ml-100k/*.pkl, binary CSR) in a compact form: data, not code."""
    import pickle
    out = {}
    for tag in ("train_test", "valid"):
        m = pickle.load(open(os.path.join(REFERENCE, "data", "ml-100k", f"ml-100k_{tag}.pkl"), "rb")).tocsr()
        m.sort_indices()
        assert set(np.unique(m.data)) <= {0, 1} and m.shape[1] < 65536     # explicit zeros are kept: the split helper
        out[tag + "_data"] = m.data.astype(np.uint8)                        # counts stored entries, not values
        out[tag + "_indptr"] = m.indptr.astype(np.int32)
        out[tag + "_indices"] = m.indices.astype(np.uint16)
        out[tag + "_shape"] = np.asarray(m.shape, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "ml100k.npz"), **out)


def fixture_e2e(ref):
    """End-to-end Recall@k / NDCG@k band of the reference on ML-100k / SVD (README hyper-parameters,
    main.py:143-200 restated as a driver), seeds 0..4, CPU.  ~2 minutes per seed."""
    import dataloaders as refdl
    import svd_benchmark as refsvd
    import pandas as pd
    ref.VAE.get_l2_reg = lambda self: 0.0 * next(self.parameters()).sum()   # the original calls .cuda() (Q12)
    hp = dict(epochs=265, batch=550, lr=2.1e-5, T=83, nd=1.0, H=2, vae_batch=780, vae_hidden=930, latent=830, vae_lr=6e-4)
    train, train_partial, valid = refdl.load_data("ml-100k", os.path.join(REFERENCE, "data"))
    n_items, n_users = train.shape[1], train.shape[0]
    sparsity = 1 - train.nnz / (train.shape[0] * train.shape[1])
    res = {"seeds": [], "M_recall": [], "M_ndcg": [], "F_recall": [], "F_ndcg": [], "V_recall": [], "V_ndcg": []}
    seeds = [int(x) for x in os.environ.get("E2E_SEEDS", "0,1,2,3,4").split(",")]
    for seed in seeds:
        torch.manual_seed(seed)
        np.random.seed(seed)
        ds = refdl.SparseDataset(train_partial, train_partial)
        sampler = torch.utils.data.sampler.BatchSampler(
            torch.utils.data.sampler.RandomSampler(ds, generator=torch.Generator(device="cpu")), batch_size=hp["batch"], drop_last=False)
        dl = torch.utils.data.DataLoader(ds, batch_size=1, collate_fn=refdl.sparse_batch_collate,
                                         generator=torch.Generator(device="cpu"), sampler=sampler, shuffle=False)
        net, vae = ref.train_SDRM(dl=dl, N_ITEMS=n_items, VAE_LATENT=hp["latent"], VAE_HIDDEN=hp["vae_hidden"], VAE_LR=hp["vae_lr"],
                                  VAE_BATCH_SIZE=hp["vae_batch"], DIFF_LATENT=hp["latent"], DIFF_TRAINING_EPOCHS=hp["epochs"],
                                  DIFF_LR=hp["lr"], N_HIDDEN_MLP_LAYERS=hp["H"], TIMESTEPS=hp["T"], noise_divider=hp["nd"],
                                  VAE_DIR_PATH=f"/tmp/ref_vae_{seed}", TRAIN_PARTIAL_VALID_DATA=train_partial, VALID_DATA=valid,
                                  OPTIMIZATION_OBJECTIVE="Recall@10", verbose=False)
        M = ref.sample_ddpm(n_users, net, vae, hp["latent"], hp["nd"], timesteps="random", n_timesteps=hp["T"]).detach().cpu().numpy()
        F_ = ref.sample_ddpm(n_users, net, vae, hp["latent"], hp["nd"], n_timesteps=hp["T"]).detach().cpu().numpy()
        V = vae.sample(n_users)
        for tag, raw in (("M", M), ("F", F_), ("V", V)):
            binar = pd.DataFrame((raw >= np.quantile(raw.flatten(), sparsity)).astype(int))
            rec, ndcg = refsvd.compute_mf_results(train, valid, synthetic_data=binar, nnmf=False, only_synthetic=True)
            res[tag + "_recall"].append(rec); res[tag + "_ndcg"].append(ndcg)
        res["seeds"].append(seed)
        print("seed", seed, "M", res["M_recall"][-1][3], "F", res["F_recall"][-1][3], "V", res["V_recall"][-1][3], flush=True)
    out = {k: np.asarray(v) for k, v in res.items()}
    out["k"] = np.asarray([1, 3, 5, 10, 20, 50])
    out["hyper"] = np.asarray([hp["epochs"], hp["batch"], hp["lr"], hp["T"], hp["nd"], hp["H"], hp["vae_batch"], hp["vae_hidden"],
                               hp["latent"], hp["vae_lr"]])
    # E2E_APPEND=1: the runs of this call are appended to the committed file (round 4 added seeds 5..9 to round 1's 0..4)
    path = os.path.join(HERE, "e2e_ml100k_svd.npz")
    if os.environ.get("E2E_APPEND") == "1" and os.path.exists(path):
        old = np.load(path)
        assert np.array_equal(old["hyper"], out["hyper"]) and not set(old["seeds"].tolist()) & set(out["seeds"].tolist())
        for key in res:
            out[key] = np.concatenate([old[key], out[key]])
    np.savez_compressed(path, **out)


def fixture_rank_metrics(ref):
    """utilities.py:116-171 as svd_benchmark.py:58-66 chains them: mask_training_examples, then recall_at_k_batch and
    NDCG_binary_at_k_batch for k = 1,3,5,10,20,50 - run by the reference's own functions on synthetic scores (rows
    checked to be tie-free, since bn.argpartition leaves ties unspecified) and synthetic interactions."""
    import utilities as refutil
    out = {}
    cases = [(48, 300, 11), (20, 64, 12), (9, 2000, 13)]
    ks = [1, 3, 5, 10, 20, 50]
    out["ks"] = np.asarray(ks)
    out["n_cases"] = np.asarray(len(cases))
    for ci, (users, items, seed) in enumerate(cases):
        scores = synth.synth_scores(users, items, seed=seed)
        assert all(len(np.unique(r)) == items for r in scores), "tie in a score row: pick another seed"
        train, held = synth.synth_interactions(users, items, seed=seed)
        masked = refutil.mask_training_examples(train, scores.copy())
        rec, ndcg = [], []
        with np.errstate(all="ignore"):
            for k in ks:
                rec.append(refutil.recall_at_k_batch(masked.copy(), held, k=k))
                ndcg.append(refutil.NDCG_binary_at_k_batch(masked.copy(), held, k=k))
        out[f"c{ci}_shape"] = np.asarray([users, items, seed])
        out[f"c{ci}_recall"] = np.asarray(rec, dtype=np.float64)
        out[f"c{ci}_ndcg"] = np.asarray(ndcg, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "rank_metrics.npz"), **out)


def fixture_equal_sparsity(ref=None):
    """main.py:177-180 verbatim on synthetic decoder outputs: threshold = np.quantile(M.flatten(), SPARSITY);
    M_equal_sparsity = (M >= threshold).astype(int).  The arithmetic is numpy's (this container: numpy 2.2.6)."""
    out = {}
    cases = [("normal", 37, 101, 0.937), ("ties", 64, 250, 0.9), ("narrow", 50, 333, 0.0634), ("normal", 1, 1, 0.5),
             ("normal", 3, 5, 0.0), ("normal", 3, 5, 1.0), ("ties", 40, 100, 0.5), ("normal", 211, 997, 0.99)]
    out["n_cases"] = np.asarray(len(cases))
    for i, (kind, users, items, sparsity) in enumerate(cases):
        M = synth.synth_scores(users, items, seed=100 + i, kind=kind)
        threshold = np.quantile(M.flatten(), sparsity)
        eq = (M >= threshold).astype(int)
        out[f"c{i}_kind"] = np.asarray(kind)
        out[f"c{i}_shape"] = np.asarray([users, items, 100 + i])
        out[f"c{i}_sparsity"] = np.asarray(sparsity)
        out[f"c{i}_threshold"] = np.asarray(threshold)
        out[f"c{i}_bits"] = np.packbits(eq.astype(np.uint8).ravel())
        out[f"c{i}_ones"] = np.asarray(int(eq.sum()))
    np.savez_compressed(os.path.join(HERE, "equal_sparsity.npz"), **out)


VAE_DECODE_CASES = [(24, 37, 101, 13), (83, 60, 1008, 7), (40, 185, 257, 33), (1, 1, 1, 1), (340, 96, 64, 5)]   # latent, hidden, items, n


def fixture_vae_decode(ref):
    """The reference's own VAE class (train_SDRM.py:206-268) with injected decoder tensors: `vae.decode(z)` (:252-254)."""
    out = {"n_cases": np.asarray(len(VAE_DECODE_CASES))}
    for i, (latent, hidden, items, n) in enumerate(VAE_DECODE_CASES):
        vae = ref.VAE(items, hidden, latent).eval()
        w1, b1, w2, b2 = synth.synth_vae_decoder(latent, hidden, items, seed=200 + i)
        with torch.no_grad():
            vae.decoder[0].weight.copy_(torch.from_numpy(w1)); vae.decoder[0].bias.copy_(torch.from_numpy(b1))
            vae.decoder[2].weight.copy_(torch.from_numpy(w2)); vae.decoder[2].bias.copy_(torch.from_numpy(b2))
            z = torch.from_numpy(synth.synth_latents(n, latent, seed=300 + i) * 2.0)
            y = vae.decode(z)
        out[f"c{i}_dims"] = np.asarray([latent, hidden, items, n, 200 + i, 300 + i])
        out[f"c{i}_out"] = y.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "vae_decode.npz"), **out)


def main():
    torch.set_num_threads(8)
    ref = load_reference()
    which = sys.argv[1:] or ["schedule", "temb", "forward", "train", "train_wide", "elementwise", "sampling", "fullsize", "host", "equal_sparsity", "rank_metrics", "vae_decode"]
    if which == ["equal_sparsity"]:     # numpy only: no need to import the reference
        fixture_equal_sparsity()
        print("wrote equal_sparsity")
        return
    table = {"schedule": fixture_schedule, "temb": fixture_timestep_embedding, "forward": fixture_forward,
             "train": fixture_train, "train_wide": fixture_train_wide, "elementwise": fixture_elementwise, "sampling": fixture_sampling,
             "fullsize": fixture_fullsize, "host": fixture_host, "ml100k": fixture_ml100k, "e2e": fixture_e2e,
             "equal_sparsity": fixture_equal_sparsity, "rank_metrics": fixture_rank_metrics, "vae_decode": fixture_vae_decode}
    for w in which:
        table[w](ref)
        print("wrote", w)


if __name__ == "__main__":
    main()
