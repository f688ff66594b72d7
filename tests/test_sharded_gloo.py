"""The N>1 path on CPU: 2 processes, gloo backend, `ShardedTrainer` from the product driving an
oracle-backed stand-in for the three C-ABI phases.  Checks that sharding rows over ranks (with the
scalar all-reduce before backward and the gradient all-reduce before Adam) reproduces the
single-process step, incl. a ragged split and Philox keyed by the global row."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import philox_ref as pr  # noqa: E402
from oracle import sdrm_oracle as orc  # noqa: E402
from sdrm_amd import synth  # noqa: E402
from sdrm_amd.parallel import ShardedTrainer, shard_rows  # noqa: E402

L, W, T, H, B = 12, 16, 7, 2, 11
SEED, LR, ND = 4242, 2e-3, 0.9


class OraclePhases:
    """Same three phases as include/sdrm_hip.h (train_forward / train_backward / adam_step), CPU oracle
    inside.  Test-only stand-in for `sdrm_amd.engine.Engine`."""

    def __init__(self, init):
        self.o = orc.Oracle(L, W, T, H, init)
        self.names = synth.param_names(H)
        self.P = synth.param_count(L, W, T, H)
        self.device = "cpu"

    def train_forward(self, x0, seed=0, step=0, nd=1.0, row0=0, sums=None, noise=None, t=None, keep=None):
        x0 = torch.as_tensor(x0)
        if noise is None:
            noise, t, keep = pr.train_randoms(seed, step, row0, x0.shape[0], L, T, nd)
        o = self.o
        eps, tt = torch.as_tensor(noise), torch.as_tensor(t)
        self.c = [dict(), dict(), dict()]
        xp = orc.q_sample(x0, tt, eps, o.alphabar)
        P = o.forward(xp, tt, torch.as_tensor(keep[0]).float(), self.c[0])
        S = o.forward(x0, tt, torch.as_tensor(keep[1]).float(), self.c[1])
        Q = o.forward(x0 + orc.MU * eps, tt, torch.as_tensor(keep[2]).float(), self.c[2])
        self.R, self.S, self.D = P - x0, S, (Q - S) / orc.MU ** 2 - (P - x0)
        R, D = self.R.double(), self.D.double()
        sums[:5] = torch.tensor([(D * D).sum(), ((R - S.double()) ** 2).sum(), R.sum(), (R * R).sum(), R.numel()])
        return sums

    def train_backward(self, sums=None, grad=None):
        sD, sC, sR, sR2, N = (float(v) for v in sums[:5])
        A, C, Rbar = sD / N, sC / N, sR / N
        V = (sR2 - N * Rbar * Rbar) / (N - 1)
        den = 1e-8 + V
        k = 0.5 / den
        gD = (2 * k / N) * self.D
        gC = (2 * k / N) * (self.R - self.S)
        gV = (-(0.5 * (A + C) / den ** 2) * 2 / (N - 1)) * (self.R - Rbar)
        grads = {n: torch.zeros_like(v) for n, v in self.o.p.items()}
        for cache, g in zip(self.c, (-gD + gC + gV, -gD / orc.MU ** 2 - gC, gD / orc.MU ** 2)):
            self.o.backward(cache, g.float(), grads)
        grad.copy_(torch.cat([grads[n].reshape(-1) for n in self.names]))
        return torch.tensor(0.5 * (A + C) / den)

    # the two-bucket form of phase 2 (sdrm_train_backward_begin / _finish)
    def grad_buckets(self):
        shapes = synth.param_shapes(L, W, T, H)
        first = sum(int(np.prod(shapes[n])) for n in ("emb_layer.weight", "emb_layer.bias", "dnn.0.weight", "dnn.0.bias"))
        return (0, first), (first, self.P - first)

    def train_backward_begin(self, sums=None, grad=None):
        full = torch.zeros(self.P)
        loss = self.train_backward(sums=sums, grad=full)
        first = self.grad_buckets()[0][1]
        grad[:first] = full[:first]
        self._second = full[first:].clone()
        return loss

    def train_backward_finish(self, grad=None):
        grad[self.P - self._second.numel():] = self._second

    def adam_step(self, lr, grad=None):
        shapes = synth.param_shapes(L, W, T, H)
        out, off = {}, 0
        for n in self.names:
            k = int(np.prod(shapes[n]))
            out[n] = grad[off:off + k].reshape(shapes[n]).clone()
            off += k
        self.o.adam_step(out, lr)

    def train_step(self, x0, lr, seed=0, step=0, nd=1.0, **kw):
        sums, grad = torch.zeros(8, dtype=torch.float64), torch.zeros(self.P)
        self.train_forward(x0, seed=seed, step=step, nd=nd, sums=sums, **kw)
        loss = self.train_backward(sums=sums, grad=grad)
        self.adam_step(lr, grad=grad)
        return loss


def _worker(rank, world, port, init, x0, q, overlap=True):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    eng = OraclePhases(init)
    tr = ShardedTrainer(eng, rank, world, device="cpu", n_params=eng.P, overlap=overlap)
    r0, rows = shard_rows(B, rank, world)
    losses = []
    for step in range(3):
        loss = tr.train_step(torch.from_numpy(x0[r0:r0 + rows]), LR * (1 - step / 3), row0=r0, step=step, seed=SEED, nd=ND)
        losses.append(float(loss))
    q.put((rank, eng.o.flat(eng.names), losses))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(180)
@pytest.mark.parametrize("overlap", [True, False])
def test_two_rank_gloo_matches_single_process(overlap):
    init = synth.init_params(L, W, T, H, seed=13)
    x0 = synth.synth_latents(B, L, seed=14)
    single = OraclePhases(init)
    ref_losses = [float(single.train_step(torch.from_numpy(x0), LR * (1 - s / 3), seed=SEED, step=s, nd=ND)) for s in range(3)]
    ref = single.o.flat(single.names)

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, init, x0, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, flat, losses in results:
        assert np.sqrt(((flat - ref) ** 2).sum() / (ref ** 2).sum()) < 1e-5, rank
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-5)
    assert np.array_equal(results[0][1], results[1][1])       # replicas stay bit-identical


def _gpu_worker(rank, world, port, init_flat, x0, dims, q, overlap):
    """One rank of the real thing on a shared GPU: the product's Engine + ShardedTrainer, gloo carrying the collectives
    (RCCL refuses two ranks on one device; the RCCL path itself is driven by tests/test_rccl_exchange.py)."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sdrm_amd.engine import Engine
    gl, gw, gt, gh, gb = dims
    torch.cuda.set_device(0)
    eng = Engine(gl, gw, gt, gh, max_rows=gb)
    eng.set_params(init_flat)
    tr = ShardedTrainer(eng, rank, world, overlap=overlap)
    r0, rows = shard_rows(gb, rank, world)
    losses = []
    for step in range(3):
        loss = tr.train_step(torch.from_numpy(x0[r0:r0 + rows]).cuda(), LR * (1 - step / 3), row0=r0, step=step, seed=SEED, nd=ND)
        losses.append(float(loss.cpu()))
    q.put((rank, eng.get_params().cpu().numpy(), losses))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("overlap", [True, False])
def test_two_rank_gloo_real_engine_on_shared_gpu(overlap):
    """The same two-rank run with the REAL engine in each process (both on cuda:0): the user-sharded step - ragged split,
    Philox keyed by the global row, loss sums and gradient buckets exchanged between the C-ABI phases - reproduces the
    single-engine sdrm_train_step trajectory, and the replicas stay bit-identical."""
    from sdrm_amd.engine import Engine
    dims = (72, 80, 9, 2, 45)
    gl, gw, gt, gh, gb = dims
    init_flat = synth.flatten_params(synth.init_params(gl, gw, gt, gh, seed=13), gh)
    x0 = synth.synth_latents(gb, gl, seed=14)
    single = Engine(gl, gw, gt, gh, max_rows=gb)
    single.set_params(init_flat)
    ref_losses = [float(single.train_step(x0, LR * (1 - s / 3), seed=SEED, step=s, nd=ND).cpu()) for s in range(3)]
    ref = single.get_params().cpu().numpy()
    single.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, init_flat, x0, dims, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, flat, losses in results:
        assert np.sqrt(((flat - ref) ** 2).sum() / (ref ** 2).sum()) < 1e-5, rank
        np.testing.assert_allclose(losses, ref_losses, rtol=2e-5)
    assert np.array_equal(results[0][1], results[1][1])


def test_oracle_phases_equal_oracle_step():
    """The three-phase decomposition (sums -> seeds) is the same arithmetic as Oracle.train_step."""
    init = synth.init_params(L, W, T, H, seed=13)
    x0 = synth.synth_latents(B, L, seed=14)
    eps, t, keep = pr.train_randoms(SEED, 0, 0, B, L, T, ND)
    a, b = OraclePhases(init), orc.Oracle(L, W, T, H, init)
    la = float(a.train_step(torch.from_numpy(x0), LR, seed=SEED, step=0, nd=ND))
    lb, _, _ = b.train_step(x0, eps, t, list(keep), LR)
    assert abs(la - lb) <= 1e-5 * abs(lb)
    fa, fb = a.o.flat(a.names), b.flat(a.names)
    assert np.sqrt(((fa - fb) ** 2).sum() / (fb ** 2).sum()) < 1e-5
