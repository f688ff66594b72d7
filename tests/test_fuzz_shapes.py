"""Seeded random net / batch shapes through the train step: ragged edges of every tile of the tail (32x32, 16x32, 16x16
sub-tiles, 8x8 tiles of the embedding gradient), odd T, L != W, narrow nets on both sides of their envelope (padded width
<= 64, T <= 128), batches that are not multiples of the 16-user groups.  Each shape: the three-phase step against the CPU
oracle (tests/test_hip_parity.py::test_train_step_vs_oracle, the same bars), then the one-call sdrm_train_step on a second
engine, which must land on the SAME parameters bit for bit (Adam fused into the tail == the three phases).  `pytest -m gpu`."""
import numpy as np
import pytest

from sdrm_amd import synth
from test_hip_parity import test_train_step_vs_oracle as check_against_oracle

pytestmark = pytest.mark.gpu


def _shapes(n=20, seed=20261004):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        narrow = k % 3 == 0
        hi = 64 if narrow else 150
        L, W = int(rng.integers(1, hi + 1)), int(rng.integers(1, hi + 1))
        if k % 4 == 1:
            W = L                                   # the reference's call sites all have L == W
        T = int(rng.integers(2, 136))               # both sides of the narrow path's T <= 128
        H = int(rng.integers(0, 4))
        B = int(rng.integers(1, 261))
        out.append((L, W, T, H, B))
    # the longest embeddings sdrm_create accepts (k_tail_emb's LDS image ends at T = 1020; T >= 1021 is refused)
    out += [(24, 24, 1020, 1, 7), (70, 70, 517, 0, 9)]
    return out


@pytest.mark.parametrize("dims", _shapes())
def test_random_shape_train_step(engine_cls, dims):
    check_against_oracle(engine_cls, dims, -1)
    L, W, T, H, B = dims
    init = synth.flatten_params(synth.init_params(L, W, T, H, seed=3), H)
    x0 = synth.synth_latents(B, L, seed=4)
    eps, t, masks = synth.synth_train_randoms(B, L, T, 0.9, seed=5)
    a, b = engine_cls(L, W, T, H, B), engine_cls(L, W, T, H, B)
    a.set_params(init); b.set_params(init)
    for step in range(2):
        la = a.train_step(x0, 1e-3, noise=eps, t=t, keep=masks)
        b.train_forward(x0, noise=eps, t=t, keep=masks)
        lb = b.train_backward()
        b.adam_step(1e-3)
        assert float(la.cpu()) == float(lb.cpu()), (dims, step)
        assert bool((a.get_grads() == b.get_grads()).all()), (dims, step)
        assert bool((a.get_params() == b.get_params()).all()), (dims, step)
    (ma, va, ta), (mb, vb, tb) = a.get_adam_state(), b.get_adam_state()
    assert ta == tb == 2 and bool((ma == mb).all()) and bool((va == vb).all())
    a.close(); b.close()
