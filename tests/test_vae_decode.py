"""the VAE decode hook (train_SDRM.py:212-214, :252-254) and the decode + equal-sparsity chain
(main.py:170-180) on the engine.

CPU: oracle/vae_decode_ref.py against golden outputs of the reference's own VAE class (tests/golden/vae_decode.npz).
GPU (-m gpu): sdrm_vae_decode through the C ABI against those goldens, the fp64 oracle and torch's own decode on the
device (1e-4 normwise: fp32 GEMM orders differ); the decode followed by sdrm_equal_sparsity against np.quantile applied to the
engine's OWN decoded matrix (byte work: threshold bit pattern and 0/1 matrix identical), with and without the raw
matrix handed back, at the BASELINE shapes."""
import os

import numpy as np
import pytest
import torch

from sdrm_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "vae_decode.npz")
TOL = 1e-4


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def rel_max(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def golden_cases():
    g = np.load(GOLD)
    for i in range(int(g["n_cases"])):
        latent, hidden, items, n, wseed, zseed = (int(v) for v in g[f"c{i}_dims"])
        tensors = synth.synth_vae_decoder(latent, hidden, items, seed=wseed)
        z = synth.synth_latents(n, latent, seed=zseed) * np.float32(2.0)
        yield i, tensors, z, g[f"c{i}_out"]


def test_oracle_matches_reference_goldens():
    from oracle.vae_decode_ref import decode
    for i, tensors, z, want in golden_cases():
        got = decode(z, *tensors)
        assert rel_max(got, want) <= 2e-6 and rel_l2(got, want) <= 2e-6, (i, rel_max(got, want))
        got32 = decode(z, *tensors, dtype=np.float32)
        assert rel_max(got32, want) <= 1e-5, i


@pytest.fixture(scope="module")
def engine():
    from sdrm_amd.engine import Engine
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    e = Engine(8, 8, 4, 0, 16)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [-1, 0, 4])
def test_hip_decode_matches_goldens(engine, tile):
    engine.debug_set(tile=tile)
    try:
        for i, tensors, z, want in golden_cases():
            got = engine.vae_decode(z, *tensors).cpu().numpy()
            assert got.shape == want.shape
            assert rel_max(got, want) <= TOL and rel_l2(got, want) <= TOL, (i, rel_max(got, want), rel_l2(got, want))
    finally:
        engine.debug_set(tile=-1)


# (latent, hidden, items, users): the three BASELINE datasets (README hyper-parameters; N_ITEMS 1008 / 3125 / 8582) at their
# sampled-user counts, ADM cut to 2000 users to keep the fp64 oracle in seconds
SHAPES = [pytest.param(830, 930, 1008, 843, id="ml100k"), pytest.param(340, 600, 3125, 5429, id="ml1m"),
          pytest.param(40, 200, 8582, 2000, id="adm")]


@pytest.mark.gpu
@pytest.mark.parametrize("latent,hidden,items,users", SHAPES)
def test_hip_decode_vs_oracle_and_torch(engine, latent, hidden, items, users):
    from oracle.vae_decode_ref import decode
    tensors = synth.synth_vae_decoder(latent, hidden, items, seed=5)
    z = synth.synth_latents(users, latent, seed=6) * np.float32(1.5)
    got = engine.vae_decode(z, *tensors).cpu().numpy()
    ref = decode(z, *tensors)
    assert rel_max(got, ref) <= TOL and rel_l2(got, ref) <= TOL, (rel_max(got, ref), rel_l2(got, ref))
    # the PyTorch module the hook replaces, on the same device
    from sdrm_amd.train_SDRM import VAE, decoder_tensors
    vae = VAE(items, hidden, latent).cuda().eval()
    with torch.no_grad():
        for p, t in zip((vae.decoder[0].weight, vae.decoder[0].bias, vae.decoder[2].weight, vae.decoder[2].bias), tensors):
            p.copy_(torch.from_numpy(t))
        y = vae.decode(torch.from_numpy(z).cuda()).cpu().numpy()
    assert rel_max(got, y) <= TOL and rel_l2(got, y) <= TOL
    assert decoder_tensors(vae) is not None


@pytest.mark.gpu
@pytest.mark.parametrize("latent,hidden,items,users", SHAPES)
def test_decode_then_equal_sparsity(engine, latent, hidden, items, users):
    """main.py:170-180 on the device: the engine's decode, then sdrm_equal_sparsity on its matrix: threshold and 0/1 matrix ==
    np.quantile / >= on that matrix, bit for bit.  (Round 2's one-call form, whose first select sweep rode on the decode GEMM's
    epilogue, was retired in round 4: it beat the two calls at one of three shapes and had no caller.)"""
    tensors = synth.synth_vae_decoder(latent, hidden, items, seed=7)
    z = synth.synth_latents(users, latent, seed=8)
    raw = engine.vae_decode(z, *tensors)
    M = raw.cpu().numpy()
    for q in (0.937, 0.0634, 0.5):
        bits, thr = engine.equal_sparsity(raw, q, return_threshold=True)
        want_thr = np.quantile(M.flatten(), q)
        assert want_thr.dtype == np.float32
        assert np.float32(thr.cpu().numpy()).tobytes() == want_thr.tobytes(), (q, float(thr.cpu()), float(want_thr))
        np.testing.assert_array_equal(bits.cpu().numpy(), (M >= want_thr).astype(np.uint8))


@pytest.mark.gpu
def test_sample_ddpm_decodes_on_the_engine(engine):
    """The drop-in `sample_ddpm` hands the reference's decoder to the engine and any other decode hook to its module."""
    from sdrm_amd import train_SDRM as ts

    class OtherHook(torch.nn.Module):
        def __init__(self, vae):
            super().__init__()
            self.vae = vae
            self.calls = 0

        def decode(self, z):
            self.calls += 1
            return self.vae.decode(z)
    L, T, H, items, hidden, n = 48, 9, 1, 130, 70, 25
    net = ts.SDRM(L, T, L, H).to("cuda")               # (N_ITEMS of the eps-net = the VAE latent width)
    vae = ts.VAE(items, hidden, L).cuda().eval()
    torch.manual_seed(3)
    a = ts.sample_ddpm(n, net, vae, L, 1.0, n_timesteps=T)
    assert tuple(a.shape) == (n, items)
    hook = OtherHook(vae)
    net._calls -= 1                                   # replay the same sampling call (same Philox call id)
    b = ts.sample_ddpm(n, net, hook, L, 1.0, n_timesteps=T)
    assert hook.calls == 1
    assert rel_max(a.cpu().numpy(), b.cpu().numpy()) <= TOL
