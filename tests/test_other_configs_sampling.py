"""Tensor-level oracle parity of the SAMPLER at the real row counts of the other two BASELINE.json configurations
("parity size gaps"; the ML-1M one is tests/test_headline_config.py):

  * ADM / NeuMF: L = W = 40, T = 93, H = 5, n = 9558 sampled users - the persistent narrow-net sampler (csrc/skinny.h) at
    598 work-groups, more than one round of the chip;
  * ML-100k / SVD: L = W = 830, T = 83, H = 2, n = 843 - the 32x32x32 tile on the 16-wide MFMA at K = 832 with the reverse
    update in the out-layer epilogue (EPI_TANH_REV), stand-alone and forced onto the 64x64 tile as well.

Full-resolution (/root/reference/train_SDRM.py:50-61) and multi-resolution (:37-49), injected randoms (EXPLICIT) and the
on-device generator (PHILOX) replayed through oracle/philox_ref.py, start steps T_j bit-exact.  Needs a real MI355X."""
import numpy as np
import pytest

from sdrm_amd import synth
from test_hip_parity import close, rel_l2, rel_max

pytestmark = pytest.mark.gpu

ADM = (40, 40, 93, 5, 9558)
ML100K = (830, 830, 83, 2, 843)


def _case(dims, nd, seeds):
    """Explicit and Philox randoms of one configuration and the oracle's latents for both branches of each."""
    from oracle import philox_ref as pr
    from oracle import sdrm_oracle as orc
    L, W, T, H, n = dims
    init = synth.init_params(L, W, T, H, seed=seeds[0])
    o = orc.Oracle(L, W, T, H, init)
    case = {"init": init}
    xT, z, keep, Tj = synth.synth_sample_randoms(n, L, T, nd, seed=seeds[1], multires=True)
    case["explicit"] = dict(xT=xT, z=z, keep=keep, Tj=Tj, full=o.sample(xT, z, keep).numpy(), multi=o.sample(xT, z, keep, Tj).numpy())
    seed, call_id, row0 = seeds[2], 3, 1000
    xT, z, keep, Tj = pr.sample_randoms(seed, call_id, row0, n, L, T, nd, True)
    case["philox"] = dict(seed=seed, call_id=call_id, row0=row0, Tj=Tj, full=o.sample(xT, z, keep).numpy(),
                          multi=o.sample(xT, z, keep, Tj).numpy())
    return case


@pytest.fixture(scope="module")
def adm_case():
    return _case(ADM, 0.9, (31, 32, 77))


@pytest.fixture(scope="module")
def ml100k_case():
    return _case(ML100K, 0.9, (41, 42, 78))


def _run(engine_cls, dims, nd, case, rng, multires, **debug):
    L, W, T, H, n = dims
    e = engine_cls(L, W, T, H, n).debug_set(**debug)
    e.set_params(synth.flatten_params(case["init"], H))
    c = case[rng]
    if rng == "explicit":
        out = e.sample(n, nd=nd, multires=multires, xT=c["xT"], z=c["z"], keep=c["keep"], Tj=c["Tj"] if multires else None)
    else:
        res = e.sample(n, nd=nd, multires=multires, seed=c["seed"], call_id=c["call_id"], row0=c["row0"], return_Tj=multires)
        if multires:
            out, tj = res
            np.testing.assert_array_equal(tj.cpu().numpy(), c["Tj"])
        else:
            out = res
    ref = c["multi" if multires else "full"]
    got = out.cpu().numpy()
    e.close()
    assert close(got, ref), (rel_max(got, ref), rel_l2(got, ref))


@pytest.mark.parametrize("rng", ["explicit", "philox"])
@pytest.mark.parametrize("multires", [False, True])
def test_adm_sampling_fullsize_vs_oracle(engine_cls, adm_case, multires, rng):
    """(40, 40, 93, 5), n = 9558: the whole reverse loop in one persistent launch of 598 work-groups."""
    _run(engine_cls, ADM, 0.9, adm_case, rng, multires)


@pytest.mark.parametrize("rng", ["explicit", "philox"])
@pytest.mark.parametrize("multires", [False, True])
def test_adm_sampling_fullsize_per_layer_path(engine_cls, adm_case, multires, rng):
    """The same configuration with the narrow-net kernels off: the general per-layer GEMM path at 9558 rows."""
    _run(engine_cls, ADM, 0.9, adm_case, rng, multires, skinny=False)


@pytest.mark.parametrize("fused", [0, 1, 2])
@pytest.mark.parametrize("tile", [-1, 0, 4])
@pytest.mark.parametrize("rng", ["explicit", "philox"])
@pytest.mark.parametrize("multires", [False, True])
def test_ml100k_sampling_vs_oracle(engine_cls, ml100k_case, multires, rng, tile, fused):
    """(830, 830, 83, 2), n = 843 on every tile and every placement of the reverse update."""
    if rng == "explicit" and fused != 1:
        pytest.skip("the fused reverse update exists for PHILOX sampling only (round 5: multi-resolution calls included)")
    _run(engine_cls, ML100K, 0.9, ml100k_case, rng, multires, tile=tile, fused_reverse=fused)
