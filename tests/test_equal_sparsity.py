"""equal-sparsity binarisation of sampled data (main.py:177-180).

CPU: the restatement in oracle/equal_sparsity_ref.py against the golden vectors (numpy's own np.quantile at the
reference's call expression) and against np.quantile on seeded cases.  GPU (-m gpu): sdrm_equal_sparsity through the
C ABI against the same golden vectors, the oracle, and np.quantile at the BASELINE shapes.  Bar: the threshold is the
same float32 bit pattern and the binarised matrix is identical (byte work: bit-exact)."""
import os

import numpy as np
import pytest
import torch

from sdrm_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "equal_sparsity.npz")


def golden_cases():
    g = np.load(GOLD)
    for i in range(int(g["n_cases"])):
        users, items, seed = (int(v) for v in g[f"c{i}_shape"])
        M = synth.synth_scores(users, items, seed=seed, kind=str(g[f"c{i}_kind"]))
        bits = np.unpackbits(g[f"c{i}_bits"])[:users * items].reshape(users, items)
        yield i, M, float(g[f"c{i}_sparsity"]), np.float32(g[f"c{i}_threshold"]), bits, int(g[f"c{i}_ones"])


def test_oracle_matches_golden():
    from oracle.equal_sparsity_ref import equal_sparsity, quantile_f32
    for i, M, q, thr, bits, ones in golden_cases():
        t = quantile_f32(M, q)
        assert t.dtype == np.float32 and t.tobytes() == thr.tobytes(), (i, t, thr)
        eq = equal_sparsity(M, q)
        assert int(eq.sum()) == ones
        np.testing.assert_array_equal(eq, bits)


@pytest.mark.parametrize("n", [1, 2, 5, 1000, 65537, 300001])
@pytest.mark.parametrize("kind", ["normal", "ties", "narrow"])
def test_oracle_matches_numpy(n, kind):
    from oracle.equal_sparsity_ref import quantile_f32
    M = synth.synth_scores(1, n, seed=n, kind=kind)
    for q in (0.0, 1.0, 0.5, 0.937, 0.0634, 0.999999, 1e-7):
        want = np.quantile(M.flatten(), q)
        got = quantile_f32(M, q)
        assert want.dtype == np.float32 and got.tobytes() == want.tobytes(), (n, kind, q, got, want)


@pytest.fixture(scope="module")
def engine():
    from sdrm_amd.engine import Engine
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    e = Engine(8, 8, 4, 0, 16)
    yield e
    e.close()


@pytest.mark.gpu
def test_hip_matches_golden(engine):
    for i, M, q, thr, bits, ones in golden_cases():
        out, t = engine.equal_sparsity(torch.from_numpy(M).cuda(), q, return_threshold=True)
        assert np.float32(t.item()).tobytes() == thr.tobytes(), (i, t.item(), thr)
        np.testing.assert_array_equal(out.cpu().numpy(), bits)
        assert int(out.sum().item()) == ones


@pytest.mark.gpu
@pytest.mark.parametrize("shape,kind,q", [
    ((843, 1008), "normal", 0.937),        # ML-100k  (users x items)
    ((5429, 3125), "normal", 0.9553),      # ML-1M
    ((5429, 3125), "ties", 0.9553),
    ((9558, 8582), "normal", 0.9877),      # ADM: 82 M elements
    ((1, 1), "normal", 0.3), ((7, 3), "ties", 0.5), ((1000, 1001), "narrow", 0.25),
    ((3, 7), "normal", 0.0), ((3, 7), "normal", 1.0), ((4099, 4099), "narrow", 0.999999),
])
def test_hip_matches_numpy(engine, shape, kind, q):
    M = synth.synth_scores(shape[0], shape[1], seed=7, kind=kind)
    want_t = np.quantile(M.flatten(), q)                       # the reference's expression, main.py:177
    want = (M >= want_t)
    out, t = engine.equal_sparsity(torch.from_numpy(M).cuda(), q, return_threshold=True)
    assert np.float32(t.item()).tobytes() == np.float32(want_t).tobytes(), (t.item(), want_t)
    got = out.cpu().numpy()
    assert got.dtype == np.uint8 and got.shape == M.shape
    assert np.array_equal(got.astype(bool), want)


@pytest.mark.gpu
def test_hip_negative_zero_and_unaligned_tail(engine):
    x = np.asarray([0.0, -0.0, 1.0, -1.0, 0.0, -0.0, 2.0], dtype=np.float32)   # n % 4 != 0; +-0 compare equal
    for q in (0.0, 0.2, 0.5, 0.8, 1.0):
        want_t = np.quantile(x, q)
        out, t = engine.equal_sparsity(torch.from_numpy(x).cuda(), q, return_threshold=True)
        assert float(t.item()) == float(want_t)
        np.testing.assert_array_equal(out.cpu().numpy().astype(bool), x >= want_t)


@pytest.mark.gpu
def test_hip_argument_errors(engine):
    from sdrm_amd.engine import SdrmError
    x = torch.zeros(8, device="cuda")
    with pytest.raises(SdrmError):
        engine.equal_sparsity(x, 1.5)
    with pytest.raises(SdrmError):
        engine.equal_sparsity(x[1:], 0.5)          # not 16-byte aligned
