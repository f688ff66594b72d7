"""Statistics of the on-device generator (csrc/philox.h: Philox4x32-10 + Box-Muller on 24-bit uniforms through the hardware's
log2 / sin / cos).  The integer draws are pinned bit for bit against oracle/philox_ref.py elsewhere; the normals are what
the reference takes from torch.randn (train_SDRM.py:326, :38, :56) and what the end-to-end quality rests on, so they are
checked here against the distribution itself: moments, tails, a Kolmogorov-Smirnov distance, independence along every axis
of the counter (row, column quad, step / purpose, the four outputs of one call) and from the dropout keep bits that ride on
the same Philox words.  Needs a GPU: `pytest -m gpu`."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROWS, QUADS = 5000, 512          # 1.024e7 normals per draw


@pytest.fixture(scope="module")
def eng():
    from sdrm_amd.engine import Engine
    e = Engine(40, 40, 9, 1, max_rows=64)
    yield e
    e.close()


def corr(a, b):
    a = a.double().flatten() - a.double().mean()
    b = b.double().flatten() - b.double().mean()
    return float((a * b).mean() / (a.std(unbiased=False) * b.std(unbiased=False)))


@pytest.mark.parametrize("purpose", [1, 3, 4 | (7 << 8)])
def test_normals_have_the_right_distribution(eng, purpose):
    z, _ = eng.philox_draws(0x5EED1234ABCD, purpose, step=3, rows=ROWS, quads=QUADS)
    n = z.numel()
    x = z.double()
    assert bool(torch.isfinite(z).all())
    mean, var = float(x.mean()), float(x.var())
    skew = float(((x - mean) ** 3).mean() / var ** 1.5)
    kurt = float(((x - mean) ** 4).mean() / var ** 2)
    # standard errors at n = 1e7: mean 3.2e-4, variance 4.5e-4, skewness 7.7e-4, kurtosis 1.5e-3 -> five sigma each
    assert abs(mean) < 1.6e-3 and abs(var - 1) < 2.3e-3 and abs(skew) < 3.9e-3 and abs(kurt - 3) < 7.7e-3, (mean, var, skew, kurt)
    # tails: P(|z| > 3) = 2.6998e-3, P(|z| > 4) = 6.334e-5; the 24-bit uniform caps |z| at sqrt(2 ln 2^24) = 5.77
    p3, p4 = float((x.abs() > 3).double().mean()), float((x.abs() > 4).double().mean())
    assert abs(p3 - 2.6998e-3) < 5 * (2.6998e-3 / n) ** 0.5 and abs(p4 - 6.334e-5) < 5 * (6.334e-5 / n) ** 0.5, (p3, p4)
    assert float(x.abs().max()) <= 5.78
    # Kolmogorov-Smirnov distance on a 2e6 subsample against the normal CDF (critical value at alpha = 1e-3: 1.95 / sqrt(n))
    sub = torch.sort(z.flatten()[:: n // 2_000_000][:2_000_000].double()).values
    cdf = 0.5 * (1 + torch.erf(sub / 2 ** 0.5))
    k = sub.numel()
    grid = torch.arange(1, k + 1, device=sub.device, dtype=torch.float64) / k
    d = float(torch.maximum((grid - cdf).abs(), (grid - 1.0 / k - cdf).abs()).max())
    assert d < 1.95 / k ** 0.5, d


def test_normals_are_independent_along_the_counter(eng):
    seed = 0x0123456789AB
    z, bits = eng.philox_draws(seed, 1, step=11, rows=ROWS, quads=QUADS)
    n = z.numel()
    tol = 5 / n ** 0.5        # five sigma of a sample correlation of independent draws
    zq = z.view(ROWS, QUADS, 4)
    # the four outputs of one call (two Box-Muller pairs: cos / sin of one angle, and the second pair)
    for i in range(4):
        for j in range(i + 1, 4):
            assert abs(corr(zq[:, :, i], zq[:, :, j])) < 2 * tol, (i, j)
            assert abs(corr(zq[:, :, i] ** 2, zq[:, :, j] ** 2)) < 2 * tol, (i, j)      # ... and their magnitudes
    # neighbouring column quads, neighbouring rows (lag 1 and lag 7)
    for lag in (1, 7):
        assert abs(corr(zq[:, :-lag], zq[:, lag:])) < tol and abs(corr(zq[:-lag], zq[lag:])) < tol, lag
    # the next step, another purpose, the next seed: the same counters otherwise
    for other in (eng.philox_draws(seed, 1, step=12, rows=ROWS, quads=QUADS, with_bits=False)[0],
                  eng.philox_draws(seed, 3, step=11, rows=ROWS, quads=QUADS, with_bits=False)[0],
                  eng.philox_draws(seed + 1, 1, step=11, rows=ROWS, quads=QUADS, with_bits=False)[0]):
        assert abs(corr(z, other)) < tol and abs(corr(z ** 2, other ** 2)) < tol
    # the dropout keep bits ride on the low bits of the words whose upper 24 bits make the normals: fair, mutually independent,
    # and independent of the normal made from the same word and of its magnitude
    for b in range(3):
        kb = ((bits >> b) & 1).float()
        assert abs(float(kb.double().mean()) - 0.5) < 5 * 0.5 / n ** 0.5
        assert abs(corr(kb, z)) < tol and abs(corr(kb, z.abs())) < tol, b
        for b2 in range(b + 1, 3):
            assert abs(corr(kb, ((bits >> b2) & 1).float())) < tol, (b, b2)
