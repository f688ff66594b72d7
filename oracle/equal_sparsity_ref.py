"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the equal-sparsity binarisation the
reference applies to sampled data,

    threshold = np.quantile(M.flatten(), SPARSITY)          # main.py:177, :179, :184
    M_equal_sparsity = (M >= threshold).astype(int)         # main.py:178, :180, :185

The arithmetic lives in numpy (third party; the container and the GPU box carry numpy 2.2.6, the reference pins
numpy~=1.19.5 in requirements.txt:1).  For a float32 array and a Python-float q, numpy 2.x computes in float32
(numpy/lib/_function_base_impl.py: `quantile` casts q to a.dtype; method "linear" uses virtual = (n-1)*q,
previous = floor(virtual), gamma = virtual - previous, `_lerp(a, b, t)` = a + (b-a)*t, or b - (b-a)*(1-t) when
t >= 0.5).  This file restates exactly that with explicit float32 steps and two order statistics, so that the HIP
radix-select (csrc/select.h) has a checker that does not depend on how a numpy build partitions; the tests pin it
against np.quantile itself on every case."""
import numpy as np


def quantile_ranks(n: int, q: float):
    """(previous index, next index, gamma) of np.quantile(a, q) for a float32 `a` of n elements."""
    f = np.float32
    q32 = f(q)
    nm1 = f(n - 1)
    virt = f(nm1 * q32)
    if virt >= nm1:
        return n - 1, n - 1, f(0)
    if virt < 0:
        return 0, 0, f(0)
    prev = np.floor(virt)
    return int(prev), min(int(prev) + 1, n - 1), f(virt - prev)


def quantile_f32(a: np.ndarray, q: float) -> np.float32:
    a = np.asarray(a, dtype=np.float32).ravel()
    r0, r1, g = quantile_ranks(a.size, q)
    part = np.partition(a, sorted({r0, r1}))
    lo, hi = part[r0], part[r1]
    f = np.float32
    diff = f(hi - lo)
    t = f(lo + f(diff * g))
    if g >= f(0.5):
        t = f(hi - f(diff * f(f(1) - g)))
    return t


def equal_sparsity(raw: np.ndarray, sparsity: float) -> np.ndarray:
    raw = np.asarray(raw, dtype=np.float32)
    return (raw >= quantile_f32(raw, sparsity)).astype(np.uint8)
