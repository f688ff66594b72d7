"""CPU oracle of the VAE decode hook - TEST INFRASTRUCTURE, NOT PRODUCT.

Restates /root/reference/train_SDRM.py:212-214 (`decoder = Sequential(Linear(latent, hidden), Tanh(), Linear(hidden,
input_dim))`) and :252-254 (`decode(z) = decoder(z)`) in numpy.  nn.Linear is y = x W^T + b with W [out, in].

Pinned by tests/golden/vae_decode.npz (made by tests/golden/make_golden.py from the reference's own VAE class with
injected weights).  Only tests/ may import this module."""
import numpy as np


def decode(z, w1, b1, w2, b2, dtype=np.float64):
    """decoder(z) in `dtype` arithmetic (float64 by default: the parity tests bound both fp32 implementations against it)."""
    z, w1, b1, w2, b2 = (np.asarray(a, dtype=dtype) for a in (z, w1, b1, w2, b2))
    return np.tanh(z @ w1.T + b1) @ w2.T + b2
