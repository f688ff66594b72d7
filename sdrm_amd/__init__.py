"""sdrm_amd — MI355X-native denoising engine for SDRM (hot path of train_SDRM.py).

Importing the package is cheap (no torch, no GPU, no library load).  The HIP
library is loaded by `sdrm_amd._lib.load()` the first time an engine is built.
"""
__version__ = "0.1.0"
