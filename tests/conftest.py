import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


@pytest.fixture(scope="module")
def engine_cls():
    """The engine class of the GPU tests.  Its debug_set() also takes tile="row": the row-owned train forward
    (csrc/rowchain.h) forced on with the automatic tile for everything else; the test is skipped when the shape lies
    outside that kernel's envelope (L == W, padded width 128..352)."""
    import torch

    from sdrm_amd.engine import Engine
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"

    class TestEngine(Engine):
        def debug_set(self, **kw):
            if kw.get("tile") in ("row48x2", "row48x4"):
                # ... with 2 / 4 work-groups of one XCD per row group (column-split, exchanged through that XCD's L2); batches whose
                # groups x parts do not fit the chip fall back to fewer parts (csrc/sdrm_hip.hip: rows48_parts)
                if not self.rows48_split_available:
                    self.close()
                    pytest.skip("shape outside the row-owned forward's envelope (or no block -> XCD mapping)")
                kw = dict(kw, tile=None, rowchain=0, rows48=2, rows48_split=int(kw["tile"][-1]), wgrad_strips=True, dgrad_rows=1)
            if kw.get("tile") in ("row48", "row48-tiles", "row48-plain"):
                # the row-owned step on 48-row work-groups (csrc/rows48.h) forced on: "row48" with its dgrad chain and the strip-owned
                # weight gradients behind it, "row48-tiles" the same forward with k_loss_seed + tile dgrads + batched tile weight gradients,
                # "row48-plain": "row48" without the shared-tile form of its kernels
                if not self.rowchain_available:
                    self.close()
                    pytest.skip("shape outside the row-owned forward's envelope")
                mode = kw["tile"]
                kw = dict(kw, tile=None, rowchain=0, rows48=2, rows48_split=0, wgrad_strips=mode != "row48-tiles", dgrad_rows=0 if mode == "row48-tiles" else 1,
                          rows48_share=mode != "row48-plain")
            if kw.get("tile") in ("row", "row-layers", "row-tiles"):
                if not self.rowchain_available:
                    self.close()
                    pytest.skip("shape outside the row-owned forward's envelope")
                # "row": the row-owned forward with the row-owned dgrad chain (one launch) and the strip-owned weight gradients
                # behind it (the default pairing); "row-layers": the same with k_loss_seed + one row-owned dgrad launch per layer;
                # "row-tiles": the same forward with the 64x64-tile dgrad and batched split-K launches
                mode = kw["tile"]
                kw = dict(kw, tile=None, rowchain=2, wgrad_strips=mode != "row-tiles", dgrad_rows={"row": 1, "row-layers": 2, "row-tiles": 0}[mode])
            return super().debug_set(**kw)

    return TestEngine
