"""the sparse batch feed on the device (dataloaders.py:46-79 + `.to_dense()`, train_SDRM.py:323).
Bar: the dense batch is identical to scipy's `csr[rows].toarray()` (a copy of stored values: bit-exact)."""
import numpy as np
import pytest
import torch
from scipy.sparse import csr_matrix

from sdrm_amd import synth


@pytest.fixture(scope="module")
def engine():
    from sdrm_amd.engine import Engine
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    e = Engine(8, 8, 4, 0, 16)
    yield e
    e.close()


def test_interaction_generator_is_disjoint_and_has_edge_users():
    train, held = synth.synth_interactions(30, 200, seed=3)
    assert train.multiply(held).nnz == 0
    assert held[0].nnz == 0 and held[1].nnz > 50 and train[2].nnz == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows,n_items", [(938, 1008), (300, 8582), (50, 3), (7, 1)])   # 8582*4 is not a multiple of 16
@pytest.mark.parametrize("weighted", [False, True])
def test_rows_to_dense_matches_scipy(engine, n_rows, n_items, weighted):
    rs = np.random.RandomState(n_rows + n_items)
    dense = (rs.random_sample((n_rows, n_items)) < 0.06).astype(np.float32)
    if weighted:
        dense *= rs.random_sample((n_rows, n_items)).astype(np.float32) + 0.5
    dense[n_rows // 2] = 0          # an empty row
    m = csr_matrix(dense)
    dev = engine.csr_to_device(m)
    assert (dev[2] is None) == (not weighted)
    rows = rs.permutation(n_rows)[: max(1, n_rows // 3)].astype(np.int64)
    got = engine.csr_rows_to_dense(dev, rows=torch.from_numpy(rows)).cpu().numpy()
    np.testing.assert_array_equal(got, m[rows].toarray().astype(np.float32))
    lo, b = n_rows // 4, max(1, n_rows // 2)
    got = engine.csr_rows_to_dense(dev, row0=lo, b=b).cpu().numpy()
    np.testing.assert_array_equal(got, dense[lo:lo + b])


@pytest.mark.gpu
def test_device_feed_yields_what_the_host_feed_yields(engine):
    from sdrm_amd.pipeline import DeviceFeed, EpochFeed
    train, _ = synth.synth_interactions(101, 257, seed=4, p_train=0.1)
    host, dev = EpochFeed(train, 32, seed=11, device="cuda"), DeviceFeed(train, 32, engine, seed=11)
    for epoch in range(2):                       # a fresh permutation per epoch on both sides
        hb, db = list(host), list(dev)
        assert len(hb) == len(db) == 4 and db[-1][0].shape[0] == 101 - 3 * 32
        for (hx, _), (dx, _) in zip(hb, db):
            assert torch.equal(hx.to_dense(), dx.to_dense())


@pytest.mark.gpu
def test_feed_argument_errors(engine):
    from sdrm_amd.engine import SdrmError
    dev = engine.csr_to_device(csr_matrix(np.eye(5, dtype=np.float32)))
    with pytest.raises(SdrmError):
        engine.csr_rows_to_dense(dev, rows=torch.tensor([0, 5]))
    with pytest.raises(SdrmError):
        engine.csr_rows_to_dense(dev, row0=3, b=3)
