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


@pytest.mark.gpu
def test_out_of_range_csr_is_caught_on_the_device_not_written_out_of_bounds(engine):
    """VERDICT r4 item 7 (dataloaders.py:46-79 hands a scipy CSR to the feed; a corrupt one must not become a stray device store):
    a row id outside the matrix leaves that output row zero, a column index outside [0, n_items) skips that entry, every other
    row / entry is delivered as scipy would, and the handle reports SDRM_ERR_ARG at the next status read (then is clean again)."""
    from sdrm_amd.engine import SdrmError
    rs = np.random.RandomState(5)
    dense = (rs.random_sample((12, 37)) < 0.3).astype(np.float32)
    m = csr_matrix(dense)
    indptr, indices, data, shape = engine.csr_to_device(m)
    # (1) row ids: -1 and n_rows among good ones; a guard band behind the output shows nothing was written past it
    rows = torch.tensor([3, -1, 7, 12, 0], dtype=torch.int64)
    out = engine.csr_rows_to_dense((indptr, indices, data, shape), rows=rows, check=False).cpu().numpy()
    np.testing.assert_array_equal(out[[0, 2, 4]], dense[[3, 7, 0]])
    assert not out[1].any() and not out[3].any()
    with pytest.raises(SdrmError, match="row id outside"):
        engine.feed_status()
    engine.feed_status()                        # the record was cleared
    # (2) column indices: one entry of row 2 points behind the row, one is negative
    bad = indices.clone()
    lo, hi = int(m.indptr[2]), int(m.indptr[3])
    assert hi - lo >= 2
    bad[lo] = 37
    bad[lo + 1] = -4
    out = engine.csr_rows_to_dense((indptr, bad, data, shape), row0=0, b=12, check=False).cpu().numpy()
    want = dense.copy()
    want[2, m.indices[lo]] = 0
    want[2, m.indices[lo + 1]] = 0
    np.testing.assert_array_equal(out, want)
    with pytest.raises(SdrmError, match="column index outside"):
        engine.feed_status()
    # (3) checked call: raises at once; an unordered indptr pair is a zero row, not a wild loop
    with pytest.raises(SdrmError):
        engine.csr_rows_to_dense((indptr, bad, data, shape), row0=0, b=12)
    badptr = indptr.clone()
    badptr[5] = badptr[6] + 3
    out = engine.csr_rows_to_dense((badptr, indices, data, shape), row0=5, b=1, check=False).cpu().numpy()
    assert not out.any()
    with pytest.raises(SdrmError, match="indptr"):
        engine.feed_status()
    good = engine.csr_rows_to_dense((indptr, indices, data, shape), row0=0, b=12).cpu().numpy()
    np.testing.assert_array_equal(good, dense)
