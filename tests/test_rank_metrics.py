"""Recall@k / NDCG@k of generated-data scores (utilities.py:116-171).

CPU: oracle/rank_metrics_ref.py against the golden vectors the reference's own functions produced.  GPU (-m gpu):
sdrm_rank_metrics through the C ABI against the golden vectors and against the oracle on further seeded cases (ties,
wide rows, no masking).  Bar: float64 results identical bit for bit (nan where the reference has 0/0)."""
import os

import numpy as np
import pytest
import torch

from sdrm_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "rank_metrics.npz")


def golden_cases():
    g = np.load(GOLD)
    ks = tuple(int(k) for k in g["ks"])
    for i in range(int(g["n_cases"])):
        users, items, seed = (int(v) for v in g[f"c{i}_shape"])
        scores = synth.synth_scores(users, items, seed=seed)
        train, held = synth.synth_interactions(users, items, seed=seed)
        yield i, scores, train, held, ks, g[f"c{i}_recall"], g[f"c{i}_ndcg"]


def same(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def test_oracle_matches_golden():
    from oracle.rank_metrics_ref import rank_metrics
    for i, scores, train, held, ks, rec, ndcg in golden_cases():
        r, n = rank_metrics(scores, held, train, ks)
        assert same(r, rec), i
        assert same(n, ndcg), i
        assert np.isnan(rec[:, 0]).all() and not np.isnan(rec[:, 1]).any()      # the edge users are in the fixture


def test_pairwise_sum_is_numpys():
    from oracle.rank_metrics_ref import np_pairwise_sum
    rs = np.random.RandomState(0)
    for n in list(range(1, 129)):
        a = rs.random_sample(n) * rs.choice([1e-3, 1.0, 1e3], n)
        assert np_pairwise_sum(list(a)) == np.tile(a, (2, 1)).sum(axis=1)[0]


@pytest.fixture(scope="module")
def engine():
    from sdrm_amd.engine import Engine
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    e = Engine(8, 8, 4, 0, 16)
    yield e
    e.close()


@pytest.mark.gpu
def test_hip_matches_golden(engine):
    for i, scores, train, held, ks, rec, ndcg in golden_cases():
        r, n = engine.rank_metrics(torch.from_numpy(scores).cuda(), held, train, ks)
        assert same(r.cpu().numpy(), rec), i
        assert same(n.cpu().numpy(), ndcg), i


@pytest.mark.gpu
@pytest.mark.parametrize("users,items,kind,mask,ks", [
    (40, 500, "ties", True, (1, 3, 5, 10, 20, 50)),        # exact duplicates: lower index ranks first on both sides
    (12, 8582, "normal", True, (10, 128)),                 # ADM-wide rows, the largest k
    (30, 200, "normal", False, (1, 2, 7)),                 # no masking
    (5, 16, "narrow", True, (16,)),                        # k = I
])
def test_hip_matches_oracle(engine, users, items, kind, mask, ks):
    from oracle.rank_metrics_ref import rank_metrics
    scores = synth.synth_scores(users, items, seed=21, kind=kind)
    train, held = synth.synth_interactions(users, items, seed=22, p_train=0.1, p_held=0.05)
    want_r, want_n = rank_metrics(scores, held, train if mask else None, ks)
    r, n = engine.rank_metrics(scores, held, train if mask else None, ks)
    assert same(r.cpu().numpy(), want_r)
    assert same(n.cpu().numpy(), want_n)


@pytest.mark.gpu
def test_hip_matches_host_restatement_at_ml100k_scale(engine):
    """Full ML-100k-sized call against sdrm_amd/metrics.py (the numpy restatement pinned by tests/golden/host.npz)."""
    from sdrm_amd import metrics
    users, items = 843, 1008
    scores = synth.synth_scores(users, items, seed=5)
    train, held = synth.synth_interactions(users, items, seed=6, p_train=0.06, p_held=0.015)
    masked = metrics.mask_training_examples(train, scores.copy())
    r, n = engine.rank_metrics(scores, held, train, (1, 3, 5, 10, 20, 50))
    with np.errstate(all="ignore"):
        for q, k in enumerate((1, 3, 5, 10, 20, 50)):
            assert same(r[q].cpu().numpy(), metrics.recall_at_k_batch(masked.copy(), held, k=k)), k
            assert same(n[q].cpu().numpy(), metrics.NDCG_binary_at_k_batch(masked.copy(), held, k=k)), k


@pytest.mark.gpu
def test_hip_argument_errors(engine):
    from sdrm_amd.engine import SdrmError
    scores = synth.synth_scores(4, 10, seed=1)
    train, held = synth.synth_interactions(4, 10, seed=1)
    with pytest.raises(SdrmError):
        engine.rank_metrics(scores, held, train, (11,))            # k > I
    with pytest.raises(SdrmError):
        engine.rank_metrics(scores, held, train, tuple(range(1, 10)))   # more than 8 cut-offs
    with pytest.raises(SdrmError):
        engine.rank_metrics(scores, held[:3], train, (1,))         # shape mismatch
