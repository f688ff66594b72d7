"""Ranking metrics and the per-user hold-out split used around the denoising path.

Own numpy restatement of the helpers in /root/reference/utilities.py that the VAE pre-stage and the
Recall@10 parity harness need: `recall_at_k_batch` (:149-171),
`NDCG_binary_at_k_batch` (:123-146), `mask_training_examples` (:116-120) and
`split_train_test_proportion_from_csr_matrix` (:174-236).  `bottleneck` is not required
(np.argpartition gives the same top-k set)."""
from __future__ import annotations

import math

import numpy as np
from scipy.sparse import csr_matrix, vstack


def mask_training_examples(sparse_training_set, dense_matrix):
    """Scores of already-seen items -> -inf so they can never be ranked (in place, returned)."""
    dense_matrix[sparse_training_set.nonzero()] = -np.inf
    return dense_matrix


def _topk_indices(scores: np.ndarray, k: int) -> np.ndarray:
    return np.argpartition(-scores, k, axis=1)[:, :k]


def recall_at_k_batch(X_pred, heldout_batch, k=100):
    """hits in the top-k / min(k, number of held-out items), per user."""
    users = X_pred.shape[0]
    top = _topk_indices(X_pred, k)
    picked = np.zeros_like(X_pred, dtype=bool)
    picked[np.arange(users)[:, None], top] = True
    truth = (heldout_batch > 0) if isinstance(heldout_batch, np.ndarray) else (heldout_batch > 0).toarray()
    hits = np.logical_and(truth, picked).sum(axis=1).astype(np.float32)
    return hits / np.minimum(k, truth.sum(axis=1))


def NDCG_binary_at_k_batch(X_pred, heldout_batch, k=100):
    """Binary-relevance NDCG@k (every zero of heldout_batch is irrelevant)."""
    users = X_pred.shape[0]
    rows = np.arange(users)[:, None]
    part = _topk_indices(X_pred, k)
    order = np.argsort(-X_pred[rows, part], axis=1)
    topk = part[rows, order]
    discount = 1.0 / np.log2(np.arange(2, k + 2))
    dcg = (heldout_batch[rows, topk].toarray() * discount).sum(axis=1)
    idcg = np.array([discount[:min(int(n), k)].sum() for n in heldout_batch.getnnz(axis=1)])
    return dcg / idcg


def split_train_test_proportion_from_csr_matrix(csr_data, test_prop=0.2, batch_size=None, random_seed=None,
                                                ignore_zeros=False):
    """Per user, move ceil(test_prop * n_items) random items to the test matrix.  Users with fewer than
    two items are dropped.  Draws with the global numpy generator, one `np.random.choice` per kept user
    in row order, so a given `random_seed` reproduces the reference's split."""
    if random_seed:
        np.random.seed(random_seed)
    if type(csr_data) is not csr_matrix:
        raise TypeError("Input data is not of type csr_matrix")
    if ignore_zeros:
        csr_data.eliminate_zeros()
    n_cols = csr_data.shape[1]
    indptr, indices = csr_data.indptr, csr_data.indices
    tr_rows, tr_cols, te_rows, te_cols = [], [], [], []
    kept = 0
    for u in range(csr_data.shape[0]):
        items = indices[indptr[u]:indptr[u + 1]]
        n_items = items.shape[0]
        if n_items < 2:
            print(f"Warning: skipping user with {n_items} items rated")
            continue
        held = np.zeros(n_items, dtype=bool)
        held[np.random.choice(n_items, size=math.ceil(test_prop * n_items), replace=False).astype("int32")] = True
        tr_cols.append(items[~held]); tr_rows.append(np.full(int((~held).sum()), kept))
        te_cols.append(items[held]); te_rows.append(np.full(int(held.sum()), kept))
        kept += 1

    def build(rows, cols):
        if not rows:
            return csr_matrix((0, n_cols))
        r, c = np.concatenate(rows), np.concatenate(cols)
        m = csr_matrix((np.ones(r.shape[0]), (r, c)), shape=(kept, n_cols))
        m.data[:] = 1.0  # duplicates collapse to 1, as np.put(v=1) does
        return m

    return build(tr_rows, tr_cols), build(te_rows, te_cols)
