"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the ranking metrics the reference scores
generated data with (utilities.py:116-171, chained as in svd_benchmark.py:58-66):

    mask_training_examples   :116-120   already-seen items -> -inf
    recall_at_k_batch        :154-171   |top-k  ∩ held-out| / min(k, |held-out|)
    NDCG_binary_at_k_batch   :123-151   sum over the held-out items in the top-k of 1/log2(rank+2), over the ideal sum

restated through the RANK of each held-out item (number of better-scored items; ties towards the lower index)
instead of a partition of the whole row - the formulation csrc/rank.h uses.  Plain Python loops: small cases only.
Pinned by tests/golden/rank_metrics.npz, which the reference's own functions produced."""
import numpy as np


def np_pairwise_sum(a):
    """numpy's pairwise add-reduce of a contiguous float64 row of length <= 128."""
    n = len(a)
    if n < 8:
        res = 0.0
        for x in a:
            res = res + x
        return res
    r = [a[j] for j in range(8)]
    i = 8
    while i < n - (n % 8):
        for j in range(8):
            r[j] = r[j] + a[i + j]
        i += 8
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < n:
        res = res + a[i]
        i += 1
    return res


def rank_metrics(scores, heldout, train=None, ks=(1, 3, 5, 10, 20, 50)):
    """(recall[nk,U], ndcg[nk,U]) float64; scores [U,I] float32, heldout/train scipy CSR."""
    scores = np.array(scores, dtype=np.float32, copy=True)
    U, I = scores.shape
    heldout = heldout.tocsr()
    if train is not None:
        train = train.tocsr()
        for u in range(U):
            scores[u, train.indices[train.indptr[u]:train.indptr[u + 1]]] = -np.inf        # utilities.py:118-119
    kmax = max(ks)
    tp = 1.0 / np.log2(np.arange(2, kmax + 2))                                              # :145
    recall = np.empty((len(ks), U)); ndcg = np.empty((len(ks), U))
    with np.errstate(all="ignore"):
        for u in range(U):
            row = scores[u]
            items = heldout.indices[heldout.indptr[u]:heldout.indptr[u + 1]]
            hit = np.zeros(kmax, dtype=bool)
            for j in items:
                rank = int(np.sum(row > row[j]) + np.sum(row[:j] == row[j]))
                if rank < kmax:
                    hit[rank] = True
            for q, k in enumerate(ks):
                m = min(k, len(items))
                recall[q, u] = np.float64(np.float32(hit[:k].sum())) / np.float64(m) if m else np.nan   # :167-169
                dcg = np_pairwise_sum([tp[r] if hit[r] else 0.0 for r in range(k)])                        # :147-148
                idcg = tp[:m].sum()                                                                        # :149-150
                ndcg[q, u] = dcg / idcg if m else np.nan
    return recall, ndcg
