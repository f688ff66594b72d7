"""The callers either side of the denoising path, restated so the end-to-end metric of the reference
(Recall@k / NDCG@k of a downstream SVD recommender trained on the generated data) can be reproduced on
the GPU box, where the reference itself cannot travel.  SURVEY.md §8f ("next" rows).

    load_split / partial_valid      dataloaders.py:82-116  (train_test + the 80 % part of a seeded per-user split of valid)
    batch_feed                      main.py:126-137        (BatchSampler(RandomSampler) index batches -> sparse COO tensors)
    equal_sparsity                  main.py:177-185        (threshold at the data's sparsity quantile)
    compute_mf_results              svd_benchmark.py:17-70 (TruncatedSVD(20, n_iter=100) reconstruction, masked, Recall/NDCG@k)
    run_experiment                  main.py:143-200        (train -> multi-res + full-res sampling -> evaluate)
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.sparse import csr_matrix, vstack

from . import metrics

K_LIST = (1, 3, 5, 10, 20, 50)


def csr_from_npz(z, tag):
    shape = tuple(int(v) for v in z[tag + "_shape"])
    return csr_matrix((z[tag + "_data"].astype(np.float64), z[tag + "_indices"].astype(np.int32), z[tag + "_indptr"]), shape=shape)


def load_split(npz_path):
    """(TRAIN_DATA, TRAIN_PARTIAL_VALID_DATA, VALID_DATA) as dataloaders.load_data returns them."""
    z = np.load(npz_path)
    train_test, valid = csr_from_npz(z, "train_test"), csr_from_npz(z, "valid")
    val_train, _ = metrics.split_train_test_proportion_from_csr_matrix(valid, batch_size=1000, random_seed=123, test_prop=0.2)
    return train_test, vstack((train_test, val_train)).tocsr(), valid


def batch_feed(data: csr_matrix, batch_size: int, generator: torch.Generator, device="cuda"):
    """One epoch of the reference's feed: a random permutation cut into index batches (last one short),
    each delivered as a sparse COO float tensor [b, N_ITEMS] twice (data, target)."""
    perm = torch.randperm(data.shape[0], generator=generator).numpy()
    for lo in range(0, len(perm), batch_size):
        coo = data[perm[lo:lo + batch_size]].tocoo()
        x = torch.sparse_coo_tensor(np.vstack([coo.row, coo.col]), coo.data.astype(np.float32), coo.shape).to(device)
        yield x, x


class EpochFeed:
    """Re-iterable like a DataLoader: every `iter()` draws a fresh permutation from the carried generator."""

    def __init__(self, data, batch_size, seed=None, device="cuda"):
        self.data, self.batch_size, self.device = data, batch_size, device
        self.gen = torch.Generator(device="cpu")
        if seed is not None:
            self.gen.manual_seed(seed)

    def __iter__(self):
        return batch_feed(self.data, self.batch_size, self.gen, self.device)


class DeviceFeed:
    """The same epoch feed with the CSR matrix resident on the device: every `iter()` draws the
    permutation the reference's RandomSampler would (torch.randperm on the carried CPU generator) and each batch is
    densified in HBM by `sdrm_csr_rows_to_dense` - what `x.to_dense()` (train_SDRM.py:323) would have produced from the
    host-built COO tensor.  Yields dense tensors (`.to_dense()` of a dense tensor is the tensor itself)."""

    def __init__(self, data, batch_size, engine, seed=None):
        self.engine, self.batch_size, self.n_rows = engine, batch_size, data.shape[0]
        self.csr = engine.csr_to_device(data)
        self.gen = torch.Generator(device="cpu")
        if seed is not None:
            self.gen.manual_seed(seed)

    def __iter__(self):
        perm = torch.randperm(self.n_rows, generator=self.gen).to(self.engine.device)
        for lo in range(0, self.n_rows, self.batch_size):
            x = self.engine.csr_rows_to_dense(self.csr, rows=perm[lo:lo + self.batch_size], check=False)
            yield x, x
        self.engine.feed_status()   # the device-side range checks of the epoch's batches, read once (one sync per epoch)


def equal_sparsity(raw, sparsity: float, engine) -> np.ndarray:
    """main.py:177-180, `(raw >= np.quantile(raw.flatten(), SPARSITY)).astype(int)`, on the device
    (`sdrm_equal_sparsity`: radix select of the two order statistics + binarise; csrc/select.h).  `raw` may be a
    device tensor (what `sample_ddpm` returns) or a host array; the 0/1 matrix comes back as the host int array the
    downstream recommenders take."""
    return engine.equal_sparsity(raw, float(sparsity)).cpu().numpy().astype(int)


def compute_mf_results(training_dataset, testing_dataset, synthetic_data, only_synthetic=True):
    """Recall@k and NDCG@k (k = 1,3,5,10,20,50) of a rank-20 truncated SVD fitted on
    [synthetic | 80 % of each test user's items] and scored on the held-out 20 %."""
    from sklearn.decomposition import TruncatedSVD
    test_data, valid_data = metrics.split_train_test_proportion_from_csr_matrix(testing_dataset, batch_size=1000, random_seed=123)
    synthetic = np.asarray(synthetic_data)
    head = synthetic if only_synthetic else training_dataset.toarray()
    training = np.concatenate([head, test_data.toarray()], axis=0)
    combined = training if only_synthetic else np.concatenate([training, synthetic], axis=0)
    svd = TruncatedSVD(n_components=20, n_iter=100)
    recon = svd.inverse_transform(svd.fit_transform(combined))
    masked = metrics.mask_training_examples(sparse_training_set=training, dense_matrix=recon[:training.shape[0]].copy())
    lo = head.shape[0]
    scored = masked[lo:lo + valid_data.shape[0]]
    rec = [np.round(np.nanmean(metrics.recall_at_k_batch(scored, valid_data, k=k)), 4) for k in K_LIST]
    ndcg = [np.round(np.nanmean(metrics.NDCG_binary_at_k_batch(scored, valid_data, k=k)), 4) for k in K_LIST]
    return np.array(rec), np.array(ndcg)


def run_experiment(split, hp, seed, vae_dir, verbose=False):
    """One run of main.py's loop body with the MI355X engine; returns {"M": (recall, ndcg), "F": ..., "V": ...}."""
    from . import train_SDRM as ts
    train, train_partial, valid = split
    n_users, n_items = train.shape
    sparsity = 1 - train.nnz / (n_users * n_items)
    torch.manual_seed(seed)
    np.random.seed(seed)
    from .engine import utility_engine
    dl = DeviceFeed(train_partial, hp["batch"], utility_engine())      # CSR resident in HBM, batches densified there
    net, vae = ts.train_SDRM(dl, N_ITEMS=n_items, VAE_HIDDEN=hp["vae_hidden"], VAE_LATENT=hp["latent"], VAE_BATCH_SIZE=hp["vae_batch"],
                             VAE_LR=hp["vae_lr"], DIFF_LATENT=hp["latent"], N_HIDDEN_MLP_LAYERS=hp["H"], DIFF_LR=hp["lr"],
                             DIFF_TRAINING_EPOCHS=hp["epochs"], TIMESTEPS=hp["T"], noise_divider=hp["nd"], VAE_DIR_PATH=vae_dir,
                             TRAIN_PARTIAL_VALID_DATA=train_partial, VALID_DATA=valid, OPTIMIZATION_OBJECTIVE="Recall@10",
                             verbose=verbose, cache_latents=hp.get("cache_latents", False))
    out = {}
    M = ts.sample_ddpm(n_users, net, vae, hp["latent"], hp["nd"], timesteps="random", n_timesteps=hp["T"]).detach()
    F = ts.sample_ddpm(n_users, net, vae, hp["latent"], hp["nd"], n_timesteps=hp["T"]).detach()
    V = vae.sample(n_users)
    for tag, raw in (("M", M), ("F", F), ("V", V)):
        out[tag] = compute_mf_results(train, valid, equal_sparsity(raw, sparsity, net.engine()), only_synthetic=True)
    return out
