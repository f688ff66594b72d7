"""The PyTorch side of the reference's pipeline that the denoising engine sits between: the MultiVAE++ model, its pre-training
and its checkpoint helpers (/root/reference/train_SDRM.py:66-83, :115-188, :206-268).

NOT part of the hot path and not engine code (SURVEY.md section 2 marks the VAE "HOOK - kept in PyTorch" and its pre-training
out of scope): this module exists so that `sdrm_amd.train_SDRM.train_SDRM()` runs end to end when the caller brings no VAE of its
own, and so that the names the reference's module exports (`VAE`, `train_variational_autoencoder`, `checkpoint`, `resume`) stay
importable from the drop-in module, which re-exports them.  A caller that already has a trained VAE - any object with
`encode(x) -> (z, kl)`, `decode(z)`, `eval()`, `parameters()` and `model_is_trained` - passes it as `variational_ae=` and nothing
in here runs."""
from __future__ import annotations

import os
import time

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import metrics as utilities
from .engine import utility_engine


def _pruned(msg):
    try:  # the reference signals VAE checkpoint IO failures to Optuna (:72,:83)
        import optuna  # type: ignore
        return optuna.TrialPruned(msg)
    except Exception:
        return RuntimeError(msg)


def checkpoint(model, filename, VAE_DIR_PATH):
    """Save model parameters to file (:75-83)."""
    try:
        torch.save(model.state_dict(), os.path.normpath(os.path.join(VAE_DIR_PATH, filename)))
    except Exception:
        print("Failed to save model parameters to %s" % filename)
        raise _pruned("checkpoint failed")


def resume(model, filename, VAE_DIR_PATH):
    """Load model parameters from file (:66-72)."""
    try:
        model.load_state_dict(torch.load(os.path.normpath(os.path.join(VAE_DIR_PATH, filename))))
    except Exception:
        print("Failed to load model parameters from %s" % filename)
        raise _pruned("resume failed")


class VAE(nn.Module):
    """MultiVAE++ (:206-268), PyTorch: the encode/decode hooks the denoising engine sits between."""

    def __init__(self, input_dim, hidden_dim, latent_dim, p_drop=0.5):
        super().__init__()
        self.latent_dim = latent_dim
        self.encoder = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.Tanh(), nn.Linear(hidden_dim, 2 * latent_dim))
        self.decoder = nn.Sequential(nn.Linear(latent_dim, hidden_dim), nn.Tanh(), nn.Linear(hidden_dim, input_dim))
        self.dropout = nn.Dropout(p=p_drop)
        self.model_is_trained = False
        self.is_training = 0
        self.weight_decay = 0
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight.data)
                m.bias.data.normal_(0.0, 0.001)

    def encode(self, x):
        h = self.encoder(self.dropout(F.normalize(x, p=2, dim=1)))
        mu, logvar = torch.chunk(h, chunks=2, dim=1)
        kl = -0.5 * torch.mean(torch.sum(1 + logvar - mu.pow(2) - logvar.exp(), dim=1))
        eps = torch.randn_like(mu)  # consumed even in eval, like the reference (Q12)
        return mu + self.is_training * eps * torch.exp(0.5 * logvar), kl

    def decode(self, z):
        return self.decoder(z)

    def forward(self, x):
        z, kl = self.encode(x)
        return self.decode(z), kl

    def get_l2_reg(self):
        if self.weight_decay <= 0:
            return torch.zeros((), device=next(self.parameters()).device)
        return self.weight_decay * sum(torch.norm(p, p=2) ** 2 for n, p in self.named_parameters() if n.endswith(".weight"))

    def sample(self, n_samples):
        z = torch.randn(n_samples, self.latent_dim, device=next(self.parameters()).device)
        return self.decode(z).cpu().detach().numpy()


def train_variational_autoencoder(model, train_data, test_data, epochs, batch_size, lr, early_stop_metric="NDCG@50",
                                  VAE_DIR_PATH="./", verbose=False):
    """VAE pre-stage (:115-188): multinomial NLL + annealed KL, early stopping on Recall/NDCG@k of a
    per-user hold-out of `test_data`, best epoch restored.  Plain PyTorch (not part of the hot path)."""
    os.makedirs(os.path.normpath(VAE_DIR_PATH), exist_ok=True)
    dev = next(model.parameters()).device
    anneal_cap, anneal_count = 0.2, 0.0
    best_metric, best_epoch, stale = -np.inf, 0, 0
    optimizer = torch.optim.Adam(model.parameters(), lr=lr)
    k = int(early_stop_metric.split("@")[1])
    start = time.time()
    for epoch in range(epochs):
        losses = []
        model.train()
        model.is_training = 1
        train_data = train_data[np.random.permutation(train_data.shape[0])]
        for lo in range(0, train_data.shape[0], batch_size):
            hi = min(lo + batch_size, train_data.shape[0])
            anneal = min(anneal_cap, 1.0 * anneal_count / 20_000)
            X = torch.tensor(train_data[lo:hi].toarray(), dtype=torch.float32, device=dev)
            optimizer.zero_grad()
            out, kl = model(X)
            neg_ll = -torch.mean(torch.sum(F.log_softmax(out, dim=1) * X, dim=1))
            loss = neg_ll + anneal * kl + model.get_l2_reg()
            losses.append(loss.item())
            loss.backward()
            optimizer.step()
            anneal_count += 1
        model.eval()
        model.is_training = 0
        scores = []
        valid_train, valid_test = utilities.split_train_test_proportion_from_csr_matrix(test_data, batch_size=1000)
        with torch.no_grad():
            for lo in range(0, valid_train.shape[0], 500):
                hi = min(lo + 500, valid_train.shape[0])
                X = valid_train[lo:hi]
                pred, _ = model(torch.tensor(X.toarray(), dtype=torch.float32, device=dev))
                if dev.type == "cuda":
                    # utilities.py:116-171 on the device (sdrm_rank_metrics): the [500, N_ITEMS] scores stay in HBM
                    rec, ndcg = utility_engine(dev).rank_metrics(pred, valid_test[lo:hi], train=X, ks=(k,))
                    scores.append((rec if "Recall" in early_stop_metric else ndcg)[0].cpu().numpy())
                else:
                    pred = utilities.mask_training_examples(X, pred.cpu().numpy())
                    fn = utilities.recall_at_k_batch if "Recall" in early_stop_metric else utilities.NDCG_binary_at_k_batch
                    scores.append(fn(pred, valid_test[lo:hi], k=k))
        avg = np.nanmean(np.concatenate(scores))
        if verbose:
            print(f"Epoch: {epoch}, Loss: {np.round(np.mean(losses), 4)}, {early_stop_metric}: {np.round(avg, 4)}", end="\r")
        if avg > best_metric:
            best_metric, best_epoch, stale = avg, epoch, 0
            checkpoint(model, f"epoch-{epoch}.pth", VAE_DIR_PATH)
        else:
            stale += 1
            if stale > 20:
                if verbose:
                    print(f"MultiVAE++ training complete. Early stopping at epoch {epoch}, "
                          f"Training took {np.round((time.time() - start) / 60, 2)} minutes")
                break
    resume(model, f"epoch-{best_epoch}.pth", VAE_DIR_PATH)
    model.model_is_trained = True
    model.is_training = 0


