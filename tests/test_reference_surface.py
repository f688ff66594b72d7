"""The reference's call surface (`train_SDRM`, `sample_ddpm`, `SDRM`, `VAE`) driven the way main.py:126-185
drives it, on a tiny synthetic dataset.  Needs a GPU."""
import numpy as np
import pytest
import torch
from scipy.sparse import csr_matrix

pytestmark = pytest.mark.gpu


def make_feed(data, batch):
    """What main.py:126-137 builds: a DataLoader over index batches yielding sparse COO tensors."""
    feed = []
    for lo in range(0, data.shape[0], batch):
        coo = data[lo:lo + batch].tocoo()
        x = torch.sparse_coo_tensor(np.vstack([coo.row, coo.col]), coo.data.astype(np.float32), coo.shape).cuda()
        feed.append((x, x))
    return feed


def test_train_and_sample_like_main(tmp_path):
    import sdrm_amd.train_SDRM as ts
    rs = np.random.RandomState(0)
    n_users, n_items = 96, 60
    data = csr_matrix((rs.random_sample((n_users, n_items)) < 0.25).astype(np.float64))
    valid = csr_matrix((rs.random_sample((40, n_items)) < 0.25).astype(np.float64))
    torch.manual_seed(0)
    np.random.seed(0)
    dl = make_feed(data, 40)                   # 40 + 40 + 16 rows: short last batch (Q15)
    DIFF, vae = ts.train_SDRM(dl, N_ITEMS=n_items, VAE_HIDDEN=32, VAE_LATENT=20, VAE_BATCH_SIZE=32, VAE_LR=1e-3,
                              DIFF_LATENT=20, N_HIDDEN_MLP_LAYERS=2, DIFF_LR=1e-3, DIFF_TRAINING_EPOCHS=3,
                              TIMESTEPS=8, noise_divider=0.5, VAE_DIR_PATH=str(tmp_path / "vae"),
                              TRAIN_PARTIAL_VALID_DATA=data, VALID_DATA=valid, OPTIMIZATION_OBJECTIVE="Recall@10")
    assert isinstance(vae, torch.nn.Module) and vae.model_is_trained and not vae.training
    assert np.isfinite(float(DIFF.last_loss.cpu()))
    m, v, step = DIFF.engine().get_adam_state()
    assert step == 3 * 3
    assert ts.ab_t.shape[0] == 9 and float(ts.ab_t[0]) == 1.0
    sparsity = 1 - data.nnz / (n_users * n_items)
    for mode in ("random", None):
        out = ts.sample_ddpm(n_users, DIFF, vae, 20, 0.5, timesteps=mode, n_timesteps=8)
        arr = out.detach().cpu().numpy()
        assert arr.shape == (n_users, n_items) and np.isfinite(arr).all()
        binar = arr >= np.quantile(arr.flatten(), sparsity)          # main.py:177-180
        assert abs(binar.mean() - (1 - sparsity)) < 0.02
    # the module helpers work on the schedule left behind by train_SDRM (Q10)
    x = torch.randn(5, 20, device="cuda")
    t = torch.randint(1, 9, (5,), device="cuda")
    xp = ts.perturb_input(x, t, torch.zeros_like(x))
    assert torch.allclose(xp, ts.ab_t.sqrt()[t, None] * x)
    y = DIFF.forward(x, t)
    assert y.shape == (5, 20) and float(y.abs().max()) <= 1.0
    assert not torch.equal(y, DIFF.forward(x, t))                       # dropout always on (Q2)
    stepped = ts.denoise_add_noise(x, 3, y, 0)
    assert stepped.shape == x.shape
    loss = ts.score_matching_loss(DIFF, x, t, y, torch.randn_like(x), 0.1)
    assert np.isfinite(float(loss))


def test_state_dict_through_engine():
    import sdrm_amd.train_SDRM as ts
    a = ts.SDRM(16, 6, 24, 3)
    a.engine(4)
    sd = a.state_dict()
    assert sd["dnn.4.weight"].shape == (24, 24) and torch.equal(sd["dnn.4.weight"], sd["dnn.2.weight"])
    b = ts.SDRM(16, 6, 24, 3)
    b.load_state_dict(sd)
    b.engine(4)
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p.cpu(), q.cpu()), n
    # growing the engine keeps parameters and optimiser state
    x = torch.randn(4, 16, device="cuda")
    a.engine(4).train_step(x, 1e-3, seed=1, step=0)
    before = a.engine().get_params().clone()
    a.engine(64)
    assert torch.equal(before, a.engine().get_params()) and a.engine().get_adam_state()[2] == 1


def test_parameters_are_live_views_and_refuse_writes():
    """`SDRM.parameters()` (hyperparameter_search.py:53 passes nets around like any nn.Module): with a live engine the tensors alias
    the engine's parameter vector - they follow a train step without being fetched again - and an in-place write (what an external
    torch.optim would do) raises instead of silently training a copy; `.cuda()`, `.zero_grad()`, `.modules()` exist."""
    import sdrm_amd.train_SDRM as ts
    from sdrm_amd.engine import SdrmError
    net = ts.SDRM(16, 6, 24, 2).cuda()
    eng = net.engine(8)
    params = dict(net.named_parameters())
    before = {n: p.as_subclass(torch.Tensor).clone() for n, p in params.items()}
    assert float((params["dnn.0.weight"] * 2).sum()) == float(before["dnn.0.weight"].sum() * 2)      # reads are ordinary tensor ops
    eng.train_step(torch.randn(8, 16, device="cuda"), 1e-2, seed=1, step=0)
    torch.cuda.synchronize()
    assert all(not torch.equal(p.as_subclass(torch.Tensor), before[n]) for n, p in params.items())    # the same objects see the update
    assert torch.equal(torch.cat([p.as_subclass(torch.Tensor).reshape(-1) for p in net.parameters()]), eng.get_params())
    with pytest.raises((SdrmError, RuntimeError)):
        params["dnn.0.bias"].add_(1.0)
    with pytest.raises(ValueError, match="non-leaf"):       # an external optimiser cannot be built over them
        torch.optim.SGD(net.parameters(), lr=0.1)
    net.zero_grad()
    assert list(net.modules()) == [net]
    with pytest.raises(SdrmError):
        net.cpu()


def test_parameter_views_survive_an_engine_rebuild():
    """`train_SDRM` (batch <= 1024) then `sample_ddpm(n_sample=5429)` rebuilds the engine with more rows.  Tensors
    taken from `parameters()` before that alias the OLD engine's master vector: it must stay allocated until they are gone (reads
    give the parameters as they were at the rebuild - never freed memory), and be freed then."""
    import gc
    import sdrm_amd.train_SDRM as ts
    net = ts.SDRM(16, 6, 24, 1, max_rows=8).cuda()
    old = net.engine(8)
    old.train_step(torch.randn(8, 16, device="cuda"), 1e-2, seed=1, step=0)
    held = net.parameters()
    snap = [p.detach().clone() for p in held]              # (detached: a clone WITH autograd history would itself keep the base alive)
    new = net.engine(64)                                   # rows > max_rows: a new handle, the old one closed
    assert new is not old and not old.closed               # ... but deferred: `held` still aliases its memory
    new.train_step(torch.randn(64, 16, device="cuda"), 1e-2, seed=1, step=1)
    filler = [torch.full((1 << 20,), 7.0, device="cuda") for _ in range(4)]   # would land in freed memory, were it freed
    torch.cuda.synchronize()
    for p, q in zip(held, snap):
        assert torch.equal(p.as_subclass(torch.Tensor), q)
    assert not torch.equal(torch.cat([q.reshape(-1) for q in snap]), new.get_params())   # the new engine has moved on
    del held, p, q, filler
    gc.collect()
    assert old.closed                                      # the last view is gone: the deferred sdrm_destroy ran
    fresh = net.parameters()
    assert torch.equal(torch.cat([p.as_subclass(torch.Tensor).reshape(-1) for p in fresh]), new.get_params())


def test_parameter_write_guard_covers_aliases_and_no_grad():
    """`.data`, `.detach()` and views of a parameter share its storage and keep the guard; parameters handed out under
    torch.no_grad() are still non-leaf (torch.optim refuses them); `requires_grad_` is not a write."""
    import sdrm_amd.train_SDRM as ts
    from sdrm_amd.engine import SdrmError
    net = ts.SDRM(16, 6, 24, 1).cuda()
    eng = net.engine(8)
    with torch.no_grad():
        params = dict(net.named_parameters())
        with pytest.raises(ValueError, match="non-leaf"):
            torch.optim.Adam(list(params.values()), lr=1e-3)
    p = params["dnn.0.bias"]
    before = eng.get_params().clone()
    for write in (lambda: p.data.add_(1.0), lambda: p.detach().mul_(2.0), lambda: p.view(-1).zero_(), lambda: p[:2].fill_(3.0),
                  lambda: p.__setitem__(0, 1.0), lambda: torch.add(p, 1.0, out=p), lambda: p.copy_(torch.zeros_like(p))):
        with pytest.raises((SdrmError, RuntimeError)):
            write()
    torch.cuda.synchronize()
    assert torch.equal(eng.get_params(), before)           # nothing got through
    p.requires_grad_(True)                                 # harmless: no data is written (not refused as an in-place write)
    assert float((p.detach() + 1).sum()) == float(p.as_subclass(torch.Tensor).sum() + p.numel())   # reads through aliases are ordinary ops
    assert type(p.detach().clone()) is torch.Tensor        # a copy is the caller's own tensor


def test_cache_latents_matches_per_batch_encoding(tmp_path):
    """encoding the feed once gives the same trained eps-net as encoding per batch."""
    import sdrm_amd.train_SDRM as ts
    rs = np.random.RandomState(1)
    data = csr_matrix((rs.random_sample((50, 30)) < 0.3).astype(np.float64))
    vae = ts.VAE(30, 16, 10).cuda()
    vae.model_is_trained = True
    outs = []
    for cache in (False, True):
        torch.manual_seed(3)
        D, _ = ts.train_SDRM(make_feed(data, 25), 30, 16, 10, 25, 1e-3, 10, 1, 1e-3, 2, 5, 1.0, str(tmp_path), data, data,
                             "Recall@10", variational_ae=vae, cache_latents=cache)
        outs.append(D.engine().get_params().cpu())
    assert torch.allclose(outs[0], outs[1], rtol=0, atol=0)
