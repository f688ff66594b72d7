"""Step-level parity AT THE LAUNCH CONFIGURATION bench.py measures (BASELINE.json configs[2]: ML-1M shaped, L = W = 340,
T = 78, H = 1, train batch 8192 = 24576 stacked rows, 5429 sampled users), through the C ABI, against the CPU oracle -
and over EVERY kernel variant a size threshold could select there: the GEMM tile is forced to the default 64x64x16 and
to 32x32x32 as well as left automatic, the DDPM reverse update runs stand-alone, fused by the size rule and always
fused.  A threshold retune therefore cannot change which code this suite covers.

Reference lines: /root/reference/train_SDRM.py:326-337 (train step), :50-61 (full-resolution sampling), :37-49
(multi-resolution sampling).  Needs a real MI355X: `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from sdrm_amd import synth
from test_hip_parity import TOL, close, engine_branch_masks, per_tensor, rel_l2, rel_max

pytestmark = pytest.mark.gpu

L, W, T, H = 340, 340, 78, 1
B_TRAIN, N_SAMPLE = 8192, 5429
TILES = [-1, 0, 4]          # automatic, 64x64x16 on the 32-wide MFMA, 32x32x32 on the 16-wide MFMA
TRAIN_PATHS = TILES + ["row", "row-layers", "row-tiles", "row48"]   # ... and the row-owned forward forced on (what the automatic choice takes at this batch, whatever
                                # its size rule becomes)
ND = 0.9


# ------------------------------------------------------------------------------------------------ train step
@pytest.fixture(scope="module")
def train_case():
    """Inputs of one B = 8192 step and the oracle's un-steered result (computed once for all tile variants)."""
    from oracle import sdrm_oracle as orc
    init = synth.init_params(L, W, T, H, seed=3)
    x0 = synth.synth_latents(B_TRAIN, L, seed=4)
    eps, t, masks = synth.synth_train_randoms(B_TRAIN, L, T, ND, seed=5)
    o = orc.Oracle(L, W, T, H, init)
    caches = []
    loss, grads, outs, _ = o.loss_and_grads(x0, eps, t, list(masks), caches=caches)
    return dict(init=init, x0=x0, eps=eps, t=t, masks=masks, oracle=o, caches=caches, loss=float(loss), grads=grads,
                outs=outs)


@pytest.mark.parametrize("tile", TRAIN_PATHS)
def test_train_step_headline_vs_oracle(engine_cls, train_case, tile):
    """B = 8192, L = 340 on each tile and on the row-owned forward: P/S/Q, loss, every gradient tensor, the parameters
    after Adam.

    Two gradient comparisons on the same step:
      * UN-STEERED, the oracle exactly as the reference computes it.  The tensors downstream of the last PReLU (the
        output layer's weight and bias) do not see the PReLU derivative at all and must agree to 5e-5 of max|ref| (the
        reference's own summation-reorder floor is 2.2e-5).  Every tensor upstream sees PReLU'(pre),
        which jumps at 0: an element whose pre-activation is zero within fp32 rounding may take the other branch (each
        such element is verified to sit at |pre| <= 2e-5 max|pre|), and ONE flip moves an upstream gradient by about
        1/sqrt(B W) (DESIGN.md "kink flips").  The effect of exactly those flips is computable - the oracle backward
        with the engine's branch choice at those elements (STEERED) minus the un-steered one - so the statement is exact,
        without a cushion: (engine - un-steered) equals (steered - un-steered) to 1e-4 of the un-steered tensor.
      * STEERED: the same residual, engine - steered, at 1e-4 of the steered tensor."""
    c = train_case
    o = c["oracle"]
    e = engine_cls(L, W, T, H, B_TRAIN).debug_set(tile=tile)
    e.set_params(synth.flatten_params(c["init"], H))
    e.train_forward(c["x0"], noise=c["eps"], t=c["t"], keep=c["masks"])
    branch, flips = engine_branch_masks(e, o, c["caches"], B_TRAIN)
    loss = float(e.train_backward().cpu())
    assert abs(loss - c["loss"]) <= TOL * abs(c["loss"])
    psq = e.train_outputs(B_TRAIN).cpu().numpy()
    for j in range(3):
        assert close(psq[j], c["outs"][j].numpy()), (j, rel_max(psq[j], c["outs"][j].numpy()))
    grads = e.get_grads().cpu().numpy()
    last = 2 + 2 * H
    downstream = {f"dnn.{last}.weight", f"dnn.{last}.bias"}
    # steered: same step, the engine's branch choice at the (verified) kink elements
    _, grads_st, _, _ = o.loss_and_grads(c["x0"], c["eps"], c["t"], list(c["masks"]), neg_override=branch)
    for n, got in per_tensor(grads, (L, W, T, H)):
        ref = c["grads"][n].numpy().ravel().astype(np.float64)
        st = grads_st[n].numpy().ravel().astype(np.float64)
        if n in downstream:
            assert rel_max(got, ref) <= 5e-5 and rel_l2(got, ref) <= 5e-5, (n, rel_max(got, ref), rel_l2(got, ref))
        else:
            # what the verified flips do to this tensor, exactly: steered - un-steered; the engine must show that and nothing else
            resid = (got.astype(np.float64) - ref) - (st - ref)
            assert np.sqrt((resid ** 2).sum()) <= TOL * np.sqrt((ref ** 2).sum()) and np.abs(resid).max() <= TOL * np.abs(ref).max(), \
                (n, flips, float(np.sqrt((resid ** 2).sum() / (ref ** 2).sum())), float(np.abs(resid).max() / np.abs(ref).max()))
        assert rel_l2(got, st) <= TOL and rel_max(got, st) <= TOL, (n, flips, rel_l2(got, st), rel_max(got, st))
    # Adam on the steered gradient (a sign flip of a ~0 gradient element moves a weight by 2 lr: normwise bar)
    from oracle import sdrm_oracle as orc
    o2 = orc.Oracle(L, W, T, H, c["init"])
    o2.adam_step(grads_st, 9.8e-5)
    e.adam_step(9.8e-5)
    assert rel_l2(e.get_params().cpu().numpy(), o2.flat(synth.param_names(H))) <= TOL
    e.close()


@pytest.mark.parametrize("tile", TRAIN_PATHS)
def test_train_step_headline_philox(engine_cls, tile):
    """The bench's own mode at its own size: PHILOX-mode step == EXPLICIT-mode step fed the numpy restatement of the
    device generator, at B = 8192 with a shard-style row offset."""
    from oracle import philox_ref as pr
    seed, step, row0 = 1234, 7, 4096
    init = synth.flatten_params(synth.init_params(L, W, T, H, seed=6), H)
    x0 = synth.synth_latents(B_TRAIN, L, seed=7)
    eps, t, keep = pr.train_randoms(seed, step, row0, B_TRAIN, L, T, 1.0)
    out = []
    for explicit in (False, True):
        e = engine_cls(L, W, T, H, B_TRAIN).debug_set(tile=tile)
        e.set_params(init)
        if explicit:
            e.train_forward(x0, noise=eps, t=t, keep=keep)
        else:
            e.train_forward(x0, seed=seed, step=step, nd=1.0, row0=row0)
        pre = [e.preacts(k, B_TRAIN).cpu() for k in range(H + 1)]
        loss = float(e.train_backward().cpu())
        e.adam_step(9.8e-5)
        out.append((loss, e.train_outputs(B_TRAIN).cpu().numpy(), e.get_grads().cpu().numpy(), e.get_params().cpu().numpy(), pre))
        e.close()
    (l1, p1, g1, w1, pre1), (l2, p2, g2, w2, pre2) = out
    assert abs(l1 - l2) <= 2e-5 * abs(l2)
    assert close(p1, p2, 2e-5)
    # the device's normals differ from numpy's in the last bits (hardware log2 / sin / cos), so the two runs may sit on
    # different sides of a PReLU kink at pre-activations that are zero within rounding: count those, bound as above
    flips = 0
    for a, b in zip(pre1, pre2):
        diff = (a <= 0) != (b <= 0)
        if bool(diff.any()):
            assert float(b[diff].abs().max()) <= 2e-5 * float(b.abs().max()), "branch flip away from the kink"
            flips += int(diff.sum())
    tol = TOL + 4.0 * flips / np.sqrt(B_TRAIN * W)
    for (n, a), (_, b) in zip(per_tensor(g1, (L, W, T, H)), per_tensor(g2, (L, W, T, H))):
        assert rel_l2(a, b) <= tol and rel_max(a, b) <= tol, (n, flips, rel_l2(a, b), rel_max(a, b))
    assert rel_l2(w1, w2) <= TOL


# ------------------------------------------------------------------------------------------------ sampling
@pytest.fixture(scope="module")
def sample_case():
    """Explicit and Philox randoms for n = 5429 rows x 78 steps, and the oracle's latents for the full-resolution and the
    multi-resolution branch of each (four oracle runs, ~4 s each, shared by every kernel variant below)."""
    from oracle import philox_ref as pr
    from oracle import sdrm_oracle as orc
    init = synth.init_params(L, W, T, H, seed=21)
    o = orc.Oracle(L, W, T, H, init)
    case = {"init": init}
    xT, z, keep, Tj = synth.synth_sample_randoms(N_SAMPLE, L, T, ND, seed=22, multires=True)
    case["explicit"] = dict(xT=xT, z=z, keep=keep, Tj=Tj,
                            full=o.sample(xT, z, keep).numpy(), multi=o.sample(xT, z, keep, Tj).numpy())
    seed, call_id, row0 = 99, 5, 2715
    xT, z, keep, Tj = pr.sample_randoms(seed, call_id, row0, N_SAMPLE, L, T, ND, True)
    case["philox"] = dict(seed=seed, call_id=call_id, row0=row0, Tj=Tj,
                          full=o.sample(xT, z, keep).numpy(), multi=o.sample(xT, z, keep, Tj).numpy())
    return case


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("multires", [False, True])
def test_sampling_headline_explicit(engine_cls, sample_case, multires, tile):
    """n = 5429 (> 4096 rows: the default-tile path with the B0tab bias row, lda = LP != ldw = K0, the stand-alone
    reverse update), injected randoms, full-resolution (:50-61) and multi-resolution with prefix compaction (:37-49)."""
    c = sample_case["explicit"]
    e = engine_cls(L, W, T, H, N_SAMPLE).debug_set(tile=tile)
    e.set_params(synth.flatten_params(sample_case["init"], H))
    out = e.sample(N_SAMPLE, nd=ND, multires=multires, xT=c["xT"], z=c["z"], keep=c["keep"], Tj=c["Tj"] if multires else None)
    ref = c["multi" if multires else "full"]
    assert close(out, ref), (rel_max(out.cpu().numpy(), ref), rel_l2(out.cpu().numpy(), ref))
    e.close()


@pytest.mark.parametrize("fused", [0, 1, 2])
@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("multires", [False, True])
def test_sampling_headline_philox(engine_cls, sample_case, multires, tile, fused):
    """The bench's own sampling mode at its own size: on-device Philox, replayed through the oracle with
    oracle/philox_ref.py (start steps bit-exact).  `fused` = sdrm_debug_set_fused_reverse: the reverse update as its own
    kernel (0), fused into the out-layer GEMM epilogue by the size rule (1) or always (2: on the 64x64 tile this is the
    32-wide-MFMA branch of EPI_TANH_REV).  Multi-resolution calls fuse by the same rules since round 5 (their active prefix is a
    launch like any other; the epilogue keys Philox by the slot's original row)."""
    c = sample_case["philox"]
    e = engine_cls(L, W, T, H, N_SAMPLE).debug_set(tile=tile, fused_reverse=fused)
    e.set_params(synth.flatten_params(sample_case["init"], H))
    res = e.sample(N_SAMPLE, nd=ND, multires=multires, seed=c["seed"], call_id=c["call_id"], row0=c["row0"], return_Tj=multires)
    if multires:
        out, tj = res
        np.testing.assert_array_equal(tj.cpu().numpy(), c["Tj"])
    else:
        out = res
    ref = c["multi" if multires else "full"]
    assert close(out, ref), (rel_max(out.cpu().numpy(), ref), rel_l2(out.cpu().numpy(), ref))
    e.close()


def test_sampling_interleaved_with_train_steps(engine_cls, sample_case):
    """bench.py walks train steps between sdrm_sample_steps calls.  The sampler keeps its own state and its own snapshot
    of the net, so the interleaved run - the parameters moving under it with every Adam step - must reproduce the
    uninterrupted one bit for bit at the measured size."""
    c = sample_case["philox"]
    flat = synth.flatten_params(sample_case["init"], H)
    x0 = synth.synth_latents(2048, L, seed=8)
    e = engine_cls(L, W, T, H, N_SAMPLE)
    e.set_params(flat)
    ref = e.sample(N_SAMPLE, nd=ND, seed=c["seed"], call_id=c["call_id"], row0=c["row0"])
    assert close(ref, c["full"])
    e.sample_begin(N_SAMPLE, nd=ND, seed=c["seed"], call_id=c["call_id"], row0=c["row0"])
    k = 0
    while e.sample_steps(7) > 0:
        e.train_step(x0, 1e-3, seed=3, step=k)     # clobbers the shared activation buffers, moves the parameters
        k += 1
    out = e.sample_end()
    assert k >= 10 and bool((out == ref).all())
    e.close()


@pytest.mark.parametrize("multires", [False, True])
def test_sampler_row_chains_change_no_bit(engine_cls, sample_case, multires):
    """Sampling calls of this size run as two row chains on two streams by default (csrc/sdrm_hip.hip: chains_for): rows are independent
    through the whole reverse loop and the generator is keyed by row, so one to four chains all reproduce the oracle's latents (the
    tile and the fusion of the reverse update go by a chain's row count, so the bits may differ between chain counts), and the same
    chain count gives the same bits - also when an event profile, which serialises the chains, begins and ends inside the call."""
    c = sample_case["philox"]
    flat = synth.flatten_params(sample_case["init"], H)
    kw = dict(nd=ND, multires=multires, seed=c["seed"], call_id=c["call_id"], row0=c["row0"])
    outs = {}
    for chains in (-1, 1, 2, 3, 4):
        e = engine_cls(L, W, T, H, N_SAMPLE).debug_set(chains=chains)
        e.set_params(flat)
        outs[chains] = e.sample(N_SAMPLE, **kw).cpu()
        assert e.sampler_chains == (2 if chains == -1 else chains)
        e.close()
    for chains in (-1, 1, 2, 3, 4):
        assert close(outs[chains], c["multi" if multires else "full"]), chains
    assert bool((outs[2] == outs[-1]).all())
    # the size rule: one chain for one rank of eight, two for one rank of two
    for n, want in ((679, 1), (2715, 2)):
        e = engine_cls(L, W, T, H, n)
        e.set_params(flat)
        e.sample(n, nd=ND, seed=1)
        assert e.sampler_chains == want
        e.close()
    # an event profile that begins and ends inside the call
    e = engine_cls(L, W, T, H, N_SAMPLE)
    e.set_params(flat)
    e.sample_begin(N_SAMPLE, **kw)
    e.sample_steps(20)
    e.profile_begin(capacity=4096)
    e.sample_steps(25)
    prof = e.profile_end()
    assert prof
    while e.sample_steps(9) > 0:
        pass
    assert bool((e.sample_end().cpu() == outs[-1]).all())
    e.close()


@pytest.mark.parametrize("multires", [False, True])
def test_small_sampling_call_runs_beside_per_layer_train_steps(engine_cls, multires):
    """A sampling call below the two-chain size (679 rows: one rank of eight) is ONE chain on an auxiliary stream, and train steps of the
    per-layer path queued between its steps do not hold it (csrc/sdrm_hip.hip: chains_for, hold_chains): the call reads its own snapshot
    of the net and runs in its own buffers, so the interleaved call - parameters moving under it with every Adam step, its launches
    sharing the chip with the train steps' - reproduces the uninterrupted one bit for bit, and so do the parameters."""
    n, B = 679, 512
    init = synth.init_params(L, W, T, H, seed=31)
    flat = synth.flatten_params(init, H)
    x0 = synth.synth_latents(B, L, seed=9)
    kw = dict(nd=ND, multires=multires, seed=17, call_id=3, row0=1358)
    e = engine_cls(L, W, T, H, max(n, B))
    e.set_params(flat)
    ref = e.sample(n, **kw).cpu()
    assert e.sampler_chains == 1
    for k in range(12):
        e.train_step(x0, 1e-3, seed=3, step=k)
    p_ref = e.get_params().cpu()
    e.close()
    e = engine_cls(L, W, T, H, max(n, B))
    e.set_params(flat)
    e.sample_begin(n, **kw)
    k = 0
    while e.sample_steps(7) > 0:
        if k < 12:
            e.train_step(x0, 1e-3, seed=3, step=k)
        k += 1
    out = e.sample_end().cpu()
    assert k >= 8 and bool((out == ref).all())
    while k < 12:
        e.train_step(x0, 1e-3, seed=3, step=k)
        k += 1
    assert bool((e.get_params().cpu() == p_ref).all())
    e.close()


def test_train_steps_beside_a_small_call_at_the_column_split_size(engine_cls):
    """2048 users, 1358 sampled rows (one rank of four): a train step queued beside the detached chain takes the per-layer path instead of
    the column-split kernels (csrc/sdrm_hip.hip: rows48_parts).  The sampling call is the plain call bit for bit; the first such step's
    loss and gradient are those of the sequential run's first step - on the column-split path - within the rounding of the two paths
    (later steps are only exercised: Adam's first updates are lr * sign(g), which turns a last-bit difference of a near-zero gradient
    into 2 lr)."""
    n, B = 1358, 2048
    flat = synth.flatten_params(synth.init_params(L, W, T, H, seed=33), H)
    x0 = synth.synth_latents(B, L, seed=10)
    kw = dict(nd=ND, seed=19, call_id=4, row0=2716)
    e = engine_cls(L, W, T, H, B)
    e.set_params(flat)
    ref = e.sample(n, **kw).cpu()
    loss_ref = float(e.train_step(x0, 1e-3, seed=3, step=0).cpu())
    g_ref = e.get_grads().cpu().numpy()
    e.close()
    e = engine_cls(L, W, T, H, B)
    e.set_params(flat)
    e.sample_begin(n, **kw)
    k = 0
    while e.sample_steps(11) > 0:
        if k < 6:
            loss = float(e.train_step(x0, 1e-3, seed=3, step=k).cpu())
            if k == 0:
                g = e.get_grads().cpu().numpy()
                assert abs(loss - loss_ref) <= 2e-5 * abs(loss_ref)
                for (name, a), (_, b) in zip(per_tensor(g, (L, W, T, H)), per_tensor(g_ref, (L, W, T, H))):
                    assert rel_l2(a, b) <= TOL and rel_max(a, b) <= 10 * TOL, (name, rel_l2(a, b), rel_max(a, b))
        k += 1
    out = e.sample_end().cpu()
    assert k >= 6 and bool((out == ref).all())
    e.close()


def test_abandoned_and_closed_sampling_calls_with_chains_in_flight(engine_cls, sample_case):
    """A sampling call that is dropped while its chains are in flight - a new sdrm_sample_begin, or the engine closed - leaves nothing behind:
    the next call is the plain call, bit for bit."""
    c = sample_case["philox"]
    flat = synth.flatten_params(sample_case["init"], H)
    x0 = synth.synth_latents(1024, L, seed=8)
    for n in (679, N_SAMPLE):     # one detached chain / two chains
        kw = dict(nd=ND, seed=c["seed"], call_id=c["call_id"], row0=c["row0"])
        e = engine_cls(L, W, T, H, max(n, 1024))
        e.set_params(flat)
        ref = e.sample(n, **kw).cpu()
        e.sample_begin(n, **kw)
        e.sample_steps(10)
        e.train_step(x0, 0.0, seed=3, step=0)      # lr = 0: the net stays what it is; the chain is detached / held
        e.sample_steps(5)
        e.sample_begin(n, **kw)                     # abandons the call above
        while e.sample_steps(13) > 0:
            pass
        assert bool((e.sample_end().cpu() == ref).all())
        e.sample_begin(n, **kw)
        e.sample_steps(20)
        e.train_step(x0, 0.0, seed=3, step=1)
        e.sample_steps(3)
        e.close()                                   # chains in flight
