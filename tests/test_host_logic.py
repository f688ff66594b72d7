"""Host-side logic that needs no GPU: C-ABI symbol export, parameter layout, metric/split helpers
against vectors produced by the reference's own utilities.py, schedule/step-mix helpers."""
import ctypes
import os
import re

import numpy as np
import pytest
from scipy.sparse import csr_matrix

from sdrm_amd import _lib, metrics, synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    """The library loads (no GPU needed) and exports exactly what include/sdrm_hip.h (the boundary) and
    include/sdrm_hip_debug.h (test / tuning hooks) declare; no debug hook sits in the public header."""
    declared = set()
    for name in ("sdrm_hip.h", "sdrm_hip_debug.h"):
        header = open(os.path.join(REPO, "include", name)).read()
        found = set(re.findall(r"\b(sdrm_[a-z_0-9]+)\s*\(", header)) - {"sdrm_engine"}
        if name == "sdrm_hip.h":
            assert not [f for f in found if f.startswith("sdrm_debug_")], "debug hooks belong in sdrm_hip_debug.h"
        else:
            assert all(f.startswith("sdrm_debug_") for f in found), found
        declared |= found
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in a header but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert b"gfx950" in lib.sdrm_build_info()


def test_binary_is_bound_to_its_sources(tmp_path, monkeypatch):
    """The loaded library carries the SHA-256 of csrc/ + include/; touching a source makes the loader see a stale binary
    (and rebuild it), whatever the file times say."""
    from sdrm_amd import _build
    lib = _lib.load()
    assert lib.sdrm_source_hash().decode() == _build.source_hash() == _build.binary_hash()
    assert _build.source_hash()[:16].encode() in lib.sdrm_build_info()
    assert not _build.is_stale()
    # a copy of the source tree with one edited header: the same binary is stale against it
    src, inc = tmp_path / "csrc", tmp_path / "include"
    src.mkdir(), inc.mkdir()
    for f in _build.source_files():
        dst = (src if os.path.dirname(f) == _build.SRC_DIR else inc) / os.path.basename(f)
        dst.write_bytes(open(f, "rb").read())
    monkeypatch.setattr(_build, "SRC_DIR", str(src))
    monkeypatch.setattr(_build, "INC_DIR", str(inc))
    assert not _build.is_stale()
    with open(src / "philox.h", "ab") as f:
        f.write(b"\n// touched\n")
    os.utime(src / "philox.h", (0, 0))              # an OLD file time must not hide the edit
    assert _build.is_stale()
    # load() on a stale binary rebuilds (here: into a scratch path, from the edited copy) or refuses
    monkeypatch.setattr(_build, "LIB_PATH", str(tmp_path / "libsdrm_hip.so"))
    monkeypatch.setattr(_lib, "_LIB", None)
    with pytest.raises(RuntimeError, match="is missing"):
        _lib.load(build_if_missing=False)
    (tmp_path / "libsdrm_hip.so").write_bytes(open(os.path.join(REPO, "sdrm_amd", "libsdrm_hip.so"), "rb").read())
    with pytest.raises(RuntimeError, match="different sources"):
        _lib.load(build_if_missing=False)
    # the edited header sits two levels below its include/ (csrc/x.hip includes "../../include/sdrm_hip.h"): mirror that
    deep = tmp_path / "a" / "csrc"
    deep.mkdir(parents=True)
    for f in os.listdir(src):
        (deep / f).write_bytes((src / f).read_bytes())
    monkeypatch.setattr(_build, "SRC_DIR", str(deep))
    rebuilt = _lib.load()
    assert rebuilt.sdrm_source_hash().decode() == _build.source_hash() != lib.sdrm_source_hash().decode()


def test_cabi_rejects_bad_arguments_without_a_gpu():
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.sdrm_create(0, 8, 5, 1, 4, 0, ctypes.byref(h)) == -2          # SDRM_ERR_SHAPE
    assert lib.sdrm_create(8, 8, 5, 1, 4, 0, None) == -1                     # SDRM_ERR_ARG
    # k_tail_emb's LDS image grows with T and passes 160 KB at T = 1021: refused up front, not at the first launch
    assert lib.sdrm_create(8, 8, 1021, 1, 4, 0, ctypes.byref(h)) == -2 and not h.value
    assert lib.sdrm_create(8, 8, 1024, 1, 4, 0, ctypes.byref(h)) == -2 and not h.value
    assert lib.sdrm_param_count(None) == -1
    assert lib.sdrm_destroy(None) == -1


def test_device_check_refuses_anything_but_the_chip_the_grids_are_sized_for():
    """the one-round launches (one work-group per CU) and the XCD mapping assume a gfx950 with 256 compute
    units; sdrm_create asks this pure function (hipDeviceProp_t::gcnArchName, multiProcessorCount) and refuses the rest."""
    lib = _lib.load()
    assert lib.sdrm_debug_device_check(b"gfx950:sramecc+:xnack-", 256) == 0
    assert lib.sdrm_debug_device_check(b"gfx950", 256) == 0
    for arch, cus in ((b"gfx950:sramecc+:xnack-", 32), (b"gfx950", 128), (b"gfx950", 304), (b"gfx942:sramecc+:xnack-", 304),
                      (b"gfx942", 256), (b"gfx9500", 256), (b"gfx90a", 104), (b"", 256)):
        assert lib.sdrm_debug_device_check(arch, cus) == -7, (arch, cus)      # SDRM_ERR_DEVICE
    assert lib.sdrm_debug_device_check(None, 256) == -1
    assert _lib.STATUS[-7] == "SDRM_ERR_DEVICE"
    assert b"256 compute units" in lib.sdrm_build_info()


@pytest.mark.parametrize("rows,n_out,k_in", [(24576, 352, 448), (24576, 352, 352), (1664, 832, 928), (1664, 832, 832),
                                             (2560, 64, 160), (512, 352, 352), (3072, 352, 352), (12288, 352, 352),
                                             (64, 32, 32), (135040, 96, 96), (192, 4096, 4096)])
def test_wgrad_split_plan(rows, n_out, k_in):
    """Split-K planning of a weight-gradient launch (host logic only): the slices cover the rows with no empty slice, are
    whole K-steps of 32 rows, at least four of them per slice when the rows allow, and slices x tiles stays within the
    work-group target (2.2 rounds of the resident work-groups) unless one slice already exceeds it."""
    lib = _lib.load()
    s, kc = ctypes.c_int(), ctypes.c_int()
    assert lib.sdrm_debug_plan_wgrad(rows, n_out, k_in, ctypes.byref(s), ctypes.byref(kc)) == 0
    S, KC = s.value, kc.value
    assert 1 <= S <= 64 and KC % 32 == 0 and KC >= 32
    assert S * KC >= rows and (S - 1) * KC < rows                 # covers the rows, no empty slice
    tiles = -(-n_out // 64) * -(-k_in // 64)
    assert S == 1 or S * tiles <= 2816, (S, tiles)
    assert S == 1 or KC >= 128, (S, KC)
    if tiles * 2 <= 2816 and rows >= 256:
        assert S >= 2, (S, KC)                                     # a small gradient over many rows is split
    assert lib.sdrm_debug_plan_wgrad(0, 8, 8, ctypes.byref(s), ctypes.byref(kc)) == -1


@pytest.mark.parametrize("H", [0, 1, 2, 5])
def test_parameter_layout_matches_reference(golden, H):
    g = golden("host")
    assert list(g[f"param_keys_H{H}"]) == synth.param_names(H)
    from sdrm_amd.train_SDRM import SDRM
    net = SDRM(6, 4, 6, H)
    assert list(net.state_dict().keys()) == list(g[f"state_keys_H{H}"])
    assert [n for n, _ in net.named_parameters()] == list(g[f"param_keys_H{H}"])
    sd = net.state_dict()
    for alias, target in synth.alias_keys(H).items():
        assert sd[alias].data_ptr() == sd[target].data_ptr() or bool((sd[alias] == sd[target]).all())


def test_param_count_formula():
    # P = 2,145,054 for ML-100k/SVD, 380,504 for ML-1M/MLP, 17,384 for ADM/NeuMF
    assert synth.param_count(830, 830, 83, 2) == 2_145_054
    assert synth.param_count(340, 340, 78, 1) == 380_504
    assert synth.param_count(40, 40, 93, 5) == 17_384


def test_state_dict_roundtrip_cpu():
    from sdrm_amd.train_SDRM import SDRM
    a, b = SDRM(10, 5, 12, 3), SDRM(10, 5, 12, 3)
    b.load_state_dict(a.state_dict())
    for (_, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert bool((p == q).all())
    bound = 1 / np.sqrt(10 + 5)
    w0 = dict(a.named_parameters())["dnn.0.weight"]
    assert float(w0.abs().max()) <= bound and float(dict(a.named_parameters())["dnn.1.weight"]) == 0.25


def test_metrics_against_reference(golden):
    g = golden("host")
    held, train = csr_matrix(g["held"]), csr_matrix(g["train"])
    for k in (1, 5, 10):
        with np.errstate(all="ignore"):
            np.testing.assert_allclose(metrics.recall_at_k_batch(g["pred"].copy(), held, k=k), g[f"recall_{k}"], rtol=1e-6)
            np.testing.assert_allclose(metrics.NDCG_binary_at_k_batch(g["pred"].copy(), held, k=k), g[f"ndcg_{k}"], rtol=1e-6)
    np.testing.assert_array_equal(metrics.mask_training_examples(train, g["pred"].copy()), g["masked"])


def test_split_against_reference(golden):
    g = golden("host")
    tr, te = metrics.split_train_test_proportion_from_csr_matrix(csr_matrix(g["split_in"]), batch_size=7,
                                                                 random_seed=123, test_prop=0.2)
    np.testing.assert_array_equal(tr.toarray(), g["split_train"])
    np.testing.assert_array_equal(te.toarray(), g["split_test"])


def test_bench_step_mix():
    import bench
    n_train, cycle = 15, 93
    kinds = [bench.is_train(k, n_train, cycle) for k in range(2 * cycle)]
    assert sum(kinds[:cycle]) == n_train and sum(kinds) == 2 * n_train
    gaps = np.diff(np.flatnonzero(kinds))
    assert gaps.min() >= 6 and gaps.max() <= 7            # evenly spread
    assert bench.train_flops(8192, 340, 78, 1) == pytest.approx(4.996e10, rel=1e-3)
    assert bench.sample_flops(5429, 340, 78, 1) == pytest.approx(4.12e9, rel=1e-2)


def test_bench_walk_in_runs_is_the_walk_in_steps():
    """bench.py's timed windows take the walk's consecutive sampling steps in one sdrm_sample_steps call (Job.advance): exactly K steps
    per window whatever K is, the same train / sampling sequence as one step per call, a run never past the end of its sampling call."""
    import bench

    class Rec:
        def __init__(self):
            self.log, self.left = [], 0
        def sample_begin(self, n, **kw):
            self.log.append(("begin",)); self.left = 78
        def sample_steps(self, k):
            assert 1 <= k <= self.left
            self.log.append(("steps", k)); self.left -= k
            return self.left
        def sample_end(self):
            assert self.left == 0
            self.log.append(("end",))
    class Tr:
        def __init__(self, rec): self.rec = rec
        def train_step(self, *a, **kw): self.rec.log.append(("train",))
    wl = {"epochs": 3, "batches_per_epoch": 5, "T": 78, "nd": 1.0, "lr": 1e-3}
    def flat(log):
        out = []
        for ev in log:
            out += [("steps", 1)] * ev[1] if ev[0] == "steps" else [ev]
        return out
    for K in (1, 7, 20, 93, 186, 250):
        r1, r2 = Rec(), Rec()
        j1, j2 = bench.Job(r1, Tr(r1), None, 0, 10, 0, wl), bench.Job(r2, Tr(r2), None, 0, 10, 0, wl)
        for w in range(11):
            done = trains = 0
            while done < K:
                m, tr = j1.advance(K - done)
                assert m >= 1 and done + m <= K
                done += m; trains += tr
            kinds = [j2.step() for _ in range(K)]
            assert trains == kinds.count("train") and j1.k == j2.k and j1.train_count == j2.train_count
        assert flat(r1.log) == flat(r2.log)
        assert len(r1.log) < len(r2.log) or K == 1


def test_bench_window_rule():
    """bench.py times several K-step windows along the cyclic walk: their number covers whole job cycles and the
    combination is unbiased in the train:sample mix even when one window cannot hold the mix (K = 20: 3 or 4 train steps)."""
    import bench
    cycle, n_train = 93, 15
    for K in (20, 93, 186, 100, 7):
        w = bench.pick_windows(K, cycle)
        assert 5 <= w <= 24
        trains, k = [], cycle + 5          # pre-heat + warm-up
        for _ in range(w):
            trains.append(sum(bench.is_train(k + i, n_train, cycle) for i in range(K)))
            k += K
        t_train, t_sample = 0.6, 0.054
        ms = [n * t_train + (K - n) * t_sample for n in trains]
        ms[0] *= 1.5                       # one slow outlier window must not move the figure much
        dt, kinds = bench.combine_windows(ms, trains, K)
        exact = K * (n_train * t_train + (cycle - n_train) * t_sample) / cycle
        assert abs(kinds["train"] / K - n_train / cycle) <= 0.012, (K, w, trains)
        assert abs(dt * 1e3 - exact) <= 0.03 * exact, (K, w, dt * 1e3, exact)


def test_shard_rows_partition():
    from sdrm_amd.parallel import shard_rows
    for n, w in [(8192, 8), (5429, 8), (7, 3), (3, 8)]:
        spans = [shard_rows(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(s[1] for s in spans) == n
        for (a, la), (b, _) in zip(spans, spans[1:]):
            assert a + la == b
        assert max(s[1] for s in spans) - min(s[1] for s in spans) <= 1


def test_tile_coordinate_magic_is_exact_inside_its_range():
    """csrc/gemm.h takes a work-group's tile coordinates from n / d = umulhi(n, floor(2^32 / d) + 1): exact while n * d < 2^32,
    the bound gemm_set_grid() enforces on every launch (tiles x tiles_n, and slices x tiles^2 for the split-K unit index)."""
    rng = np.random.RandomState(0)

    def magic(d):
        return 0 if d <= 1 else ((1 << 32) // d + 1) & 0xFFFFFFFF

    def div(n, m):
        return (n * m) >> 32 if m else n

    for d in list(range(1, 70)) + [114, 1352, 4096, 65535, 65536, 92681]:
        m = magic(d)
        top = ((1 << 32) - 1) // d          # largest n with n * d < 2^32
        ns = np.unique(np.concatenate([np.arange(0, min(top, 4096) + 1), rng.randint(0, top + 1, 4096), [top, max(top - 1, 0)]]))
        for n in ns.tolist():
            assert div(n, m) == n // d, (n, d)
    # just past the bound the quotient can be one too large: that is why the launch helpers refuse such grids
    assert any(div(n, magic(d)) != n // d for d in (3, 7, 11) for n in range(((1 << 32)) // d * d - 1, (1 << 32), 1) if n < (1 << 32))
