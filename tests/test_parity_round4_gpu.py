"""Round-4 parity additions.

* Deferred weight gradients (``hipvg.functional.defer_vec_grads`` bracket, ``vg_gemm_grouped`` with up to 48 products):
  the queue of several backward nodes leaves as few large launches.  Exact on small integers (every partial sum is an
  exactly representable integer), so a (tile, K range) segment added twice, dropped, or two products racing on one
  gradient change the result:
    - four Transformer layers = 768 tiles = three whole rounds of 256 CUs (flushed by the whole-rounds rule),
    - ~240 tiles of mixed shapes = one nearly full round run as whole tiles (the plan's full-tail round),
    - the ragged small products of the heads (8 x 1024, 200 x 1024, 512 x 80 ...) inside such a launch,
    - a weight that comes back a second time inside one bracket (shared module: the queue must flush first).
  Reference: the nn.Linear / k = 1 Conv1d weight gradients autograd computes for
  modules/transformer/layers.py:52,79,82,151 and modules/conv/layers.py of the reference.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def F():
    import hipvg
    hipvg.lib()
    from hipvg import functional
    return functional


def _make(shapes, frames, g):
    """[(N, Ktot, col0, cols)] -> (items, weights, refs); a repeated (N, Ktot, id) addresses the same weight."""
    weights, items, refs = {}, [], {}
    for i, (N, Ktot, col0, cols, wid) in enumerate(shapes):
        key = (N, Ktot, wid)
        if key not in weights:
            w = torch.nn.Parameter(torch.zeros(N, Ktot, device=dev()))
            w.grad = torch.randint(-3, 4, (N, Ktot), generator=g).float().to(dev())
            weights[key] = w
            refs[key] = w.grad.double().clone()
        w = weights[key]
        dy = torch.randint(-2, 3, (frames, N), generator=g).float().to(dev()).bfloat16()
        x = torch.randint(-2, 3, (frames, cols), generator=g).float().to(dev()).bfloat16()
        refs[key][:, col0:col0 + cols] += dy.double().T @ x.double()
        items.append((w, dy, x, col0))
    return items, weights, refs


LAYER = [(4096, 1024), (1024, 4096), (3072, 1024), (1024, 1024)]


@pytest.mark.parametrize("frames", [4096, 16000])
def test_deferred_layer_groups_leave_in_whole_rounds(F, frames, monkeypatch):
    """Four layers' products queued node by node (tag "layer"): flushed as ONE launch of 768 tiles when the fourth
    arrives; a fifth layer stays queued until the bracket closes."""
    g = torch.Generator().manual_seed(frames)
    launches = []
    real = F._launch_wgrad_items
    monkeypatch.setattr(F, "_launch_wgrad_items", lambda items: (launches.append(len(items)), real(items))[1])
    shapes = [(N, K, 0, K, layer) for layer in range(5) for (N, K) in LAYER]
    items, weights, refs = _make(shapes, frames, g)
    F.defer_vec_grads(True)
    try:
        for layer in range(5):
            F.sink_wgrad_group(items[4 * layer:4 * layer + 4], tag="layer")
            assert launches == ([] if layer < 3 else [16])
    finally:
        F.defer_vec_grads(False)
    assert launches == [16, 4]
    for key, w in weights.items():
        assert torch.equal(w.grad.double(), refs[key]), key


def test_deferred_mixed_products_as_one_nearly_full_round(F, monkeypatch):
    """The conv blocks' products (with a column slice) and the small, ragged products of the heads in one bracket:
    ~240 tiles -> one launch whose last round is run as whole tiles."""
    frames = 16000
    g = torch.Generator().manual_seed(7)
    shapes = []
    for blk in range(6):
        shapes += [(512, 2048, 0, 2048, blk), (2048, 608, 0, 512, blk), (2048, 608, 512, 96, blk)]
    shapes += [(512, 512, 0, 512, 100 + i) for i in range(4)]
    shapes += [(1024, 1024, 0, 1024, 200), (1024, 1024, 0, 1024, 201), (512, 1024, 0, 1024, 202),
               (200, 1024, 0, 1024, 203), (1024, 64, 0, 64, 204), (32, 192, 0, 192, 205), (80, 512, 0, 512, 206),
               (8, 1024, 0, 1024, 207), (512, 80, 0, 80, 208)]
    items, weights, refs = _make(shapes, frames, g)
    launches = []
    real = F._launch_wgrad_items
    monkeypatch.setattr(F, "_launch_wgrad_items", lambda its: (launches.append(len(its)), real(its))[1])
    F.defer_vec_grads(True)
    try:
        i = 0
        for blk in range(6):
            F.sink_wgrad_group(items[i:i + 3], fire=False)
            i += 3
        for it in items[i:]:
            F.sink_wgrad(it[0], it[1], it[2])
    finally:
        F.defer_vec_grads(False)
    assert launches == [len(items)], launches
    for key, w in weights.items():
        assert torch.equal(w.grad.double(), refs[key]), key


def test_a_weight_that_comes_back_flushes_the_queue_first(F, monkeypatch):
    """Two products into the same gradient must not share a launch (plain read-modify-write of whole tiles)."""
    frames = 2048
    g = torch.Generator().manual_seed(11)
    shapes = [(512, 2048, 0, 2048, 0), (2048, 512, 0, 512, 1), (512, 2048, 0, 2048, 0), (2048, 512, 0, 512, 1)]
    items, weights, refs = _make(shapes, frames, g)
    launches = []
    real = F._launch_wgrad_items
    monkeypatch.setattr(F, "_launch_wgrad_items", lambda its: (launches.append(len(its)), real(its))[1])
    F.defer_vec_grads(True)
    try:
        F.sink_wgrad_group(items[:2])
        F.sink_wgrad_group(items[2:])
        assert launches == [2]
    finally:
        F.defer_vec_grads(False)
    assert launches == [2, 2]
    for key, w in weights.items():
        assert torch.equal(w.grad.double(), refs[key]), key


def test_mixed_reduction_lengths_leave_in_separate_launches(F, monkeypatch):
    g = torch.Generator().manual_seed(13)
    a, wa, ra = _make([(512, 2048, 0, 2048, 0), (2048, 512, 0, 512, 1)], 4096, g)
    b, wb, rb = _make([(512, 2048, 0, 2048, 2), (2048, 512, 0, 512, 3)], 2048, g)
    launches = []
    real = F._launch_wgrad_items
    monkeypatch.setattr(F, "_launch_wgrad_items", lambda its: (launches.append(len(its)), real(its))[1])
    F.defer_vec_grads(True)
    try:
        F.sink_wgrad_group(a)
        F.sink_wgrad_group(b)
    finally:
        F.defer_vec_grads(False)
    assert sorted(launches) == [2, 2]
    for weights, refs in ((wa, ra), (wb, rb)):
        for key, w in weights.items():
            assert torch.equal(w.grad.double(), refs[key]), key


def test_masked_means_match_tensormask_mean(F):
    """vg_masked_means == TensorMask(x, mask).mean() / .abs().mean() (utils/tensormask.py:135-140 of the reference) for
    strided column slices, ragged lengths, an empty sequence and the no-mask case."""
    from utils.tensormask import TensorMask
    g = torch.Generator().manual_seed(5)
    B, T = 5, 333
    lengths = torch.tensor([333, 200, 0, 1, 77])
    mask = (torch.arange(T)[None] < lengths[:, None]).to(dev())
    both = torch.randn(B, T, 8, generator=g).to(dev())
    four = torch.randn(B, T, 4, generator=g).to(dev())
    lens32 = lengths.to(dev()).int()
    items = [(both.view(B * T, 8)[:, 4:], False), (both.view(B * T, 8)[:, :4], False), (four.view(B * T, 4), False),
             (four.view(B * T, 4), True)]
    got = F.masked_means(items, lens32, T)
    want = [TensorMask(both[..., 4:], mask).mean(), TensorMask(both[..., :4], mask).mean(), TensorMask(four, mask).mean(),
            TensorMask(four, mask).abs().mean()]
    for a, b in zip(got, want):
        assert abs(float(a) - float(b)) <= 1e-6 + 1e-5 * abs(float(b)), (float(a), float(b))
    got = F.masked_means(items[2:], None, T)
    want = [TensorMask(four).mean(), TensorMask(four).abs().mean()]
    for a, b in zip(got, want):
        assert abs(float(a) - float(b)) <= 1e-6 + 1e-5 * abs(float(b)), (float(a), float(b))


@pytest.mark.parametrize("fused", ["0", "1"])
def test_full_config_decode_session_fp32_matches_oracle(full_cfg, fused, monkeypatch):
    """BASELINE config 4 at the size the decode bench runs it (VERDICT r03 "What's weak" 1): the FULL model (L = 16,
    d = 1024, H = 16), a 150-frame prompt through ``DecodeSession.prefill`` and 24 teacher-forced single-frame steps on
    the pre-allocated KV cache, both decode paths (five launches per layer / the fused attention sub-layer), fp32,
    against ``oracle.lvtr_oracle`` run with its own KV cache on the CPU (reference: models/speech/lvtr.py:227-286,
    modules/attention/attention.py:56-73).  Tolerances of ``test_decode_session_fp32_matches_reference``."""
    import copy

    import numpy as np

    import hipvg
    from hparams.hp import Hparams
    from inference.speech.session import DecodeSession
    from models.speech.lvtr import LVTR
    from oracle import lvtr_oracle as O
    from oracle.weights import fill_like
    monkeypatch.setenv("VG_DECODE_FUSED", fused)
    cfg = full_cfg["model"]
    rng = np.random.default_rng(404)
    B, Tp, n = 4, 150, 24
    x = torch.cat([torch.from_numpy(rng.integers(0, 200, (B, Tp + n, 1))).float(),
                   torch.from_numpy(rng.standard_normal((B, Tp + n, 4)).astype(np.float32))], -1)
    init = torch.from_numpy(rng.random((B, 1, 64)).astype(np.float32)) * 2 - 1
    filled = fill_like(O.param_shapes(cfg), 20250620)
    sd = {k: torch.from_numpy(v) for k, v in filled.items()}
    torch.set_num_threads(min(32, torch.get_num_threads() if torch.get_num_threads() > 1 else 32))
    # ---- oracle: prefill (init state pushed in front) + single-frame steps with the KV cache
    tr = cfg["transformer"]
    past, want = None, dict(lat=[], mean=[], logstd=[], logits=[])
    with torch.no_grad():
        for i in range(n + 1):
            xi = x[:, :Tp] if i == 0 else x[:, Tp + i - 1: Tp + i]
            tok = sd["token_embedding.weight"][xi[..., 0].long()]
            fused_in = tok + torch.relu(O.dense(sd, "token_fuser.linear", xi[..., 1:]))
            if i == 0:
                fused_in = torch.cat([init, fused_in], 1)
            m = torch.ones(B, fused_in.shape[1], dtype=torch.bool)
            hT, past, _ = O.transformer_stack(sd, "transformer.0", fused_in, m, tr, past)
            c = torch.relu(O.dense(sd, "q_spliter.linear", hT))
            want["lat"].append(hT[:, -1])
            want["mean"].append(O.dense(sd, "transformer.1.mean", c)[:, -1])
            want["logstd"].append(O.dense(sd, "transformer.1.logstd", c)[:, -1])
            want["logits"].append(O.dense(sd, "token_predictor.linear",
                                          torch.relu(O.dense(sd, "token_spliter.linear", hT)))[:, -1])
    # ---- the HIP session
    hipvg.set_precision("fp32")
    model = LVTR(Hparams.from_dict(copy.deepcopy(cfg)), input_dim=80)
    model.load_state_dict(sd, strict=False)
    model = model.cuda().eval()
    xd, zeros = x.to(dev()), torch.zeros(B, 4, device=dev())
    sess = DecodeSession(model, B, Tp + n + 2, use_graph=False, keep_latent=True)
    assert sess._fused == (fused == "1")
    sess.prefill(xd[:, :Tp], init_state=init.to(dev()), noise=torch.zeros(B, Tp + 1, 4, device=dev()))
    got = dict(lat=[sess._last["transformer_latent"][:, -1].float()], mean=[sess._last["prior"].mean.value[:, -1].float()],
               logstd=[sess._last["prior"].logstd.value[:, -1].float()], logits=[sess._last["logits"][:, -1].float()])
    for i in range(1, n + 1):
        sess.force_frame(xd[:, Tp + i - 1: Tp + i])
        sess.step(noise=zeros)
        got["lat"].append(sess._last["transformer_latent"][:, 0].float())
        got["mean"].append(sess._last["mu_ls"][:, 0, :4])
        got["logstd"].append(sess._last["mu_ls"][:, 0, 4:])
        got["logits"].append(sess._last["logits"][:, 0])
    for key, tol in (("lat", 1e-4), ("mean", 1e-4), ("logstd", 1e-4), ("logits", 5e-4)):
        np.testing.assert_allclose(torch.stack(got[key], 1).cpu().numpy(), torch.stack(want[key], 1).numpy(),
                                   atol=tol, rtol=2e-4, err_msg=key)
    # the layer-0 key cache holds the oracle's keys of all Tp + 1 + n frames
    cache = sess.kc[0][:, : Tp + 1 + n].float().cpu().numpy()
    np.testing.assert_allclose(cache[:, ::7, ::13], past[0][0].numpy()[:, ::7, ::13], atol=2e-5)


def test_fresh_pass_stores_and_second_contributions_accumulate(F):
    """vg_gemm_grouped with accumulate = 0 (hipvg.functional.begin_backward_pass(True): the gradients hold zeros): whole-K
    tiles are stored without reading C; a second product into the same region in the same pass (a module applied twice)
    accumulates; a product that took the single-launch route first is seen by the grouped one; a pass that is not fresh
    adds to what is there.  Exact on small integers."""
    frames = 4096
    g = torch.Generator().manual_seed(21)
    items, weights, refs = _make([(1024, 1024, 0, 1024, 0), (512, 2048, 0, 2048, 1), (2048, 608, 0, 512, 2),
                                  (2048, 608, 512, 96, 2)], frames, g)
    prods = {}
    for (w, dy, x, col0), key in zip(items, [(1024, 1024, 0), (512, 2048, 1), (2048, 608, 2), (2048, 608, 2)]):
        prods.setdefault(key, torch.zeros_like(weights[key].grad, dtype=torch.float64))[:, col0:col0 + x.shape[1]] += dy.double().T @ x.double()
    # fresh pass on zeroed gradients: the result is the product itself
    for w in weights.values():
        w.grad.zero_()
    F.begin_backward_pass(True)
    try:
        F.sink_wgrad_group(items)
        for key, w in weights.items():
            assert torch.equal(w.grad.double(), prods[key]), key
        F.sink_wgrad_group(items)                       # the same regions again in the same pass: accumulate
        for key, w in weights.items():
            assert torch.equal(w.grad.double(), 2 * prods[key]), key
    finally:
        F.end_backward_pass()
    # a product that went through the single-launch route first, then the same region in a grouped launch
    for w in weights.values():
        w.grad.zero_()
    F.begin_backward_pass(True)
    try:
        F.sink_wgrad(items[0][0], items[0][1], items[0][2])
        F.sink_wgrad_group(items)
    finally:
        F.end_backward_pass()
    assert torch.equal(weights[(1024, 1024, 0)].grad.double(), 2 * prods[(1024, 1024, 0)])
    assert torch.equal(weights[(512, 2048, 1)].grad.double(), prods[(512, 2048, 1)])
    # not fresh: adds to what is there
    before = {k: w.grad.double().clone() for k, w in weights.items()}
    F.begin_backward_pass(False)
    F.sink_wgrad_group(items)
    F.end_backward_pass()
    for key, w in weights.items():
        assert torch.equal(w.grad.double(), before[key] + prods[key]), key
