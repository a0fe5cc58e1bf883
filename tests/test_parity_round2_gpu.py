"""GPU parity tests added in round 2 (all through the C ABI):

* the bf16 fast-path kernels at the SHAPES THE BENCH RUNS (attention at B x T x H = 2 x 1000 x 16 and 1 x 2000 x 16;
  GEMMs at M = 16000 / 10240 with the tile configuration the library picks itself) against fp64 references;
* the reference's MODULE-LEVEL golden vectors (tests/golden/modules.npz) through the drop-in module objects;
* ``LVTRTrainer._training_loop`` and ``validation_step`` against the reference step golden;
* ``LVTR.likelihood`` against the reference (tests/golden/extras_c1.npz);
* hipGraph replay of ragged batches against eager launches on the 2nd and 3rd batch of one padded shape;
* one full-configuration bf16 training step at T = 2000 (BASELINE config 5).
"""
import copy

import numpy as np
import pytest
import torch

from test_kernels_gpu import attn_reference, dev, lengths_for, rnd, row_mask
from test_model_parity_gpu import SEED, build_model, make_inputs, rel, small_model_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    import hipvg
    hipvg.lib()
    from hipvg import functional
    return functional


# ------------------------------------------------------------------ attention at the bench shapes
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 1000, 16), (1, 2000, 16)], ids=["B2-T1000-H16", "B1-T2000-H16"])
def test_attention_fwd_bwd_at_bench_shapes(F, dtype, shape):
    """Forward and backward of the flash-style kernels against the dense fp64 reference at the sequence lengths and
    head count of BASELINE configs 2 and 5 (the bf16 kernels stage K/V by LDS-DMA, the fp32 ones through registers:
    different code, and 16 heads x 16 tiles per sequence in longest-first order)."""
    B_, T, H = shape
    D = H * 64
    lens = lengths_for(B_, T)
    slopes = torch.tensor(F.alibi_slopes(H), dtype=torch.float32, device=dev())
    qkv = rnd(B_ * T, 3 * D, dtype=dtype, seed=T).requires_grad_(True)
    out = F.attention(qkv, slopes, B_, T, H, lens)
    mask = row_mask(lens, T)[:, None]
    go = torch.where(mask, rnd(B_ * T, D, seed=3), 0.0)
    (out.float() * go).sum().backward()
    qr = qkv.detach().double().requires_grad_(True)
    ref = attn_reference(qr, B_, T, H, lens, slopes)
    (ref * go.double()).sum().backward()
    if dtype == torch.float32:
        torch.testing.assert_close(out.double(), ref.detach(), atol=2e-5, rtol=2e-5)
        torch.testing.assert_close(qkv.grad.double(), qr.grad, atol=1e-4, rtol=1e-4)
    else:
        # bf16 operands and probabilities: 2^-8 relative per element; errors are judged against the tensor's scale
        assert (out.double() - ref.detach()).abs().max() / ref.detach().abs().max() < 2e-2
        assert (qkv.grad.double() - qr.grad).abs().max() / qr.grad.abs().max() < 3e-2
        assert (qkv.grad.double() - qr.grad).norm() / qr.grad.norm() < 1e-2


# ------------------------------------------------------------------ GEMMs at the bench shapes, library-chosen tiles
@pytest.mark.parametrize("M", [16000, 13312, 10240])
def test_gemm_layer_shapes_exact_small_integers(F, M):
    """The layer's products at the row counts of the step (16 x 1000 and 16 x 640 frames; 13,312 = the packed rows of the
    ragged bench at 77 % fill, where the N = 4096 products run as two launches: whole rounds + the remaining row band,
    round 6), tile_cfg = 0 (whatever
    pick_cfg selects: 256x256 phase-pipelined, 192x256, 128x128 + split-K), on small-integer operands whose exact
    result is representable: any mis-mapped fragment, dropped K tile or lost split-K slice changes it."""
    g = torch.Generator().manual_seed(M)
    D, Fd = 1024, 4096

    def ints(*s):
        return torch.randint(-2, 3, s, generator=g).to(dev()).bfloat16()

    x, h = ints(M, D), ints(M, Fd)
    w1, w2 = ints(Fd, D), ints(D, Fd)
    # NT forward: x @ W1^T (K = 1024) and h @ W2^T (K = 4096)
    for A, W, N, K in ((x, w1, Fd, D), (h, w2, D, Fd)):
        out = F.gemm(A, W, M, N, K, out_f32=True)
        assert torch.equal(out.double(), A.double() @ W.double().t())
    # NN dgrad: dh = dy @ W2 (K = 1024) and dx = dh @ W1 (K = 4096)
    for A, W, N, K in ((x, w2, Fd, D), (h, w1, D, Fd)):
        out = F.gemm(A, W, M, N, K, b_tr=True, out_f32=True)
        assert torch.equal(out.double(), A.double() @ W.double())
    # TN wgrad: dW1 = dh^T @ x with the split the library's heuristic picks (reduction over the frames)
    sk = F.wgrad_splits(Fd, D, M, torch.bfloat16)
    out = F.gemm(h, x, Fd, D, M, a_tr=True, b_tr=True, out_f32=True, split_k=sk)
    assert torch.equal(out.double(), h.double().t() @ x.double())
    out = F.gemm(x, x, D, D, M, a_tr=True, b_tr=True, out_f32=True, split_k=F.wgrad_splits(D, D, M, torch.bfloat16))
    assert torch.equal(out.double(), x.double().t() @ x.double())


@pytest.mark.parametrize("M", [16000, 13312, 10240])
def test_gemm_layer_shapes_random_vs_fp64(F, M):
    """Same products on random bf16 operands with their real epilogues, against fp64 on the same bf16 inputs."""
    g = torch.Generator().manual_seed(M + 1)
    D, Fd = 1024, 4096
    x = torch.randn(M, D, generator=g).to(dev()).bfloat16()
    w1 = (torch.randn(Fd, D, generator=g) * D ** -0.5).to(dev()).bfloat16()
    w2 = (torch.randn(D, Fd, generator=g) * Fd ** -0.5).to(dev()).bfloat16()
    b1 = torch.randn(Fd, generator=g).to(dev())
    # FFN-in: bias + exact GELU (+ stored derivative)
    aux = torch.empty(M, Fd, device=dev(), dtype=torch.bfloat16)
    hdn = F.gemm(x, w1, M, Fd, D, bias=b1, act=2 | 16, aux_out=aux)
    pre = x.double() @ w1.double().t() + b1.double()
    ref = torch.nn.functional.gelu(pre)
    assert (hdn.double() - ref).abs().max() < 2e-2 * max(1.0, float(ref.abs().max()))
    assert (hdn.double() - ref).norm() / ref.norm() < 4e-3
    # FFN-out: K = 4096 with the residual epilogue
    y = F.gemm(hdn, w2, M, D, Fd, residual=x)
    ref2 = hdn.double() @ w2.double().t() + x.double()
    assert (y.double() - ref2).norm() / ref2.norm() < 4e-3
    # dgrad with the stored-derivative epilogue and wgrad (fp32, split-K)
    dh = F.gemm(y, w2, M, Fd, D, b_tr=True, dact=4, aux_in=aux)
    ref3 = (y.double() @ w2.double()) * aux.double()
    assert (dh.double() - ref3).norm() / ref3.norm() < 4e-3
    dw = F.gemm(dh, x, Fd, D, M, a_tr=True, b_tr=True, out_f32=True, split_k=F.wgrad_splits(Fd, D, M, torch.bfloat16))
    ref4 = dh.double().t() @ x.double()
    assert (dw.double() - ref4).norm() / ref4.norm() < 1e-4


# ------------------------------------------------------------------ the reference's module-level vectors
def _fill(module, seed):
    from oracle.weights import fill_like
    sd = module.state_dict()
    filled = fill_like([(k, tuple(v.shape)) for k, v in sd.items()], seed)
    with torch.no_grad():
        for k, arr in filled.items():
            sd[k].copy_(torch.from_numpy(arr))
    return module.cuda()


def test_module_objects_match_reference_module_vectors(golden):
    """tests/golden/modules.npz (outputs of the REFERENCE's RMSNorm, SelfAttention, TransformerLayer,
    GaussianParameterize, masked_ce_loss and masked_loss) through the drop-in module objects in fp32 mode, at the
    tolerances the CPU oracle is held to (tests/test_oracle_golden.py::test_modules)."""
    import hipvg
    from hparams.hp import Hparams
    from modules.attention.attention import SelfAttention
    from modules.linear.layers import GaussianParameterize
    from modules.norm import RMSNorm
    from modules.position.alibi import ALiBi
    from modules.transformer.layers import TransformerLayer
    from training_lib.losses import masked_ce_loss, masked_loss
    from utils.tensormask import TensorMask
    hipvg.set_precision("fp32")
    g = golden("modules")
    x = torch.from_numpy(g["x"]).cuda()
    B, T, D = x.shape
    H = 4
    lengths = torch.from_numpy(g["lengths"]).cuda()
    mask = torch.arange(T, device=x.device)[None] < lengths[:, None]
    xm = TensorMask(x, mask).apply_mask()
    # RMSNorm (modules/norm.py:22-32)
    rn = _fill(RMSNorm(D, eps=1e-6), 11)
    np.testing.assert_allclose(rn(x).detach().cpu().numpy(), g["rmsnorm_y"], atol=2e-6, rtol=2e-6)
    # SelfAttention with ALiBi (modules/attention/attention.py:36-93)
    sa = _fill(SelfAttention(D, Hparams.from_dict(dict(nheads=H, causal=True))), 12)
    alibi = ALiBi(H, 64).cuda()
    o = sa(xm, rpe_pair=("ALiBi", alibi), return_kv=True)
    np.testing.assert_allclose(o["output"].value.detach().cpu().numpy(), g["attn_y"], atol=5e-6, rtol=2e-5)
    assert o["kv"]["key"].shape == (B, T, D)
    # TransformerLayer (modules/transformer/layers.py:41-93): the fused layer function
    lhp = Hparams.from_dict(dict(dim=D, ffd_size=512, norm=dict(identifier="RMSNorm", eps=1e-6),
                                 activation=dict(identifier="GELU"), self_attn=dict(nheads=H, causal=True)))
    tl = _fill(TransformerLayer(lhp), 13)
    y = tl(xm, rpe_pair=("ALiBi", alibi))["output"].value
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["layer_y"], atol=3e-5, rtol=2e-5)
    # ... and its decode / debug path (module by module) gives the same layer
    y2 = tl(xm, rpe_pair=("ALiBi", alibi), return_kv=True)["output"].value
    np.testing.assert_allclose(y2.detach().cpu().numpy(), g["layer_y"], atol=3e-5, rtol=2e-5)
    # GaussianParameterize with injected noise (modules/linear/layers.py:54-134)
    gp = _fill(GaussianParameterize(D, 4), 14)
    out = gp(xm, temperature=0.85, noise=torch.from_numpy(g["gauss_eps"]).cuda())
    np.testing.assert_allclose(out.mean.value.detach().cpu().numpy(), g["gauss_mean"], atol=3e-6)
    np.testing.assert_allclose(out.logstd.value.detach().cpu().numpy(), g["gauss_logstd"], atol=3e-6)
    np.testing.assert_allclose(out.sample.value.detach().cpu().numpy(), g["gauss_sample"], atol=1e-5)
    # losses (training_lib/losses.py:9-41)
    ce = masked_ce_loss(TensorMask(torch.from_numpy(g["ce_logits"]).cuda(), mask),
                        TensorMask(torch.from_numpy(g["ce_target"]).cuda(), mask))
    assert rel(ce, g["ce_sum"]) < 1e-6
    ml = masked_loss(TensorMask(torch.from_numpy(g["ml_a"]).cuda(), mask), TensorMask(torch.from_numpy(g["ml_b"]).cuda(), mask),
                     fn=lambda p, q: p - q)
    assert rel(ml, g["ml_sum"]) < 1e-5


# ------------------------------------------------------------------ the trainer object against the reference step
def _trainer_c1(full_cfg, graph=False, precision="fp32", accumulation=1, coalesce=False):
    from hparams.hp import Hparams
    from oracle.lvtr_oracle import small_config
    from oracle.weights import fill_like
    from trainers.speech.lvtr import LVTRTrainer
    cfg = copy.deepcopy(full_cfg)
    cfg["model"] = small_config(cfg["model"])
    cfg["hip"]["precision"] = precision
    cfg["hip"]["graph"] = graph
    cfg["hip"]["coalesce_accumulation"] = coalesce
    cfg["training"]["gradient_accumulation"] = accumulation
    tr = LVTRTrainer(Hparams.from_dict(cfg))
    sd = tr.model.state_dict()
    filled = fill_like([(k, tuple(v.shape)) for k, v in sd.items()], SEED)
    with torch.no_grad():
        for k, arr in filled.items():
            sd[k].copy_(torch.from_numpy(arr))
    tr = tr.cuda()
    tr.configure_optimizers()
    tr.attach_reducer()
    return tr


def _golden_batch(g):
    from utils.tensormask import TensorMask
    d = torch.device("cuda:0")
    lengths = torch.from_numpy(g["in_lengths"]).to(d)
    T = g["in_tokens"].shape[1]
    mask = torch.arange(T, device=d)[None] < lengths[:, None]
    batch = {"tokens": TensorMask(torch.from_numpy(g["in_tokens"]).to(d), mask),
             "mel": TensorMask(torch.from_numpy(g["in_mel"]).to(d), mask),
             "cropped_mel_utt": TensorMask(torch.from_numpy(g["in_utt"]).to(d))}
    noise = dict(eps_q=torch.from_numpy(g["noise_eps_q"]).to(d),
                 init_state=torch.from_numpy(g["noise_init_rand"]).to(d) * 2 - 1,
                 t_diff=torch.from_numpy(g["noise_t_diff"]).to(d),
                 eps_diff=torch.from_numpy(g["noise_eps_diff"]).to(d))
    return batch, noise, mask


def test_training_loop_matches_reference_step(golden, full_cfg):
    """``LVTRTrainer._training_loop(batch, idx, noise)`` itself (input packing, KL weight, loss assembly, the
    monitors; reference trainers/speech/lvtr.py:103-145) on the step_c1 golden: loss, kld, rec_loss, token_kld and
    the logged means, plus the gradients its backward leaves in the buckets."""
    g = golden("step_c1")
    tr = _trainer_c1(full_cfg)
    tr.global_step = 10 ** 9                       # past the KL warm-up: weight = fixed_beta, as in the golden
    batch, noise, mask = _golden_batch(g)
    out = tr._training_loop(batch, 0, noise)
    assert abs(out["kld_weight"] - float(g["kld_weight"])) < 1e-12
    for mine, key in ((out["loss"], "loss"), (out["kld"], "kld"), (out["rec_loss"], "rec_loss"),
                      (out["token_kld"], "ce_loss")):
        assert rel(mine, g[key]) < 1e-4, (key, float(mine), float(g[key]))
    for mine, key in ((out["logstd"], "logstd"), (out["q_logstd"], "q_logstd"), (out["q_mean_abs"], "q_mean_abs")):
        assert abs(float(mine) - float(g[key])) < 2e-5, key
    n = int(g["in_lengths"].sum())
    assert int(out["length"]) == n
    # 'log_p' / 'log_q' monitors are -TensorMask.mean(): sum over valid frames of the per-frame channel mean / frames
    m = mask.cpu().numpy()
    for mine, key in ((out["log_p"], "log_p"), (out["log_q"], "log_q")):
        want = -(g[key][m].mean(-1).sum() / n)
        assert abs(float(mine) - want) < 1e-4 * max(1.0, abs(want)), key
    keys = list(g["keys"])
    grads = dict(tr.model.named_parameters())
    gn = np.array([float(grads[k].grad.double().norm()) for k in keys])
    ref = g["grad_norm"]
    big = ref > 1e-6 * ref.max()
    assert np.max(np.abs(gn[big] - ref[big]) / ref[big]) < 1e-3


def test_validation_step_matches_reference_scalars(golden, full_cfg):
    """``validation_step`` + ``on_validation_end`` (reference :182-286): val/kld, val/rec_loss, val/token_kld are
    the golden sums per valid frame."""
    g = golden("step_c1")
    tr = _trainer_c1(full_cfg)
    batch, noise, _ = _golden_batch(g)
    tr.on_validation_start()
    tr.validation_step(batch, 0, noise)
    tr.validation_step(batch, 1, noise)
    logged = tr.on_validation_end()
    n = float(g["in_lengths"].sum())
    for name, key in (("val/kld", "kld"), ("val/rec_loss", "rec_loss"), ("val/token_kld", "ce_loss")):
        assert abs(logged[name] - float(g[key]) / n) < 1e-4 * abs(float(g[key]) / n), name
    assert all(p.grad is None or float(p.grad.abs().sum()) == 0.0 for p in tr.model.parameters())


def test_likelihood_matches_reference(golden, full_cfg):
    """``LVTR.likelihood`` (reference models/speech/lvtr.py:337-388) on the step_c1 batch, temperature 0."""
    g, e = golden("step_c1"), golden("extras_c1")
    model, _ = build_model(small_model_cfg(full_cfg), "fp32")
    x, _, noise, _ = make_inputs(g)
    with torch.no_grad():
        ll = model.likelihood(x, temperature=0.0, init_state=noise["init_state"])
    np.testing.assert_allclose(ll.double().cpu().numpy(), e["likelihood"], rtol=1e-5)


def _fixed_random_draws(monkeypatch, seed=11):
    """Replace torch.randn / randn_like / rand / randint by draws from per-shape tables (generated once from a CPU
    generator), so that eager and captured runs -- and two trainers -- see the same noise.  A table is created on its
    first use (an H2D copy): run an eager pass over every shape before capturing."""
    d = torch.device("cuda:0")
    table = {}
    cpu = torch.Generator().manual_seed(seed)
    o_randn, o_rand, o_randint = torch.randn, torch.rand, torch.randint

    def fixed(kind, shape, gen_fn):
        key = (kind, tuple(int(s) for s in shape))
        if key not in table:
            table[key] = gen_fn().to(d)
        return table[key].clone()

    def shape_of(s):
        return tuple(s[0]) if len(s) == 1 and isinstance(s[0], (tuple, list, torch.Size)) else tuple(s)

    monkeypatch.setattr(torch, "randn", lambda *s, **kw: fixed("randn", shape_of(s), lambda: o_randn(*shape_of(s), generator=cpu)))
    monkeypatch.setattr(torch, "randn_like", lambda x, **kw: fixed("randn", x.shape, lambda: o_randn(*x.shape, generator=cpu)).to(x.dtype))
    monkeypatch.setattr(torch, "rand", lambda *s, **kw: fixed("rand", shape_of(s), lambda: o_rand(*shape_of(s), generator=cpu)))
    monkeypatch.setattr(torch, "randint", lambda lo, hi, size, **kw: fixed("randint", size, lambda: o_randint(lo, hi, tuple(size), generator=cpu)))


# ------------------------------------------------------------------ hipGraph replay with changing lengths
def test_ragged_graph_replay_uses_each_batch_lengths(full_cfg, monkeypatch):
    """Three ragged batches that pad to the SAME graph shape but have different sequence lengths: replaying the
    captured micro-step must mask with each batch's own lengths (they are recomputed inside the graph), i.e. give
    what eager launches give.  The random draws are replaced by fixed tables in both modes so that the two runs see
    the same noise."""
    from training_lib.synthetic import make_batch
    from utils.tensormask import TensorMask
    d = torch.device("cuda:0")

    def ragged(seed, lens, T=128):
        b = make_batch(len(lens), T, d, seed=seed)
        mask = torch.arange(T, device=d)[None] < torch.tensor(lens, device=d)[:, None]
        return {"tokens": TensorMask(b["tokens"].value, mask), "mel": TensorMask(b["mel"].value, mask),
                "cropped_mel_utt": b["cropped_mel_utt"]}

    batches = [ragged(1, [128, 90, 64]), ragged(2, [70, 128, 33]), ragged(3, [128, 5, 101])]
    _fixed_random_draws(monkeypatch)
    results = {}
    for mode in ("eager", "graph"):
        tr = _trainer_c1(full_cfg, graph=(mode == "graph"))
        tr.global_step = 10 ** 9
        outs = []
        for i, b in enumerate(batches):
            o = tr._graphed_micro_step(b, i, True) if mode == "graph" else tr._training_loop(b, i)
            grads = torch.cat([bk["flat"] for bk in tr.reducer.buckets]).clone()
            outs.append((float(o["loss"]), float(o["kld"]), float(o["token_kld"]), float(o["rec_loss"]), int(o["length"]), grads))
            tr.reducer.zero_grad()
        results[mode] = outs
    for i, (e, gph) in enumerate(zip(results["eager"], results["graph"])):
        assert e[4] == gph[4] == sum(int(v) for v in batches[i]["mel"].mask.sum(-1)), (i, e[4], gph[4])
        for a, b in zip(e[:4], gph[:4]):
            assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (i, e[:4], gph[:4])
        assert (e[5] - gph[5]).norm() <= 1e-4 * e[5].norm(), i


def test_new_graph_shape_mid_window_keeps_accumulated_gradients(full_cfg, monkeypatch):
    """A padded shape that first appears on the SECOND micro-batch of an accumulation window (warm-up pass and
    capture happen there) must not discard the gradients the first micro-batch left in the buckets: afterwards the
    buckets hold the first micro-batch's gradient PLUS the new batch's (same noise tables in both trainers)."""
    from training_lib.synthetic import make_batch
    d = torch.device("cuda:0")
    b64, b128 = make_batch(2, 64, d, seed=2), make_batch(2, 128, d, seed=3)
    _fixed_random_draws(monkeypatch)
    flat = lambda t: torch.cat([b["flat"] for b in t.reducer.buckets]).clone()
    # what each batch contributes on its own (eager launches; also fills the noise tables before any capture)
    ref = _trainer_c1(full_cfg, graph=False, accumulation=2)
    ref.global_step = 10 ** 9
    ref._training_loop(b64, 0)
    g64 = flat(ref)
    ref.reducer.zero_grad()
    ref._training_loop(b128, 1)
    g128 = flat(ref)
    tr = _trainer_c1(full_cfg, graph=True, accumulation=2)
    tr.global_step = 10 ** 9
    tr._graphed_micro_step(b64, 0, False)              # window start: shape captured here
    kept = flat(tr)
    assert (kept - g64).norm() <= 1e-4 * g64.norm()
    tr._graphed_micro_step(b128, 1, False)             # new shape in mid-window: warm-up pass + capture + first replay
    after = flat(tr)
    assert (after - (g64 + g128)).norm() <= 1e-4 * (g64 + g128).norm()


# ------------------------------------------------------------------ BASELINE config 5: full model, bf16, T = 2000
def test_full_config_bf16_step_at_T2000(full_cfg):
    """One training micro-step of the FULL configuration at seq_len 2000 in bf16 (finite loss and gradients), with its
    loss terms next to the fp32 path on the same weights, batch and noise: KL within 5 %, CE and reconstruction within
    1 % (the reference's own fp32 vs 16-bit drift is of this size)."""
    from training_lib.synthetic import make_batch
    d = torch.device("cuda:0")
    B, T = 2, 2000
    batch = make_batch(B, T, d, seed=77)
    g = torch.Generator().manual_seed(5)
    noise = dict(eps_q=torch.randn(B, T, 4, generator=g).to(d), init_state=(torch.rand(B, 1, 64, generator=g) * 2 - 1).to(d),
                 t_diff=torch.randint(0, 1000, (B,), generator=g).to(d), eps_diff=torch.randn(B, T, 80, generator=g).to(d))
    terms = {}
    for precision in ("fp32", "bf16"):
        model, _ = build_model(full_cfg["model"], precision)
        x = batch["tokens"].expand().cat(batch["mel"])
        out = model(x, utterance=batch["cropped_mel_utt"], noise=noise)
        loss = out["decoder_output"] + 0.04 * out["kld"] + 0.02 * out["ce_loss"]
        loss.backward()
        assert torch.isfinite(loss)
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
        terms[precision] = {k: float(out[k]) for k in ("kld", "ce_loss", "decoder_output")}
        del model, out, loss
        torch.cuda.empty_cache()
    print("T=2000 full config, fp32 vs bf16:", terms)
    assert rel(terms["bf16"]["kld"], terms["fp32"]["kld"]) < 5e-2
    assert rel(terms["bf16"]["ce_loss"], terms["fp32"]["ce_loss"]) < 1e-2
    assert rel(terms["bf16"]["decoder_output"], terms["fp32"]["decoder_output"]) < 1e-2
