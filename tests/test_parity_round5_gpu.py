"""Round-5 parity tests (need an MI355X; everything enters through the C ABI):

* the bf16 GELU + stored-derivative epilogue (cheaper Phi, round 5) against float64 exact-erf GELU / GELU' on EVERY bf16
  pre-activation in [-9, 9] plus random tiles: within one bf16 ulp (reference: modules/activations.py:11,
  modules/transformer/layers.py:82);
* a NaN accumulator stays non-finite through the lean GEMM epilogues;
* the ALiBi window of the attention backward (vg_attn_fwd_stats / vg_attn_bwd_stats) against float64 dense attention at the
  bench shape, against the launch without the window, with ragged lengths, with large-norm q / k (window = everything)
  and with a steep-slope-only configuration (window = a few tiles); packed rows;
* 'fresh' store mode of the grouped weight gradients follows the real state of the gradient buffers:
  per-epoch batch indices that do not line up with the accumulation window, and a backward outside training_step.
"""
import copy
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    import hipvg
    hipvg.lib()
    from hipvg import functional
    return functional


def dev():
    return torch.device("cuda:0")


def _bf16_ulp(v: torch.Tensor) -> torch.Tensor:
    """bf16 ulp (8 significant bits) at the magnitude of the float64 values ``v``."""
    ex = torch.floor(torch.log2(v.abs().clamp_min(1e-300)))
    return torch.pow(torch.tensor(2.0, dtype=torch.float64, device=v.device), ex - 7)


def _all_bf16_values(lo_exp=-12, hi=9.0):
    """Every bf16 value with 2^lo_exp <= |x| <= hi, and zero."""
    bits = torch.arange(0, 1 << 15, dtype=torch.int32)
    vals = (bits << 16).view(torch.float32)
    vals = vals[(vals >= 2.0 ** lo_exp) & (vals <= hi)]
    return torch.cat([vals, -vals, torch.zeros(1)])


def test_gelu_epilogue_within_one_bf16_ulp(F):
    """h = GELU(u) and the stored GELU'(u) of the FFN-in launch (act = GELU | SAVE_DERIV, lean epilogue) for every bf16
    u in [-9, 9] (|u| >= 2^-12) and 60,000 random ones: |error| <= max(1 bf16 ulp of the exact value, floor), floors
    2^-17 for h and 2^-15 for GELU' (GELU' crosses zero at u = -0.75; both floors are the bf16 ulp of values of 2^-9 /
    2^-7, two orders below anything these tensors feed).  The pre-activation is produced exactly: u = X . I."""
    import hipvg
    grid = _all_bf16_values()
    N = 256
    g = torch.Generator().manual_seed(5)
    extra = (torch.randn(256 * N - grid.numel() % (256 * N), generator=g) * 2.5).bfloat16().float()
    u = torch.cat([grid, extra])
    M = u.numel() // N
    u = u[:M * N].view(M, N)
    x = u.to(dev()).bfloat16()
    eye = torch.eye(N, device=dev()).bfloat16()
    deriv = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    h = F.gemm(x, eye, M, N, N, act=F.ACT_GELU | F.ACT_SAVE_DERIV, aux_out=deriv)
    ud = x.double()
    phi = 0.5 * torch.erfc(-ud / math.sqrt(2.0))                       # exact in the negative tail too
    h_ref = ud * phi
    g_ref = phi + ud * torch.exp(-0.5 * ud * ud) / math.sqrt(2.0 * math.pi)
    eh = (h.double() - h_ref).abs()
    eg = (deriv.double() - g_ref).abs()
    tol_h = torch.maximum(_bf16_ulp(h_ref), torch.tensor(2.0 ** -17, dtype=torch.float64, device=dev()))
    tol_g = torch.maximum(_bf16_ulp(g_ref), torch.tensor(2.0 ** -15, dtype=torch.float64, device=dev()))
    worst_h, worst_g = float((eh / tol_h).max()), float((eg / tol_g).max())
    assert worst_h <= 1.0, f"GELU off by {worst_h:.2f} tolerances at u = {float(ud.flatten()[(eh / tol_h).argmax()])}"
    assert worst_g <= 1.0, f"GELU' off by {worst_g:.2f} tolerances at u = {float(ud.flatten()[(eg / tol_g).argmax()])}"
    # and where the values are of ordinary size the result is the correctly rounded one or its neighbour
    big = h_ref.abs() > 2.0 ** -6
    assert float((eh[big] / _bf16_ulp(h_ref[big])).max()) <= 1.0


def test_nan_stays_non_finite_through_lean_epilogues(F):
    """the plain lean epilogue clamps with v_max against a -inf floor, which turns a NaN accumulator into
    -inf instead of passing it through: accepted and documented -- what must hold is that a diverged value never comes
    out FINITE, for every lean epilogue variant."""
    M, N, K = 512, 256, 256
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).to(dev()).bfloat16()
    w = torch.randn(N, K, generator=g).to(dev()).bfloat16()
    x[7, 3] = float("nan")
    x[300, 100] = float("inf")
    res = torch.randn(M, N, generator=g).to(dev()).bfloat16()
    aux = torch.randn(M, N, generator=g).to(dev()).bfloat16()
    outs = {
        "plain": F.gemm(x, w, M, N, K),
        "relu": None,
        "residual": F.gemm(x, w, M, N, K, residual=res),
        "gelu+deriv": F.gemm(x, w, M, N, K, act=F.ACT_GELU | F.ACT_SAVE_DERIV, aux_out=torch.empty_like(res)),
        "x stored derivative": F.gemm(x, w, M, N, K, dact=F.ACT_STORED, aux_in=aux),
    }
    for name, y in outs.items():
        if y is None:
            continue
        assert not bool(torch.isfinite(y[7]).any()), f"{name}: the NaN row came out finite"
        assert not bool(torch.isfinite(y[300]).all()), f"{name}: the inf row came out finite"
        keep = torch.ones(M, dtype=torch.bool, device=dev())
        keep[7] = keep[300] = False
        assert bool(torch.isfinite(y[keep]).all()), f"{name}: a clean row is not finite"


# ---------------------------------------------------------------- attention: the ALiBi window of the backward
def _dense_attention(qkv, dout, slopes, B, T, H, lens):
    """float64 restatement of modules/attention/attention.py:60-77 with modules/position/alibi.py:9-33, and its
    gradient by autograd.  Returns out [B*T, D], dqkv [B*T, 3D] (rows past a sequence's length are zero)."""
    D = H * 64
    x = qkv.double().view(B, T, 3, H, 64).clone().requires_grad_(True)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)     # [B, H, T, 64]
    i = torch.arange(T, device=qkv.device)
    dist = (i[:, None] - i[None, :]).double()
    s = q @ k.transpose(-1, -2) / 8.0 - slopes.double()[None, :, None, None] * dist[None, None]
    ok = (i[None, :] <= i[:, None])[None, None] & (i[None, None, None, :] < lens[:, None, None, None])
    s = s.masked_fill(~ok, float("-inf"))
    p = torch.softmax(s, -1)
    rows = (i[None, :] < lens[:, None])                                                                 # [B, T]
    o = (p @ v).transpose(1, 2).reshape(B, T, D) * rows[..., None]
    (o * dout.double().view(B, T, D)).sum().backward()
    g = x.grad.view(B, T, 3 * D) * rows[..., None]
    return o.reshape(B * T, D).detach(), g.reshape(B * T, 3 * D)


def _run_attn(F, qkv, dout, slopes, B, T, H, lens, window=True):
    import hipvg
    D = H * 64
    out = torch.empty(B * T, D, dtype=qkv.dtype, device=dev())
    ws = F.attn_workspace(B, T, H, B * T, dev())
    dqkv = torch.full_like(qkv, float("nan"))
    delta = torch.empty(H, B * T, dtype=torch.float32, device=dev())
    F.attn_fwd_raw(qkv, out, ws, slopes, B, T, H, lens)
    old = os.environ.get("VG_ATTN_WINDOW")
    os.environ["VG_ATTN_WINDOW"] = "1" if window else "0"
    try:
        F.attn_bwd_raw(qkv, out, dout, ws, slopes, dqkv, delta, B, T, H, lens)
    finally:
        if old is None:
            os.environ.pop("VG_ATTN_WINDOW", None)
        else:
            os.environ["VG_ATTN_WINDOW"] = old
    torch.cuda.synchronize()
    return out, dqkv, ws


@pytest.mark.parametrize("case", ["bench shape", "ragged", "large norms", "steep only", "one tile"])
def test_attention_window_matches_dense_and_the_full_sweep(F, case):
    """bf16 attention with the round-5 window against float64 dense attention and against the same launch with the
    window switched off (VG_ATTN_WINDOW=0): the tiles the window drops are exactly those whose probabilities are below
    2^-20 of their row, so the two agree to far below bf16 resolution and both meet the dense reference."""
    H = 16
    std = 0.5
    slopes = torch.tensor(F.alibi_slopes(H), dtype=torch.float32, device=dev())
    # (B * H * ceil(T / 256) >= 512 selects the 256-query forward; below that the 128-query forward runs: the last case)
    if case == "bench shape":
        B, T, lens = 8, 1000, None
    elif case == "ragged":
        B, T = 8, 1000
        lens = torch.tensor([1000, 513, 64, 1, 999, 257, 256, 700], dtype=torch.int32, device=dev())
    elif case == "large norms":
        B, T, lens, std = 8, 1000, None, 3.0
    elif case == "steep only":
        B, T, lens = 8, 900, None
        slopes = torch.full((H,), 0.7071, dtype=torch.float32, device=dev())
    else:
        B, T, lens = 3, 64, None
    g = torch.Generator().manual_seed(T)
    qkv = (torch.randn(B * T, 3 * H * 64, generator=g) * std).to(dev()).bfloat16()
    dout = torch.randn(B * T, H * 64, generator=g).to(dev()).bfloat16()
    full = lens if lens is not None else torch.full((B,), T, dtype=torch.int32, device=dev())
    if lens is not None:
        rows = (torch.arange(T, device=dev())[None] < lens[:, None]).reshape(-1)
        dout = torch.where(rows[:, None], dout, torch.zeros_like(dout))
    o_ref, g_ref = _dense_attention(qkv, dout, slopes, B, T, H, full)
    out, dq_win, ws = _run_attn(F, qkv, dout, slopes, B, T, H, lens, window=True)
    _, dq_all, _ = _run_attn(F, qkv, dout, slopes, B, T, H, lens, window=False)
    # (logits of standard deviation 9 -- "large norms" -- carry the bf16 rounding of q and k straight into the
    # exponent: that case checks that nothing is dropped when the window covers everything, with a tolerance to match)
    to, tg = (dict(atol=3e-2, rtol=3e-2), dict(atol=6e-2, rtol=6e-2)) if case != "large norms" else \
             (dict(atol=0.6, rtol=0.1), dict(atol=4.0, rtol=0.2))
    torch.testing.assert_close(out.double(), o_ref, **to)
    torch.testing.assert_close(dq_win.double(), g_ref, **tg)
    torch.testing.assert_close(dq_all.double(), g_ref, **tg)
    scale = float(g_ref.abs().max())
    assert float((dq_win.double() - dq_all.double()).abs().max()) <= 2e-3 * scale, "the window changed the gradient"
    # the statistics are what they claim to be (sequence 0, every head): max |k|^2 over the valid keys
    import hipvg
    n = hipvg.lib().vg_attn_stats_floats(1, T, 1)
    stats = ws[H * B * T:].view(B * H, n)
    L0 = int(full[0])
    k = qkv.float().view(B, T, 3, H, 64)[0, :L0, 1]                      # [L0, H, 64]
    k2 = (k * k).sum(-1).max(0).values
    got = stats[:H, 0]
    last = ((L0 + 63) // 64) * 64                                         # the last tile's padding rows may take part
    assert bool((got >= k2 * 0.999).all()), "max |k|^2 below the true maximum: the window would not be conservative"
    kk = qkv.float().view(B, T, 3, H, 64)[0, :min(last, T), 1]
    assert bool((got <= (kk * kk).sum(-1).max(0).values * 1.001).all()), "the forward left no statistics"
    # ... and max |q|^2 over sequence 0's queries, reduced over the per-group entries (both forward kernels write them:
    # the 256-query one at B * H * ceil(T / 256) >= 512, the 128-query one below that -- the last case)
    n128 = (n - 4) // 8
    q = qkv.float().view(B, T, 3, H, 64)[0, :L0, 0]
    q2 = (q * q).sum(-1).max(0).values
    got_q = stats[:H, 4:4 + 4 * n128].max(1).values
    torch.testing.assert_close(got_q, q2, rtol=1e-3, atol=1e-4)


def test_attention_window_on_packed_rows(F):
    """vg_attn_*_stats on packed rows (cu_rows): the window applies per sequence; against the padded launch."""
    H, B, T = 16, 3, 640
    lens = torch.tensor([640, 300, 77], dtype=torch.int32, device=dev())
    slopes = torch.tensor(F.alibi_slopes(H), dtype=torch.float32, device=dev())
    g = torch.Generator().manual_seed(9)
    qkv = (torch.randn(B * T, 3 * H * 64, generator=g) * 0.5).to(dev()).bfloat16()
    dout = torch.randn(B * T, H * 64, generator=g).to(dev()).bfloat16()
    mask = (torch.arange(T, device=dev())[None] < lens[:, None]).reshape(-1)
    dout = torch.where(mask[:, None], dout, torch.zeros_like(dout))
    out, dq, _ = _run_attn(F, qkv, dout, slopes, B, T, H, lens, window=True)
    rows = F.pack_rows_bucket(int(lens.sum()), 256)
    plan = F.PackPlan(B, T, rows, dev()).fill(lens)
    qp, dop = F.pack_rows(qkv, plan), F.pack_rows(dout, plan)
    outp = torch.full((rows, H * 64), float("nan"), device=dev(), dtype=torch.bfloat16)
    dqp = torch.full((rows, 3 * H * 64), float("nan"), device=dev(), dtype=torch.bfloat16)
    ws = F.attn_workspace(plan.nseq, T, H, rows, dev())
    deltap = torch.empty(H, rows, device=dev())
    F.attn_fwd_raw(qp, outp, ws, slopes, plan.nseq, T, H, plan.lengths, plan.cu, rows)
    F.attn_bwd_raw(qp, outp, dop, ws, slopes, dqp, deltap, plan.nseq, T, H, plan.lengths, plan.cu, rows)
    n = int(lens.sum())
    assert torch.equal(outp[:n], out[mask]) and bool((outp[n:] == 0).all())
    scale = float(dq.float().abs().max())
    assert float((dqp[:n].float() - dq[mask].float()).abs().max()) <= 2e-3 * scale and bool((dqp[n:] == 0).all())


# ---------------------------------------------------------------- 'fresh' follows the state of the gradient buffers
def _small_trainer(full_cfg, graph: bool, accum: int):
    import hipvg
    from hparams.hp import Hparams
    from oracle.lvtr_oracle import small_config
    from trainers.speech.lvtr import LVTRTrainer
    cfg = copy.deepcopy(full_cfg)
    cfg["model"] = small_config(full_cfg["model"])
    cfg["training"]["gradient_accumulation"] = accum
    cfg.setdefault("hip", {})
    cfg["hip"].update(precision="bf16", graph=graph, coalesce_accumulation=False, bucket_mb=4, graph_bucket_mb=4)
    torch.manual_seed(11)
    tr = LVTRTrainer(Hparams.from_dict(cfg)).to(dev())
    tr.configure_optimizers()
    tr.attach_reducer()
    tr.global_step = cfg["training"]["scheduler"]["warmup_kld"]
    return tr


def _grads_at_step(tr, tag, sink):
    """Record the flat gradient the optimizer is about to consume."""
    orig = tr.optimizer.step

    def step(*args, **kw):
        torch.cuda.synchronize()
        sink[tag] = torch.cat([b["flat"].float().clone() for b in tr.reducer.buckets])
        return orig(*args, **kw)
    tr.optimizer.step = step


def _noise_for(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    d = dev()
    return dict(eps_q=torch.randn(B, T, 4, generator=g).to(d), init_state=(torch.rand(B, 1, 64, generator=g) * 2 - 1).to(d),
                eps_p=torch.zeros(B, T, 4, device=d), t_diff=torch.randint(0, 1000, (B,), generator=g).to(d),
                eps_diff=torch.randn(B, T, 80, generator=g).to(d))


def test_fresh_follows_the_buffers_not_the_batch_index(full_cfg):
    """ADVICE r04.  'The gradient buffers hold zeros' used to be inferred from batch_idx % window == 0; a backward pass
    outside training_step (or per-epoch batch indices that restart in the middle of a window) then met grouped
    weight-gradient launches that STORE and lost what had been accumulated.  Trainer `d` runs a window of two
    micro-batches the ordinary way; trainer `e` accumulates the first of them through a bare _training_loop call and
    then enters training_step with index 0 -- which the old rule read as "fresh" -- and a window of one: the gradients
    the two optimizers consume must agree.  Injected noise makes the two runs the same computation; M = 2 x 512 frames
    puts the weight gradients on the grouped launch."""
    from hipvg import functional as HF
    from training_lib.synthetic import make_batch
    B, T = 2, 512
    batches = [make_batch(B, T, dev(), seed=40 + i) for i in range(2)]
    noises = [_noise_for(B, T, 90 + i) for i in range(2)]
    seen = {}
    d = _small_trainer(full_cfg, False, 2)
    e = _small_trainer(full_cfg, False, 2)
    e.model.load_state_dict(d.model.state_dict())
    _grads_at_step(d, "d", seen)
    _grads_at_step(e, "e", seen)
    d.training_step(batches[0], 0, noise=noises[0])
    d.training_step(batches[1], 1, noise=noises[1])
    # the ordinary case still takes the store path: right after the optimizer cleared the buffers nothing has written
    assert d._clean_epoch == HF.write_epoch()
    e._training_loop(batches[0], 0, noises[0])       # a backward pass outside training_step
    assert d._clean_epoch != HF.write_epoch()        # (any library write, by whomever, ends "the buffers hold zeros")
    torch.cuda.synchronize()
    e.gradient_update_step = 1                       # the next call ends the window whatever its index ...
    e.training_step(batches[1], 0, noise=noises[1])  # ... and carries an index the old rule read as "gradients are zero"
    gd, ge = seen["d"], seen["e"]
    assert bool(torch.isfinite(gd).all()) and bool(torch.isfinite(ge).all()) and float(gd.norm()) > 0
    rel = float((gd - ge).norm() / gd.norm())
    assert rel < 1e-3, f"the pass after a backward outside training_step stored over it (rel {rel:.3e})"


# ---------------------------------------------------------------- decode at the reference's inference batch (64)
def _oracle_decode(cfg, sd, x, init, Tp, n):
    """oracle.lvtr_oracle with its own KV cache: prefill (init state pushed in front) + n single-frame steps."""
    from oracle import lvtr_oracle as O
    B = x.shape[0]
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
    return {k: torch.stack(v, 1) for k, v in want.items()}


def _session_decode(cfg, sd, x, init, Tp, n, precision):
    import hipvg
    from hparams.hp import Hparams
    from inference.speech.session import DecodeSession
    from models.speech.lvtr import LVTR
    B = x.shape[0]
    hipvg.set_precision(precision)
    model = LVTR(Hparams.from_dict(copy.deepcopy(cfg)), input_dim=80)
    model.load_state_dict(sd, strict=False)
    model = model.cuda().eval()
    xd, zeros = x.to(dev()), torch.zeros(B, 4, device=dev())
    sess = DecodeSession(model, B, Tp + n + 2, use_graph=False, keep_latent=True)
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
    return {k: torch.stack(v, 1).cpu() for k, v in got.items()}, sess


def test_full_config_decode_session_at_batch_64(full_cfg):
    """``DecodeSession`` at the batch the reference's inference config decodes
    (configs/infer/speech/vae-gslm.yaml:27: 64 sequences): the FULL model, a 24-frame prompt and 6 teacher-forced steps,
    against the oracle's own KV-cache decode (reference: models/speech/lvtr.py:227-286, trainers/speech/sampler.py:50-62).
    fp32 (exact dot-product rows kernel in groups of 8) with the tolerances of the B = 4 test; bf16 on the round-5
    matrix-core rows kernel (weights streamed once for all 64 rows) within bf16 drift of the oracle, token arg-max
    exact where the oracle's margin is clear.  (The kernel itself against float64: test_kernels_gpu.py::test_gemm_rows,
    whose 17..64-row bf16 cases run on it.)"""
    import numpy as np
    from oracle import lvtr_oracle as O
    from oracle.weights import fill_like
    cfg = full_cfg["model"]
    rng = np.random.default_rng(64)
    B, Tp, n = 64, 24, 6
    x = torch.cat([torch.from_numpy(rng.integers(0, 200, (B, Tp + n, 1))).float(),
                   torch.from_numpy(rng.standard_normal((B, Tp + n, 4)).astype(np.float32))], -1)
    init = torch.from_numpy(rng.random((B, 1, 64)).astype(np.float32)) * 2 - 1
    sd = {k: torch.from_numpy(v) for k, v in fill_like(O.param_shapes(cfg), 20250620).items()}
    torch.set_num_threads(min(32, torch.get_num_threads() if torch.get_num_threads() > 1 else 32))
    want = _oracle_decode(cfg, sd, x, init, Tp, n)
    got32, sess = _session_decode(cfg, sd, x, init, Tp, n, "fp32")
    assert not sess._fused                                     # 64 sequences: five launches per layer
    for key, tol in (("lat", 1e-4), ("mean", 1e-4), ("logstd", 1e-4), ("logits", 5e-4)):
        np.testing.assert_allclose(got32[key].numpy(), want[key].numpy(), atol=tol, rtol=2e-4, err_msg=key)
    got16, _ = _session_decode(cfg, sd, x, init, Tp, n, "bf16")          # matrix-core rows kernel
    for key, tol in (("lat", 0.12), ("mean", 0.05), ("logstd", 0.05), ("logits", 0.25)):
        err = (got16[key] - want[key]).abs().max().item()
        assert err <= tol, f"bf16 decode at B = 64 drifts from the oracle: {key} {err:.3f}"
    # arg-max of the token logits where the oracle's margin is clear
    top2 = want["logits"].topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 0.3
    assert bool((got16["logits"].argmax(-1) == want["logits"].argmax(-1))[clear].all())


@pytest.mark.parametrize("shape", [(64, 1024, 1024, 4), (33, 1024, 4096, 4), (17, 200, 512, 3), (64, 1024, 4096, 16)])
def test_gemm_rows_acc(F, shape):
    """vg_gemm_rows_acc against float64: bf16 rows x bf16 weights + bias + fp32 residual accumulated into a ZEROED fp32
    buffer by K slices that meet through fp32 atomics (the N = d_model products of a decode layer at 17..64 sequences);
    the launch also clears the buffer handed in as `zero`."""
    M, N, K, splits = shape
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev()).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev()).bfloat16()
    bias = torch.randn(N, generator=g).to(dev())
    res = torch.randn(M, N, generator=g).to(dev())
    out = torch.zeros(M, N, device=dev())
    junk = torch.full((M, 64), 7.0, device=dev())
    F.rows_linear_acc(x, w, bias, res, out, splits=splits, zero=junk)
    ref = x.double() @ w.double().T + bias.double() + res.double()
    torch.testing.assert_close(out.double(), ref, atol=2e-3, rtol=1e-4)
    assert bool((junk == 0).all())


@pytest.mark.parametrize("graph", [False, True])
def test_decoder_on_the_side_branch_changes_nothing_but_the_schedule(full_cfg, graph):
    """hip.side_unet: the diffusion decoder runs on the step's side stream beside the Transformer stack (its backward
    follows through autograd; its weight-gradient products queue apart from the main chain's).  Same arithmetic, another
    schedule: the losses of one step and every gradient against the default placement, eager and as a captured graph."""
    import copy
    from hparams.hp import Hparams
    from oracle.lvtr_oracle import small_config
    from trainers.speech.lvtr import LVTRTrainer
    from training_lib.synthetic import make_batch
    d = torch.device("cuda:0")
    B, T = 4, 256
    batch = make_batch(B, T, d, seed=9, lengths=[256, 200, 256, 31])
    g = torch.Generator().manual_seed(4)
    D, E = full_cfg["model"]["latent_dim"], full_cfg["model"]["tokens"]["embedding_dim"]
    noise = {"eps_q": torch.randn(B, T, D, generator=g).to(d), "eps_diff": torch.randn(B, T, 80, generator=g).to(d),
             "t_diff": torch.randint(0, 1000, (B,), generator=g).to(d), "init_state": (torch.rand(B, 1, E, generator=g) * 2 - 1).to(d)}
    res = {}
    for side in (False, True):
        cfg = copy.deepcopy(full_cfg)
        cfg["model"] = small_config(cfg["model"])
        cfg["hip"].update(precision="bf16", graph=graph, side_unet=side, coalesce_accumulation=False)
        cfg["training"]["gradient_accumulation"] = 1
        torch.manual_seed(3)
        tr = LVTRTrainer(Hparams.from_dict(cfg)).to(d)
        tr.configure_optimizers()
        tr.attach_reducer()
        tr.global_step = 10 ** 9
        assert tr.model.side_unet == side
        if graph:
            import models.speech.lvtr as M
            orig = M.LVTR.forward
            fixed = noise

            def fwd(self, x, c=None, spkr=None, utterance=None, diff_input=None, noise=None, _o=orig):
                return _o(self, x, c, spkr, utterance, diff_input, fixed if noise is None else noise)
            M.LVTR.forward = fwd
            try:
                o = tr._graphed_micro_step(batch, 0, True)
            finally:
                M.LVTR.forward = orig
        else:
            o = tr._training_loop(batch, 0, noise=noise)
        torch.cuda.synchronize()
        res[side] = (o, torch.cat([bk["flat"] for bk in tr.reducer.buckets]).clone())
    (oa, ga), (ob, gb) = res[True], res[False]
    for k in ("loss", "kld", "rec_loss", "token_kld"):
        assert abs(float(oa[k]) - float(ob[k])) <= 1e-5 * max(1.0, abs(float(ob[k]))), (k, float(oa[k]), float(ob[k]))
    assert float((ga - gb).norm()) <= 1e-3 * float(gb.norm())
