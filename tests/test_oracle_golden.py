"""Pin the CPU oracle against golden vectors produced by the reference itself
(tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import lvtr_oracle as O
from oracle.weights import fill_like

SEED = 20250620


def build_sd(cfg, requires_grad=False):
    arrs = fill_like(O.param_shapes(cfg), SEED)
    sd = {k: torch.from_numpy(v).clone() for k, v in arrs.items()}
    if requires_grad:
        for v in sd.values():
            v.requires_grad_(True)
    return sd


def batch_noise(g):
    batch = {k: torch.from_numpy(g["in_" + k]) for k in
             ("tokens", "mel", "lengths", "utt", "utt_lengths")}
    noise = dict(eps_q=torch.from_numpy(g["noise_eps_q"]),
                 init_state=torch.from_numpy(g["noise_init_rand"]) * 2 - 1,
                 eps_p=torch.from_numpy(g["noise_eps_p"]),
                 t_diff=torch.from_numpy(g["noise_t_diff"]),
                 eps_diff=torch.from_numpy(g["noise_eps_diff"]))
    return batch, noise


def rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-12)


def check_step(g, cfg, tcfg):
    sd = build_sd(cfg, requires_grad=True)
    assert sorted(sd.keys()) == list(g["keys"])
    batch, noise = batch_noise(g)
    out = O.training_loss(sd, cfg, tcfg, batch, noise)
    # scalars: 1e-4 rel is the north-star bar; the restatement is far tighter
    for name, key in (("loss", "loss"), ("kld", "kld"), ("ce_loss", "ce_loss"),
                      ("decoder_output", "rec_loss")):
        assert rel(out[name], g[key]) < 2e-5, (name, float(out[name]), float(g[key]))
    for name in ("logstd", "mean", "q_logstd", "q_mean", "q_mean_abs"):
        assert abs(float(out[name]) - float(g[name])) < 1e-5
    np.testing.assert_allclose(out["log_q"].detach().numpy(), g["log_q"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(out["log_p"].detach().numpy(), g["log_p"], atol=5e-3, rtol=2e-4)
    np.testing.assert_allclose(out["sample_q"].detach().numpy(), g["sample_q"], atol=1e-5)
    np.testing.assert_allclose(out["u_c"].detach().numpy(), g["u_c"], atol=1e-5)
    mask = out["mask"].numpy()
    am = out["logits"].argmax(-1).numpy()
    sure = mask & (g["margin"] > 1e-3)
    assert (am[sure] == g["argmax"][sure]).all()
    assert sure.sum() > 0.95 * mask.sum()
    np.testing.assert_allclose(out["transformer_latent"].detach()[:, ::5, ::9].numpy(),
                               g["latent_slice"], atol=2e-5, rtol=1e-4)
    # gradients
    out["loss"].backward()
    keys = list(g["keys"])
    gn = np.array([float(sd[k].grad.double().norm()) for k in keys])
    ref = g["grad_norm"]
    big = ref > 1e-6 * ref.max()
    assert np.max(np.abs(gn[big] - ref[big]) / ref[big]) < 1e-3
    for name in g.files:
        if name.startswith("grad::"):
            k = name[6:]
            flat = sd[k].grad.reshape(-1)
            mine = flat[:: max(1, flat.numel() // 257)][:257].numpy()
            scale = np.abs(g[name]).max() + 1e-12
            assert np.abs(mine - g[name]).max() / scale < 1e-3, k


def test_step_c1(golden, full_cfg):
    check_step(golden("step_c1"), O.small_config(full_cfg["model"]), full_cfg["training"])


def test_step_full_config(golden, full_cfg):
    check_step(golden("step_full"), full_cfg["model"], full_cfg["training"])


def test_modules(golden):
    g = golden("modules")
    x = torch.from_numpy(g["x"])
    B, T, D = x.shape
    H = 4
    mask = O.prefix_mask(torch.from_numpy(g["lengths"]), T)
    # RMSNorm
    w = fill_like([("scale", (D,))], 11)
    y = O.rmsnorm(x, torch.from_numpy(w["scale"]), 1e-6)
    np.testing.assert_allclose(y.numpy(), g["rmsnorm_y"], atol=1e-6, rtol=1e-6)
    # attention
    sd = {"sa." + k: torch.from_numpy(v) for k, v in fill_like(
        [("in_proj.weight", (3 * D, D)), ("out_proj.weight", (D, D))], 12).items()}
    xm = O.zero_pad_rows(x, mask)
    o, _, _ = O.self_attention(sd, "sa", xm, mask, H)
    np.testing.assert_allclose(o.numpy(), g["attn_y"], atol=3e-6, rtol=1e-5)
    # ALiBi table (closed form) incl. non-power-of-two head count
    for H_, name in ((16, "alibi_16"), (12, "alibi_12")):
        pos = torch.arange(8)
        rel_ = (pos[None] - pos[:, None]).abs().float()
        tab = -torch.tensor(O.alibi_slopes(H_))[:, None, None] * rel_
        np.testing.assert_allclose(tab.numpy(), g[name], atol=1e-7)
    assert np.allclose(O.alibi_slopes(16), [2 ** (-(h + 1) / 2) for h in range(16)])
    # transformer layer
    shapes = [("self_attn.in_proj.weight", (3 * D, D)), ("self_attn.out_proj.weight", (D, D)),
              ("linear1.weight", (512, D)), ("linear1.bias", (512,)),
              ("linear2.weight", (D, 512)), ("linear2.bias", (D,)),
              ("norm1.scale", (D,)), ("norm3.scale", (D,))]
    sd = {"l." + k: torch.from_numpy(v) for k, v in fill_like(shapes, 13).items()}
    y, _ = O.transformer_layer(sd, "l", xm, mask, H, 1e-6)
    np.testing.assert_allclose(y.numpy(), g["layer_y"], atol=2e-5, rtol=1e-5)
    # gaussian head
    shapes = [("mean.weight", (4, D)), ("mean.bias", (4,)),
              ("logstd.weight", (4, D)), ("logstd.bias", (4,))]
    sd = {"g." + k: torch.from_numpy(v) for k, v in fill_like(shapes, 14).items()}
    m, ls, s = O.gaussian_head(sd, "g", xm, torch.from_numpy(g["gauss_eps"]), 0.85)
    np.testing.assert_allclose(m.numpy(), g["gauss_mean"], atol=2e-6)
    np.testing.assert_allclose(ls.numpy(), g["gauss_logstd"], atol=2e-6)
    np.testing.assert_allclose(s.numpy(), g["gauss_sample"], atol=1e-5)
    # losses
    logits = torch.from_numpy(g["ce_logits"])
    tgt = torch.from_numpy(g["ce_target"])
    tg = torch.where(mask, tgt, torch.full_like(tgt, -100))
    ce = torch.nn.functional.cross_entropy(
        O.zero_pad_rows(logits, mask).reshape(B * T, -1), tg.reshape(-1),
        reduction="sum", ignore_index=-100)
    assert rel(ce, g["ce_sum"]) < 1e-6
    a, b = torch.from_numpy(g["ml_a"]), torch.from_numpy(g["ml_b"])
    kl = O.kl_sum(O.zero_pad_rows(a, mask), O.zero_pad_rows(b, mask))
    assert rel(kl, g["ml_sum"]) < 1e-5


def test_decode_kv_cache(golden, full_cfg):
    """Teacher-forced prefill(30)+10 single-frame steps with a KV cache
    (models/speech/lvtr.py:227-286) reproduce the reference step outputs."""
    g = golden("decode_c1")
    cfg = O.small_config(full_cfg["model"])
    sd = build_sd(cfg)
    x = torch.from_numpy(g["x"])
    Tp = int(g["prefill"])
    B = x.shape[0]
    init = torch.from_numpy(g["init_rand"]) * 2 - 1
    tr = cfg["transformer"]
    past = None
    lat, mean, logstd, logits = [], [], [], []
    for i in range(x.shape[1] - Tp + 1):
        xi = x[:, :Tp] if i == 0 else x[:, Tp + i - 1: Tp + i]
        tok = sd["token_embedding.weight"][xi[..., 0].long()]
        fused = tok + torch.relu(O.dense(sd, "token_fuser.linear", xi[..., 1:]))
        if i == 0:
            fused = torch.cat([init, fused], 1)
        m = torch.ones(B, fused.shape[1], dtype=torch.bool)
        hT, kvs, _ = O.transformer_stack(sd, "transformer.0", fused, m, tr, past)
        past = kvs
        c = torch.relu(O.dense(sd, "q_spliter.linear", hT))
        lat.append(hT[:, -1])
        mean.append(O.dense(sd, "transformer.1.mean", c)[:, -1])
        logstd.append(O.dense(sd, "transformer.1.logstd", c)[:, -1])
        logits.append(O.dense(sd, "token_predictor.linear",
                              torch.relu(O.dense(sd, "token_spliter.linear", hT)))[:, -1])
    np.testing.assert_allclose(torch.stack(lat, 1).numpy(), g["latent_last"], atol=3e-5, rtol=1e-4)
    np.testing.assert_allclose(torch.stack(mean, 1).numpy(), g["mean_last"], atol=3e-5, rtol=1e-4)
    np.testing.assert_allclose(torch.stack(logstd, 1).numpy(), g["logstd_last"], atol=3e-5, rtol=1e-4)
    np.testing.assert_allclose(torch.stack(logits, 1).numpy(), g["logits_last"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(past[0][0].numpy()[:, ::3, ::11], g["k_cache_l0"], atol=1e-5)
