#!/usr/bin/env python3
"""Generate the golden vectors under ``tests/golden/`` from the REFERENCE.

Runs ONLY in the build container (it imports ``/root/reference``); the GPU
box never sees the reference -- it gets the ``.npz`` files this script wrote.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

What is captured:
* ``step_c1.npz``   full training-loss forward+backward of the reference
  ``LVTR`` at BASELINE configs[0] (L=2, d=256, H=4, ffd=1024; B=2, T=200,
  ragged lengths), with key-hashed weights and injected noise;
* ``step_full.npz`` the same at the full yaml config on a short batch
  (B=2, T=96) -- pins H=16 ALiBi slopes, d=1024, 16 layers;
* ``modules.npz``   per-module vectors: RMSNorm, SelfAttention,
  TransformerLayer, GaussianParameterize, masked_ce_loss, masked_loss, ALiBi;
* ``decode_c1.npz`` teacher-forced KV-cache decode (prefill 30 + 10 steps);
* ``extras_c1.npz`` ``LVTR.likelihood`` (models/speech/lvtr.py:337-388) on the C1 batch.

Weights are not stored: both sides regenerate them with
``oracle.weights.fill_like``.  Noise tensors ARE stored (they are inputs).
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle.weights import fill_like  # noqa: E402
from oracle.lvtr_oracle import small_config  # noqa: E402

SEED = 20250620


# ---------------------------------------------------------------- reference import
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


def import_reference():
    _stub("lightning"); _stub("lightning.fabric"); _stub("lightning.fabric.utilities")
    _stub("lightning.fabric.utilities.types", _DEVICE=object)
    _stub("lightning.fabric.utilities.apply_func",
          _BLOCKING_DEVICE_TYPES=("cpu",), _TransferableDataType=object)
    _stub("lightning_utilities"); _stub("lightning_utilities.core")
    _stub("lightning_utilities.core.apply_func", apply_to_collection=lambda *a, **k: None)
    _stub("torchaudio", functional=types.SimpleNamespace())
    sys.path.insert(0, REF)
    import importlib
    mods = {}
    for n in ("hparams.hp", "models.speech.lvtr", "utils.tensormask",
              "training_lib.losses", "modules.norm", "modules.attention.attention",
              "modules.transformer.layers", "modules.linear.layers",
              "modules.position.alibi"):
        mods[n] = importlib.import_module(n)
    sys.path.remove(REF)
    return mods


def to_hparams(Hparams, d):
    import json
    return json.loads(json.dumps(d), object_hook=lambda x: Hparams(**x))


# ---------------------------------------------------------------- noise injection
class NoiseQueue:
    """Replaces torch.randn_like / torch.rand / torch.randint while active."""

    def __init__(self, items):
        self.items = list(items)
        self._orig = {}

    def _pop(self, kind, shape):
        k, t = self.items.pop(0)
        assert k == kind, (k, kind)
        assert tuple(t.shape) == tuple(shape), (kind, t.shape, shape)
        return t.clone()

    def __enter__(self):
        self._orig = dict(randn_like=torch.randn_like, rand=torch.rand,
                          randint=torch.randint)
        torch.randn_like = lambda x, **kw: self._pop("randn", x.shape)
        torch.rand = lambda *s, **kw: self._pop("rand", s if not isinstance(s[0], (list, tuple)) else s[0])
        torch.randint = lambda lo, hi, size, **kw: self._pop("randint", size)
        return self

    def __exit__(self, *a):
        torch.randn_like = self._orig["randn_like"]
        torch.rand = self._orig["rand"]
        torch.randint = self._orig["randint"]
        assert not self.items, f"unused noise: {[k for k, _ in self.items]}"


def make_batch(rng, B, T, lengths, Tu, vocab, n_mels=80):
    tokens = rng.integers(0, vocab, size=(B, T)).astype(np.int64)
    mel = rng.standard_normal((B, T, n_mels)).astype(np.float32)
    utt = rng.standard_normal((B, Tu, n_mels)).astype(np.float32)
    return dict(tokens=tokens, mel=mel, lengths=np.asarray(lengths, np.int64),
                utt=utt, utt_lengths=np.full((B,), Tu, np.int64))


def make_noise(rng, B, T, latent, emb, n_mels, timesteps):
    return dict(
        eps_q=rng.standard_normal((B, T, latent)).astype(np.float32),
        init_rand=rng.random((B, 1, emb)).astype(np.float32),
        eps_p=rng.standard_normal((B, T, latent)).astype(np.float32),
        t_diff=rng.integers(0, timesteps, size=(B,)).astype(np.int64),
        eps_diff=rng.standard_normal((B, T, n_mels)).astype(np.float32),
    )


def load_weights(model, seed=SEED):
    sd = model.state_dict()
    filled = fill_like([(k, tuple(v.shape)) for k, v in sd.items()], seed)
    with torch.no_grad():
        for k, arr in filled.items():
            sd[k].copy_(torch.from_numpy(arr))
    return sorted(filled.keys())


def run_step(mods, model_cfg, train_cfg, tag, B, T, lengths, Tu, seed):
    Hparams = mods["hparams.hp"].Hparams
    LVTR = mods["models.speech.lvtr"].LVTR
    TensorMask = mods["utils.tensormask"].TensorMask
    masked_loss = mods["training_lib.losses"].masked_loss
    model = LVTR(to_hparams(Hparams, model_cfg), input_dim=80)
    keys = load_weights(model)
    rng = np.random.default_rng(seed)
    batch = make_batch(rng, B, T, lengths, Tu, model_cfg["tokens"]["vocab_size"])
    noise = make_noise(rng, B, T, model_cfg["latent_dim"],
                       model_cfg["tokens"]["embedding_dim"], 80,
                       model_cfg["decoder"]["diffusion"]["timesteps"])
    tl = torch.from_numpy(batch["lengths"])
    mask = torch.arange(T)[None] < tl[:, None]
    tok = TensorMask(torch.from_numpy(batch["tokens"]), mask)
    mel = TensorMask(torch.from_numpy(batch["mel"]), mask)
    x = tok.expand().cat(mel)                           # trainers/speech/lvtr.py:116-118
    utt = TensorMask(torch.from_numpy(batch["utt"]))
    q = [("randn", torch.from_numpy(noise["eps_q"])),
         ("rand", torch.from_numpy(noise["init_rand"])),
         ("randn", torch.from_numpy(noise["eps_p"])),
         ("randint", torch.from_numpy(noise["t_diff"])),
         ("randn", torch.from_numpy(noise["eps_diff"]))]
    # capture logits through a forward hook on token_predictor
    cap = {}
    hk = model.token_predictor.register_forward_hook(
        lambda m, i, o: cap.__setitem__("logits", o.value.detach().clone()))
    with NoiseQueue(q):
        out = model(x, utterance=utt)
    hk.remove()
    kld = masked_loss(out["log_q"] * 1.0, out["log_p"], fn=lambda a, b: a - b)
    kw = train_cfg["fixed_beta"]                         # after KL warm-up
    loss = out["decoder_output"] * 1.0 + kld * kw
    loss = loss + out["ce_loss"] * train_cfg["token_kld_weight"] * kw
    loss.backward()
    logits = cap["logits"]
    top2 = logits.topk(2, -1).values
    grads = {k: p.grad for k, p in model.named_parameters()}
    gnorm = np.array([float(grads[k].double().norm()) for k in keys], np.float64)
    res = dict(
        **{"in_" + k: v for k, v in batch.items()},
        **{"noise_" + k: v for k, v in noise.items()},
        keys=np.array(keys),
        grad_norm=gnorm,
        loss=np.float64(loss.item()), kld=np.float64(kld.item()),
        ce_loss=np.float64(out["ce_loss"].item()),
        rec_loss=np.float64(out["decoder_output"].item()),
        kld_weight=np.float64(kw),
        log_p=out["log_p"].value.detach().numpy(),
        log_q=out["log_q"].value.detach().numpy(),
        sample_q=out["sample_q"].value.detach().numpy(),
        logstd=np.float64(out["logstd"].item()), mean=np.float64(out["mean"].item()),
        q_logstd=np.float64(out["q_logstd"].item()), q_mean=np.float64(out["q_mean"].item()),
        q_mean_abs=np.float64(out["q_mean_abs"].item()),
        u_c=out["u_c"].detach().numpy(),
        argmax=logits.argmax(-1).numpy().astype(np.int16),
        margin=(top2[..., 0] - top2[..., 1]).numpy().astype(np.float32),
        logits_slice=logits[:, ::7, ::5].numpy(),
        latent_slice=out["transformer_latent"].value.detach()[:, ::5, ::9].numpy(),
        latent_sum=np.float64(out["transformer_latent"].value.detach().double().sum().item()),
        latent_abs_sum=np.float64(out["transformer_latent"].value.detach().double().abs().sum().item()),
    )
    for k in ("transformer.0.layers.0.self_attn.in_proj.weight",
              "transformer.0.layers.1.linear2.weight",
              "transformer.0.layers.0.norm1.scale",
              "token_embedding.weight", "transformer.1.logstd.weight",
              "encoder.0.layers.0.conv1.weight", "q_spliter.linear.bias"):
        g = grads[k].detach()
        flat = g.reshape(-1)
        res["grad::" + k] = flat[:: max(1, flat.numel() // 257)][:257].numpy()
    np.savez_compressed(os.path.join(HERE, f"step_{tag}.npz"), **res)
    print(f"[{tag}] loss={res['loss']:.6f} kld={res['kld']:.6f} ce={res['ce_loss']:.6f} "
          f"rec={res['rec_loss']:.6f} min-margin={res['margin'][mask.numpy()].min():.4g}")
    return model, batch, noise


def run_modules(mods, model_cfg):
    Hparams = mods["hparams.hp"].Hparams
    TensorMask = mods["utils.tensormask"].TensorMask
    rng = np.random.default_rng(77)
    res = {}
    B, T, D, H = 2, 50, 256, 4
    lengths = np.array([50, 33], np.int64)
    mask = torch.arange(T)[None] < torch.from_numpy(lengths)[:, None]
    x = rng.standard_normal((B, T, D)).astype(np.float32)
    res["x"] = x
    res["lengths"] = lengths
    xt = torch.from_numpy(x)
    # RMSNorm
    rn = mods["modules.norm"].RMSNorm(D, eps=1e-6)
    load_weights(rn, 11)
    res["rmsnorm_y"] = rn(xt).detach().numpy()
    # SelfAttention
    sa_hp = to_hparams(Hparams, dict(nheads=H, causal=True))
    sa = mods["modules.attention.attention"].SelfAttention(D, sa_hp)
    load_weights(sa, 12)
    alibi = mods["modules.position.alibi"].ALiBi(H, 64)
    xm = TensorMask(xt, mask).apply_mask()
    o = sa(xm, rpe_pair=("ALiBi", alibi), return_kv=True)
    res["attn_y"] = o["output"].value.detach().numpy()
    res["alibi_16"] = mods["modules.position.alibi"].ALiBi(16, 8).alibi.numpy()
    res["alibi_12"] = mods["modules.position.alibi"].ALiBi(12, 8).alibi.numpy()
    # TransformerLayer
    lhp = to_hparams(Hparams, dict(
        dim=D, ffd_size=512, norm=dict(identifier="RMSNorm", eps=1e-6),
        activation=dict(identifier="GELU"), self_attn=dict(nheads=H, causal=True)))
    tl = mods["modules.transformer.layers"].TransformerLayer(lhp)
    load_weights(tl, 13)
    o = tl(xm, rpe_pair=("ALiBi", alibi))
    res["layer_y"] = o["output"].value.detach().numpy()
    # GaussianParameterize
    gp = mods["modules.linear.layers"].GaussianParameterize(D, 4)
    load_weights(gp, 14)
    eps = rng.standard_normal((B, T, 4)).astype(np.float32)
    res["gauss_eps"] = eps
    with NoiseQueue([("randn", torch.from_numpy(eps))]):
        g = gp(xm, temperature=0.85)
    res["gauss_mean"] = g.mean.value.detach().numpy()
    res["gauss_logstd"] = g.logstd.value.detach().numpy()
    res["gauss_sample"] = g.sample.value.detach().numpy()
    # losses
    logits = rng.standard_normal((B, T, 200)).astype(np.float32) * 3
    tgt = rng.integers(0, 200, (B, T)).astype(np.int64)
    res["ce_logits"] = logits
    res["ce_target"] = tgt
    ce = mods["training_lib.losses"].masked_ce_loss(
        TensorMask(torch.from_numpy(logits), mask), TensorMask(torch.from_numpy(tgt), mask))
    res["ce_sum"] = np.float64(ce.item())
    a = rng.standard_normal((B, T, 4)).astype(np.float32)
    b = rng.standard_normal((B, T, 4)).astype(np.float32)
    res["ml_a"], res["ml_b"] = a, b
    ml = mods["training_lib.losses"].masked_loss(
        TensorMask(torch.from_numpy(a), mask), TensorMask(torch.from_numpy(b), mask),
        fn=lambda p, q: p - q)
    res["ml_sum"] = np.float64(ml.item())
    np.savez_compressed(os.path.join(HERE, "modules.npz"), **res)
    print("[modules] written")


def run_decode(mods, model_cfg, model, seed=99):
    """Teacher-forced KV-cache decode on the C1 model (models/speech/lvtr.py:227-286)."""
    rng = np.random.default_rng(seed)
    B, Tp, Ns = 2, 30, 10
    latent = model_cfg["latent_dim"]
    emb = model_cfg["tokens"]["embedding_dim"]
    tot = Tp + Ns
    tok = rng.integers(0, 200, (B, tot)).astype(np.float32)
    z = rng.standard_normal((B, tot, latent)).astype(np.float32)
    x = np.concatenate([tok[..., None], z], -1)
    init_rand = rng.random((B, 1, emb)).astype(np.float32)
    res = dict(x=x, init_rand=init_rand, prefill=np.int64(Tp))
    lat, mean, logstd, logits = [], [], [], []
    kv = None
    model.eval()
    with torch.no_grad():
        for i in range(Ns + 1):
            xi = torch.from_numpy(x[:, :Tp] if i == 0 else x[:, Tp + i - 1: Tp + i])
            Tq = xi.shape[1] + (1 if i == 0 else 0)
            q = []
            if i == 0:
                q.append(("rand", torch.from_numpy(init_rand)))
            q.append(("randn", torch.zeros(B, Tq, latent)))
            cap = {}
            h1 = model.transformer[1].register_forward_hook(
                lambda m, a, o: cap.update(mean=o.mean.value.clone(), logstd=o.logstd.value.clone()))
            h2 = model.token_predictor.register_forward_hook(
                lambda m, a, o: cap.update(logits=o.value.clone()))
            with NoiseQueue(q):
                out = model.step(xi, past_kv=kv, push_init_state=(i == 0))
            h1.remove(); h2.remove()
            kv = out["kv"]
            lat.append(out["transformer_latent"].value[:, -1].numpy())
            mean.append(cap["mean"][:, -1].numpy())
            logstd.append(cap["logstd"][:, -1].numpy())
            logits.append(cap["logits"][:, -1].numpy())
    res["latent_last"] = np.stack(lat, 1)
    res["mean_last"] = np.stack(mean, 1)
    res["logstd_last"] = np.stack(logstd, 1)
    res["logits_last"] = np.stack(logits, 1)
    res["k_cache_l0"] = kv[0]["key"].numpy()[:, ::3, ::11]
    np.savez_compressed(os.path.join(HERE, "decode_c1.npz"), **res)
    print("[decode] written; cache len", kv[0]["key"].shape[1])


def run_extras(mods, model, batch, noise):
    """``LVTR.likelihood`` of the reference on the step_c1 batch (temperature 0; the start frame is the same
    injected draw as in the training step; the two Gaussian heads still draw their unused noise)."""
    TensorMask = mods["utils.tensormask"].TensorMask
    T = batch["tokens"].shape[1]
    mask = torch.arange(T)[None] < torch.from_numpy(batch["lengths"])[:, None]
    x = TensorMask(torch.from_numpy(batch["tokens"]), mask).expand().cat(TensorMask(torch.from_numpy(batch["mel"]), mask))
    q = [("randn", torch.from_numpy(noise["eps_q"])), ("rand", torch.from_numpy(noise["init_rand"])),
         ("randn", torch.from_numpy(noise["eps_p"]))]
    with torch.no_grad(), NoiseQueue(q):
        ll = model.likelihood(x, temperature=0.0)
    np.savez_compressed(os.path.join(HERE, "extras_c1.npz"), likelihood=ll.numpy().astype(np.float64))
    print("[extras] likelihood", ll.numpy())


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    mods = import_reference()
    with open(os.path.join(ROOT, "vae-gslm_amd/configs/train/speech/vae-gslm.yaml")) as f:
        cfg = yaml.safe_load(f)
    # the build's yaml must parse to the reference's values (plus the hip: block)
    with open(os.path.join(REF, "configs/train/speech/vae-gslm.yaml")) as f:
        ref_cfg = yaml.safe_load(f)
    mine = {k: v for k, v in cfg.items() if k != "hip"}
    assert mine == ref_cfg, "build yaml diverged from the reference yaml"
    c1 = small_config(cfg["model"])
    model, batch, noise = run_step(mods, c1, cfg["training"], "c1", 2, 200, [200, 163], 150, seed=1234)
    run_extras(mods, model, batch, noise)
    run_decode(mods, c1, model)
    run_step(mods, cfg["model"], cfg["training"], "full", 2, 96, [96, 61], 150, seed=4321)
    run_modules(mods, c1)


if __name__ == "__main__":
    main()
