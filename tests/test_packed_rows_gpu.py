"""Packed rows (hip.packed_rows): the Transformer stack of a ragged batch runs on its valid frames
only.  Checked against the padded path of the same kernels (which the golden / oracle tests pin to the reference):
the row gather and its adjoint, the attention kernels on packed rows (outputs bitwise on the valid frames, added rows
zero-filled), the stack's outputs and parameter gradients (fp32 and bf16, a batch with an empty sequence, a batch
that is not worth packing), and whole training steps -- eager and hipGraph replays over batches that fall into
different row buckets -- against the same steps with the option off."""
import copy

import numpy as np
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


def test_pack_plan_and_row_gather(F):
    B, T, C = 4, 37, 64
    lens = torch.tensor([37, 0, 12, 30], dtype=torch.int32, device=dev())
    rows = F.pack_rows_bucket(int(lens.sum()), 32)
    plan = F.PackPlan(B, T, rows, dev()).fill(lens)
    mask = (torch.arange(T, device=dev())[None] < lens[:, None]).reshape(-1)
    want = torch.nonzero(mask).flatten().int()
    assert torch.equal(plan.idx[:want.numel()], want) and bool((plan.idx[want.numel():] == -1).all())
    assert bool((plan.inv[~mask] == -1).all()) and torch.equal(plan.inv[mask], torch.arange(want.numel(), device=dev()).int())
    cu = plan.cu.tolist()
    assert cu[:B + 1] == [0, 37, 37, 49, 79] and cu[-1] == rows and all(b - a <= T for a, b in zip(cu, cu[1:]))
    assert plan.lengths.tolist()[:B] == lens.tolist() and sum(plan.lengths.tolist()[B:]) == 0
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.randn(B * T, C, device=dev()).to(dtype).requires_grad_(True)
        xp = F.pack_rows(x, plan)
        assert torch.equal(xp[:want.numel()], x.detach()[mask]) and bool((xp[want.numel():] == 0).all())
        back = F.unpack_rows(xp, plan)
        assert torch.equal(back[mask], x.detach()[mask]) and bool((back[~mask] == 0).all())
        g = torch.randn_like(back)
        back.backward(g)
        assert torch.equal(x.grad[mask], g[mask]) and bool((x.grad[~mask] == 0).all())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(3, 200, 4, (200, 77, 130)), (4, 333, 12, (333, 0, 1, 290)), (2, 1000, 16, (1000, 613))])
def test_attention_on_packed_rows_is_the_padded_attention(F, dtype, case):
    import hipvg
    B, T, H, lens_ = case
    D = H * 64
    L, st, p = hipvg.lib(), hipvg.stream(), hipvg.ptr
    lens = torch.tensor(lens_, dtype=torch.int32, device=dev())
    slopes = torch.tensor(F.alibi_slopes(H), dtype=torch.float32, device=dev())
    g = torch.Generator().manual_seed(T)
    qkv = torch.randn(B * T, 3 * D, generator=g).to(dev()).to(dtype)
    dout = torch.randn(B * T, D, generator=g).to(dev()).to(dtype)
    mask = (torch.arange(T, device=dev())[None] < lens[:, None]).reshape(-1)
    dout = torch.where(mask[:, None], dout, torch.zeros_like(dout))
    did = hipvg.dtype_id(dtype)
    out = torch.empty(B * T, D, device=dev(), dtype=dtype)
    lse = torch.empty(H, B * T, device=dev())
    dqkv, delta = torch.empty_like(qkv), torch.empty(H, B * T, device=dev())
    assert L.vg_attn_fwd(p(qkv), p(out), p(lse), p(slopes), B, T, H, p(lens), did, st) == 0
    assert L.vg_attn_bwd(p(qkv), p(out), p(dout), p(lse), p(slopes), p(dqkv), p(delta), B, T, H, p(lens), did, st) == 0
    rows = F.pack_rows_bucket(int(lens.sum()), 256)
    plan = F.PackPlan(B, T, rows, dev()).fill(lens)
    qp, dop = F.pack_rows(qkv, plan), F.pack_rows(dout, plan)
    outp = torch.full((rows, D), float("nan"), device=dev(), dtype=dtype)
    dqp = torch.full((rows, 3 * D), float("nan"), device=dev(), dtype=dtype)
    lsep, deltap = torch.empty(H, rows, device=dev()), torch.empty(H, rows, device=dev())
    assert L.vg_attn_fwd_varlen(p(qp), p(outp), p(lsep), p(slopes), plan.nseq, T, H, p(plan.lengths), p(plan.cu), rows, did, st) == 0
    assert L.vg_attn_bwd_varlen(p(qp), p(outp), p(dop), p(lsep), p(slopes), p(dqp), p(deltap), plan.nseq, T, H,
                                p(plan.lengths), p(plan.cu), rows, did, st) == 0
    n = int(lens.sum())
    assert torch.equal(outp[:n], out[mask]) and bool((outp[n:] == 0).all()), "forward differs on packed rows"
    assert torch.equal(dqp[:n], dqkv[mask]) and bool((dqp[n:] == 0).all()), "backward differs on packed rows"


def _stack(full_cfg, precision, small=True):
    import hipvg
    from hparams.hp import Hparams
    from modules.transformer.layers import TransformerLayerStack
    from oracle.lvtr_oracle import small_config
    hipvg.set_precision(precision)
    cfg = small_config(full_cfg["model"]) if small else copy.deepcopy(full_cfg["model"])
    hp = Hparams.from_dict(copy.deepcopy(cfg["transformer"]))
    torch.manual_seed(5)
    st = TransformerLayerStack(hp, input_dim=64).to(dev())
    with torch.no_grad():
        for p_ in st.parameters():
            if p_.dim() == 1:
                p_.add_(0.1 * torch.randn_like(p_))
    return st


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("lens_", [(300, 120, 0, 211), (300, 300, 299, 300)])
def test_stack_on_packed_rows_matches_padded_rows(full_cfg, precision, lens_):
    from utils.tensormask import TensorMask
    st = _stack(full_cfg, precision)
    B, T = len(lens_), 300
    lens = torch.tensor(lens_, device=dev())
    mask = torch.arange(T, device=dev())[None] < lens[:, None]
    g = torch.Generator().manual_seed(11)
    x = torch.where(mask[..., None], torch.randn(B, T, 64, generator=g).to(dev()), torch.zeros((), device=dev()))
    gy = torch.where(mask[..., None], torch.randn(B, T, st.hp.layer.dim, generator=g).to(dev()), torch.zeros((), device=dev()))
    res = {}
    for mode in (None, "auto"):
        st.pack_rows = mode
        st.zero_grad(set_to_none=True)
        xin = x.clone().requires_grad_(True)
        y = st(TensorMask(xin, mask)).value
        (y.float() * gy).sum().backward()
        res[mode] = (y.detach().float(), xin.grad.float(), {k: v.grad.float().clone() for k, v in st.named_parameters()})
    packed_used = any(k[2] <= int(0.94 * B * T) for k in st._pack_plans)
    assert packed_used == (sum(lens_) < 0.9 * B * T)
    tol = dict(atol=2e-5, rtol=2e-5) if precision == "fp32" else dict(atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(res["auto"][0], res[None][0], **tol)
    assert bool((res["auto"][0][~mask] == 0).all())
    torch.testing.assert_close(res["auto"][1], res[None][1], **tol)
    for k in res[None][2]:
        a, b = res["auto"][2][k], res[None][2][k]
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) / scale < (2e-4 if precision == "fp32" else 3e-2), k


def _trainer(full_cfg, packed, graph, precision="bf16"):
    import hipvg
    from hparams.hp import Hparams
    from oracle.lvtr_oracle import small_config
    from trainers.speech.lvtr import LVTRTrainer
    cfg = copy.deepcopy(full_cfg)
    cfg["model"] = small_config(cfg["model"])
    cfg["hip"].update(precision=precision, graph=graph, packed_rows=packed, packed_rows_granule=256, coalesce_accumulation=False,
                      packed_step=False)      # (the stack's own packing is what these tests are about: tests/test_packed_step_gpu.py has the other)
    cfg["training"]["gradient_accumulation"] = 1
    hp = Hparams.from_dict(cfg)
    torch.manual_seed(3)
    tr = LVTRTrainer(hp).to(dev())
    tr.configure_optimizers()
    tr.attach_reducer()
    tr.global_step = hp.training.scheduler.warmup_kld
    return tr


def test_training_steps_with_packed_rows_track_padded_steps(full_cfg):
    """Six eager optimizer steps over ragged batches whose valid-frame counts fall into different row buckets (and one
    full-length batch): the same losses as the padded run, step by step (same seeds, same order of random draws)."""
    from training_lib.synthetic import make_batch
    B, T = 4, 256
    lens_list = [[256, 100, 31, 200], [256, 90, 40, 197], [256, 256, 256, 256], [256, 17, 5, 64], [256, 99, 33, 201], [256, 20, 9, 60]]
    losses = {}
    for packed in (False, True):
        tr = _trainer(full_cfg, packed, False)
        seq = []
        for i, ls in enumerate(lens_list):
            batch = make_batch(B, T, dev(), seed=50 + i, lengths=ls)
            out = tr.training_step(batch, i)
            seq.append(float(out["loss"]))
        losses[packed] = seq
        if packed:
            assert len(tr.model.transformer[0]._pack_plans) >= 2, "the ragged batches should have used at least two row buckets"
    a, b = np.array(losses[True]), np.array(losses[False])
    assert np.all(np.isfinite(a)) and np.max(np.abs(a - b) / np.abs(b)) < 2e-2, (a, b)


def test_graph_replays_with_packed_rows_match_eager_padded_rows(full_cfg, monkeypatch):
    """hipGraph replays with packed rows against eager launches on padded rows, batch by batch, with the random draws
    replaced by fixed tables in both runs: five ragged batches of one padded shape whose valid-frame counts fall into
    two row buckets (the third and fifth batch REPLAY graphs captured for the first and second with new lengths, one
    batch is full and runs unpacked): losses, valid-frame counts and the whole gradient agree."""
    import test_parity_round2_gpu as r2
    from training_lib.synthetic import make_batch
    from utils.tensormask import TensorMask
    d = dev()

    def ragged(seed, lens, T=256):
        b = make_batch(len(lens), T, d, seed=seed)
        mask = torch.arange(T, device=d)[None] < torch.tensor(lens, device=d)[:, None]
        return {"tokens": TensorMask(b["tokens"].value, mask), "mel": TensorMask(b["mel"].value, mask),
                "cropped_mel_utt": b["cropped_mel_utt"]}

    batches = [ragged(1, [256, 90, 64, 200]), ragged(2, [256, 20, 9, 60]), ragged(3, [250, 101, 70, 180]),
               ragged(4, [256, 256, 256, 256]), ragged(5, [256, 30, 12, 70])]
    r2._fixed_random_draws(monkeypatch)
    results = {}
    for mode in ("eager padded", "graph packed"):
        tr = r2._trainer_c1(full_cfg, graph=(mode == "graph packed"))
        tr.packed_rows, tr.packed_granule, tr.packed_step = mode == "graph packed", 256, False
        tr.global_step = 10 ** 9
        outs = []
        for i, b in enumerate(batches):
            if mode == "graph packed":
                o = tr._graphed_micro_step(b, i, True)
            else:
                tr._choose_pack_rows(None)
                o = tr._training_loop(b, i)
            grads = torch.cat([bk["flat"] for bk in tr.reducer.buckets]).clone()
            outs.append((float(o["loss"]), float(o["kld"]), float(o["token_kld"]), float(o["rec_loss"]), int(o["length"]), grads))
            tr.reducer.zero_grad()
        results[mode] = outs
        if mode == "graph packed":
            rows = sorted(k[2] for k in tr.model.transformer[0]._pack_plans)
            assert rows == [512, 768] and len(tr._graphs) == 3, (rows, len(tr._graphs))   # two row buckets + the full batch
    for i, (e, gph) in enumerate(zip(results["eager padded"], results["graph packed"])):
        assert e[4] == gph[4] == sum(int(v) for v in batches[i]["mel"].mask.sum(-1)), (i, e[4], gph[4])
        for a, b in zip(e[:4], gph[:4]):
            assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (i, e[:4], gph[:4])
        assert (e[5] - gph[5]).norm() <= 2e-2 * e[5].norm(), i


def test_packed_rows_with_fewer_frames_than_one_granule(F):
    """A batch whose valid frames fill a fraction of one granule: every row the rounding adds (here 891 of 1024) must be
    covered by a pseudo sequence, i.e. come out of the varlen attention as zeros, not as unwritten memory."""
    B, T, H = 8, 192, 4
    lens = torch.tensor([1, 1, 1, 1, 0, 0, 0, 129], dtype=torch.int32, device=dev())
    rows = F.pack_rows_bucket(int(lens.sum()), 1024)
    plan = F.PackPlan(B, T, rows, dev(), 1024).fill(lens)
    assert rows == 1024 and int(plan.cu[-1]) == rows and plan.npseudo * T >= rows
    import hipvg
    L, p, st = hipvg.lib(), hipvg.ptr, hipvg.stream()
    qkv = torch.randn(rows, 3 * H * 64, device=dev()).bfloat16()
    out = torch.full((rows, H * 64), float("nan"), device=dev()).bfloat16()
    lse = torch.empty(H, rows, device=dev())
    slopes = torch.tensor(F.alibi_slopes(H), dtype=torch.float32, device=dev())
    hipvg.check(L.vg_attn_fwd_varlen(p(qkv), p(out), p(lse), p(slopes), plan.nseq, T, H, p(plan.lengths), p(plan.cu), rows, 1, st),
                "attn_fwd_varlen")
    total = int(lens.sum())
    assert torch.isfinite(out.float()).all()
    assert torch.all(out[total:] == 0)
