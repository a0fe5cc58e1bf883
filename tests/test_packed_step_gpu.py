"""The packed step (hip.packed_step): the WHOLE training step of a ragged batch on its valid frames --
posterior encoder, heads, Transformer stack, flow, losses and the diffusion UNet -- with an 18-frame halo of padding per
sequence for the UNet's three look-ahead blocks.  The reference pads (utils/helpers.py:80-135) and convolves the padding
(modules/conv/layers.py:70-135); the padded path of this build is pinned to it by the golden / oracle tests, so the checks
here are against that padded path: the segment conv kernels sequence by sequence (bitwise), the conv stack, one whole
forward / backward with injected noise (losses to 1e-5, every gradient to 1e-3 of its norm in bf16), and hipGraph replays
over batches that fall into different row buckets."""
import copy

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


def test_pack_plan_with_a_halo(F):
    B, T, halo = 4, 20, 5
    lens = torch.tensor([20, 7, 0, 12], dtype=torch.int32, device=dev())
    need = int(torch.clamp(lens + halo, max=T).sum())
    rows = F.pack_rows_bucket(need, 16)
    p = F.PackPlan(B, T, rows, dev(), 16, halo=halo).fill(lens)
    cu = p.cu.tolist()
    assert cu[:B + 1] == [0, 20, 32, 37, 54] and cu[-1] == rows             # min(len + halo, T) rows per sequence
    assert p.lengths.tolist()[:B] == lens.tolist()
    valid, seq, idx = p.valid.tolist(), p.seq.tolist(), p.idx.tolist()
    for s in range(B):
        for r in range(cu[s], cu[s + 1]):
            t = r - cu[s]
            assert seq[r] == s and valid[r] == int(t < int(lens[s])) and idx[r] == (s * T + t if t < int(lens[s]) else -1)
    assert sum(valid[cu[B]:]) == 0
    src, dst = p.shift_src.tolist(), p.shift_dst.tolist()
    for r in range(rows):
        if valid[r]:
            assert src[r] == (rows + seq[r] if r == cu[seq[r]] else r - 1) and dst[src[r]] == r
        else:
            assert src[r] == -1
    assert sum(1 for v in dst if v >= 0) == sum(valid)
    # the shift itself, against TensorMask.push().pop().apply_mask() on padded rows
    from utils.tensormask import TensorMask
    x = torch.randn(B, T, 8, device=dev())
    mask = torch.arange(T, device=dev())[None] < lens[:, None]
    x = torch.where(mask[..., None], x, torch.zeros((), device=dev()))
    start = torch.randn(B, 1, 8, device=dev())
    want = TensorMask(x, mask).push(start).pop(1).apply_mask().value
    xp = F.pack_rows(x.reshape(B * T, 8), p).requires_grad_(True)
    got = F.shift_rows(xp, start, p)
    assert torch.equal(F.unpack_rows(got, p).view(B, T, 8), want)
    g = torch.randn_like(got)
    got.backward(g)
    ref = torch.zeros_like(xp)
    for r in range(rows):
        if src[r] >= 0 and src[r] < rows:
            ref[src[r]] = g[r]
    assert torch.equal(xp.grad, ref)


@pytest.mark.parametrize("halo,shift", [(0, 6), (18, 0), (18, 6)])
def test_conv_kernels_on_packed_sequences_equal_one_sequence_at_a_time(F, halo, shift):
    """vg_dwnorm_fwd_seg / vg_dwnorm_bwd_seg on ragged sequences laid end to end (one empty, one a single run, pseudo
    sequences over the bucket's spare rows) against the uniform kernels run on each sequence alone: outputs, statistics
    and both input gradients bitwise, the parameter partial sums to rounding; vg_colsum_segments_cu against torch."""
    torch.manual_seed(0)
    d = dev()
    B, T, C = 5, 333, 512
    lens = torch.tensor([333, 7, 0, 200, 129], dtype=torch.int32, device=d)
    rows = F.pack_rows_bucket(int(torch.clamp(lens + halo, max=T).sum()), 256)
    plan = F.PackPlan(B, T, rows, d, 256, halo=halo).fill(lens)
    xp = torch.randn(rows, C, device=d).bfloat16()
    w = torch.randn(C, 7, device=d) * 0.3
    cb, gamma, beta = torch.randn(C, device=d) * 0.1, 1 + 0.1 * torch.randn(C, device=d), 0.1 * torch.randn(C, device=d)
    te = torch.randn(B, C, device=d) * 0.2
    y, mean, rstd = F.dwnorm_fwd_raw(xp, w, cb, te, gamma, beta, plan, 7, shift, 1e-6)
    dy, dxa = torch.randn(rows, C, device=d).bfloat16(), torch.randn(rows, C, device=d).bfloat16()
    du, dx, pg, pb, pw = F.dwnorm_bwd_raw(dy, xp, w, cb, te, gamma, mean, rstd, dxa, plan, 7, shift)
    cu = plan.cu.tolist()
    assert cu[-1] == rows
    spg = spb = spw = 0
    for s in range(plan.nseq):
        a, b = cu[s], cu[s + 1]
        if b == a:
            continue
        tes = te[min(s, B - 1):min(s, B - 1) + 1].contiguous()
        y1, m1, r1 = F.dwnorm_fwd_raw(xp[a:b].contiguous(), w, cb, tes, gamma, beta, b - a, 7, shift, 1e-6)
        assert torch.equal(y1, y[a:b]) and torch.equal(m1, mean[a:b]) and torch.equal(r1, rstd[a:b]), s
        du1, dx1, pg1, pb1, pw1 = F.dwnorm_bwd_raw(dy[a:b].contiguous(), xp[a:b].contiguous(), w, cb, tes, gamma, m1, r1,
                                                   dxa[a:b].contiguous(), b - a, 7, shift)
        assert torch.equal(du1, du[a:b]) and torch.equal(dx1, dx[a:b]), s
        spg, spb, spw = spg + pg1.sum(0), spb + pb1.sum(0), spw + pw1.sum(0)
    for got, want in ((pg.sum(0), spg), (pb.sum(0), spb), (pw.sum(0), spw)):
        assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    sc = F.segment_colsum(dy, plan)
    ref = torch.stack([dy[cu[s]:cu[s + 1]].float().sum(0) for s in range(plan.nseq)])
    assert float((sc - ref).abs().max()) <= 1e-5 * float(ref.abs().max() + 1)


def _trainer(full_cfg, step, graph=False):
    from hparams.hp import Hparams
    from oracle.lvtr_oracle import small_config
    from trainers.speech.lvtr import LVTRTrainer
    cfg = copy.deepcopy(full_cfg)
    cfg["model"] = small_config(cfg["model"])
    cfg["hip"].update(precision="bf16", graph=graph, packed_rows=False, packed_step=step, packed_rows_granule=256,
                      coalesce_accumulation=False)
    cfg["training"]["gradient_accumulation"] = 1
    torch.manual_seed(3)
    tr = LVTRTrainer(Hparams.from_dict(cfg)).to(dev())
    tr.configure_optimizers()
    tr.attach_reducer()
    tr.global_step = 10 ** 9
    return tr


def _noise(full_cfg, B, T, seed):
    g = torch.Generator().manual_seed(seed)
    D = full_cfg["model"]["latent_dim"]
    E = full_cfg["model"]["tokens"]["embedding_dim"]
    return {"eps_q": torch.randn(B, T, D, generator=g).to(dev()), "eps_diff": torch.randn(B, T, 80, generator=g).to(dev()),
            "t_diff": torch.randint(0, 1000, (B,), generator=g).to(dev()),
            "init_state": (torch.rand(B, 1, E, generator=g) * 2 - 1).to(dev())}


@pytest.mark.parametrize("lens_", [(256, 90, 64, 200), (256, 250, 3, 0), (256, 17, 5, 64)])
def test_packed_step_equals_the_padded_step(full_cfg, lens_):
    """One forward / backward of the C1 configuration (full-size conv stacks, bf16) with every random draw injected: the
    packed step's losses, monitors, returned tensors and gradients against the padded step's.  The valid frames go
    through the same kernels with the same operands in both layouts (row-local kernels and GEMM rows do not depend on a
    row's position), so the only differences are the summation orders of the weight gradients and loss sums."""
    from training_lib.synthetic import make_batch
    B, T = len(lens_), 256
    batch = make_batch(B, T, dev(), seed=7, lengths=list(lens_))
    noise = _noise(full_cfg, B, T, 5)
    res = {}
    for step in (False, True):
        tr = _trainer(full_cfg, step)
        tr._choose_pack_rows(batch, eager=True)
        o = tr._training_loop(batch, 0, noise=noise)
        grads = {n: p.grad.float().clone() for n, p in tr.model.named_parameters() if p.grad is not None}
        with torch.no_grad():
            tr._choose_pack_rows(batch, eager=True)
            model_in = batch["tokens"].expand().cat(batch["mel"])
            out = tr.model(model_in, noise=noise, utterance=batch["cropped_mel_utt"])
        res[step] = (o, grads, out)
        if step:
            assert tr.model._pack_plans and all(k[4] == 18 for k in tr.model._pack_plans), "the packed step did not run"
    (oa, ga, outa), (ob, gb, outb) = res[True], res[False]
    for k in ("loss", "kld", "rec_loss", "token_kld", "log_p", "log_q", "logstd", "q_logstd", "q_mean_abs"):
        x, y = float(oa[k]), float(ob[k])
        assert abs(x - y) <= 1e-5 * max(1.0, abs(y)), (k, x, y)
    assert int(oa["length"]) == int(ob["length"]) == sum(lens_)
    assert set(ga) == set(gb)
    for n in gb:
        assert float((ga[n] - gb[n]).norm()) <= 1e-3 * float(gb[n].norm()) + 1e-7, n
    mask = batch["mel"].mask
    for k in ("log_p", "log_q", "sample_q", "transformer_latent"):
        a, b = outa[k].value.float(), outb[k].value.float()
        assert a.shape == b.shape and torch.equal(outa[k].mask, mask)
        assert torch.equal(a[mask], b[mask]), k                      # the valid frames: bitwise
        assert bool((a[~mask] == 0).all()), k
    assert torch.equal(outa["token_argmax"][mask], outb["token_argmax"][mask])
    assert torch.equal(outa["logits"].reshape(B, T, -1)[mask], outb["logits"].reshape(B, T, -1)[mask])


def test_graph_replays_of_the_packed_step(full_cfg, monkeypatch):
    """hipGraph replays with the packed step against eager launches on padded rows, batch by batch, with the random draws
    replaced by per-shape tables of which the packed run sees the gathered rows: three ragged batches of one padded shape
    in two row buckets (the third REPLAYS the first one's graph with new lengths) and a full batch that runs unpacked."""
    from training_lib.synthetic import make_batch
    d = dev()
    B, T = 4, 256
    lens_list = [[256, 90, 64, 200], [256, 20, 9, 60], [250, 101, 70, 180], [256, 256, 256, 256]]
    batches = [make_batch(B, T, d, seed=20 + i, lengths=ls) for i, ls in enumerate(lens_list)]
    noises = [_noise(full_cfg, B, T, 40 + i) for i in range(len(batches))]
    results = {}
    for mode in ("eager padded", "graph packed"):
        tr = _trainer(full_cfg, mode == "graph packed", graph=(mode == "graph packed"))
        outs = []
        for i, (b, nz) in enumerate(zip(batches, noises)):
            if mode == "graph packed":
                # the captured forward draws its own noise: replace the draws by this batch's tables (packed on the fly)
                import models.speech.lvtr as M
                orig = M.LVTR.forward

                def fwd(self, x, c=None, spkr=None, utterance=None, diff_input=None, noise=None, _nz=nz, _orig=orig):
                    return _orig(self, x, c, spkr, utterance, diff_input, noise if noise is not None else _nz)
                monkeypatch.setattr(M.LVTR, "forward", fwd)
                o = tr._graphed_micro_step(b, i, True)
                monkeypatch.setattr(M.LVTR, "forward", orig)
            else:
                tr._choose_pack_rows(None)
                o = tr._training_loop(b, i, noise=nz)
            grads = torch.cat([bk["flat"] for bk in tr.reducer.buckets]).clone()
            outs.append((float(o["loss"]), float(o["kld"]), float(o["token_kld"]), float(o["rec_loss"]), int(o["length"]), grads))
            tr.reducer.zero_grad()
        results[mode] = outs
        if mode == "graph packed":
            rows = sorted(k[2] for k in tr.model._pack_plans)
            assert len(rows) == 2 and len(tr._graphs) == 3, (rows, len(tr._graphs))     # two row buckets + the full batch
    for i, (e, gph) in enumerate(zip(results["eager padded"], results["graph packed"])):
        assert e[4] == gph[4] == sum(lens_list[i])
        if i < 2 or i == 3:      # (a replay reuses the noise tables captured with its graph: batch 2 sees batch 0's draws)
            for a, b in zip(e[:4], gph[:4]):
                assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), (i, e[:4], gph[:4])
            assert (e[5] - gph[5]).norm() <= 2e-3 * e[5].norm(), i
        else:
            assert all(abs(v) < 1e9 for v in gph[:4]) and bool(torch.isfinite(gph[5]).all())
