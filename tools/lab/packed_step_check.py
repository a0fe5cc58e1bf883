#!/usr/bin/env python3
"""Lab: the packed step (hip.packed_step) against padded rows -- the segment conv kernels sequence by sequence, then one
training forward/backward of the C1 configuration with injected noise."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
sys.path.insert(0, ROOT)
import torch
import yaml

import hipvg
from hipvg import functional as F

d = torch.device("cuda:0")
hipvg.lib()
hipvg.set_precision("bf16")

# ---- 1. kernels
torch.manual_seed(0)
B, T, C = 5, 333, 512
lens = torch.tensor([333, 7, 0, 200, 129], dtype=torch.int32, device=d)
for halo, shift in ((0, 6), (18, 0), (18, 6)):
    tot = int(torch.clamp(lens + halo, max=T).sum())
    rows = F.pack_rows_bucket(tot, 256)
    plan = F.PackPlan(B, T, rows, d, 256, halo=halo).fill(lens)
    xp = torch.randn(rows, C, device=d).bfloat16()
    w = torch.randn(C, 7, device=d) * 0.3
    cb, gamma, beta = torch.randn(C, device=d) * 0.1, 1 + 0.1 * torch.randn(C, device=d), 0.1 * torch.randn(C, device=d)
    te = torch.randn(B, C, device=d) * 0.2
    y, mean, rstd = F.dwnorm_fwd_raw(xp, w, cb, te, gamma, beta, plan, 7, shift, 1e-6)
    dy = torch.randn(rows, C, device=d).bfloat16()
    dxa = torch.randn(rows, C, device=d).bfloat16()
    du, dx, pg, pb, pw = F.dwnorm_bwd_raw(dy, xp, w, cb, te, gamma, mean, rstd, dxa, plan, 7, shift)
    cu = plan.cu.tolist()
    worst = 0.0
    spg, spb, spw = 0, 0, 0
    for s in range(plan.nseq):
        a, b = cu[s], cu[s + 1]
        if b == a:
            continue
        tes = te[min(s, B - 1):min(s, B - 1) + 1].contiguous()
        y1, m1, r1 = F.dwnorm_fwd_raw(xp[a:b].contiguous(), w, cb, tes, gamma, beta, b - a, 7, shift, 1e-6)
        assert torch.equal(y1, y[a:b]), ("fwd", halo, shift, s)
        assert torch.equal(m1, mean[a:b]) and torch.equal(r1, rstd[a:b])
        du1, dx1, pg1, pb1, pw1 = F.dwnorm_bwd_raw(dy[a:b].contiguous(), xp[a:b].contiguous(), w, cb, tes, gamma, m1, r1,
                                                   dxa[a:b].contiguous(), b - a, 7, shift)
        assert torch.equal(du1, du[a:b]) and torch.equal(dx1, dx[a:b]), ("bwd", halo, shift, s)
        spg, spb, spw = spg + pg1.sum(0), spb + pb1.sum(0), spw + pw1.sum(0)
    for name, a_, b_ in (("gamma", pg.sum(0), spg), ("beta", pb.sum(0), spb), ("w", pw.sum(0), spw)):
        err = float((a_ - b_).abs().max() / (b_.abs().max() + 1e-9))
        assert err < 1e-4, (name, err)
    sc = F.segment_colsum(dy, plan)
    ref = torch.stack([dy[cu[s]:cu[s + 1]].float().sum(0) for s in range(plan.nseq)])
    assert float((sc - ref).abs().max()) < 1e-2 * float(ref.abs().max() + 1), "segment_colsum"
    print(f"kernels ok: halo={halo} shift={shift} rows={rows} nseq={plan.nseq}", flush=True)

# ---- 2. whole step
from hparams.hp import Hparams
from oracle.lvtr_oracle import small_config
from trainers.speech.lvtr import LVTRTrainer
from training_lib.synthetic import make_batch

full_cfg = yaml.safe_load(open(os.path.join(ROOT, "vae-gslm_amd/configs/train/speech/vae-gslm.yaml")))


def trainer(step):
    cfg = copy.deepcopy(full_cfg)
    cfg["model"] = small_config(cfg["model"])
    cfg["hip"].update(precision="bf16", graph=False, packed_rows=False, packed_step=step, packed_rows_granule=256,
                      coalesce_accumulation=False)
    cfg["training"]["gradient_accumulation"] = 1
    torch.manual_seed(3)
    tr = LVTRTrainer(Hparams.from_dict(cfg)).to(d)
    tr.configure_optimizers()
    tr.attach_reducer()
    tr.global_step = 10 ** 9
    return tr


Bs, Ts = 4, 256
lens_ = [256, 90, 64, 200]
batch = make_batch(Bs, Ts, d, seed=7, lengths=lens_)
g = torch.Generator().manual_seed(5)
D = full_cfg["model"]["latent_dim"]
noise = {"eps_q": torch.randn(Bs, Ts, D, generator=g).to(d), "eps_diff": torch.randn(Bs, Ts, 80, generator=g).to(d),
         "t_diff": torch.randint(0, 1000, (Bs,), generator=g).to(d),
         "init_state": (torch.rand(Bs, 1, full_cfg["model"]["tokens"]["embedding_dim"], generator=g) * 2 - 1).to(d)}
res = {}
for step in (False, True):
    tr = trainer(step)
    tr._choose_pack_rows(batch, eager=True)
    o = tr._training_loop(batch, 0, noise=noise)
    torch.cuda.synchronize()
    grads = torch.cat([bk["flat"] for bk in tr.reducer.buckets]).clone()
    res[step] = (o, grads, tr)
    print("packed" if step else "padded", {k: float(v) for k, v in o.items() if torch.is_tensor(v) and v.numel() == 1}, flush=True)
    if step:
        assert tr.model._pack_plans, "the packed step did not run"
        print("plans", [(k[2], k[4]) for k in tr.model._pack_plans])
a, b = res[True], res[False]
for k in ("loss", "kld", "rec_loss", "token_kld"):
    x, y = float(a[0][k]), float(b[0][k])
    print(f"{k}: packed {x:.6f} padded {y:.6f} rel {abs(x - y) / max(1e-9, abs(y)):.2e}")
print("grad rel diff", float((a[1] - b[1]).norm() / b[1].norm()))
# per-parameter
names = [n for n, _ in b[2].model.named_parameters()]
pa = dict(a[2].model.named_parameters())
worst = []
for n, p in b[2].model.named_parameters():
    ga, gb = pa[n].grad, p.grad
    if ga is None or gb is None:
        continue
    worst.append((float((ga - gb).norm() / (gb.norm() + 1e-12)), n))
worst.sort(reverse=True)
for w_, n in worst[:12]:
    print(f"  {w_:.3e} {n}")
