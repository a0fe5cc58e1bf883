import copy, sys
sys.path.insert(0, "/root/repo/vae-gslm_amd"); sys.path.insert(0, "/root/repo")
import torch, yaml
from hparams.hp import Hparams
from trainers.speech.lvtr import LVTRTrainer
from training_lib.synthetic import make_batch
from oracle.lvtr_oracle import small_config
cfg = yaml.safe_load(open("/root/repo/vae-gslm_amd/configs/train/speech/vae-gslm.yaml"))
cfg["model"] = small_config(cfg["model"])
cfg["hip"].update(precision="fp32", graph=False)
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = LVTRTrainer(Hparams.from_dict(copy.deepcopy(cfg))).to(dev)
tr.configure_optimizers(); tr.attach_reducer()
tr.global_step = cfg["training"]["scheduler"]["warmup_kld"]
for (B, T, lens) in [(2, 40, [40, 17]), (1, 1, [1])]:
    batch = make_batch(B, T, dev, seed=3, lengths=lens)
    padded = tr._pad_for_graph(batch)
    Tp = padded["mel"].value.shape[1]
    g = torch.Generator(device="cpu").manual_seed(5)
    def noise(Tn):
        n = dict(eps_q=torch.randn(B, T, 4, generator=torch.Generator().manual_seed(1)),
                 init_state=torch.rand(B, 1, 64, generator=torch.Generator().manual_seed(2)) * 2 - 1,
                 t_diff=torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(3)),
                 eps_diff=torch.randn(B, T, 80, generator=torch.Generator().manual_seed(4)))
        for k in ("eps_q", "eps_diff"):
            n[k] = torch.nn.functional.pad(n[k], (0, 0, 0, Tn - T))
        return {k: v.to(dev) for k, v in n.items()}
    res = []
    for b_, Tn in ((batch, T), (padded, Tp)):
        tr.reducer.zero_grad()
        out = tr._training_loop(b_, 0, noise(Tn))
        res.append({k: float(out[k]) for k in ("loss", "kld", "rec_loss", "token_kld")})
    print(B, T, lens, "padded T =", Tp)
    print("  plain :", res[0])
    print("  padded:", res[1])
