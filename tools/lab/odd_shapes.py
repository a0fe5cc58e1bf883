import copy, sys
sys.path.insert(0, "/root/repo/vae-gslm_amd"); sys.path.insert(0, "/root/repo")
import torch, yaml
from hparams.hp import Hparams
from trainers.speech.lvtr import LVTRTrainer
from training_lib.synthetic import make_batch
from oracle.lvtr_oracle import small_config
cfg = yaml.safe_load(open("/root/repo/vae-gslm_amd/configs/train/speech/vae-gslm.yaml"))
cfg["model"] = small_config(cfg["model"])
dev = torch.device("cuda:0")
for graph in (False, True):
    for (B, T, lens) in [(3, 77, [77, 1, 40]), (1, 1, [1]), (5, 129, [129, 128, 3, 64, 65]), (2, 1003, None)]:
        cfg["hip"].update(precision="bf16", graph=graph)
        tr = LVTRTrainer(Hparams.from_dict(copy.deepcopy(cfg))).to(dev)
        tr.configure_optimizers(); tr.attach_reducer()
        tr.global_step = cfg["training"]["scheduler"]["warmup_kld"]
        for it in range(4):
            out = tr.training_step(make_batch(B, T, dev, seed=it, lengths=lens), it)
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(out["loss"])) and all(torch.isfinite(p).all() for p in tr.model.parameters())
        print(f"graph={graph} B={B} T={T} lens={lens}: loss {float(out['loss']):.3f} finite={ok}", flush=True)
