#!/usr/bin/env python3
"""Which Python lines of this repo issue the step's stock ATen ops?  One eager optimizer step under a TorchDispatchMode:
every dispatched op that is not a view / metadata op is counted against the innermost frame inside vae-gslm_amd/
(forward pass and everything else that runs on the calling thread; ops of stock backward nodes run on autograd's worker
thread and are not seen -- tools/op_attrib.py names those by node).  GPU only.
    python tools/lab/op_lines.py"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

CONFIG = os.path.join(ROOT, "vae-gslm_amd", "configs", "train", "speech", "vae-gslm.yaml")
SKIP = ("view", "reshape", "expand", "permute", "transpose", "slice", "select", "unsqueeze", "squeeze", "detach", "alias",
        "as_strided", "t.default", "split", "unbind", "size", "stride", "numel", "is_", "_local_scalar", "empty", "unfold",
        "lift_fresh", "narrow", "chunk", "_unsafe_view", "resize", "set_", "_has_", "sym_", "dim", "storage_offset")


class Lines(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in SKIP):
            where = "<outside repo>"
            for fr in reversed(traceback.extract_stack()):
                if "vae-gslm_amd/" in fr.filename and "op_lines" not in fr.filename:
                    where = f"{fr.filename[fr.filename.index('vae-gslm_amd/') + 13:]}:{fr.lineno} {fr.name}"
                    break
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            self.count[(where, name.replace("aten.", ""), str(shapes))] += 1
        return func(*args, **(kwargs or {}))


def main():
    import hipvg
    from hparams.hp import Hparams
    from trainers.speech.lvtr import LVTRTrainer
    from training_lib.synthetic import make_batch
    hipvg.lib()
    dev = torch.device("cuda:0")
    hp = Hparams.from_yamlfile(CONFIG)
    hp.hip.precision = "bf16"
    hp.hip.graph = False
    torch.manual_seed(1234)
    tr = LVTRTrainer(hp).to(dev)
    tr.configure_optimizers()
    tr.attach_reducer()
    tr.global_step = hp.training.scheduler.warmup_kld
    B, accum = hp.data.train.batch_size, tr.gradient_update_step
    batches = [make_batch(B, 1000, dev, seed=i) for i in range(2 * accum)]
    for i in range(accum):
        tr.training_step(batches[i], i)
    torch.cuda.synchronize()
    with Lines() as mode:
        for i in range(accum, 2 * accum):
            tr.training_step(batches[i], i)
    torch.cuda.synchronize()
    tot = sum(mode.count.values())
    print(f"{tot} dispatched non-view ops on the calling thread in one optimizer step")
    per_line = collections.Counter()
    for (where, op, shapes), n in mode.count.items():
        per_line[where] += n
    for where, n in per_line.most_common(70):
        ops = collections.Counter()
        for (w, op, shapes), k in mode.count.items():
            if w == where:
                ops[op] += k
        print(f"{n:4d}  {where}   " + ", ".join(f"{o} x{k}" for o, k in ops.most_common(6)))


if __name__ == "__main__":
    main()
