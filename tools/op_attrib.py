#!/usr/bin/env python3
"""Attribute the stock (non-HIP-library) kernels of one training step to the Python lines that
launch them: runs two eager micro-batches under torch.profiler with stacks and prints, per kernel
name pattern, the source lines (inside this repo) with launch counts and device time.  GPU only.

usage: python tools/op_attrib.py [pattern ...]     (default patterns: Fill add copy reduce Cat)"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch
from torch.profiler import ProfilerActivity, profile

CONFIG = os.path.join(ROOT, "vae-gslm_amd", "configs", "train", "speech", "vae-gslm.yaml")


def main():
    pats = sys.argv[1:] or ["Fill", "add", "copy", "reduce", "Cat", "mul", "Cijk"]
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
    # label every module two levels below the model so that forward ops can be attributed to a sub-system
    ranges = {}

    def pre(name):
        def f(mod, args):
            r = torch.profiler.record_function("MOD:" + name)
            r.__enter__()
            ranges.setdefault(name, []).append(r)
        return f

    def post(name):
        def f(mod, args, out):
            ranges[name].pop().__exit__(None, None, None)
        return f

    for name, mod in tr.model.named_modules():
        if name and name.count(".") <= int(os.environ.get("DEPTH", "1")):
            mod.register_forward_pre_hook(pre(name))
            mod.register_forward_hook(post(name))
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for i in range(accum, 2 * accum):
            tr.training_step(batches[i], i)
        torch.cuda.synchronize()
    events = prof.events()
    # kernel events are linked to the CPU op that launched them through the correlation id
    by_pat = {p: collections.Counter() for p in pats}
    time_pat = {p: collections.Counter() for p in pats}
    repo = ROOT + os.sep
    for ev in events:
        ks = getattr(ev, "kernels", None)
        if not ks or ev.cpu_parent is not None and getattr(ev.cpu_parent, "kernels", None):
            continue
        where = None
        for fr in (ev.stack or []):
            if "vae-gslm_amd/" in fr:
                where = fr[fr.index("vae-gslm_amd/"):]
                break
        if where is None:      # no Python frames: innermost module label (forward) or autograd node (backward)
            par, chain = ev.cpu_parent, []
            while par is not None:
                chain.append(par.name)
                par = par.cpu_parent
            mods = [c[4:] for c in chain if c.startswith("MOD:")]
            if mods:
                where = "fwd " + mods[0]
            else:
                where = " < ".join(c.replace("autograd::engine::evaluate_function: ", "") for c in chain[-2:]) or None
        for k in ks:
            for p in pats:
                if p.lower() in k.name.lower():
                    key = f"{where or '<outside repo>'}  [{ev.name}]"
                    by_pat[p][key] += 1
                    time_pat[p][key] += k.duration
    mod_n, mod_t = collections.Counter(), collections.Counter()
    for ev in events:
        ks = getattr(ev, "kernels", None)
        if not ks or ev.cpu_parent is not None and getattr(ev.cpu_parent, "kernels", None):
            continue
        par, chain = ev.cpu_parent, []
        while par is not None:
            chain.append(par.name)
            par = par.cpu_parent
        mods = [c[4:] for c in chain if c.startswith("MOD:")]
        key = ("fwd " + mods[0]) if mods else ("bwd " + (chain[-1].replace("autograd::engine::evaluate_function: ", "") if chain else ev.name))
        for k in ks:
            if "GLOBAL__N" in k.name or "anonymous namespace)::gemm" in k.name or "dwnorm" in k.name:
                continue        # this library's kernels
            mod_n[key] += 1
            mod_t[key] += k.duration
    print(f"=== stock kernels by forward module / backward node: {sum(mod_n.values()) / accum:.0f} launches, {sum(mod_t.values()) / accum / 1e3:.3f} ms per micro-batch")
    for key, n in mod_n.most_common(40):
        print(f"  {n / accum:6.1f} x {mod_t[key] / accum / 1e3:7.3f} ms  {key}")
    for p in pats:
        tot = sum(by_pat[p].values())
        print(f"\n=== kernels matching '{p}': {tot / accum:.0f} launches, {sum(time_pat[p].values()) / accum / 1e3:.3f} ms per micro-batch")
        for key, n in by_pat[p].most_common(25):
            print(f"  {n / accum:6.1f} x {time_pat[p][key] / accum / 1e3:7.3f} ms  {key}")


if __name__ == "__main__":
    main()
