#!/usr/bin/env python3
"""Phase-pipelined 256x256 GEMM (tile_cfg 10 / 11) against the 2-stage 256x256 tile (cfg 3) and 128x128 (cfg 1):
exact-integer correctness in all three operand modes (ragged M / N, split-K, epilogues), then interleaved timing on
the layer shapes with rotating (cold) operands.  GPU only.   CFGS=3,10,11  M=16000  ITERS=6"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
CFGS = [int(c) for c in os.environ.get("CFGS", "3,10,11").split(",")]
M0 = int(os.environ.get("M", "16000"))
R = int(os.environ.get("R", "6"))
ITERS = int(os.environ.get("ITERS", "6"))


def ints(shape, g, lo=-3, hi=4):
    return torch.randint(lo, hi, shape, generator=g).float()


def check():
    g = torch.Generator().manual_seed(1)
    bad = 0
    for cfg in [c for c in CFGS if c >= 10]:
        for (M, N, K) in ((256, 256, 64), (256, 256, 128), (256, 256, 192), (520, 392, 320), (1000, 1024, 1024),
                          (384, 3072, 256), (16000, 1024, 128)):
            # NT: A [M,K], B [N,K]
            A, B = ints((M, K), g), ints((N, K), g)
            ref = A.double() @ B.double().t()
            out = F.gemm(A.to(dev).bfloat16(), B.to(dev).bfloat16(), M, N, K, out_f32=True, tile_cfg=cfg)
            e = (out.double().cpu() - ref).abs().max().item()
            # NN: A [M,K], B [K,N] (b_tr)
            Bt = B.t().contiguous()
            out2 = F.gemm(A.to(dev).bfloat16(), Bt.to(dev).bfloat16(), M, N, K, b_tr=True, out_f32=True, tile_cfg=cfg)
            e2 = (out2.double().cpu() - ref).abs().max().item()
            # TN: A [K,M], B [K,N]
            At = A.t().contiguous()
            Mp, Np = (M // 8) * 8, (N // 8) * 8
            out3 = F.gemm(At[:, :Mp].contiguous().to(dev).bfloat16(), Bt[:, :Np].contiguous().to(dev).bfloat16(), Mp, Np, K,
                          a_tr=True, b_tr=True, out_f32=True, tile_cfg=cfg)
            e3 = (out3.double().cpu() - ref[:Mp, :Np]).abs().max().item()
            # split-K (atomics) on the TN form
            e4 = 0.0
            if K >= 128:
                out4 = F.gemm(At[:, :Mp].contiguous().to(dev).bfloat16(), Bt[:, :Np].contiguous().to(dev).bfloat16(), Mp, Np,
                              K, a_tr=True, b_tr=True, split_k=2, out_f32=True, tile_cfg=cfg)
                e4 = (out4.double().cpu() - ref[:Mp, :Np]).abs().max().item()
            ok = max(e, e2, e3, e4) == 0.0
            bad += not ok
            print(f"cfg {cfg} M={M} N={N} K={K}: NT {e:g} NN {e2:g} TN {e3:g} TN-split {e4:g} {'ok' if ok else 'FAIL'}", flush=True)
        # epilogue: bias + GELU + saved derivative + residual + lengths mask, bf16 out (vs cfg 3, bitwise)
        M, N, K, T = 1000, 512, 256, 250
        A = torch.randn(M, K, generator=g).to(dev).bfloat16()
        B = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16()
        bias = torch.randn(N, generator=g).to(dev)
        res = torch.randn(M, N, generator=g).to(dev).bfloat16()
        lens = torch.tensor([250, 100, 1, 0], dtype=torch.int32, device=dev)
        outs = {}
        for c in (3, cfg):
            aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            part = []
            o = F.gemm(A, B, M, N, K, bias=bias, act=2 | 16, aux_out=aux, residual=res, lengths=lens, T=T, tile_cfg=c,
                       colpart=part)
            outs[c] = (o, aux, part[0])
        same = all(torch.equal(a, b) for a, b in zip(outs[3][:2], outs[cfg][:2]))
        cp = (outs[3][2].sum(0) - outs[cfg][2].sum(0)).abs().max().item()
        print(f"cfg {cfg} epilogue (bias, GELU + derivative, residual, mask) bitwise equal to cfg 3: {same}; colpart diff {cp:g}")
        bad += not same
        # the lean epilogues of the long-phase schedule (plain / GELU + stored derivative / times stored derivative)
        M2, N2 = 1003, 520                      # ragged edges; N a multiple of 8
        A2 = torch.randn(M2, K, generator=g).to(dev).bfloat16()
        B2 = (torch.randn(N2, K, generator=g) * K ** -0.5).to(dev).bfloat16()
        bias2 = torch.randn(N2, generator=g).to(dev)
        res2 = torch.randn(M2, N2, generator=g).to(dev).bfloat16()
        der2 = torch.randn(M2, N2, generator=g).to(dev).bfloat16()
        lens2 = torch.tensor([17, 16, 1, 0, 16], dtype=torch.int32, device=dev)       # T = 17 -> 5 sequences = 85 rows
        variants = {
            "plain": dict(),
            "bias + residual + mask + colpart": dict(bias=bias2, residual=res2, lengths=None, colpart=True),
            "GELU + stored derivative": dict(bias=bias2, act=2 | 16, aux=True),
            "times stored derivative + colpart": dict(dact=4, aux_in=der2, colpart=True),
        }
        for vname, kw in variants.items():
            for masked in (False, True):
                Mv = 85 if masked else M2
                outs = {}
                for c in (3, cfg):
                    args = dict(tile_cfg=c)
                    if "bias" in kw: args["bias"] = kw["bias"]
                    if "residual" in kw: args["residual"] = kw["residual"][:Mv]
                    if "act" in kw: args["act"] = kw["act"]
                    if "dact" in kw: args.update(dact=kw["dact"], aux_in=kw["aux_in"][:Mv].contiguous())
                    aux = torch.zeros(Mv, N2, device=dev, dtype=torch.bfloat16) if kw.get("aux") else None
                    if aux is not None: args["aux_out"] = aux
                    part = [] if kw.get("colpart") else None
                    if part is not None: args["colpart"] = part
                    if masked: args.update(lengths=lens2, T=17)
                    if "residual" in args: args["residual"] = args["residual"].contiguous()
                    o = F.gemm(A2[:Mv].contiguous(), B2, Mv, N2, K, **args)
                    outs[c] = (o, aux, part[0] if part else None)
                same = torch.equal(outs[3][0], outs[cfg][0]) and (outs[3][1] is None or torch.equal(outs[3][1], outs[cfg][1]))
                cpd = 0.0 if outs[3][2] is None else (outs[3][2].sum(0) - outs[cfg][2].sum(0)).abs().max().item()
                ok = same and cpd < 1e-2
                bad += not ok
                print(f"cfg {cfg} lean epilogue [{vname}{', masked' if masked else ''}]: bitwise {same}, colpart diff {cpd:g} {'ok' if ok else 'FAIL'}")
    # grouped weight gradients (vg_gemm_grouped): exact integers, accumulation into non-zero gradients
    layer = [(4096, 1024), (1024, 4096), (3072, 1024), (1024, 1024)]
    scenarios = [("layer M=1024", [(s, 1024) for s in layer]), ("layer M=16000", [(s, 16000) for s in layer]),
                 ("3 full rounds", [((8192, 4096), 1024), ((4096, 4096), 1024)]),
                 ("2 rounds + 22 left", [((8192, 4096), 1024), ((1024, 1024), 1024), ((520, 768), 1024)]),
                 ("mixed K (stream)", [((4096, 1024), 2048), ((1024, 4096), 1024), ((3072, 1024), 1088)]),
                 ("100 ragged tiles", [((2000, 1016), 4096), ((3072, 1024), 4096), ((1000, 2040), 4096)])]
    for name, probs in scenarios:
        items, refs = [], []
        for (N, K), Mf in probs:
            w = torch.nn.Parameter(torch.zeros(N, K, device=dev))
            w.grad = ints((N, K), g).to(dev)
            dy, x = ints((Mf, N), g, -2, 3).to(dev).bfloat16(), ints((Mf, K), g, -2, 3).to(dev).bfloat16()
            refs.append(w.grad.double() + dy.double().t() @ x.double())
            items.append((w, dy, x))
        F.sink_wgrad_group(items)
        err = max((it[0].grad.double() - r).abs().max().item() for it, r in zip(items, refs))
        print(f"grouped wgrad {name}: max err {err:g} {'ok' if err == 0 else 'FAIL'}", flush=True)
        bad += err != 0
    print("CHECK", "PASSED" if bad == 0 else f"FAILED ({bad})", flush=True)
    return bad == 0


def bench():
    g = torch.Generator().manual_seed(0)
    D, Fd, M = 1024, 4096, M0
    mk = lambda *s: [torch.randn(*s, generator=g).to(dev).bfloat16() for _ in range(R)]
    xs, hs = mk(M, D), mk(M, Fd)
    q3 = [torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    us = [torch.empty(M, Fd, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    ys = [torch.empty(M, D, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    w1 = [(torch.randn(Fd, D, generator=g) * D ** -0.5).to(dev).bfloat16() for _ in range(R)]
    w2 = [(torch.randn(D, Fd, generator=g) * Fd ** -0.5).to(dev).bfloat16() for _ in range(R)]
    wq = [(torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev).bfloat16() for _ in range(R)]
    wo = [(torch.randn(D, D, generator=g) * D ** -0.5).to(dev).bfloat16() for _ in range(R)]
    b1 = torch.randn(Fd, generator=g).to(dev)
    b2 = torch.randn(D, generator=g).to(dev)
    wg1 = [torch.zeros(Fd, D, device=dev) for _ in range(R)]
    wgq = [torch.zeros(3 * D, D, device=dev) for _ in range(R)]
    wgo = [torch.zeros(D, D, device=dev) for _ in range(R)]
    cases = {
        "QKV fwd      NT N=3072 K=1024": (2.0 * M * 3 * D * D, lambda i, c: F.gemm(xs[i], wq[i], M, 3 * D, D, out=q3[i], tile_cfg=c)),
        "out-proj fwd NT N=1024 K=1024": (2.0 * M * D * D, lambda i, c: F.gemm(xs[i], wo[i], M, D, D, residual=ys[i], out=ys[i], tile_cfg=c)),
        "FFN-in fwd   NT N=4096 K=1024 +GELU'": (2.0 * M * Fd * D, lambda i, c: F.gemm(xs[i], w1[i], M, Fd, D, bias=b1, act=2 | 16, aux_out=us[i], out=hs[i], tile_cfg=c)),
        "FFN-out fwd  NT N=1024 K=4096": (2.0 * M * Fd * D, lambda i, c: F.gemm(hs[i], w2[i], M, D, Fd, bias=b2, residual=xs[i], out=ys[i], tile_cfg=c)),
        "dgrad->hid   NN N=4096 K=1024 *GELU'": (2.0 * M * Fd * D, lambda i, c: F.gemm(ys[i], w2[i], M, Fd, D, b_tr=True, dact=4, aux_in=us[i], out=hs[i], tile_cfg=c)),
        "dgrad->model NN N=1024 K=4096": (2.0 * M * Fd * D, lambda i, c: F.gemm(hs[i], w1[i], M, D, Fd, b_tr=True, out=ys[i], tile_cfg=c)),
        "dgrad qkv    NN N=1024 K=3072": (2.0 * M * 3 * D * D, lambda i, c: F.gemm(q3[i], wq[i], M, D, 3 * D, b_tr=True, out=ys[i], tile_cfg=c)),
        "wgrad W1     TN 4096x1024 K=M s2": (2.0 * M * Fd * D, lambda i, c: F.gemm(hs[i], xs[i], Fd, D, M, a_tr=True, b_tr=True, out=wg1[i], split_k=2 if c == 1 else 4, tile_cfg=c)),
        "wgrad Wqkv   TN 3072x1024 K=M": (2.0 * M * 3 * D * D, lambda i, c: F.gemm(q3[i], xs[i], 3 * D, D, M, a_tr=True, b_tr=True, out=wgq[i], split_k=2 if c == 1 else 5, tile_cfg=c)),
        "wgrad Wo     TN 1024x1024 K=M": (2.0 * M * D * D, lambda i, c: F.gemm(ys[i], xs[i], D, D, M, a_tr=True, b_tr=True, out=wgo[i], split_k=6 if c == 1 else 10, tile_cfg=c)),
    }
    cfgs = [1] + CFGS if 1 not in CFGS else CFGS
    # the layer's four weight gradients: one by one (library's own split and tile choice) against one grouped launch
    ws = [torch.nn.Parameter(torch.zeros(n, k, device=dev)) for n, k in ((Fd, D), (D, Fd), (3 * D, D), (D, D))]
    for w in ws:
        w.grad = torch.zeros_like(w)
    def one_by_one(i):
        F.sink_wgrad(ws[0], hs[i], xs[i]); F.sink_wgrad(ws[1], ys[i], hs[i]); F.sink_wgrad(ws[2], q3[i], xs[i]); F.sink_wgrad(ws[3], ys[i], xs[i])
    def grouped(i):
        F.sink_wgrad_group([(ws[0], hs[i], xs[i]), (ws[1], ys[i], hs[i]), (ws[2], q3[i], xs[i]), (ws[3], ys[i], xs[i])])
    for name, fn in (("layer wgrads one by one", one_by_one), ("layer wgrads grouped", grouped)):
        for i in range(R):
            fn(i)
        torch.cuda.synchronize()
        ts = []
        for _ in range(ITERS):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(R):
                fn(i)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / R * 1e-3)
        t = sorted(ts)[len(ts) // 2]
        print(f"{name:40s} {t * 1e6:7.1f} us {2.0 * M * 12 * D * D / t / 1e12:6.0f} TF", flush=True)
    if os.environ.get("BENCH") == "group":
        return
    for name, (flop, fn) in cases.items():
        res = {c: [] for c in cfgs}
        for c in cfgs:
            for i in range(R):
                fn(i, c)
        torch.cuda.synchronize()
        for _ in range(ITERS):
            for c in cfgs:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(R):
                    fn(i, c)
                b.record()
                torch.cuda.synchronize()
                res[c].append(a.elapsed_time(b) / R * 1e-3)
        line = f"{name:40s}"
        for c in cfgs:
            t = sorted(res[c])[len(res[c]) // 2]
            line += f" | cfg{c:2d} {t * 1e6:6.1f} us {flop / t / 1e12:6.0f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    hipvg.lib()
    ok = check()
    if os.environ.get("BENCH", "1") != "0":
        bench()
    sys.exit(0 if ok else 1)
