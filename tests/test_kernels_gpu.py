"""Kernel-level parity tests (need an MI355X): every HIP kernel, forward and
backward, fp32 (exact-f32 MFMA) and bf16, against plain torch math evaluated
in fp64/fp32 on the same inputs.  Shapes include ragged lengths, sizes that are
not tile multiples, and T = 1."""
import math

import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    import hipvg
    hipvg.lib()
    from hipvg import functional
    return functional


def dev():
    return torch.device("cuda:0")


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dev()).to(dtype)


def tol(dtype):
    return dict(atol=2e-5, rtol=2e-5) if dtype == torch.float32 else dict(atol=3e-2, rtol=3e-2)


def lengths_for(B, T):
    ls = [T] + [max(1, T - 37 * (i + 1)) for i in range(B - 1)]
    return torch.tensor(ls[:B], dtype=torch.int32, device=dev())


def row_mask(lengths, T):
    return (torch.arange(T, device=dev())[None] < lengths[:, None]).reshape(-1)


# ------------------------------------------------------------------ MFMA operand maps (exact integers)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
@pytest.mark.parametrize("K", [192, 80, 200])      # reduction length: whole 64-tiles, shorter than one, a ragged tail
def test_gemm_exact_small_integers(F, dtype, mode, K):
    """Asymmetric small-integer operands: any row/col or k-permutation error in
    the fragment maps (or a K tail that is not zero-filled) changes the exactly representable result."""
    M, N = 200, 136
    g = torch.Generator().manual_seed(5)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = torch.randint(-3, 4, (N, K), generator=g).float()
    ref = A @ B.T
    Ad, Bd = A.to(dev()).to(dtype), B.to(dev()).to(dtype)
    if mode == "nt":
        out = F.gemm(Ad, Bd, M, N, K)
    elif mode == "nn":
        out = F.gemm(Ad, Bd.T.contiguous(), M, N, K, b_tr=True)
    else:
        out = F.gemm(Ad.T.contiguous(), Bd.T.contiguous(), M, N, K, a_tr=True, b_tr=True, out_f32=True)
    assert torch.equal(out.float().cpu(), ref)


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
def test_gemm_every_tile_shape_exact(F, cfg, mode):
    """Every LDS-DMA tile configuration (64-deep 2/3-stage rings and the 32-deep 4-stage ring), forced through
    tile_cfg, on shapes with ragged M / N edges, a K tail and more K tiles than stages: exact on small integers."""
    M, N, K = 520, 392, 328
    g = torch.Generator().manual_seed(9)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = torch.randint(-3, 4, (N, K), generator=g).float()
    ref = A @ B.T
    Ad, Bd = A.to(dev()).bfloat16(), B.to(dev()).bfloat16()
    if mode == "nt":
        out = F.gemm(Ad, Bd, M, N, K, tile_cfg=cfg, out_f32=True)
    elif mode == "nn":
        out = F.gemm(Ad, Bd.T.contiguous(), M, N, K, b_tr=True, tile_cfg=cfg, out_f32=True)
    else:
        out = torch.zeros(M, N, device=dev())
        F.gemm(Ad.T.contiguous(), Bd.T.contiguous(), M, N, K, a_tr=True, b_tr=True, out=out, split_k=3, tile_cfg=cfg)
    assert torch.equal(out.float().cpu(), ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(F, dtype):
    B_, T = 2, 100
    M, N, K = B_ * T, 320, 256
    x, w = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, scale=K ** -0.5)
    bias = rnd(N, scale=0.1)
    res = rnd(M, N, dtype=dtype)
    lens = lengths_for(B_, T)
    mask = row_mask(lens, T)[:, None]
    u_ref = x.double() @ w.double().T + bias.double()
    aux = torch.empty(M, N, dtype=dtype, device=dev())
    y = F.gemm(x, w, M, N, K, bias=bias, act=2, aux_out=aux, residual=res, lengths=lens, T=T)
    ref = torch.where(mask, torch.nn.functional.gelu(u_ref) + res.double(), 0.0)
    torch.testing.assert_close(y.double(), ref, **tol(dtype))
    torch.testing.assert_close(aux.double(), u_ref, **tol(dtype))
    # relu + fp32 output
    y = F.gemm(x, w, M, N, K, bias=bias, act=1, out_f32=True)
    assert y.dtype == torch.float32
    torch.testing.assert_close(y.double(), torch.relu(u_ref), **tol(dtype))
    # dgrad with fused GELU derivative
    dy = rnd(M, N, dtype=dtype, seed=3)
    wide = rnd(M, K, dtype=dtype, seed=4)     # plays the role of the pre-activation
    dx = F.gemm(dy, w, M, K, N, b_tr=True, dact=2, aux_in=wide, lengths=lens, T=T)
    a = wide.double()
    gp = 0.5 * (1 + torch.erf(a / math.sqrt(2))) + a * torch.exp(-0.5 * a * a) / math.sqrt(2 * math.pi)
    ref = torch.where(mask, (dy.double() @ w.double()) * gp, 0.0)
    torch.testing.assert_close(dx.double(), ref, **tol(dtype))
    # stored-derivative form: the forward launch writes act'(u) to aux_out, the dgrad multiplies by it
    for act_id, fn in ((2, torch.nn.functional.gelu), (3, torch.nn.functional.silu)):
        ud = u_ref.clone().requires_grad_(True)
        fn(ud).sum().backward()
        deriv = torch.empty(M, N, dtype=dtype, device=dev())
        y = F.gemm(x, w, M, N, K, bias=bias, act=act_id | 16, aux_out=deriv)
        torch.testing.assert_close(y.double(), fn(u_ref), **tol(dtype))
        torch.testing.assert_close(deriv.double(), ud.grad, **tol(dtype))
        dx = F.gemm(dy, w, M, K, N, b_tr=True, dact=4, aux_in=wide)
        torch.testing.assert_close(dx.double(), (dy.double() @ w.double()) * wide.double(), **tol(dtype))
    # per-row-tile column sums of the stored result from the same launch (bf16 LDS-DMA path only)
    parts = []
    dx = F.gemm(dy, w, M, K, N, b_tr=True, dact=4, aux_in=wide, lengths=lens, T=T, colpart=parts)
    if dtype == torch.bfloat16:
        assert parts[0] is not None and parts[0].shape[1] == K
        torch.testing.assert_close(parts[0].sum(0).double(), torch.where(mask, (dy.double() @ w.double()) * wide.double(), 0.0).sum(0),
                                   atol=0.15, rtol=2e-2)
        torch.testing.assert_close(parts[0].sum(0), dx.float().sum(0), atol=0.15, rtol=2e-2)
    else:
        assert parts == [None]
    # wgrad, split-K (atomic) and single pass agree with the reference
    for s in (1, 3):
        dW = F.gemm(dy, x, N, K, M, a_tr=True, b_tr=True, out_f32=True, split_k=s)
        torch.testing.assert_close(dW.double(), dy.double().T @ x.double(),
                                   atol=1e-4 if dtype == torch.float32 else 5e-2, rtol=1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("frames", [8 * 125, 8 * 111])     # reduction length: 1000 (tail of 40) and 888 (tail of 56)
def test_wgrad_with_fused_bias_gradient(F, dtype, frames):
    """Weight-gradient launch that also accumulates the bias gradient (colsum_out), split-K and
    single pass, accumulate-into-existing-gradient semantics, ragged frame counts."""
    M, N, K = frames, 328, 192            # N = out features (not a tile multiple), K = in features
    dy, x = rnd(M, N, dtype=dtype), rnd(M, K, dtype=dtype, seed=1)
    ref_w = dy.double().T @ x.double()
    ref_b = dy.double().sum(0)
    wt = dict(atol=1e-3, rtol=1e-4) if dtype == torch.float32 else dict(atol=8e-2, rtol=2e-2)
    for s in (1, 4):
        gw = torch.ones(N, K, device=dev())
        gb = torch.full((N,), 2.0, device=dev())
        F.gemm(dy, x, N, K, M, a_tr=True, b_tr=True, out=gw, split_k=s, accumulate=(s == 1), colsum_out=gb)
        torch.testing.assert_close(gw.double(), ref_w + 1.0, **wt)
        torch.testing.assert_close(gb.double(), ref_b + 2.0, **wt)


LEAN_VARIANTS = {
    "plain": dict(),
    "bias + residual + colpart": dict(bias=True, residual=True, colpart=True),
    "GELU + stored derivative": dict(bias=True, act=2 | 16, aux=True),
    "times stored derivative + colpart": dict(dact=4, aux_in=True, colpart=True),
    "bias + ReLU": dict(bias=True, act=1),
    "SiLU + stored derivative": dict(bias=True, act=3 | 16, aux=True),
    "pre-add + SiLU + stored derivative": dict(bias=True, act=3 | 16, aux=True, pre_add=True),
    "ReLU derivative from the stored output + colpart": dict(dact=1, aux_in=True, colpart=True),
}


@pytest.mark.parametrize("variant", list(LEAN_VARIANTS))
@pytest.mark.parametrize("mode", ["nt", "nn"])
@pytest.mark.parametrize("masked", [False, True])
def test_lean_epilogues_are_bitwise_the_generic_one(F, variant, mode, masked):
    """tile_cfg 13 ends in a compile-time epilogue (tile_epilogue_lean) when the options fit one of three shapes;
    tile_cfg 3 always runs the all-options epilogue.  Same arithmetic in the same order: results must be bitwise
    equal, on ragged tile edges and with a row mask whose sequences (T = 17) end inside 16-row bands."""
    kw = LEAN_VARIANTS[variant]
    M, N, K, T = (85 if masked else 1003), 520, 256, 17
    A = rnd(M, K, dtype=torch.bfloat16)
    B = rnd(N, K, dtype=torch.bfloat16, scale=K ** -0.5, seed=1)
    if mode == "nn":
        B = B.T.contiguous()
    bias, res, der = rnd(N, seed=2), rnd(M, N, dtype=torch.bfloat16, seed=3), rnd(M, N, dtype=torch.bfloat16, seed=4)
    lens = torch.tensor([17, 16, 1, 0, 16], dtype=torch.int32, device=dev())
    outs = {}
    for cfg in (3, 13, 15, 1):
        args = dict(tile_cfg=cfg, b_tr=(mode == "nn"))
        if kw.get("bias"): args["bias"] = bias
        if kw.get("residual"): args["residual"] = res
        if "act" in kw: args["act"] = kw["act"]
        if "dact" in kw: args.update(dact=kw["dact"], aux_in=der)
        if kw.get("pre_add"): args["pre_add"] = res
        aux = torch.zeros(M, N, device=dev(), dtype=torch.bfloat16) if kw.get("aux") else None
        if aux is not None: args["aux_out"] = aux
        part = [] if kw.get("colpart") else None
        if part is not None: args["colpart"] = part
        if masked: args.update(lengths=lens, T=T)
        out = F.gemm(A, B, M, N, K, **args)
        outs[cfg] = (out, aux, part[0] if part else None)
    for cfg in (13, 15, 1):          # 13 / 15 (192-row tiles): every variant lean; 1 (128x128 tiles): the plain / ReLU variants lean
        assert torch.equal(outs[3][0], outs[cfg][0]), cfg
        if outs[3][1] is not None:
            assert torch.equal(outs[3][1], outs[cfg][1]), cfg
    if outs[3][2] is not None:      # per-row-tile partial sums: both tile shapes have 256 rows, the summation order inside differs
        torch.testing.assert_close(outs[3][2].sum(0), outs[13][2].sum(0), atol=2e-2, rtol=1e-3)
    ref = A.float() @ (B.float() if mode == "nn" else B.float().T)
    if variant == "plain" and not masked:
        torch.testing.assert_close(outs[13][0].float(), ref, atol=3e-2, rtol=3e-2)


GROUPED_CASES = {
    # name: [((out features, in features), frames)]   -- frames = the reduction length of that product
    "layer (192 tiles: heads + tails in lockstep)": [((4096, 1024), 2048), ((1024, 4096), 2048), ((3072, 1024), 2048),
                                                     ((1024, 1024), 2048)],
    "three whole rounds": [((8192, 4096), 1024), ((4096, 4096), 1024)],
    "two rounds and 22 tiles left": [((8192, 4096), 1024), ((1024, 1024), 1024), ((520, 768), 1024)],
    "mixed reduction lengths (stream plan)": [((4096, 1024), 2048), ((1024, 4096), 1024), ((3072, 1024), 1088)],
    "ragged tile edges": [((2000, 1016), 1024), ((3072, 1024), 1024), ((1000, 2040), 1024)],
}


@pytest.mark.parametrize("case", list(GROUPED_CASES))
def test_grouped_weight_gradients_exact(F, case):
    """vg_gemm_grouped (one persistent launch for the weight gradients of one backward node): small-integer operands,
    gradients that already hold a value, so every (tile, K range) segment must be added exactly once -- plain
    accumulate for whole-K segments, fp32 atomics for the head / tail pieces of either work plan."""
    g = torch.Generator().manual_seed(11)
    items, refs = [], []
    for (N, K), frames in GROUPED_CASES[case]:
        w = torch.nn.Parameter(torch.zeros(N, K, device=dev()))
        w.grad = torch.randint(-3, 4, (N, K), generator=g).float().to(dev())
        dy = torch.randint(-2, 3, (frames, N), generator=g).float().to(dev()).bfloat16()
        x = torch.randint(-2, 3, (frames, K), generator=g).float().to(dev()).bfloat16()
        refs.append(w.grad.double() + dy.double().T @ x.double())
        items.append((w, dy, x))
    F.sink_wgrad_group(items)
    for (w, _, _), ref in zip(items, refs):
        assert torch.equal(w.grad.double(), ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_and_ffn_autograd(F, dtype):
    B_, T, D, Fd = 2, 75, 256, 512
    M = B_ * T
    lens = lengths_for(B_, T)
    mask = row_mask(lens, T)[:, None]
    x = rnd(M, D, dtype=dtype).requires_grad_(True)
    w1, b1 = rnd(Fd, D, scale=D ** -0.5).requires_grad_(True), rnd(Fd, scale=0.1).requires_grad_(True)
    w2, b2 = rnd(D, Fd, scale=Fd ** -0.5).requires_grad_(True), rnd(D, scale=0.1).requires_grad_(True)
    res = rnd(M, D, dtype=dtype, seed=9).requires_grad_(True)
    y = F.ffn(x, w1, b1, w2, b2, residual=res, lengths=lens, T=T)
    gy = torch.where(mask, rnd(M, D, seed=11), 0.0)
    (y.float() * gy).sum().backward()
    got = [t.grad.clone() for t in (x, w1, b1, w2, b2, res)]
    xs = [t.detach().double().requires_grad_(True) for t in (x, w1, b1, w2, b2, res)]
    x_, w1_, b1_, w2_, b2_, r_ = xs
    if dtype == torch.bfloat16:   # the kernel consumes bf16-rounded weights
        w1q, w2q = w1_.float().bfloat16().double(), w2_.float().bfloat16().double()
        w1q, w2q = w1_ + (w1q - w1_).detach(), w2_ + (w2q - w2_).detach()
    else:
        w1q, w2q = w1_, w2_
    h = torch.nn.functional.gelu(x_ @ w1q.T + b1_)
    yr = torch.where(mask, r_ + h @ w2q.T + b2_, 0.0)
    (yr * gy.double()).sum().backward()
    torch.testing.assert_close(y.double(), yr.detach(), **tol(dtype))
    for a, b, name in zip(got, xs, "x w1 b1 w2 b2 res".split()):
        t = tol(dtype) if dtype == torch.float32 else dict(atol=0.15, rtol=5e-2)
        if dtype == torch.float32:
            t = dict(atol=2e-4, rtol=1e-4)
        torch.testing.assert_close(a.double(), b.grad, msg=lambda m: f"{name}: {m}", **t)
    # Linear + ReLU head with fp32 output
    xx = rnd(M, D, dtype=dtype, seed=21).requires_grad_(True)
    y = F.linear(xx, w1, b1, act="relu", out_f32=True)
    y.sum().backward()
    ref = torch.relu(xx.detach().double() @ w1.detach().double().T + b1.detach().double())
    torch.testing.assert_close(y.double(), ref, **tol(dtype))


# ------------------------------------------------------------------ RMSNorm
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [256, 1024])
def test_rmsnorm(F, dtype, C):
    B_, T = 3, 67
    M = B_ * T
    lens = lengths_for(B_, T)
    mask = row_mask(lens, T)[:, None]
    x = rnd(M, C, dtype=dtype).requires_grad_(True)
    sc = (1 + 0.1 * rnd(C)).requires_grad_(True)
    y = F.rmsnorm(x, sc, 1e-6, lengths=lens, T=T)
    gy = rnd(M, C, seed=2)
    (y.float() * gy).sum().backward()
    xr = x.detach().double().requires_grad_(True)
    sr = sc.detach().double().requires_grad_(True)
    yr = torch.where(mask, sr * xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6), 0.0)
    (yr * gy.double()).sum().backward()
    torch.testing.assert_close(y.double(), yr.detach(), **tol(dtype))
    t = dict(atol=1e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=6e-2, rtol=5e-2)
    torch.testing.assert_close(x.grad.double(), xr.grad, **t)
    torch.testing.assert_close(sc.grad.double(), sr.grad, atol=1e-3 if dtype == torch.float32 else 0.5, rtol=2e-2)


# ------------------------------------------------------------------ attention
def attn_reference(qkv, B, T, H, lengths, slopes):
    D = H * 64
    q, k, v = qkv.double().reshape(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    pos = torch.arange(T, device=qkv.device)
    rel = (pos[:, None] - pos[None, :]).double()
    bias = -slopes.double()[:, None, None] * rel.abs()
    s = q @ k.transpose(-1, -2) / 8.0 + bias[None]
    s = s.masked_fill(pos[None, :] > pos[:, None], float("-inf"))
    o = torch.softmax(s, -1) @ v
    o = o.permute(0, 2, 1, 3).reshape(B * T, D)
    return torch.where(row_mask(lengths, T)[:, None], o, 0.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 200, 4), (1, 1, 4), (2, 130, 16), (1, 333, 12)])
def test_attention_fwd_bwd(F, dtype, shape):
    B_, T, H = shape
    D = H * 64
    lens = lengths_for(B_, T)
    slopes = torch.tensor(F.alibi_slopes(H), dtype=torch.float32, device=dev())
    qkv = rnd(B_ * T, 3 * D, dtype=dtype, seed=T).requires_grad_(True)
    out = F.attention(qkv, slopes, B_, T, H, lens)
    mask = row_mask(lens, T)[:, None]
    go = torch.where(mask, rnd(B_ * T, D, seed=3), 0.0)
    (out.float() * go).sum().backward()
    qr = qkv.detach().double().requires_grad_(True)
    ref = attn_reference(qr, B_, T, H, lens, slopes)
    (ref * go.double()).sum().backward()
    torch.testing.assert_close(out.double(), ref.detach(), **tol(dtype))
    # reference grads on padded rows are exactly zero by the masking invariants
    t = dict(atol=1e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=6e-2, rtol=5e-2)
    torch.testing.assert_close(qkv.grad.double(), qr.grad, **t)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_softmax_rescale_branch(F, dtype):
    """Force the running max to jump in the middle of the key sweep (a spiked key) so the online-softmax rescale
    path is exercised against a full fp64 reference.  The forward sweeps the key tiles from the diagonal DOWN to
    tile 0, so the spike sits on an EARLY key (tile 0) for late queries: their maximum jumps at the last tile of
    the sweep; a second spike on a late key covers an ascending sweep as well."""
    B_, T, H = 1, 256, 4
    D = H * 64
    slopes = torch.tensor(F.alibi_slopes(H), dtype=torch.float32, device=dev())
    for spike, queries in ((10, slice(200, None)), (200, slice(201, None))):
        qkv = rnd(B_ * T, 3 * D, dtype=dtype, scale=0.5)
        with torch.no_grad():
            qkv[spike, D + 192:D + 256] = 12.0        # key `spike` of head 3 (the flattest ALiBi slope) stands out ...
            qkv[queries, 192:256] = 1.5               # ... for these queries
        out = F.attention(qkv, slopes, B_, T, H, None)
        ref = attn_reference(qkv, B_, T, H, torch.tensor([T], device=dev()), slopes)
        # the spiked key must actually dominate those rows (otherwise the branch under test was not taken late)
        q, k = qkv[queries, 192:256].double(), qkv[:, D + 192:D + 256].double()
        sc = q @ k.t() / 8.0 - float(slopes[3]) * (torch.arange(T, device=dev())[queries][:, None]
                                                    - torch.arange(T, device=dev())[None]).clamp_min(0).double()
        causal = torch.arange(T, device=dev())[None] <= torch.arange(T, device=dev())[queries][:, None]
        assert bool((sc.masked_fill(~causal, -1e9).argmax(-1) == spike).all())
        torch.testing.assert_close(out.double(), ref, **tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_decode(F, dtype):
    B_, H, Tmax = 3, 4, 50
    D = H * 64
    slopes = torch.tensor(F.alibi_slopes(H), dtype=torch.float32, device=dev())
    q, kc, vc = rnd(B_, D, dtype=dtype), rnd(B_, Tmax, D, dtype=dtype, seed=1), rnd(B_, Tmax, D, dtype=dtype, seed=2)
    pos = torch.tensor([50, 17, 1], dtype=torch.int32, device=dev())
    out = F.attention_decode(q, kc, vc, slopes, pos, H)
    for b in range(B_):
        n = int(pos[b])
        qq = q[b].double().reshape(H, 1, 64)
        kk = kc[b, :n].double().reshape(n, H, 64).permute(1, 0, 2)
        vv = vc[b, :n].double().reshape(n, H, 64).permute(1, 0, 2)
        dist = (n - 1 - torch.arange(n, device=dev())).double()
        s = qq @ kk.transpose(-1, -2) / 8.0 - slopes.double()[:, None, None] * dist
        ref = (torch.softmax(s, -1) @ vv).reshape(D)
        torch.testing.assert_close(out[b].double(), ref, **tol(dtype))


# ------------------------------------------------------------------ losses / VAE terms
def test_cross_entropy(F):
    B_, T, V = 2, 90, 200
    M = B_ * T
    lens = lengths_for(B_, T)
    mask = row_mask(lens, T)
    logits = rnd(M, V, scale=3.0).requires_grad_(True)
    tgt = torch.randint(0, V, (M,), device=dev())
    loss, amax = F.cross_entropy_sum(logits, tgt, lens, T)
    (loss * 0.37).backward()
    lr = logits.detach().double().requires_grad_(True)
    tr = torch.where(mask, tgt, torch.full_like(tgt, -100))
    ref = torch.nn.functional.cross_entropy(lr, tr, reduction="sum", ignore_index=-100)
    (ref * 0.37).backward()
    assert abs(loss.item() - ref.item()) / abs(ref.item()) < 1e-6
    torch.testing.assert_close(logits.grad.double(), lr.grad, atol=1e-6, rtol=1e-5)
    assert torch.equal(amax.long(), logits.detach().argmax(-1))


def test_vae_terms(F):
    B_, T, Dl = 2, 77, 4
    M = B_ * T
    lens = lengths_for(B_, T)
    mask = row_mask(lens, T)[:, None]
    mu, ls, eps = (rnd(M, Dl, seed=s).requires_grad_(s < 2) for s in range(3))
    z, lq = F.reparameterize(mu, ls, eps, 0.85, lens, T)
    mu_ls = rnd(M, 2 * Dl, seed=5).requires_grad_(True)
    u = rnd(M, Dl, seed=6).requires_grad_(True)
    ldet = rnd(M, seed=7).requires_grad_(True)
    log_p, kl = F.prior_logp_kl(mu_ls, u, ldet, lq, lens, T)
    w = rnd(M, Dl, seed=8)
    (kl * 0.04 + (z * w).sum() + (log_p * w).sum() * 0.1).backward()
    got = [t.grad.clone() for t in (mu, ls, mu_ls, u, ldet)]
    d = [t.detach().double().requires_grad_(True) for t in (mu, ls, mu_ls, u, ldet)]
    mu_, ls_, ml_, u_, ld_ = d
    c = 0.5 * math.log(2 * math.pi)
    zr = torch.where(mask, mu_ + eps.double() * torch.exp(ls_) * 0.85, 0.0)
    lqr = torch.where(mask, -ls_ - 0.5 - c, 0.0)
    mp, lp_ = ml_[:, :Dl], ml_[:, Dl:]
    lpr = ld_[:, None] / Dl - lp_ - c - 0.5 * torch.exp(-2 * lp_) * (u_ - mp) ** 2
    lpr = torch.where(mask, lpr, 0.0)
    klr = (lqr - lpr).mean(-1).sum()
    (klr * 0.04 + (zr * w.double()).sum() + (lpr * w.double()).sum() * 0.1).backward()
    torch.testing.assert_close(z.double(), zr.detach(), atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(log_p.double(), lpr.detach(), atol=1e-5, rtol=1e-5)
    assert abs(kl.item() - klr.item()) / abs(klr.item()) < 1e-5
    for a, b in zip(got, d):
        torch.testing.assert_close(a.double(), b.grad, atol=1e-5, rtol=1e-4)


def test_small_reductions(F):
    x = rnd(1000, 200, dtype=torch.bfloat16)
    torch.testing.assert_close(F.colsum(x).double(), x.double().sum(0), atol=1e-2, rtol=1e-3)
    y = rnd(12345)
    assert abs(F.sum_f32(y).item() - y.double().sum().item()) < 1e-2
    w = rnd(333, 77)
    torch.testing.assert_close(F.shadow(torch.nn.Parameter(w), torch.bfloat16).float(), w.bfloat16().float())


# ------------------------------------------------------------------ conv bottleneck stack (channels-last HIP path)
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("which", ["encoder", "unet"])
def test_bottleneck_resnet_hip_vs_stock(F, full_cfg, precision, which, monkeypatch):
    """BottleNeckResNet on the HIP row/GEMM kernels == the same module on stock torch conv ops
    (forward values and every parameter / input gradient), causal + look-ahead blocks,
    conditioning, time embedding and concat skips included."""
    import copy
    import hipvg
    from hparams.hp import Hparams
    from modules.conv.layers import BottleNeckResNet
    from utils.tensormask import TensorMask
    hipvg.set_precision(precision)
    torch.manual_seed(0)
    B, T = 2, 70
    lens = torch.tensor([70, 41], device=dev())
    mask = torch.arange(T, device=dev())[None] < lens[:, None]
    if which == "encoder":
        hp = Hparams.from_dict(copy.deepcopy(full_cfg["model"]["encoder"]))
        net = BottleNeckResNet(hp, input_dim=80, output_dim=4).to(dev())
        cond = temb = None
    else:
        hp = Hparams.from_dict(copy.deepcopy(full_cfg["model"]["decoder"]["cond_unet"]["unet"]))
        hp.time_dim = 256
        net = BottleNeckResNet(hp, input_dim=80, output_dim=80).to(dev())
        cond = TensorMask(torch.randn(B, T, 32, device=dev()), mask).apply_mask()
        temb = torch.randn(B, 256, device=dev())
    with torch.no_grad():
        for p in net.parameters():
            if p.ndim == 1:
                p.add_(0.05 * torch.randn_like(p))
    x = TensorMask(torch.randn(B, T, 80, device=dev()), mask).apply_mask()
    gy = torch.randn(B, T, net.out_linear.out_features, device=dev())

    def run(stock):
        monkeypatch.setenv("VG_CONV_STOCK", "1" if stock else "0")
        net.zero_grad(set_to_none=True)
        xin = TensorMask(x.value.clone().requires_grad_(True), mask)
        c_in = None if cond is None else TensorMask(cond.value.clone().requires_grad_(True), mask)
        t_in = None if temb is None else temb.clone().requires_grad_(True)
        y = net(xin, c_in, t_in)
        (y.value.float() * gy).sum().backward()
        grads = {k: p.grad.clone() for k, p in net.named_parameters()}
        grads["__x"] = xin.value.grad.clone()
        if c_in is not None:
            grads["__c"] = c_in.value.grad.clone()
            grads["__t"] = t_in.grad.clone()
        return y.value.detach().float(), grads

    y_ref, g_ref = run(stock=True)
    y_hip, g_hip = run(stock=False)
    tight = precision == "fp32"
    torch.testing.assert_close(y_hip, y_ref, atol=2e-4 if tight else 0.15, rtol=1e-4 if tight else 0.05)
    for k in g_ref:
        a, b = g_hip[k].double(), g_ref[k].double()
        if tight:     # fp32: element-wise against the largest entry
            err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-6)
            assert err < 2e-3, (k, err)
        else:         # bf16 storage between kernels: relative L2 error of each gradient tensor
            err = (a - b).norm().item() / (b.norm().item() + 1e-9)
            assert err < 0.06, (k, err)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cnnstack_hip_vs_stock(F, full_cfg, precision, monkeypatch):
    """Utterance encoder: strided convs as window-gather + MFMA GEMM == stock conv1d."""
    import copy
    import hipvg
    from hparams.hp import Hparams
    from modules.conv.layers import CNNStack
    from utils.tensormask import TensorMask
    hipvg.set_precision(precision)
    torch.manual_seed(1)
    hp = Hparams.from_dict(copy.deepcopy(full_cfg["model"]["utterance_encoder"]))
    net = CNNStack(hp, input_dim=80, output_dim=128).to(dev())
    B, T = 3, 150
    lens = torch.tensor([150, 97, 2], device=dev())
    x = TensorMask.fromlength(torch.randn(B, T, 80, device=dev()), lens).apply_mask()
    gy = torch.randn(B, 18, 128, device=dev())

    def run(stock):
        monkeypatch.setenv("VG_CONV_STOCK", "1" if stock else "0")
        net.zero_grad(set_to_none=True)
        y = net(x)
        (y.value.float() * gy).sum().backward()
        return y, {k: p.grad.clone() for k, p in net.named_parameters()}

    y_ref, g_ref = run(True)
    y_hip, g_hip = run(False)
    assert torch.equal(y_ref.mask, y_hip.mask)
    tight = precision == "fp32"
    torch.testing.assert_close(y_hip.value.float(), y_ref.value.float(), atol=2e-4 if tight else 0.1,
                               rtol=1e-4 if tight else 0.05)
    for k in g_ref:
        err = (g_hip[k].double() - g_ref[k].double()).norm().item() / (g_ref[k].double().norm().item() + 1e-9)
        assert err < (1e-3 if tight else 0.12), (k, err)   # bf16: activations and grads are stored in bf16 between kernels


# ------------------------------------------------------------------ coupling flow (row kernel vs stock tensor ops)
def test_coupling_flow_hip_vs_stock(F, full_cfg, monkeypatch):
    """CouplingStack on vg_flow_fwd / vg_flow_bwd == the same module on stock tensor ops: transformed
    latent, masked log-det, gradients of the latent, the conditioning state and every parameter; and
    reverse(forward(z)) == z through vg_flow_reverse.  fp32 (the flow never runs in bf16)."""
    import copy
    import hipvg
    from hparams.hp import Hparams
    from modules.flow.layers import CouplingStack
    from modules.flow.utils import TensorLogdet
    from utils.tensormask import TensorMask
    hipvg.set_precision("fp32")
    torch.manual_seed(3)
    hp = Hparams.from_dict(copy.deepcopy(full_cfg["model"]["transformer"]["flow"]))
    net = CouplingStack(4, hp, condition_dim=256).to(dev())
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.1 * torch.randn_like(p))
    B, T = 3, 50
    lens = torch.tensor([50, 17, 1], device=dev())
    z = TensorMask.fromlength(torch.randn(B, T, 4, device=dev()), lens).apply_mask()
    c = TensorMask.fromlength(torch.randn(B, T, 256, device=dev()), lens).apply_mask()
    gu, gl = torch.randn(B, T, 4, device=dev()), torch.randn(B, T, device=dev())
    valid = z.mask[..., None].float()

    def run(stock):
        monkeypatch.setenv("VG_FLOW_STOCK", "1" if stock else "0")
        net.zero_grad(set_to_none=True)
        zin = TensorMask(z.value.clone().requires_grad_(True), z.mask)
        cin = TensorMask(c.value.clone().requires_grad_(True), c.mask)
        out = net(TensorLogdet(zin, 0.0), c=cin)
        u, ld = out.tensor.value, out.logdet.sum(-1)
        ((u * gu * valid).sum() + (ld * gl).sum()).backward()
        g = {k: p.grad.clone() for k, p in net.named_parameters()}
        g["__z"], g["__c"] = zin.value.grad.clone(), cin.value.grad.clone()
        back = net.reverse(TensorMask(u.detach(), z.mask), c=c).value
        return (u * valid).detach(), ld.detach(), g, back

    u_ref, ld_ref, g_ref, back_ref = run(stock=True)
    u_hip, ld_hip, g_hip, back_hip = run(stock=False)
    torch.testing.assert_close(u_hip, u_ref, atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(ld_hip, ld_ref, atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(back_hip * valid, z.value * valid, atol=1e-4, rtol=1e-4)
    for k in g_ref:
        a, b = g_hip[k].double(), g_ref[k].double()
        err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-9)
        assert err < 1e-3, (k, err)


# ------------------------------------------------------------------ flat AdamW (one launch per gradient bucket)
def test_flat_adamw_matches_torch(F):
    """FlatAdamW bound to the reducer's buckets == torch.optim.AdamW (two parameter groups, decoupled
    weight decay, bias correction) over several steps; the bf16 weight copies follow the parameters
    and the gradient buckets come back zeroed."""
    import copy
    import torch.nn as nn
    from training_lib.dp import GradReducer
    from training_lib.optimizer import FlatAdamW
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(300, 129), nn.LayerNorm(129), nn.Linear(129, 7)).to(dev())
    ref = copy.deepcopy(net)

    def groups(m):
        ps = list(m.parameters())
        return [{"params": [p for p in ps if p.ndim != 1]}, {"params": [p for p in ps if p.ndim == 1], "weight_decay": 0}]

    kw = dict(lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
    opt = FlatAdamW(groups(net), fused=True, **kw)
    opt_ref = torch.optim.AdamW(groups(ref), **kw)
    red = GradReducer(net.parameters(), bucket_mb=0.05)      # several small buckets
    assert len(red.buckets) > 1
    opt.bind(red)
    for step in range(4):
        x = torch.randn(16, 300, device=dev())
        for m in (net, ref):
            m(x).square().sum().backward()
        for g in opt.param_groups + opt_ref.param_groups:     # a schedule: per-step learning rate
            g["lr"] = 3e-3 / (step + 1)
        opt.step()
        opt_ref.step()
        opt_ref.zero_grad(set_to_none=True)
        for b in red.buckets:
            assert float(b["flat"].abs().max()) == 0.0
        for (k, p), q in zip(net.named_parameters(), ref.parameters()):
            torch.testing.assert_close(p.data, q.data, atol=2e-6, rtol=2e-5, msg=lambda m: f"{k} step {step}: {m}")
            assert torch.equal(p._vg_flat_shadow, p.data.bfloat16())
    sd = opt.state_dict()
    assert len(sd["state"]) == len(list(net.parameters())) and float(sd["state"][0]["step"]) == 4.0
    # resume: a fresh optimizer bound to a copy of the model, loaded from that state dict, continues identically
    net2 = copy.deepcopy(net)
    with torch.no_grad():
        for p2, p in zip(net2.parameters(), net.parameters()):
            p2.data = p.data.clone()              # deepcopy kept views into net's flat storage: detach them
    opt2 = FlatAdamW(groups(net2), fused=True, **kw)
    red2 = GradReducer(net2.parameters(), bucket_mb=0.05)
    opt2.bind(red2)
    opt2.load_state_dict(copy.deepcopy(sd))
    x = torch.randn(16, 300, device=dev())
    for m in (net, net2):
        m(x).square().sum().backward()
    opt.step()
    opt2.step()
    for (k, p), q in zip(net.named_parameters(), net2.parameters()):
        assert torch.equal(p.data, q.data), k


# ------------------------------------------------------------------ decode-step kernels
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(1, 200, 1024), (3, 8, 1024), (8, 3072, 1024), (8, 1024, 4096), (16, 136, 64),
                                   (5, 24, 8192), (9, 72, 512), (33, 256, 1024), (64, 40, 256)])
def test_gemm_rows(F, dtype, shape):
    """vg_gemm_rows (the Linear of the autoregressive step) vs fp64: row counts 1..64 (groups of 8 along grid.y), column counts that are
    not multiples of 8, reduction lengths from 64 to 8192, bias / GELU / residual epilogue, fp32 output."""
    M, N, K = shape
    x, w = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, scale=K ** -0.5)
    bias, res = rnd(N, scale=0.1), rnd(M, N, dtype=dtype, seed=2)
    ref = x.double() @ w.double().T + bias.double()
    y = F.rows_linear(x, w, bias, out_f32=True)
    torch.testing.assert_close(y.double(), ref, atol=2e-5 if dtype == torch.float32 else 2e-3, rtol=1e-4)
    y = F.rows_linear(x, w, bias, act=2, residual=res)
    assert y.dtype == dtype
    torch.testing.assert_close(y.double(), torch.nn.functional.gelu(ref) + res.double(), **tol(dtype))
    # fused RMSNorm prologue
    g = 1 + 0.1 * rnd(K, seed=5)
    xn = x.double() * torch.rsqrt(x.double().square().mean(-1, keepdim=True) + 1e-6) * g.double()
    y = F.rows_linear(x, w, bias, out_f32=True, norm_scale=g, norm_eps=1e-6)
    # (bf16: the matrix-core kernel -- every row count since round 5; VG_ROWS_MFMA=17 restores the 8-row kernel below 17
    # rows -- rounds the scaled inputs x * g to bf16, what the training path does when it stores the normed activations,
    # where the 8-row kernel multiplies in fp32)
    from_rows = int(os.environ.get("VG_ROWS_MFMA", "1"))
    mfma = dtype == torch.bfloat16 and from_rows > 0 and M >= from_rows
    torch.testing.assert_close(y.double(), xn @ w.double().T + bias.double(),
                               atol=5e-5 if dtype == torch.float32 else (2e-2 if mfma else 5e-3), rtol=2e-4)


def test_decode_noise(F):
    """vg_decode_noise: a function of (seed, sequence, pos) only; standard normals and uniforms in [0, 1) by their
    moments over 64 sequences x 400 frames; a different seed, sequence or frame gives different numbers."""
    B, n = 64, 4
    pos = torch.zeros(B, dtype=torch.int32, device=dev())
    draws_n, draws_u = [], []
    for t in range(400):
        pos.fill_(t)
        a, u = F.decode_noise(1234, pos, n)
        draws_n.append(a)
        draws_u.append(u)
    a2, u2 = F.decode_noise(1234, pos, n)
    assert torch.equal(a2, draws_n[-1]) and torch.equal(u2, draws_u[-1])          # same key, same numbers
    a3, _ = F.decode_noise(1235, pos, n)
    assert not torch.equal(a3, a2)
    z, u = torch.stack(draws_n).double(), torch.stack(draws_u).double()
    assert torch.isfinite(z).all() and float(u.min()) >= 0.0 and float(u.max()) < 1.0
    assert abs(float(z.mean())) < 0.02 and abs(float(z.var()) - 1.0) < 0.03 and abs(float((z ** 4).mean()) - 3.0) < 0.2
    assert abs(float(u.mean()) - 0.5) < 0.01 and abs(float(u.var()) - 1.0 / 12.0) < 0.005
    # no two (sequence, frame) cells share their numbers; the four normals of a cell are uncorrelated
    flat = z.reshape(-1, n)
    assert torch.unique(flat[:, 0]).numel() > 0.999 * flat.shape[0]
    c = torch.corrcoef(flat.T)
    assert float((c - torch.eye(n, dtype=c.dtype, device=c.device)).abs().max()) < 0.03
    # odd counts: 1 and 7 normals per sequence
    for nn in (1, 7):
        a, _ = F.decode_noise(7, pos, nn)
        assert a.shape == (B, nn) and torch.isfinite(a).all()


def test_decode_sampling_kernels(F):
    """vg_embed_fuse == embedding + relu(Linear); vg_sample_token draws by inverse CDF (exact index for given
    uniforms, frame counter advanced, empirical frequencies follow softmax(logits / T))."""
    B, V, E, Lz = 5, 200, 64, 4
    emb, wf, bf = rnd(V, E), rnd(E, Lz, seed=1), rnd(E, seed=2)
    frame = torch.cat([torch.tensor([[3.], [199.], [0.], [77.], [12.]], device=dev()), rnd(B, Lz, seed=3)], 1)
    out = F.embed_fuse(frame, emb, wf, bf, torch.float32)
    ref = emb[frame[:, 0].long()] + torch.relu(frame[:, 1:] @ wf.T + bf)
    torch.testing.assert_close(out, ref, atol=1e-5, rtol=1e-5)
    logits = rnd(B, V, scale=2.0, seed=4)
    T = 0.7
    probs = torch.softmax(logits.double() / T, -1)
    cdf = probs.cumsum(-1)
    u = torch.tensor([0.0, 0.25, 0.5, 0.75, 0.999], device=dev())
    pos = torch.zeros(B, dtype=torch.int32, device=dev())
    fr = frame.clone()
    F.sample_token(logits, T, u, fr, pos)
    want = (cdf > u.double()[:, None]).float().argmax(-1)
    got = fr[:, 0].long()
    # allow the neighbouring id where u falls within rounding of a CDF step
    for b in range(B):
        assert got[b] == want[b] or abs(float(cdf[b, got[b]]) - float(u[b])) < 1e-5 or \
            (got[b] > 0 and abs(float(cdf[b, got[b] - 1]) - float(u[b])) < 1e-5), (b, got[b], want[b])
    assert torch.equal(pos, torch.ones_like(pos)) and torch.equal(fr[:, 1:], frame[:, 1:])
    # distribution check on one row
    n = 20000
    lg = logits[:1].expand(n, V).contiguous()
    frn = torch.zeros(n, 5, device=dev())
    F.sample_token(lg, T, torch.rand(n, device=dev()), frn, None)
    freq = torch.bincount(frn[:, 0].long(), minlength=V).double() / n
    assert (freq - probs[0]).abs().max() < 0.015


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_decode_append(F, dtype):
    """Cache append + single-query attention == dense softmax over the grown cache, per-sequence positions."""
    B, H, Tmax = 3, 4, 300
    D = H * 64
    kc, vc = rnd(B, Tmax, D, dtype=dtype), rnd(B, Tmax, D, dtype=dtype, seed=1)
    qkv = rnd(B, 3 * D, dtype=dtype, seed=2)
    pos = torch.tensor([0, 129, 298], dtype=torch.int32, device=dev())
    slopes = torch.tensor(F.alibi_slopes(H), dtype=torch.float32, device=dev())
    kref, vref = kc.clone(), vc.clone()
    out = F.attention_decode_append(qkv, kc, vc, slopes, pos, H)
    for b in range(B):
        n = int(pos[b]) + 1
        kref[b, n - 1], vref[b, n - 1] = qkv[b, D:2 * D], qkv[b, 2 * D:]
        assert torch.equal(kc[b, n - 1], kref[b, n - 1]) and torch.equal(vc[b, n - 1], vref[b, n - 1])
        q = qkv[b, :D].double().view(H, 1, 64)
        k = kref[b, :n].double().view(n, H, 64).transpose(0, 1)
        v = vref[b, :n].double().view(n, H, 64).transpose(0, 1)
        s = q @ k.transpose(1, 2) / 8.0 - slopes.double()[:, None, None] * torch.arange(n - 1, -1, -1, device=dev())[None, None]
        ref = (torch.softmax(s, -1) @ v).reshape(D)
        torch.testing.assert_close(out[b].double(), ref, **tol(dtype))
    assert torch.equal(kc[:, 299], kref[:, 299])          # untouched rows stay untouched


@pytest.mark.parametrize("H", [4, 16])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_layer_decode_fused(F, dtype, H):
    """vg_attn_layer_decode (RMSNorm + QKV rows of a head + cache append + attention + out-projection band, accumulated
    into x1 with fp32 atomics) against a float64 restatement of the sub-layer on the same (rounded) weights, per-sequence
    positions including an empty cache, and against the three-launch path; the buffer it is told to clear is cleared,
    cache rows other than pos[b] are untouched.  vg_gemm_rows_mixed (fp32 rows x weights in dtype) rides along."""
    import hipvg
    B, Tmax = 5, 500
    D = H * 64
    g = torch.Generator().manual_seed(7 + H)
    x = torch.randn(B, D, generator=g).to(dev())
    g1 = (1 + 0.1 * torch.randn(D, generator=g)).to(dev())
    wqkv = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev()).to(dtype)
    wo = (torch.randn(D, D, generator=g) * D ** -0.5).to(dev()).to(dtype)
    bq, bo = (0.1 * torch.randn(3 * D, generator=g)).to(dev()), (0.1 * torch.randn(D, generator=g)).to(dev())
    kc, vc = rnd(B, Tmax, D, dtype=dtype), rnd(B, Tmax, D, dtype=dtype, seed=1)
    pos = torch.tensor([0, 1, 129, 400, 498], dtype=torch.int32, device=dev())
    slopes = torch.tensor(F.alibi_slopes(H), dtype=torch.float32, device=dev())
    k0, v0 = kc.clone(), vc.clone()
    x1 = torch.zeros(B, D, device=dev())
    dirty = torch.full((B, D), 3.0, device=dev())
    F.attention_layer_decode(x, g1, 1e-6, wqkv, bq, wo, bo, kc, vc, slopes, pos, H, x1, zero=dirty)
    assert torch.all(dirty == 0)
    # float64 restatement; the projection and the context are rounded to dtype as the kernels do
    xd = x.double()
    xn = xd * torch.rsqrt((xd * xd).mean(-1, keepdim=True) + 1e-6) * g1.double()
    if dtype == torch.bfloat16:        # the bf16 kernel multiplies bf16(x * g) and applies rstd to the sums
        rstd = torch.rsqrt((xd * xd).mean(-1, keepdim=True) + 1e-6)
        xn = (x * g1).to(dtype).double() * rstd
    qkv = (xn @ wqkv.double().T + bq.double()).to(dtype)
    for b in range(B):
        n = int(pos[b]) + 1
        assert torch.equal(kc[b, n - 1].float(), qkv[b, D:2 * D].float()) or torch.allclose(
            kc[b, n - 1].float(), qkv[b, D:2 * D].float(), **tol(dtype))
        kk, vv = k0[b].clone(), v0[b].clone()
        kk[n - 1], vv[n - 1] = kc[b, n - 1], vc[b, n - 1]
        q = qkv[b, :D].double().view(H, 1, 64)
        k = kk[:n].double().view(n, H, 64).transpose(0, 1)
        v = vv[:n].double().view(n, H, 64).transpose(0, 1)
        s = q @ k.transpose(1, 2) / 8.0 - slopes.double()[:, None, None] * torch.arange(n - 1, -1, -1, device=dev())[None, None]
        ctx = (torch.softmax(s, -1) @ v).reshape(D).to(dtype).double()
        ref = xd[b] + bo.double() + wo.double() @ ctx
        torch.testing.assert_close(x1[b].double(), ref, **tol(dtype))
        keep = torch.ones(Tmax, dtype=torch.bool, device=dev())
        keep[n - 1] = False
        assert torch.equal(kc[b][keep], k0[b][keep]) and torch.equal(vc[b][keep], v0[b][keep])
    # the three-launch path on the same inputs (input rows in dtype)
    kc2, vc2 = k0.clone(), v0.clone()
    q3 = F.rows_linear(x.to(dtype), wqkv, bq, norm_scale=g1, norm_eps=1e-6)
    c3 = F.attention_decode_append(q3, kc2, vc2, slopes, pos, H)
    y3 = F.rows_linear(c3, wo, bo, residual=x.to(dtype), out_f32=True)
    torch.testing.assert_close(x1, y3, atol=3e-2 if dtype == torch.bfloat16 else 1e-4, rtol=3e-2 if dtype == torch.bfloat16 else 1e-4)
    # fp32 rows x weights in dtype, norm fused, residual fp32, a buffer to clear
    w1 = (torch.randn(2 * D, D, generator=g) * D ** -0.5).to(dev()).to(dtype)
    dirty.fill_(1.0)
    y = F.rows_linear_mixed(x1, w1, None, act=hipvg.ACT_GELU, norm_scale=g1, norm_eps=1e-6, out_f32=True, zero=dirty)
    x1d = x1.double()
    n1 = x1d * torch.rsqrt((x1d * x1d).mean(-1, keepdim=True) + 1e-6) * g1.double()
    torch.testing.assert_close(y.double(), torch.nn.functional.gelu(n1 @ w1.double().T), **tol(dtype))
    assert torch.all(dirty == 0)
    z = F.rows_linear_mixed(y, (torch.randn(D, 2 * D, generator=g) * D ** -0.5).to(dev()).to(dtype), bo, residual=x1, out_f32=True)
    assert z.dtype == torch.float32 and torch.isfinite(z).all()


def test_rccl_bucket_allreduce_through_the_c_abi():
    """vg_comm_unique_id / vg_comm_init / vg_allreduce_bucket / vg_comm_destroy on a one-rank communicator (two
    ranks cannot share a device under RCCL): mean and sum over one rank leave fp32 and bf16 buckets unchanged, the
    call is stream-ordered, and the reducer's 'abi' backend drives the same entry point."""
    from hipvg import comm
    from training_lib.dp import GradReducer
    comm.init(0, 1)
    assert comm.world() == 1
    side = torch.cuda.Stream()
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.randn(3 * 256 * 1024 + 256, device=dev()).to(dtype)
        ref = x.clone()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            comm.all_reduce_(x, average=True)
            comm.all_reduce_(x, average=False)
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(x, ref)
    lin = torch.nn.Linear(512, 512).to(dev())
    red = GradReducer(lin.parameters(), bucket_mb=0.5, comm="abi")
    red.buckets[0]["flat"].fill_(3.0)
    red._launch(red.buckets[0])
    red.finish()
    torch.cuda.synchronize()
    assert float(red.buckets[0]["flat"].min()) == 3.0 and float(red.buckets[0]["flat"].max()) == 3.0
    comm.destroy()
    assert comm.world() == 0


def test_embed_fuse_train_matches_stock_modules(F):
    """vg_embed_fuse_fwd / _bwd against Embedding (masked) + Linear(4 -> 64) + ReLU (not masked) + add in stock
    PyTorch: output, d z, and the gradients of the table, Wf and bf; ragged lengths incl. an empty sequence."""
    B, T, D, E, V = 3, 70, 4, 64, 200
    g = torch.Generator().manual_seed(31)
    ids = torch.randint(0, V, (B, T), generator=g).to(dev())
    z = torch.randn(B, T, D, generator=g).to(dev()).requires_grad_(True)
    emb = torch.nn.Parameter((torch.rand(V, E, generator=g) * 2 - 1).to(dev()))
    wf = torch.nn.Parameter((torch.randn(E, D, generator=g) * 0.5).to(dev()))
    bf = torch.nn.Parameter((torch.randn(E, generator=g) * 0.3).to(dev()))
    lens = torch.tensor([70, 0, 33], dtype=torch.int32, device=dev())
    mask = (torch.arange(T, device=dev())[None] < lens[:, None])
    dout = torch.randn(B * T, E, generator=g).to(dev())
    ref = torch.nn.functional.embedding(ids, emb) * mask[..., None] + torch.relu(z @ wf.T + bf)
    ref.reshape(-1, E).backward(dout)
    want = [t.grad.clone() for t in (z, emb, wf, bf)]
    for t in (z, emb, wf, bf):
        t.grad = None
    out = F.embed_fuse_train(ids.reshape(-1), z.reshape(-1, D), emb, wf, bf, lens, T)
    torch.testing.assert_close(out, ref.detach().reshape(-1, E), rtol=1e-6, atol=1e-6)
    out.backward(dout)
    for name, t, w in zip(("dz", "demb", "dWf", "dbf"), (z, emb, wf, bf), want):
        torch.testing.assert_close(t.grad, w, rtol=2e-5, atol=2e-5, msg=name)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_diffusion_loss_kernels_match_stock_arithmetic(F, dtype):
    """vg_qsample and vg_l1_rows_fwd/bwd against the tensor expressions of GaussianDiffusion1D.q_sample / p_losses
    with masked_l1_loss (ragged lengths incl. an empty sequence): x_t exact, loss and d pred to rounding."""
    B, T, C = 3, 50, 80
    g = torch.Generator().manual_seed(17)
    x0 = torch.randn(B, T, C, generator=g).to(dev())
    noise = torch.randn(B, T, C, generator=g).to(dev())
    ca, cs = torch.rand(1000, generator=g).to(dev()), torch.rand(1000, generator=g).to(dev())
    t = torch.randint(0, 1000, (B,), generator=g).to(dev())
    lens = torch.tensor([50, 0, 21], dtype=torch.int32, device=dev())
    mask = (torch.arange(T, device=dev())[None] < lens[:, None])[..., None]
    want = torch.where(mask, ca[t][:, None, None] * x0 + cs[t][:, None, None] * noise, torch.zeros(()).to(dev()))
    got = F.qsample(x0.reshape(-1, C), noise.reshape(-1, C), ca, cs, t, lens, T)
    torch.testing.assert_close(got, want.reshape(-1, C), rtol=1e-6, atol=1e-6)
    pred = torch.randn(B, T, C, generator=g).to(dev()).to(dtype).requires_grad_(True)
    ref = (torch.where(mask, pred.float(), torch.zeros(()).to(dev())) - torch.where(mask, noise, torch.zeros(()).to(dev()))).abs().mean(-1).sum()
    ref.backward()
    want_g, pred.grad = pred.grad.clone(), None
    loss = F.masked_l1_sum(pred.reshape(-1, C), noise.reshape(-1, C), lens, T)
    torch.testing.assert_close(loss, ref.detach(), rtol=1e-5, atol=1e-4)
    (loss * 1.0).backward()
    torch.testing.assert_close(pred.grad.float(), want_g.float(), rtol=1e-2 if dtype == torch.bfloat16 else 1e-6, atol=1e-6)


def test_colsum_multi_folds_several_partial_arrays_in_one_launch(F):
    """vg_colsum_multi through hipvg.functional.vec_grads: three partial-sum arrays of different shapes -> two
    sunk parameters (accumulated into existing .grad) and one dense result; an ineligible source (bf16) takes the
    single path inside the same call."""
    g = torch.Generator().manual_seed(2)
    mk = lambda r, c: torch.randn(r, c, generator=g).to(dev())
    srcs = [mk(512, 1024), mk(63, 4096), mk(7, 64), mk(300, 128).bfloat16()]
    ps = [torch.nn.Parameter(torch.zeros(1024, device=dev())), torch.nn.Parameter(torch.zeros(4096, device=dev())),
          torch.zeros(64, device=dev()), torch.nn.Parameter(torch.zeros(128, device=dev()))]
    ps[0].grad = torch.full_like(ps[0], 2.0)
    outs = F.vec_grads(list(zip(ps, srcs)))
    assert outs[0] is None and outs[1] is None and outs[3] is None
    torch.testing.assert_close(ps[0].grad, 2.0 + srcs[0].sum(0), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(ps[1].grad, srcs[1].sum(0), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(outs[2], srcs[2].sum(0), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ps[3].grad, srcs[3].float().sum(0), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nseg,T,C", [(16, 1000, 512), (3, 77, 1024), (5, 1, 512)])
def test_segment_column_sums(F, dtype, nseg, T, C):
    """vg_colsum_segments (time-embedding gradient of a conv block): per-sequence sums over time of [B * T, C] rows."""
    x = rnd(nseg * T, C, dtype=dtype)
    got = F.segment_colsum(x, nseg)
    want = x.view(nseg, T, C).float().sum(1)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum_partials_of_several_large_matrices_in_one_launch(F, dtype):
    """vg_colsum_partials_multi (the three or four bias gradients of a Transformer layer's backward): partial sums of
    matrices of different widths, strided rows included, folded by vec_grads into sunk gradients."""
    g = torch.Generator().manual_seed(4)
    M = 4001                                         # not a multiple of the row blocking
    wide = torch.randn(M, 3072 + 64, generator=g).to(dev()).to(dtype)
    mats = [torch.randn(M, 1024, generator=g).to(dev()).to(dtype), wide[:, :3072], torch.randn(M, 72 * 8, generator=g).to(dev()).to(dtype)]
    parts = F.colsum_partials(mats)
    assert parts is not None and all(q.shape[1] == x.shape[1] for q, x in zip(parts, mats))
    ps = [torch.nn.Parameter(torch.zeros(x.shape[1], device=dev())) for x in mats]
    ps[1].grad = torch.full_like(ps[1], -1.0)
    outs = F.vec_grads(list(zip(ps, parts)))
    assert all(o is None for o in outs)
    tl = dict(rtol=1e-5, atol=2e-3) if dtype == torch.float32 else dict(rtol=1e-5, atol=2e-3)
    for i, (p, x) in enumerate(zip(ps, mats)):
        torch.testing.assert_close(p.grad, x.float().sum(0) + (-1.0 if i == 1 else 0.0), **tl)
    assert F.colsum_partials([mats[0], mats[0][:100]]) is None       # different row counts: caller takes the single path


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,relu", [(128, True), (256, False), (72, True)])
def test_narrow_channel_norm_matches_tensor_ops(F, dtype, C, relu):
    """vg_chnorm_fwd / _bwd (unbiased variance, optional fused ReLU) against the tensor expression the narrow
    utterance-encoder layers used before: output, dx, dgamma, dbeta."""
    M = 333
    g = torch.Generator().manual_seed(C)
    x = (torch.randn(M, C, generator=g) * 1.5 + 0.3).to(dev()).to(dtype).requires_grad_(True)
    w = torch.nn.Parameter((torch.rand(C, generator=g) + 0.5).to(dev()))
    b = torch.nn.Parameter((torch.randn(C, generator=g) * 0.2).to(dev()))
    dy = torch.randn(M, C, generator=g).to(dev()).to(dtype)
    xf = x.float()
    var, mean = torch.var_mean(xf, dim=-1, keepdim=True)
    ref = w * ((xf - mean) * torch.rsqrt(var + 1e-5)) + b
    if relu:
        ref = torch.relu(ref)
    ref.backward(dy.float())
    want = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    y = F.narrow_channel_norm(x, w, b, eps=1e-5, relu=relu)
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(y.float(), ref.detach(), **tol)
    y.backward(dy)
    if dtype == torch.float32:
        torch.testing.assert_close(x.grad, want[0], rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(w.grad, want[1], rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(b.grad, want[2], rtol=1e-3, atol=1e-3)
    else:       # bf16: the ReLU mask is taken from the rounded output, a handful of near-zero elements may differ
        assert (x.grad.float() - want[0].float()).abs().mean() < 2e-2 * want[0].float().abs().mean() + 1e-3
        torch.testing.assert_close(w.grad, want[1], rtol=5e-2, atol=0.5)
        torch.testing.assert_close(b.grad, want[2], rtol=5e-2, atol=0.5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,stride,pl,pr", [(4, 2, 1, 1), (4, 2, 3, 0), (3, 1, 1, 1), (5, 3, 2, 1)])
def test_conv_gather_and_its_adjoint(F, dtype, k, stride, pl, pr):
    """vg_conv_gather / vg_conv_scatter against pad + unfold + permute (and autograd's backward of that)."""
    B, T, C = 3, 37, 24
    g = torch.Generator().manual_seed(k * 10 + stride)
    x = torch.randn(B, T, C, generator=g).to(dev()).to(dtype).requires_grad_(True)
    win = torch.nn.functional.pad(x, (0, 0, pl, pr)).unfold(1, k, stride)
    ref = win.permute(0, 1, 3, 2).reshape(B * win.shape[1], k * C)
    dr = torch.randn(ref.shape, generator=g).to(dev()).to(dtype)
    ref.backward(dr)
    want, x.grad = x.grad.clone(), None
    rows = F.conv_gather(x, k, stride, pl, pr)
    assert torch.equal(rows, ref.detach())
    rows.backward(dr)
    torch.testing.assert_close(x.grad.float(), want.float(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-6, atol=1e-2 if dtype == torch.bfloat16 else 1e-6)


def test_kernels_without_atomics_are_bitwise_repeatable(F):
    """SURVEY section 5 (no GPU sanitizer on ROCm): determinism checks.  The same inputs twice through the kernels
    that do not use floating-point atomics -- tile GEMM forward / dgrad (bf16, every operand mode, GELU + stored
    derivative epilogue), attention forward and backward, RMSNorm forward and backward -- must give bitwise
    identical results (an uninitialised LDS read or a race shows up as a flipped bit)."""
    import hipvg
    L, p, st = hipvg.lib(), hipvg.ptr, hipvg.stream()
    B, T, H = 3, 200, 4
    D = H * 64
    M = B * T
    lens = torch.tensor([200, 77, 1], dtype=torch.int32, device=dev())
    x = rnd(M, D, dtype=torch.bfloat16)
    w = rnd(3 * D, D, dtype=torch.bfloat16, scale=D ** -0.5)
    bias = rnd(3 * D, scale=0.1)

    def once():
        out = {}
        aux = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev())
        out["nt"] = F.gemm(x, w, M, 3 * D, D, bias=bias, act=hipvg.ACT_GELU | hipvg.ACT_SAVE_DERIV, aux_out=aux,
                           lengths=lens, T=T)
        out["aux"] = aux
        qkv = out["nt"]
        out["nn"] = F.gemm(qkv, w, M, D, 3 * D, b_tr=True, lengths=lens, T=T)
        slopes = torch.tensor(F.alibi_slopes(H), dtype=torch.float32, device=dev())
        att = torch.empty(M, D, dtype=torch.bfloat16, device=dev())
        lse = torch.empty(H, B, T, dtype=torch.float32, device=dev())      # [H][B * T] (include/vaegslm_hip.h)
        hipvg.check(L.vg_attn_fwd(p(qkv), p(att), p(lse), p(slopes), B, T, H, p(lens), 1, st), "attn_fwd")
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        hipvg.check(L.vg_attn_bwd(p(qkv), p(att), p(x), p(lse), p(slopes), p(dqkv), p(delta), B, T, H, p(lens), 1, st),
                    "attn_bwd")
        out.update(att=att, lse=lse, dqkv=dqkv)
        sc = torch.rand(D, device=dev()) + 0.5 if False else torch.full((D,), 1.25, device=dev())
        y, rstd = F.rmsnorm_fwd_raw(x, sc, 1e-6, lens, T)
        dx, part = F.rmsnorm_bwd_raw(att, x, sc, rstd, None, lens, T)
        out.update(rms_y=y, rms_dx=dx, rms_part=part)
        return out

    a, b = once(), once()
    torch.cuda.synchronize()
    valid = (torch.arange(T, device=dev())[None] < lens[:, None])[None].expand(H, B, T)
    for k in a:
        if k == "lse":          # the log-sum-exp of padded query rows is never written (nor read)
            assert torch.equal(a[k][valid], b[k][valid]), k
        else:
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("variant", list(LEAN_VARIANTS))
@pytest.mark.parametrize("mode", ["nt", "nn"])
def test_lean_epilogues_bitwise_on_launches_of_more_than_one_round(F, variant, mode):
    """The same comparison on launches of more than one round of tiles (289 tiles of 256 rows, 374 of 192: every CU gets
    a second block), ragged in both directions, with a row mask of 100 random lengths whose sequences (T = 41) end
    inside 16-row bands: every lean variant bitwise equal to the all-options epilogue of tile_cfg 3."""
    kw = LEAN_VARIANTS[variant]
    M, N, K, T = 4100, 4104, 192, 41
    A = rnd(M, K, dtype=torch.bfloat16)
    B = rnd(N, K, dtype=torch.bfloat16, scale=K ** -0.5, seed=1)
    if mode == "nn":
        B = B.T.contiguous()
    bias, res, der = rnd(N, seed=2), rnd(M, N, dtype=torch.bfloat16, seed=3), rnd(M, N, dtype=torch.bfloat16, seed=4)
    lens = torch.randint(0, T + 1, (M // T,), generator=torch.Generator().manual_seed(5)).to(torch.int32).to(dev())
    outs = {}
    for cfg in (3, 13, 15):
        args = dict(tile_cfg=cfg, b_tr=(mode == "nn"), lengths=lens, T=T)
        if kw.get("bias"): args["bias"] = bias
        if kw.get("residual"): args["residual"] = res
        if "act" in kw: args["act"] = kw["act"]
        if "dact" in kw: args.update(dact=kw["dact"], aux_in=der)
        if kw.get("pre_add"): args["pre_add"] = res
        aux = torch.zeros(M, N, device=dev(), dtype=torch.bfloat16) if kw.get("aux") else None
        if aux is not None: args["aux_out"] = aux
        part = [] if kw.get("colpart") else None
        if part is not None: args["colpart"] = part
        out = F.gemm(A, B, M, N, K, **args)
        outs[cfg] = (out, aux, part[0] if part else None)
    for cfg in (13, 15):
        assert torch.equal(outs[3][0], outs[cfg][0]), cfg
        if outs[3][1] is not None:
            assert torch.equal(outs[3][1], outs[cfg][1]), cfg
    if outs[3][2] is not None:
        torch.testing.assert_close(outs[3][2].sum(0), outs[13][2].sum(0), atol=5e-2, rtol=1e-3)
