"""N > 1 data-parallel path on CPU: two processes, gloo backend.  The bucketed
reducer must (a) average gradients over ranks exactly like DDP, (b) reduce only
on the last micro-batch of an accumulation window, (c) keep param.grad as a
view into its flat bucket."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, bucket_mb, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from training_lib.dp import GradReducer
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8))
    red = GradReducer(model.parameters(), bucket_mb=bucket_mb)
    assert red.world == world
    for p in model.parameters():
        assert p.grad is not None and p.grad.untyped_storage().data_ptr() in {
            b["flat"].untyped_storage().data_ptr() for b in red.buckets}
    g = torch.Generator().manual_seed(100 + rank)
    xs = [torch.randn(4, 16, generator=g) for _ in range(2)]
    # accumulation window of two micro-batches: reduce only on the second
    red.sync_now = False
    model(xs[0]).pow(2).sum().backward()
    local_first = [p.grad.clone() for p in model.parameters()]
    red.sync_now = True
    model(xs[1]).pow(2).sum().backward()
    red.finish()
    reduced = [p.grad.clone() for p in model.parameters()]
    # reference: plain autograd on every rank's data, averaged by hand
    ref_model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8))
    ref_model.load_state_dict(model.state_dict())
    total = [torch.zeros_like(p) for p in ref_model.parameters()]
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        for _ in range(2):
            x = torch.randn(4, 16, generator=gr)
            ref_model.zero_grad()
            ref_model(x).pow(2).sum().backward()
            for t, p in zip(total, ref_model.parameters()):
                t += p.grad / world
    ok = all(torch.allclose(a, b, atol=1e-5, rtol=1e-5) for a, b in zip(reduced, total))
    # first micro-batch stayed local (differs across ranks)
    gathered = [torch.zeros_like(local_first[0]) for _ in range(world)]
    dist.all_gather(gathered, local_first[0])
    local_only = not torch.allclose(gathered[0], gathered[1])
    red.zero_grad()
    zeroed = all(float(p.grad.abs().sum()) == 0.0 for p in model.parameters())
    out[rank] = bool(ok and local_only and zeroed and len(red.buckets) >= 1)
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb", [50.0, 0.001])
def test_reducer_two_ranks_gloo(bucket_mb):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), bucket_mb, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


class _SinkLinear(torch.autograd.Function):
    """Mimics hipvg's gradient sink on CPU: the weight gradient is accumulated straight into ``weight.grad``
    and reported through ``_vg_grad_hooks``; backward returns None for it (autograd still runs the
    parameter's post-accumulate hooks for a None gradient)."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x)
        ctx.weight = weight
        return x @ weight.t()

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        w = ctx.weight
        w.grad.add_(g.t() @ x)
        for h in getattr(w, "_vg_grad_hooks", ()):
            h(w)
        return g @ w, None


def _sink_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from training_lib.dp import GradReducer
    torch.manual_seed(0)
    w1 = torch.nn.Parameter(torch.randn(6, 5))
    lin2 = torch.nn.Linear(6, 3)
    params = [w1, lin2.weight, lin2.bias]
    red = GradReducer(params, bucket_mb=50.0)          # one bucket: fires when all three have reported
    launches = []
    orig = red._launch
    red._launch = lambda b: (launches.append(1), orig(b))[1]
    g = torch.Generator().manual_seed(7 + rank)
    ok = True
    for step in range(3):
        x = torch.randn(4, 5, generator=g)
        red.new_backward()
        lin2(_SinkLinear.apply(x, w1)).pow(2).sum().backward()
        red.finish()
        ok &= all(b["pending"] == b["need"] for b in red.buckets)       # counters balanced after every pass
        ok &= len(launches) == step + 1                                   # exactly one all-reduce per pass
        # reference gradient, averaged over ranks by hand
        tot = torch.zeros_like(w1)
        for r in range(world):
            gr = torch.Generator().manual_seed(7 + r)
            for _ in range(step + 1):
                xr = torch.randn(4, 5, generator=gr)
            wr = w1.detach().clone().requires_grad_(True)
            torch.nn.functional.linear(xr @ wr.t(), lin2.weight.detach(), lin2.bias.detach()).pow(2).sum().backward()
            tot += wr.grad / world
        ok &= bool(torch.allclose(w1.grad, tot, atol=1e-5, rtol=1e-5))
        red.zero_grad()
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_sunk_gradients_report_once_per_backward():
    """A parameter whose gradient is sunk reports readiness twice per backward (sink + autograd's hook for the
    None gradient); the reducer must count it once, or the bucket's all-reduce starts before the bucket is full."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sink_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def _reuse_worker(rank, world, port, out):
    """A sunk parameter that is used TWICE in the forward (its gradient sink reports after each contribution) next
    to a parameter that gets no gradient at all: the bucket must be reduced after the last contribution, exactly
    once per pass, and the result must be the rank average of the full gradient."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from training_lib.dp import GradReducer
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(5, 5))
    unused = torch.nn.Parameter(torch.randn(3))
    lin = torch.nn.Linear(5, 2)
    red = GradReducer([w, unused, lin.weight, lin.bias], bucket_mb=50.0)
    launches, snapshots = [], []
    orig = red._launch

    def spy(b):
        launches.append(1)
        snapshots.append(w.grad.clone())          # local (un-reduced) gradient of `w` at launch time
        return orig(b)
    red._launch = spy
    g = torch.Generator().manual_seed(3 + rank)
    ok = True
    for step in range(3):
        x = torch.randn(4, 5, generator=g)
        red.new_backward(signature=("twice",))
        y = lin(_SinkLinear.apply(_SinkLinear.apply(x, w), w))      # w used twice
        y.pow(2).sum().backward()
        red.finish()
        ok &= len(launches) == step + 1
        wr = w.detach().clone().requires_grad_(True)
        torch.nn.functional.linear((x @ wr.t()) @ wr.t(), lin.weight.detach(), lin.bias.detach()).pow(2).sum().backward()
        ok &= bool(torch.allclose(snapshots[-1], wr.grad, atol=1e-5, rtol=1e-5))     # BOTH contributions were in
        gathered = [torch.zeros_like(wr.grad) for _ in range(world)]
        dist.all_gather(gathered, wr.grad)
        ok &= bool(torch.allclose(w.grad, sum(gathered) / world, atol=1e-5, rtol=1e-5))
        ok &= float(unused.grad.abs().sum()) == 0.0
        ok &= all(b["pending"] == b["need"] for b in red.buckets)
        red.zero_grad()
    # a different step structure under the SAME signature is detected instead of silently racing
    x = torch.randn(4, 5, generator=g)
    red.new_backward(signature=("twice",))
    lin(_SinkLinear.apply(_SinkLinear.apply(_SinkLinear.apply(x, w), w), w)).pow(2).sum().backward()
    try:
        red.finish()
        ok = False
    except RuntimeError as exc:
        ok &= "more often" in str(exc)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_reused_sunk_parameter_is_reduced_after_its_last_contribution():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_reuse_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def _wire_worker(rank, world, port, out):
    """hip.comm_dtype = bf16 (round 6): the same two-rank accumulation window through the bf16 wire and the fp32 wire."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from training_lib.dp import GradReducer
    res = {}
    for wire in ("fp32", "bf16"):
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8))
        red = GradReducer(model.parameters(), bucket_mb=0.004, wire_dtype=wire)       # several buckets
        assert all((b["wire"] is not None) == (wire == "bf16") for b in red.buckets)
        g = torch.Generator().manual_seed(100 + rank)
        red.sync_now = False
        model(torch.randn(4, 16, generator=g)).pow(2).sum().backward()
        red.sync_now = True
        model(torch.randn(4, 16, generator=g)).pow(2).sum().backward()
        red.finish()
        res[wire] = torch.cat([b["flat"].clone() for b in red.buckets])
        # every rank holds the same averaged gradient, as fp32 values the optimizer reads in place
        gathered = [torch.zeros_like(res[wire]) for _ in range(world)]
        dist.all_gather(gathered, res[wire])
        assert torch.equal(gathered[0], gathered[1]) and res[wire].dtype == torch.float32
    a, b = res["fp32"], res["bf16"]
    # each rank's gradient is rounded to bf16 (2^-9 relative), the sum of two again, the halving is exact: the difference
    # is <= 2^-7 of the gradient's norm (element-wise the two ranks' terms may cancel), and not zero (the wire really was
    # bf16: every value the optimizer reads is a bf16 number)
    rel = float((a - b).norm() / a.norm())
    out[rank] = (rel <= 2.0 ** -7, not torch.equal(a, b), bool((b == b.bfloat16().float()).all()), rel)
    dist.destroy_process_group()


def test_bf16_wire_two_ranks_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_wire_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        assert tuple(out[r][:3]) == (True, True, True), dict(out)
