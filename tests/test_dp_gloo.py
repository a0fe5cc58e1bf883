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
