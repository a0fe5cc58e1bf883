"""Round-4 parity additions.

* Deferred weight gradients (``hipvg.functional.defer_vec_grads`` bracket, ``vg_gemm_grouped`` with up to 48 products):
  the queue of several backward nodes leaves as few large launches.  Exact on small integers (every partial sum is an
  exactly representable integer), so a (tile, K range) segment added twice, dropped, or two products racing on one
  gradient change the result:
    - four Transformer layers = 768 tiles = three whole rounds of 256 CUs (flushed by the whole-rounds rule),
    - ~240 tiles of mixed shapes = one nearly full round run as whole tiles (the plan's full-tail round),
    - the ragged small products of the heads (8 x 1024, 200 x 1024, 512 x 80 ...) inside such a launch,
    - a weight that comes back a second time inside one bracket (shared module: the queue must flush first).
  Reference: the nn.Linear / k = 1 Conv1d weight gradients autograd computes for
  modules/transformer/layers.py:52,79,82,151 and modules/conv/layers.py of the reference.
"""
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


def _make(shapes, frames, g):
    """[(N, Ktot, col0, cols)] -> (items, weights, refs); a repeated (N, Ktot, id) addresses the same weight."""
    weights, items, refs = {}, [], {}
    for i, (N, Ktot, col0, cols, wid) in enumerate(shapes):
        key = (N, Ktot, wid)
        if key not in weights:
            w = torch.nn.Parameter(torch.zeros(N, Ktot, device=dev()))
            w.grad = torch.randint(-3, 4, (N, Ktot), generator=g).float().to(dev())
            weights[key] = w
            refs[key] = w.grad.double().clone()
        w = weights[key]
        dy = torch.randint(-2, 3, (frames, N), generator=g).float().to(dev()).bfloat16()
        x = torch.randint(-2, 3, (frames, cols), generator=g).float().to(dev()).bfloat16()
        refs[key][:, col0:col0 + cols] += dy.double().T @ x.double()
        items.append((w, dy, x, col0))
    return items, weights, refs


LAYER = [(4096, 1024), (1024, 4096), (3072, 1024), (1024, 1024)]


@pytest.mark.parametrize("frames", [4096, 16000])
def test_deferred_layer_groups_leave_in_whole_rounds(F, frames, monkeypatch):
    """Four layers' products queued node by node (tag "layer"): flushed as ONE launch of 768 tiles when the fourth
    arrives; a fifth layer stays queued until the bracket closes."""
    g = torch.Generator().manual_seed(frames)
    launches = []
    real = F._launch_wgrad_items
    monkeypatch.setattr(F, "_launch_wgrad_items", lambda items: (launches.append(len(items)), real(items))[1])
    shapes = [(N, K, 0, K, layer) for layer in range(5) for (N, K) in LAYER]
    items, weights, refs = _make(shapes, frames, g)
    F.defer_vec_grads(True)
    try:
        for layer in range(5):
            F.sink_wgrad_group(items[4 * layer:4 * layer + 4], tag="layer")
            assert launches == ([] if layer < 3 else [16])
    finally:
        F.defer_vec_grads(False)
    assert launches == [16, 4]
    for key, w in weights.items():
        assert torch.equal(w.grad.double(), refs[key]), key


def test_deferred_mixed_products_as_one_nearly_full_round(F, monkeypatch):
    """The conv blocks' products (with a column slice) and the small, ragged products of the heads in one bracket:
    ~240 tiles -> one launch whose last round is run as whole tiles."""
    frames = 16000
    g = torch.Generator().manual_seed(7)
    shapes = []
    for blk in range(6):
        shapes += [(512, 2048, 0, 2048, blk), (2048, 608, 0, 512, blk), (2048, 608, 512, 96, blk)]
    shapes += [(512, 512, 0, 512, 100 + i) for i in range(4)]
    shapes += [(1024, 1024, 0, 1024, 200), (1024, 1024, 0, 1024, 201), (512, 1024, 0, 1024, 202),
               (200, 1024, 0, 1024, 203), (1024, 64, 0, 64, 204), (32, 192, 0, 192, 205), (80, 512, 0, 512, 206),
               (8, 1024, 0, 1024, 207), (512, 80, 0, 80, 208)]
    items, weights, refs = _make(shapes, frames, g)
    launches = []
    real = F._launch_wgrad_items
    monkeypatch.setattr(F, "_launch_wgrad_items", lambda its: (launches.append(len(its)), real(its))[1])
    F.defer_vec_grads(True)
    try:
        i = 0
        for blk in range(6):
            F.sink_wgrad_group(items[i:i + 3], fire=False)
            i += 3
        for it in items[i:]:
            F.sink_wgrad(it[0], it[1], it[2])
    finally:
        F.defer_vec_grads(False)
    assert launches == [len(items)], launches
    for key, w in weights.items():
        assert torch.equal(w.grad.double(), refs[key]), key


def test_a_weight_that_comes_back_flushes_the_queue_first(F, monkeypatch):
    """Two products into the same gradient must not share a launch (plain read-modify-write of whole tiles)."""
    frames = 2048
    g = torch.Generator().manual_seed(11)
    shapes = [(512, 2048, 0, 2048, 0), (2048, 512, 0, 512, 1), (512, 2048, 0, 2048, 0), (2048, 512, 0, 512, 1)]
    items, weights, refs = _make(shapes, frames, g)
    launches = []
    real = F._launch_wgrad_items
    monkeypatch.setattr(F, "_launch_wgrad_items", lambda its: (launches.append(len(its)), real(its))[1])
    F.defer_vec_grads(True)
    try:
        F.sink_wgrad_group(items[:2])
        F.sink_wgrad_group(items[2:])
        assert launches == [2]
    finally:
        F.defer_vec_grads(False)
    assert launches == [2, 2]
    for key, w in weights.items():
        assert torch.equal(w.grad.double(), refs[key]), key


def test_mixed_reduction_lengths_leave_in_separate_launches(F, monkeypatch):
    g = torch.Generator().manual_seed(13)
    a, wa, ra = _make([(512, 2048, 0, 2048, 0), (2048, 512, 0, 512, 1)], 4096, g)
    b, wb, rb = _make([(512, 2048, 0, 2048, 2), (2048, 512, 0, 512, 3)], 2048, g)
    launches = []
    real = F._launch_wgrad_items
    monkeypatch.setattr(F, "_launch_wgrad_items", lambda its: (launches.append(len(its)), real(its))[1])
    F.defer_vec_grads(True)
    try:
        F.sink_wgrad_group(a)
        F.sink_wgrad_group(b)
    finally:
        F.defer_vec_grads(False)
    assert sorted(launches) == [2, 2]
    for weights, refs in ((wa, ra), (wb, rb)):
        for key, w in weights.items():
            assert torch.equal(w.grad.double(), refs[key]), key
