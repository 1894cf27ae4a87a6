"""The N > 1 host logic on CPU with the gloo backend, world_size 2: tile planning / stitching
of sharded inference and the data-parallel gradient all-reduce.  The models here are stock
torch stand-ins (the kernels need a GPU); what is tested is the sharding, not the kernels."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visiontransformer_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _ToySR(torch.nn.Module):
    """x2 SR stand-in with a 5x5 receptive field (two 3x3 convs)"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Conv2d(3, 8, 3, 1, 1)
        self.b = torch.nn.Conv2d(8, 12, 3, 1, 1)
        self.up = torch.nn.PixelShuffle(2)

    def forward(self, x):
        return self.up(self.b(torch.relu(self.a(x))))


def test_plan_tiles_cover_the_image_once():
    H, W = 70, 45
    cover = torch.zeros(H, W, dtype=torch.int32)
    tiles = parallel.plan_tiles(H, W, tile=32, halo=4)
    for t in tiles:
        cover[t.y0:t.y1, t.x0:t.x1] += 1
        assert t.py0 == t.y0 - 4 and t.px1 == t.x1 + 4
    assert int(cover.min()) == 1 and int(cover.max()) == 1
    assert len(tiles) == 3 * 2


def test_plan_tiles_rectangular_clipped():
    H, W = 70, 45
    cover = torch.zeros(H, W, dtype=torch.int32)
    tiles = parallel.plan_tiles(H, W, tile=(48, 16), halo=4, clip=True)
    assert len(tiles) == 2 * 3
    for t in tiles:
        cover[t.y0:t.y1, t.x0:t.x1] += 1
        assert t.py0 == max(t.y0 - 4, 0) and t.py1 == min(t.y1 + 4, H)
        assert t.px0 == max(t.x0 - 4, 0) and t.px1 == min(t.x1 + 4, W)
    assert int(cover.min()) == 1 and int(cover.max()) == 1


def test_balanced_grid():
    # one rank, image within the limit: a single tile = whole-image inference
    assert parallel.balanced_grid(540, 960, 1) == (544, 960)
    assert len(parallel.plan_tiles(540, 960, (544, 960), 32, clip=True)) == 1
    for world in (2, 4, 8):
        th, tw = parallel.balanced_grid(540, 960, world)
        assert th % 16 == 0 and tw % 16 == 0
        tiles = parallel.plan_tiles(540, 960, (th, tw), 32, clip=True)
        assert len(tiles) % world == 0
        assert max(max(t.py1 - t.py0, t.px1 - t.px0) for t in tiles) <= 1024
    # the side limit forces more tiles
    th, tw = parallel.balanced_grid(540, 960, 1, max_side=320)
    assert max(th, tw) + 64 <= 320 + 64 and len(parallel.plan_tiles(540, 960, (th, tw), 32, clip=True)) >= 8
    with pytest.raises(ValueError):
        parallel.balanced_grid(64, 64, 3, max_side=16, halo=32)


def test_balanced_tiling_matches_whole_image_everywhere():
    """clipped halo: border tiles see the image border as whole-image inference does, so with
    halo >= receptive-field radius the tiled result is exact everywhere"""
    model = _ToySR().eval()
    lr = torch.randn(1, 3, 40, 52)
    full = model(lr)
    one = parallel.tiled_super_resolve(model, lr, 2, tile="balanced", halo=4)
    assert torch.allclose(one, full, atol=1e-6)
    parts = sum(parallel.tiled_super_resolve(model, lr, 2, tile="balanced", halo=4, rank=r, world=4, reduce=False)
                for r in range(4))
    assert torch.allclose(parts, full, atol=1e-6)
    tiled = parallel.tiled_super_resolve(model, lr, 2, tile=(16, 32), halo=4, clip=True)
    assert torch.allclose(tiled, full, atol=1e-6)


def test_reflect_crop_matches_padding():
    img = torch.arange(2 * 3 * 6 * 7, dtype=torch.float32).view(2, 3, 6, 7)
    ref = torch.nn.functional.pad(img, (3, 3, 3, 3), mode="reflect")
    got = parallel._reflect_crop(img, -3, 9, -3, 10)
    assert torch.equal(got, ref)


def test_tiled_inference_single_process_exact_in_the_interior():
    """halo >= receptive-field radius: tiling reproduces whole-image inference away from the
    image border (at the border the reflect padding differs from the conv's zero padding)"""
    model = _ToySR().eval()
    lr = torch.randn(1, 3, 40, 52)
    full = model(lr)
    tiled = parallel.tiled_super_resolve(model, lr, 2, tile=16, halo=4)
    assert tiled.shape == full.shape
    assert torch.allclose(tiled[..., 8:-8, 8:-8], full[..., 8:-8, 8:-8], atol=1e-6)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    r, w, dev = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    # (1) sharded tiled inference: every rank ends with the whole image
    model = _ToySR().eval()
    gen = torch.Generator().manual_seed(5)
    lr = torch.randn(1, 3, 40, 52, generator=gen)
    out = parallel.tiled_super_resolve(model, lr, 2, tile=16, halo=4, rank=r, world=w)
    ref = parallel.tiled_super_resolve(model, lr, 2, tile=16, halo=4)
    ok_tiles = torch.allclose(out, ref, atol=1e-6)
    # the balanced plan for this world size, halo clipped at the border: equal to whole-image inference
    bal = parallel.tiled_super_resolve(model, lr, 2, tile="balanced", halo=4, rank=r, world=w)
    ok_tiles = ok_tiles and torch.allclose(bal, model(lr), atol=1e-6)
    # (2) data parallel: per-rank batches, gradients averaged over the group
    net = parallel.data_parallel(_ToySR(), dev)
    gen2 = torch.Generator().manual_seed(100 + r)
    x = torch.randn(2, 3, 8, 8, generator=gen2)
    net(x).square().mean().backward()
    g_ddp = net.module.a.weight.grad.clone()
    single = _ToySR()
    xs = torch.cat([torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(100 + i)) for i in range(w)])
    single(xs).square().mean().backward()
    ok_grad = torch.allclose(g_ddp, single.a.weight.grad, atol=1e-6)
    # (3) the flat-buffer wrapper: same averaged gradients for every parameter, over three steps - .grad dropped
    # between steps, kept (accumulating into the buffer's views), and a step inside no_sync followed by a synced one
    start = _ToySR()
    if r != 0:                   # a different start on the other ranks: the constructor must bring rank 0's values
        with torch.no_grad():
            for prm in start.parameters():
                prm.add_(1.0)
    flat = parallel.data_parallel(start, dev, flat=True)
    twin = _ToySR()
    ok_flat = isinstance(flat, parallel.FlatDataParallel)
    ok_flat = ok_flat and all(torch.equal(a, b) for a, b in zip(flat.module.state_dict().values(), twin.state_dict().values()))

    def grads(m):
        return [p.grad.clone() for p in m.parameters()]

    def same(m):
        return all(torch.allclose(a, b, atol=1e-6) for a, b in zip(grads(m), grads(twin)))
    flat(x).square().mean().backward()
    twin(xs).square().mean().backward()
    ok_flat = ok_flat and same(flat.module)
    flat.zero_grad(set_to_none=True)
    twin.zero_grad(set_to_none=True)
    flat(x).square().mean().backward()
    twin(xs).square().mean().backward()
    ok_flat = ok_flat and same(flat.module)
    # gradients kept: the next pass accumulates into the views; the reduced sum is avg(g1 + local g2 ...) - compare
    # the no_sync + sync pair instead, whose result is defined: avg over ranks of (g + g) = 2 * avg g
    flat.zero_grad(set_to_none=True)
    twin.zero_grad(set_to_none=True)
    with flat.no_sync():
        flat(x).square().mean().backward()
    flat(x).square().mean().backward()
    twin(xs).square().mean().backward()
    twin(xs).square().mean().backward()
    ok_flat = ok_flat and same(flat.module)
    # a parameter no pass reaches (an early return) contributes zeros and does not hold the collective up
    class _Spare(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = _ToySR()
            self.spare = torch.nn.Parameter(torch.ones(5))

        def forward(self, t):
            return self.net(t)
    sp = parallel.FlatDataParallel(_Spare())
    sp(x).square().mean().backward()
    ok_flat = ok_flat and torch.equal(sp.module.spare.grad, torch.zeros(5)) \
        and torch.allclose(sp.module.net.a.weight.grad, single.a.weight.grad, atol=1e-6)
    # a backward pass that raises after the output hook has fired (every rank alike, e.g. an out-of-memory error the
    # trainer catches and skips) never runs its end-of-pass callback; the next pass must still average its gradients
    class _Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    class _Fragile(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = _ToySR()
            self.fail = False

        def forward(self, t):
            return self.net(_Boom.apply(t) if self.fail else t)
    fr = parallel.FlatDataParallel(_Fragile())
    fr.module.fail = True
    raised = False
    try:
        fr(x.clone().requires_grad_()).square().mean().backward()
    except RuntimeError:
        raised = True
    fr.module.fail = False
    fr.zero_grad(set_to_none=True)
    fr(x).square().mean().backward()
    ok_flat = ok_flat and raised and torch.allclose(fr.module.net.a.weight.grad, single.a.weight.grad, atol=1e-6)
    q.put((rank, ok_tiles, ok_grad and ok_flat))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world_size_two_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True)]
