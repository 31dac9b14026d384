"""world_size-2 gloo test of the data-parallel step's host logic (runs on CPU, no GPU kernels).

What is checked: the flat gradient buffer + all-reduce + scale-by-1/world gives every rank the mean of the per-rank
gradients, and identical parameters after the update -- the only exchange step of the batch-sharded path
(SURVEY.md section 8e).  The GPU kernels themselves are covered by tests/test_gpu_ops.py.
"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jova_amd.train import FlatParameters
    torch.manual_seed(1)                                        # same initial weights on every rank
    net = nn.Sequential(nn.Linear(6, 5), nn.BatchNorm1d(5), nn.ReLU(), nn.Linear(5, 1))
    flat = FlatParameters(net)
    assert all(p.data_ptr() >= flat.data.data_ptr() for p in net.parameters())
    g = torch.Generator().manual_seed(100 + rank)               # different shard per rank
    x, y = torch.randn(16, 6, generator=g), torch.randn(16, 1, generator=g)
    flat.zero_grad()
    loss = ((net(x) - y) ** 2).mean()
    loss.backward()                                             # autograd accumulates into the flat views
    local = flat.grad.clone()
    dist.all_reduce(flat.grad)
    lr, wd = 0.1, 0.01
    flat.data -= lr * (flat.grad / world + wd * flat.data)      # what jova_sgd_step does with grad_scale = 1/world
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    mean = sum(gathered) / world
    ok_mean = torch.allclose(flat.grad / world, mean, atol=1e-7)
    params = [torch.zeros_like(flat.data) for _ in range(world)]
    dist.all_gather(params, flat.data)
    ok_same = all(torch.equal(params[0], p) for p in params)
    ok_views = all(torch.equal(p.grad.reshape(-1), flat.grad[o:o + p.numel()])
                   for p, o in zip(flat.params, _offsets(flat)))
    out[rank] = bool(ok_mean and ok_same and ok_views and not torch.equal(gathered[0], gathered[1]))
    dist.destroy_process_group()


def _offsets(flat):
    offs, total = [], 0
    for p in flat.params:
        offs.append(total)
        total += (p.numel() + 63) // 64 * 64
    return offs


def test_flat_gradient_allreduce_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def _worker_overlap(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jova_amd.train import FlatParameters, OverlappedGradSync, encoder_param_split

    class Views(nn.Module):                                     # stands for NwayForward: returns a list of tensors
        def __init__(self):
            super(Views, self).__init__()
            self.a, self.b = nn.Linear(6, 4), nn.Linear(3, 4)

        def forward(self, xs):
            return [self.a(xs[0]), self.b(xs[1]), xs[2]]        # the third view has no parameters (like ECFP)

    class Fuse(nn.Module):
        def __init__(self):
            super(Fuse, self).__init__()
            self.l = nn.Linear(12, 5)

        def forward(self, xs):
            return self.l(torch.cat(xs, dim=1))

    torch.manual_seed(1)
    net = nn.Sequential(Views(), Fuse(), nn.ReLU(), nn.Linear(5, 1))
    flat = FlatParameters(net)
    split = encoder_param_split(net, flat)
    n_enc = sum((p.numel() + 63) // 64 * 64 for p in net[0].parameters())
    g = torch.Generator().manual_seed(100 + rank)
    xs = [torch.randn(16, 6, generator=g), torch.randn(16, 3, generator=g), torch.randn(16, 4, generator=g)]
    y = torch.randn(16, 1, generator=g)

    def backward(sync):
        flat.zero_grad()
        kids = list(net.children())
        h = kids[0](xs)
        if sync is not None:
            sync.watch(h)
        for m in kids[1:]:
            h = m(h)
        ((h - y) ** 2).mean().backward()

    backward(None)
    dist.all_reduce(flat.grad)
    want = flat.grad.clone()
    sync = OverlappedGradSync(flat, split)
    fired = []
    for _ in range(2):                                           # twice: the watch re-arms every step
        backward(sync)
        fired.append(sync.work is not None)
        sync.finish()
    out[rank] = bool(split == n_enc and all(fired) and torch.allclose(flat.grad, want, atol=1e-7) and sync.work is None)
    dist.destroy_process_group()


def test_overlapped_gradient_allreduce_world2():
    """The two-piece all-reduce (late layers' slice started from a hook during backward) gives the same sums as one
    all-reduce of the flat buffer."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_overlap, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
