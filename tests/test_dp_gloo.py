"""world_size-2 data-parallel logic on CPU (gloo): bucketed gradient averaging, summed EMA statistics,
the differentiable AllReduce, parameter broadcast and the rank-strided sampler."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return dict(ret)


def _reducer_case(rank, world):
    from lvt_amd.engine.grad_reducer import BucketedGradReducer
    torch.manual_seed(100 + rank)                       # different initial weights per rank (SEED + rank)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 2))
    red = BucketedGradReducer(net.parameters(), bucket_bytes=64)      # tiny buckets -> several of them
    assert len(red.buckets) >= 3
    w0 = net[0].weight.detach().clone()
    outs = []
    for step in range(2):                               # two steps: buckets must re-arm
        x = torch.full((3, 8), float(rank + 1 + step))
        net.zero_grad()
        # this rank's own gradient, taken without touching .grad (whose buckets are reduced while backward runs)
        local = [g.clone() for g in torch.autograd.grad(net(x).sum(), list(net.parameters()))]
        net(x).sum().backward()
        red.wait()
        outs.append(([p.grad.clone() for p in net.parameters()], local))
    return w0, outs


def test_bucketed_reducer_broadcasts_and_averages():
    res = _run(_reducer_case)
    w0a, outa = res[0]
    w0b, outb = res[1]
    assert torch.equal(w0a, w0b)                        # rank 0's weights everywhere
    for step in range(2):
        (avg_a, loc_a), (avg_b, loc_b) = outa[step], outb[step]
        for ga, gb, la, lb in zip(avg_a, avg_b, loc_a, loc_b):
            assert torch.equal(ga, gb)
            assert torch.allclose(ga, (la + lb) / 2, rtol=1e-6, atol=1e-7)


def _reference_loop_case(rank, world):
    """The reference's own loop (vidgen/engine/trainer.py:79-87): forward, backward, every ACCUMULATION_STEPS
    `optimizer.step()` then `optimizer.zero_grad()` -- no join call.  The reducer must behave like torch DDP
    there: every backward averages, the step sees averaged gradients."""
    from lvt_amd.engine.grad_reducer import BucketedGradReducer
    out = {}
    for acc, set_to_none in ((1, True), (2, True), (2, False), (3, True)):
        torch.manual_seed(5)
        net = torch.nn.Sequential(torch.nn.Linear(6, 10), torch.nn.Tanh(), torch.nn.Linear(10, 3))
        red = BucketedGradReducer(net.parameters(), bucket_bytes=128)
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        seen = []
        for it in range(2 * acc):
            g = torch.Generator().manual_seed(1000 * it + rank)
            x = torch.randn(4, 6, generator=g)
            if it > 0:
                red.wait()              # what the meta-architecture's forward does (finish_gradient_sync)
            net(x).pow(2).sum().backward()
            if (it + 1) % acc == 0:
                opt.step()              # pre-step hook joins the all-reduce
                seen.append([p.grad.clone() for p in net.parameters()])
                opt.zero_grad(set_to_none=set_to_none)
        out[(acc, set_to_none)] = ([p.detach().clone() for p in net.parameters()], seen)
        red.remove()
    return out


def test_reference_loop_with_accumulation_matches_big_batch():
    res = _run(_reference_loop_case)
    for key in res[0]:
        acc, _ = key
        wa, ga = res[0][key]
        wb, gb = res[1][key]
        for a, b in zip(wa, wb):
            assert torch.equal(a, b), key                          # replicas stay identical
        # one process, both ranks' data: gradient = mean over ranks of the per-rank accumulated gradient
        torch.manual_seed(5)
        net = torch.nn.Sequential(torch.nn.Linear(6, 10), torch.nn.Tanh(), torch.nn.Linear(10, 3))
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        k = 0
        for it in range(2 * acc):
            for rank in range(2):
                x = torch.randn(4, 6, generator=torch.Generator().manual_seed(1000 * it + rank))
                (net(x).pow(2).sum() / 2).backward()
            if (it + 1) % acc == 0:
                for g1, g2 in zip(ga[k], [p.grad for p in net.parameters()]):
                    assert torch.allclose(g1, g2, rtol=1e-5, atol=1e-6), key
                k += 1
                opt.step()
                opt.zero_grad()
        for a, p in zip(wa, net.parameters()):
            assert torch.allclose(a, p.detach(), rtol=1e-5, atol=1e-6), key


def _ema_case(rank, world):
    from lvt_amd.layers import AllReduce, all_reduce_sum_
    stats = torch.arange(4 * 8 * 5, dtype=torch.float32).view(4, 8, 5) * (rank + 1)
    all_reduce_sum_(stats)
    x = torch.full((3,), float(rank + 1), requires_grad=True)
    y = AllReduce.apply(x)
    (y * torch.tensor([1.0, 2.0, 3.0])).sum().backward()
    return stats, y.detach(), x.grad


def test_ema_statistics_and_allreduce_fn():
    res = _run(_ema_case)
    base = torch.arange(4 * 8 * 5, dtype=torch.float32).view(4, 8, 5)
    for r in (0, 1):
        stats, y, g = res[r]
        assert torch.equal(stats, base * 3)             # 1x + 2x
        assert torch.equal(y, torch.full((3,), 3.0))
        assert torch.equal(g, torch.tensor([2.0, 4.0, 6.0]))    # backward all-reduces the gradient


def test_training_sampler_shards_are_disjoint_and_cover():
    from lvt_amd.data.samplers import TrainingSampler
    import itertools
    a = list(itertools.islice(iter(TrainingSampler(10, seed=7, rank=0, world_size=2)), 10))
    b = list(itertools.islice(iter(TrainingSampler(10, seed=7, rank=1, world_size=2)), 10))
    full = list(itertools.islice(iter(TrainingSampler(10, seed=7, rank=0, world_size=1)), 20))
    assert a == full[0::2] and b == full[1::2]
    assert sorted(full[:10]) == list(range(10))


def _second_backward_case(rank, world):
    from lvt_amd.engine.grad_reducer import BucketedGradReducer
    torch.manual_seed(3)
    net = torch.nn.Linear(5, 4)
    red = BucketedGradReducer(net.parameters(), bucket_bytes=1 << 20)
    x = torch.randn(2, 5, generator=torch.Generator().manual_seed(rank))
    net(x).sum().backward()                      # launches the all-reduce of the single bucket
    try:
        net(x).sum().backward()                  # no join in between
        raised = False
    except RuntimeError as e:
        raised = "still in flight" in str(e)
    red.wait()
    red.remove()
    return raised


def test_second_backward_without_join_raises():
    """a gradient landing in a bucket whose all-reduce is in flight must fail loudly, not be joined silently."""
    res = _run(_second_backward_case)
    assert res[0] is True and res[1] is True
