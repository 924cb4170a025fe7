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
    net(x).sum().backward()                      # the first backward only measures the arrival order (launched by the join)
    red.wait()
    net.zero_grad()
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


def _unused_on_one_rank_case(rank, world):
    """Rank 1 never uses `side`: its gradient there is None.  Every rank must still issue the same collectives in the same order,
    rank 1 contributes zeros, and BOTH ranks end up with the same mean (VERDICT r5 item 12 / ADVICE r5: rounds 2-5 reduced such a
    bucket "as it is" -- stale data in the slot, a different launch order on the two ranks)."""
    from lvt_amd.engine.grad_reducer import BucketedGradReducer
    torch.manual_seed(5)
    trunk = torch.nn.Sequential(torch.nn.Linear(6, 6), torch.nn.Linear(6, 3))
    side = torch.nn.Linear(6, 3)
    params = list(trunk.parameters()) + list(side.parameters())
    red = BucketedGradReducer(params, bucket_bytes=32)               # one bucket per parameter or two
    outs = []
    for step in range(4):                                            # steps 0-2 measure (rank 1 never completes), 3 runs ordered
        x = torch.full((2, 6), float(1 + rank + step))
        for p in params:
            p.grad = None
        y = trunk(x).sum()
        if rank == 0:
            y = y + side(x).sum() * 3.0
        y.backward()
        red.wait()
        outs.append([None if p.grad is None else p.grad.clone() for p in params])
    local_side = [torch.full((3, 6), 3.0 * 2 * float(1 + 3)), torch.full((3,), 3.0 * 2)]       # rank 0's own gradient at step 3
    pos = {id(q): i for i, q in enumerate(params)}
    red.remove()
    return outs, local_side, [[pos[id(q)] for q in b["params"]] for b in red.buckets]


def test_parameter_without_gradient_on_one_rank():
    res = _run(_unused_on_one_rank_case)
    (o0, side0, ord0), (o1, _, ord1) = res[0], res[1]
    assert ord0 == ord1                                              # the same buckets in the same order on both ranks
    for step in range(4):
        for a, b in zip(o0[step], o1[step]):
            assert a is not None and b is not None and torch.equal(a, b)          # replicas cannot drift
    # the side layer: (rank 0's gradient + 0) / 2
    assert torch.allclose(o0[3][4], side0[0] / 2) and torch.allclose(o0[3][5], side0[1] / 2)


def _arrival_order_case(rank, world):
    """The bucket order follows the measured arrival order: a layer registered FIRST but used LAST in the forward pass (its
    gradient arrives first) moves to the front, and later backwards launch from the hooks, in index order."""
    from lvt_amd.engine.grad_reducer import BucketedGradReducer
    torch.manual_seed(9)
    last_used = torch.nn.Linear(4, 4)        # registered first, applied last
    first_used = torch.nn.Linear(4, 4)
    params = list(last_used.parameters()) + list(first_used.parameters())
    red = BucketedGradReducer(params, bucket_bytes=16)
    pos = {id(q): i for i, q in enumerate(params)}
    before = [[pos[id(q)] for q in b["params"]] for b in red.buckets]
    launched = []
    orig = red._launch
    red._launch = lambda b: (launched.append(b["index"]), orig(b))[1]
    grads = []
    for step in range(3):
        for p in params:
            p.grad = None
        x = torch.full((2, 4), float(rank + 1))
        last_used(first_used(x)).sum().backward()
        n_hook = len(launched)
        red.wait()
        grads.append((n_hook, list(launched), [p.grad.clone() for p in params]))
        launched.clear()
    after = [[pos[id(q)] for q in b["params"]] for b in red.buckets]
    red.remove()
    return before, after, grads


def test_buckets_follow_the_arrival_order():
    res = _run(_arrival_order_case)
    before, after, grads = res[0]
    assert res[1][1] == after
    flat = [i for b in after for i in b]
    assert set(flat[:2]) == {0, 1} and set(flat[2:]) == {2, 3}       # last_used's parameters (indices 0, 1) now come first
    assert [i for b in before for i in b] == [3, 2, 1, 0]            # before the measurement: reverse registration
    n_hook0, launched0, _ = grads[0]
    assert n_hook0 == 0 and launched0 == sorted(launched0)           # the measuring backward launches from the join only
    n_hook1, launched1, _ = grads[1]
    assert n_hook1 == len(after) and launched1 == list(range(len(after)))        # afterwards: from the hooks, in index order
    for (_, _, g0), (_, _, g1) in zip(res[0][2], res[1][2]):
        for a, b in zip(g0, g1):
            assert torch.equal(a, b)
