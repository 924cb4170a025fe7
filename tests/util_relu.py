"""Gradient comparison that is exact about ReLU units sitting on their threshold.

A forward pass of the full-size models has millions of ReLU units; about one of them has a pre-activation within fp32
round-off (|x| ~ 1e-7 of the tensor's scale) of zero, and two correct fp32 evaluations may put it on different sides.  The
unit then passes or blocks its token's gradient entirely, which moves individual weight-gradient entries by ~1e-3 -- far
above the 2e-5 the kernels are held to.  Instead of widening the tolerance, `assert_grads_match` finds out WHICH units are
undecided (from an fp64 run of the oracle, which sees every pre-activation), determines on which side the path under test
put each of them, and then holds every gradient to the strict bound against an fp64 run that makes the same decisions:

    1. g64 = fp64 gradients, with the list of undecided units (|x| < delta * max |x| of that ReLU call; delta grows from
       2e-7 to 2e-6 only while the mismatch is not explained);
    2. if the strict bound already holds against g64 -> done (no unit flipped);
    3. else one fp64 run per undecided unit with that unit forced to the other side gives the direction d_u in which a flip
       moves the gradients; the least-squares coefficients of (grad_under_test - g64) on {d_u} are ~1 for the units that
       flipped and ~0 for the others (the same for the CPU fp32 oracle run, whose distance to fp64 scales the bound);
    4. the strict bound must hold against the fp64 run with exactly those units flipped.  No escape hatch.

The oracle is test infrastructure (oracle/lvt_oracle.py); it applies every ReLU through `torch.relu`, which is what the
probe replaces while a run is in progress.
"""
import contextlib

import os

import torch

_ORIG_RELU = torch.relu


class ReluProbe:
    """Stand-in for torch.relu: records the units whose input is within `delta` (relative to the call's max |x|) of zero, and
    forces the units listed in `force` ({(call index, position tuple): on?}) to pass (y = x) or block (y = 0)."""

    def __init__(self, delta=None, force=None, record=False):
        self.delta, self.force, self.calls, self.near = delta, dict(force or {}), 0, []
        self.masks = [] if record else None            # record: the sign pattern (x > 0) of every call

    def __call__(self, x):
        i = self.calls
        self.calls += 1
        y = _ORIG_RELU(x)
        if self.masks is not None:
            self.masks.append(x.detach() > 0)
        if self.delta is not None:
            xd = x.detach()
            top = float(xd.abs().max())
            for pos in (xd.abs() < self.delta * top).nonzero().tolist():
                self.near.append((i, tuple(pos), float(xd[tuple(pos)]), top))
        for (ci, pos), on in self.force.items():
            if ci == i:
                m = torch.zeros_like(x, dtype=torch.bool)
                m[pos] = True
                y = torch.where(m, x if on else torch.zeros_like(x), y)
        return y


class ReluFollow:
    """Stand-in for torch.relu that takes EVERY decision from recorded masks (`masks`: one bool tensor per ReLU call in the
    oracle's call order and element order -- lvt_amd.hip.binding.RELU_TRACE): y = x where the mask passes, 0 elsewhere.
    An oracle run under it makes exactly the decisions of the path under test (used step by step along a training trajectory,
    where a unit on its threshold would otherwise fork the two trajectories)."""

    def __init__(self, masks):
        self.masks, self.calls, self.differ = masks, 0, 0

    def __call__(self, x):
        m = self.masks[self.calls].detach().to(x.device).reshape(x.shape)
        self.calls += 1
        self.differ += int((m != (x.detach() > 0)).sum())
        return torch.where(m, x, torch.zeros_like(x))


@contextlib.contextmanager
def relu_probe(probe):
    torch.relu = probe
    try:
        yield probe
    finally:
        torch.relu = _ORIG_RELU


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def assert_grads_match(mine, run_oracle, names, deltas=(2e-7, 6e-7, 2e-6), slack=4.0, floor=2e-5, max_units=40):
    """mine: {name: gradient under test}; run_oracle(dtype) -> {name: gradient} (called under a ReluProbe).
    The undecided set is grown (`deltas`) until the strict bound holds; past the last delta the mismatch is a failure.
    Returns (number of undecided units considered, units the path under test resolved differently from fp64)."""
    with relu_probe(ReluProbe(deltas[-1])) as base:
        g64 = run_oracle(torch.float64)
    g32 = run_oracle(torch.float32)

    def violations(ref, ref_cpu):
        out = []
        for n in names:
            e_mine, e_cpu = _rel(mine[n], ref[n]), _rel(g32[n], ref_cpu[n])
            if not e_mine < max(slack * e_cpu, floor):
                out.append((n, e_mine, e_cpu))
        return out

    bad = violations(g64, g64)
    if not bad:
        return 0, []
    # direction of a single flip, over all compared tensors, each tensor scaled by its own norm
    scale = {n: float(g64[n].norm()) + 1e-300 for n in names}

    def flat(g, sub=None):
        return torch.cat([((g[n].detach().double().cpu() - (sub[n].detach().double().cpu() if sub is not None else 0)) / scale[n]).reshape(-1)
                          for n in names])

    def run_with(fl):
        if not fl:
            return g64
        with relu_probe(ReluProbe(force={(ci, pos): not (x > 0) for (ci, pos, x) in fl})):
            return run_oracle(torch.float64)

    dirs = {}
    for delta in deltas:
        # a call's max |x| is not kept by the probe: re-derive the threshold from the ratio of the deltas
        units = [u for u in base.near if abs(u[2]) < u[3] * delta]
        if not units:
            continue
        assert len(units) <= max_units, "%d undecided units at delta %g" % (len(units), delta)
        for u in units:
            if u[:2] not in dirs:
                dirs[u[:2]] = flat(run_with([u[:3]]), g64)
        D = torch.stack([dirs[u[:2]] for u in units], 1)                      # (entries, units)

        def flipped(g):
            c = torch.linalg.lstsq(D, flat(g, g64).unsqueeze(1)).solution.reshape(-1)
            return [u[:3] for u, cu in zip(units, c.tolist()) if cu > 0.5], c.tolist()

        f_mine, c_mine = flipped(mine)
        f_cpu, _ = flipped(g32)
        bad = violations(run_with(f_mine), run_with(f_cpu))
        if not bad:
            return len(units), f_mine
    raise AssertionError(("gradient mismatch that no undecided ReLU unit explains", bad, "undecided units", len(base.near)))


def assert_grads_match_decisions(mine, mine_masks, run_oracle, names, slack=4.0, floor=2e-5, max_flips=64):
    """The same bound when the path under test can REPORT its ReLU decisions (`mine_masks`: one bool tensor per ReLU call,
    in the oracle's call order, any shape with the oracle's element order -- lvt_amd.hip.binding.RELU_TRACE): the units it
    resolved differently from the fp64 run are read off directly, the fp64 run is repeated with exactly those decisions,
    and every gradient is held to max(slack x the CPU fp32 oracle's distance from ITS matching fp64 run, floor)."""
    with relu_probe(ReluProbe(record=True)) as p64:
        g64 = run_oracle(torch.float64)
    with relu_probe(ReluProbe(record=True)) as p32:
        g32 = run_oracle(torch.float32)
    assert len(mine_masks) == len(p64.masks) == len(p32.masks), (len(mine_masks), len(p64.masks), len(p32.masks))

    def differing(masks):
        force = {}
        for ci, (m, ref) in enumerate(zip(masks, p64.masks)):
            m = m.detach().cpu().reshape(ref.shape)
            for pos in (m != ref).nonzero().tolist():
                force[(ci, tuple(pos))] = bool(m[tuple(pos)])
        return force

    f_mine, f_cpu = differing(mine_masks), differing(p32.masks)
    assert len(f_mine) <= max_flips and len(f_cpu) <= max_flips, (len(f_mine), len(f_cpu))

    def run_with(force):
        if not force:
            return g64
        with relu_probe(ReluProbe(force=force)):
            return run_oracle(torch.float64)

    ref_mine, ref_cpu = run_with(f_mine), run_with(f_cpu)
    bad = []
    for n in names:
        e_mine, e_cpu = _rel(mine[n], ref_mine[n]), _rel(g32[n], ref_cpu[n])
        if os.environ.get("LVT_TEST_VERBOSE"):
            print("grad %-55s this path %.3e  CPU fp32 %.3e  (vs fp64)" % (n, e_mine, e_cpu))
        if not e_mine < max(slack * e_cpu, floor):
            bad.append((n, e_mine, e_cpu))
    assert not bad, (bad, "units decided differently from fp64: this path %d, CPU fp32 %d" % (len(f_mine), len(f_cpu)))
    return len(f_mine), len(f_cpu)
