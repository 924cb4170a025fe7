"""Optimizer-in-the-loop trajectory parity: the reference's unit of work is run_step + optimizer.step()
(vidgen/engine/trainer.py:79-87, solver/build.py:46-74).  Eight train steps of PR-DVQVAE2 (B = 4 frames, Adam beta = (0.9, 0.9),
config/defaults.py:113-114) and of DSFVT (b = 2, RMSprop alpha 0.95, momentum 0.9, configs/vt/DSFVT.yaml:28-32) in the
default arithmetic (f16x2: the max |.| records of every weight are invalidated and re-derived after every optimizer step)
against the CPU oracle stepped by torch.optim with the same hyperparameters on the same seeds:
  * the loss of every step within 1e-4 (relative) of an fp64 oracle trajectory;
  * the weights after the last step as close to the fp64 trajectory as the CPU fp32 oracle's trajectory is (x 4; floor 1e-5 of
    the tensor's max), per tensor in the l2 norm -- an Adam / RMSprop update is ~lr * sign(g) wherever |g| is at the rounding
    level, so single entries of ANY two fp32 trajectories differ by up to 2 lr there and a max-norm bound would test luck.
    The DSFVT oracle runs (fp32 and fp64) take their ReLU decisions from the device (binding.RELU_TRACE, tests/util_relu.py):
    among the 4 M units of a forward pass one or two sit within round-off of zero, and a unit resolved differently moves whole
    gradient tensors by 1e-4 .. 1e-3 (tools/ubench/grad_accuracy_dsfvt.py: 300 x the CPU fp32 oracle's distance from fp64 at step 0,
    with the plane and the flash attention kernels alike) -- a fork of the trajectories that says nothing about the arithmetic;
  * code indices: no flip against the oracle's own search on rows with a clear margin, at every step (the oracle trajectory is
    run with the device's indices forced, so that a sub-margin row cannot fork the two trajectories)."""
import pytest
import torch

import seeded
from oracle import lvt_oracle as O
from util_models import MEAN, STD, dsfvt_cfg, margin_ok, vqvae_seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
STEPS = 8


def _l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def test_vqvae_train_trajectory_vs_oracle():
    from lvt_amd.hip import binding as L
    from lvt_amd.utils.events import EventStorage
    assert L.get_math_mode() == "f16x2"
    seed = 31
    model, enc, dec, st = vqvae_seeded(seed, scale=0.05)
    model.train()
    opts, _ = model.configure_optimizers_and_checkpointers()
    s = model.cfg.SOLVER
    assert s.OPTIMIZER_NAME == "adam" and (s.ADAM.BETA1_G, s.ADAM.BETA2_G) == (0.9, 0.9)
    xs = [seeded.seeded_input("traj.%d" % i, (4, 3, 64, 64), seed) for i in range(STEPS)]

    def oracle_side(dtype):
        pe = {k: v.clone().to(dtype).requires_grad_(True) for k, v in enc.items()}
        pd = {k: v.clone().to(dtype).requires_grad_(True) for k, v in dec.items()}
        state = {k: v.clone().to(dtype) for k, v in st.items()}
        opt = [torch.optim.Adam([{"params": [v], "lr": s.LR_G, "weight_decay": 0.0} for v in d.values()], s.LR_G,
                                betas=(s.ADAM.BETA1_G, s.ADAM.BETA2_G)) for d in (pe, pd)]
        return pe, pd, state, opt

    sides = {dt: oracle_side(dt) for dt in (torch.float32, torch.float64)}
    flips_clear = 0
    for i in range(STEPS):
        cpu32 = {}
        with EventStorage(i):
            losses = model([{"image": xs[i][j].numpy()} for j in range(4)], mode="supervised")
        sum(losses.values()).backward()
        idx = model.codebook.last_indices.cpu()                                    # (N, 4, 16, 16)
        force, flat = idx, idx.transpose(0, 1).reshape(4, -1)      # the oracle's force_idx is (N, num, H, W); its aux idx (num, N H W)
        for dt, (pe, pd, state, opt) in sides.items():
            x = O.normalize(xs[i], MEAN, STD).to(dt)
            # the oracle's own search first (the flip count), then the step with the device's indices forced
            if dt == torch.float64:
                with torch.no_grad():
                    _, _, aux = O.vqvae_supervised_loss(pe, pd, state, x)
                mine = aux["idx"].view(4, -1)
                ze = aux["z_e"].permute(0, 2, 3, 1).reshape(-1, 256).float()
                for g in range(4):
                    ok = margin_ok(ze[:, 64 * g:64 * g + 64], state["ve.%d.embedding.weight" % g].float())
                    flips_clear += int((mine[g][ok] != flat[g][ok]).sum())
            ref, new_state, _ = O.vqvae_supervised_loss(pe, pd, state, x, force_idx=force)
            for o in opt:
                o.zero_grad()
            sum(ref.values()).backward()
            for o in opt:
                o.step()
            sides[dt] = (pe, pd, {k: v.detach() for k, v in new_state.items()}, opt)
            if dt == torch.float32:
                cpu32 = {k: float(v.detach()) for k, v in ref.items()}
            if dt == torch.float64:
                for k in ("loss_reconstruction", "loss_commitment"):
                    a, b = float(losses[k].detach()), float(ref[k].detach())
                    # (the commitment loss is a mean of squared differences of nearly equal tensors, ~1e-4 after a few steps: its
                    # fp32 evaluations scatter by more than 1e-4 of it, so the CPU fp32 trajectory's own distance is the yardstick)
                    assert abs(a - b) < max(1e-4 * abs(b), 4 * abs(cpu32[k] - b)), (i, k, a, b, cpu32[k])
        for o in opts:
            o["optimizer"].step()
        for o in opts:
            o["optimizer"].zero_grad()
    assert flips_clear == 0
    mine = {**{"enc." + k: v for k, v in model.encoder.state_dict().items()},
            **{"dec." + k: v for k, v in model.generator.state_dict().items()}}
    r32 = {**{"enc." + k: v for k, v in sides[torch.float32][0].items()}, **{"dec." + k: v for k, v in sides[torch.float32][1].items()}}
    r64 = {**{"enc." + k: v for k, v in sides[torch.float64][0].items()}, **{"dec." + k: v for k, v in sides[torch.float64][1].items()}}
    for k in r64:
        e_mine, e_cpu = _l2(mine[k].cpu(), r64[k].detach()), _l2(r32[k].detach(), r64[k].detach())
        assert e_mine < max(4 * e_cpu, 1e-5), (k, e_mine, e_cpu)
    cb, cb64, cb32 = model.codebook.state_dict(), sides[torch.float64][2], sides[torch.float32][2]
    for k in cb64:
        assert _l2(cb[k].cpu(), cb64[k]) < max(4 * _l2(cb32[k], cb64[k]), 1e-4), k


def test_dsfvt_train_trajectory_vs_oracle():
    from lvt_amd.hip import binding as L
    from lvt_amd.modeling import build_model
    from lvt_amd.utils.events import EventStorage
    assert L.get_math_mode() == "f16x2"
    seed = 47
    model = build_model(dsfvt_cfg())
    params = seeded.seeded_params(seeded.dsfvt_shapes(), seed)
    missing, unexpected = model.model.load_state_dict(params, strict=False)
    assert not unexpected
    model.train()
    opts, _ = model.configure_optimizers_and_checkpointers()
    s = model.cfg.SOLVER
    assert s.OPTIMIZER_NAME == "rmsprop" and (s.RMSPROP.ALPHA_G, s.RMSPROP.MOMENTUM_G) == (0.95, 0.9)
    ds = dict(blocks_e=((1, 16, 16),) * 8, blocks_d=((1, 16, 16),) * 8, stride=(16, 1, 1))
    batches = []
    for i in range(STEPS):
        data = [O.prepare_slices(seeded.seeded_codes("traj.codes%d.%d" % (i, j), (16, 4, 16, 16), seed), (a, 0, 0), (16, 1, 1),
                                 (7, 1, 1), 1) for j, a in enumerate(((3 * i + 1) % 15 + 1, (5 * i + 7) % 15 + 1))]
        batches.append(data)

    def oracle_side(dtype):
        p = {k: v.clone().to(dtype).requires_grad_(True) for k, v in params.items()}
        opt = torch.optim.RMSprop([{"params": [v], "lr": s.LR_G, "weight_decay": 0.0} for v in p.values()], s.LR_G,
                                  alpha=s.RMSPROP.ALPHA_G, momentum=s.RMSPROP.MOMENTUM_G)
        return p, opt

    from util_relu import ReluFollow, relu_probe
    sides = {dt: oracle_side(dt) for dt in (torch.float32, torch.float64)}
    forks = 0
    for i, data in enumerate(batches):
        trace = L.RELU_TRACE = []
        try:
            with EventStorage(i):
                loss = model(data, mode="supervised")["loss_cross_entropy"]
        finally:
            L.RELU_TRACE = None
        loss.backward()
        ctx = torch.stack([d["context"] for d in data]); sl = torch.stack([d["slice"] for d in data])
        si = torch.stack([d["slice_idx"] for d in data]); ig = torch.stack([d["ignore_mask"] for d in data])
        for dt, (p, opt) in sides.items():
            opt.zero_grad()
            with torch.no_grad():       # MaskedConv3d.forward zeroes the causal taps in weight.data (vt_utils.py:197-199): the
                w = p["decoder.conv.conv.weight"]       # optimizer's update of those taps lives until the next forward only
                w[:, :, -1, -1, w.shape[-1] // 2:] = 0
            with relu_probe(ReluFollow(trace)) as follow:
                lo, _ = O.vt_supervised_loss(p, ctx, sl, si, ig, **ds)
            assert follow.calls == len(trace)
            lo.backward()
            opt.step()
            if dt == torch.float64:
                forks += follow.differ
                assert follow.differ <= 16, follow.differ      # units on their threshold, not a different network
                a, b = float(loss.detach()), float(lo.detach())
                assert abs(a - b) < 1e-4 * abs(b), (i, a, b)
        for o in opts:
            o["optimizer"].step()
        for o in opts:
            o["optimizer"].zero_grad()
    named = dict(model.model.named_parameters())
    worst = []
    for k, r64 in sides[torch.float64][0].items():
        e_mine, e_cpu = _l2(named[k].detach().cpu(), r64.detach()), _l2(sides[torch.float32][0][k].detach(), r64.detach())
        worst.append((e_mine / max(4 * e_cpu, 1e-5), k, e_mine, e_cpu))
    worst.sort(reverse=True)
    assert worst[0][0] < 1.0, worst[:5]
