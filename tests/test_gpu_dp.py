"""Model-level data parallelism on the GPU box: two ranks share cuda:0 and talk over gloo (a single-GPU box
cannot host two RCCL ranks), exercising exactly the code the 8-GPU run uses -- wrap_parallel (parameter +
codebook broadcast), bucketed gradient averaging on a side stream, the packed EMA-statistics all-reduce --
and comparing with one process that sees the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import seeded
from conftest import rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _vq_worker(rank, world, port, ret, backend="gloo"):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from util_models import vqvae_seeded
    from lvt_amd.utils.events import EventStorage
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = rank if backend == "nccl" else 0               # RCCL: one GPU per rank; gloo: both ranks share cuda:0
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        model, _, _, _ = vqvae_seeded(50 + rank, scale=0.05 * (1 + rank), device="cuda:%d" % dev)      # ranks start different
        model.train()
        model.wrap_parallel(device_ids=[dev], broadcast_buffers=False)
        x = seeded.seeded_input("dp", (8, 3, 64, 64), 9)[4 * rank:4 * rank + 4]
        with EventStorage(0):
            losses = model([{"image": x[i].numpy()} for i in range(4)], mode="supervised")
        sum(losses.values()).backward()
        model.finish_gradient_sync()
        torch.cuda.synchronize()
        ret[rank] = {
            "w0": model.encoder.layers[0].weight.detach().cpu(),
            "g_enc": model.encoder.layers[4].weight.grad.cpu(), "g_dec": model.generator.layers[6].weight.grad.cpu(),
            "g_b": model.encoder.layers[0].bias.grad.cpu(),
            "cb": {k: v.cpu() for k, v in model.codebook.state_dict().items()},
            "loss": {k: float(v.detach()) for k, v in losses.items()},
        }
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_vqvae_two_ranks_equal_one_big_batch(backend):
    """backend "nccl": two REAL RCCL ranks on two GPUs -- gradient buckets averaged with ReduceOp.AVG on the side stream, the
    EMA-statistics all-reduce started asynchronously behind the quantiser and joined after the decoder forward, parameter
    and codebook broadcast -- collected everywhere, skipped on boxes with a single GPU (the first multi-GPU box runs it)."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs for two RCCL ranks (this box has %d)" % torch.cuda.device_count())
    from util_models import vqvae_seeded
    from lvt_amd.utils.events import EventStorage
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_vq_worker, args=(r, 2, port, ret, backend)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    a, b = ret[0], ret[1]
    # replicas are identical after the step: same weights, same averaged grads, same codebook
    assert torch.equal(a["w0"], b["w0"])
    for k in ("g_enc", "g_dec", "g_b"):
        assert torch.equal(a[k], b[k]), k
    for k in a["cb"]:
        assert torch.equal(a["cb"][k], b["cb"][k]), k
    # and equal to a single process on the 8-frame batch starting from rank 0's weights
    model, _, _, _ = vqvae_seeded(50, scale=0.05)
    model.train()
    x = seeded.seeded_input("dp", (8, 3, 64, 64), 9)
    with EventStorage(0):
        losses = model([{"image": x[i].numpy()} for i in range(8)], mode="supervised")
    sum(losses.values()).backward()
    assert rel_err(a["g_enc"], model.encoder.layers[4].weight.grad) < 1e-4
    assert rel_err(a["g_dec"], model.generator.layers[6].weight.grad) < 1e-4
    assert rel_err(a["g_b"], model.encoder.layers[0].bias.grad) < 1e-4
    one = model.codebook.state_dict()
    for k in one:
        assert rel_err(a["cb"][k], one[k]) < 1e-5, k
    mean_loss = 0.5 * (a["loss"]["loss_reconstruction"] + b["loss"]["loss_reconstruction"])
    assert abs(mean_loss - float(losses["loss_reconstruction"].detach())) < 1e-5 * mean_loss


def _trained_codebook_model(seed, scale, dev):
    """PR-DVQVAE2 with CODEBOOK.EMA False (the codebooks are parameters), seeded weights."""
    from util_models import vqvae_cfg
    from lvt_amd.modeling import build_model
    cfg = vqvae_cfg(dev)
    cfg.MODEL.CODEBOOK.EMA = False
    model = build_model(cfg)
    model.encoder.load_state_dict(seeded.seeded_params(seeded.VQVAE_ENCODER_SHAPES, seed, "enc."))
    model.generator.load_state_dict(seeded.seeded_params(seeded.VQVAE_DECODER_SHAPES, seed, "dec."))
    model.codebook.load_state_dict({k: v for k, v in seeded.seeded_codebook_state(seed, scale=scale).items()
                                    if k.endswith("embedding.weight")})
    model.train()
    return model


def _vq_trained_codebook_worker(rank, world, port, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from lvt_amd.utils.events import EventStorage
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _trained_codebook_model(50 + rank, 0.05 * (1 + rank), "cuda:0")          # ranks start different
        model.wrap_parallel(device_ids=[0], broadcast_buffers=False)
        x = seeded.seeded_input("dp", (8, 3, 64, 64), 9)[4 * rank:4 * rank + 4]
        with EventStorage(0):
            losses = model([{"image": x[i].numpy()} for i in range(4)], mode="supervised")
        sum(losses.values()).backward()
        model.finish_gradient_sync()
        torch.cuda.synchronize()
        ret[rank] = {"cb_w": model.codebook.ve[1].embedding.weight.detach().cpu(),
                     "cb_g": [v.embedding.weight.grad.cpu() for v in model.codebook.ve],
                     "g_enc": model.encoder.layers[4].weight.grad.cpu(), "keys": sorted(losses)}
    finally:
        dist.destroy_process_group()


def test_vqvae_trained_codebook_two_ranks_equal_one_big_batch():
    """CODEBOOK.EMA False under data parallelism (the reference wraps the codebook in DDP, vqvae.py:46-49): the codebook is
    broadcast with the other parameters and its gradients are averaged by a reducer of its own."""
    from lvt_amd.utils.events import EventStorage
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_vq_trained_codebook_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    a, b = ret[0], ret[1]
    assert a["keys"] == ["loss_commitment", "loss_dict", "loss_reconstruction"]
    assert torch.equal(a["cb_w"], b["cb_w"]) and torch.equal(a["g_enc"], b["g_enc"])
    for i in range(4):
        assert torch.equal(a["cb_g"][i], b["cb_g"][i]), i
    model = _trained_codebook_model(50, 0.05, "cuda:0")
    x = seeded.seeded_input("dp", (8, 3, 64, 64), 9)
    with EventStorage(0):
        losses = model([{"image": x[i].numpy()} for i in range(8)], mode="supervised")
    sum(losses.values()).backward()
    for i in range(4):
        assert rel_err(a["cb_g"][i], model.codebook.ve[i].embedding.weight.grad) < 1e-5, i
    assert rel_err(a["g_enc"], model.encoder.layers[4].weight.grad) < 1e-4


# ---- BASELINE configs[3]: DSFVT on Kinetics codes (configs/vt/KDSFVT.yaml), data parallel -------------------------
def _vt_cfg(acc):
    from lvt_amd.config import get_cfg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(root, "configs/vt/KDSFVT.yaml"))
    cfg.MODEL.DEVICE = "cuda:0"
    cfg.OUTPUT_DIR = "/tmp/lvt_test_out_dp"
    cfg.SOLVER.ACCUMULATION_STEPS = acc
    cfg.SOLVER.MAX_ITER = 2 * acc
    cfg.SOLVER.CHECKPOINT_PERIOD = 10 ** 6
    return cfg


def _vt_batches(cfg, ranks, per_rank, iters):
    """Deterministic mapper output: micro-step `it` of rank r = clips seeded by (it, r) with forced slice offsets."""
    from lvt_amd.data.dataset_mapper import prepare_slices
    v = cfg.MODEL.AUTOREGRESSIVE.VT
    for it in range(iters):
        batch = []
        for r in ranks:
            g = torch.Generator().manual_seed(977 * it + 31 * r + 5)
            for j in range(per_rank):
                codes = torch.randint(0, v.NV, (16, v.NC, 16, 16), generator=g)
                a = int(torch.randint(v.N_PRIME, 16, (1,), generator=g))
                batch.append(prepare_slices(codes, (a, 0, 0), v.STRIDE, v.KERNEL, v.N_PRIME, v.PAD_VALUE))
        yield batch


def _vt_train(cfg, ranks, per_rank, seed):
    """The product Trainer -- the reference's loop shape (vidgen/engine/trainer.py:79-87): forward, backward,
    every ACCUMULATION_STEPS step + zero_grad; NO gradient-sync call anywhere."""
    from lvt_amd.engine.trainer import Trainer
    from lvt_amd.modeling import build_model
    torch.manual_seed(seed)
    model = build_model(cfg)
    tr = Trainer(cfg, model, _vt_batches(cfg, ranks, per_rank, cfg.SOLVER.MAX_ITER), log_period=10 ** 6)
    keys = ["encoder.conv.weight", "decoder.block_local_attention.7.mha.w_q", "decoder.block_local_attention.0.dh_bank",
            "encoder.block_local_attention.3.ffn.1.weight", "ch_predictor.U.2.weight", "ch_predictor.P.0.bias",
            "decoder.ch_embedder.1.weight"]
    named = dict(model.model.named_parameters())
    grads = []
    # what the optimizer is about to consume at its FIRST step: registered after the reducer's own pre-step hook
    # (Trainer -> wrap_parallel), so it sees the joined, averaged gradients
    from torch.optim.optimizer import register_optimizer_step_pre_hook
    h = register_optimizer_step_pre_hook(
        lambda opt, a, k: grads.append({n: named[n].grad.detach().cpu().clone() for n in keys}) if not grads else None)
    last = tr.train()
    h.remove()
    torch.cuda.synchronize()
    sd = model.model.state_dict()
    return ({k: sd[k].detach().cpu() for k in keys}, float(last["loss_cross_entropy"].detach()), sorted(sd), grads[0])


def _vt_worker(rank, world, port, acc, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _vt_train(_vt_cfg(acc), [rank], 2, 300 + rank)           # ranks start from different weights
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("acc", [1, 2])
def test_kdsfvt_two_ranks_reference_loop_equals_one_big_batch(acc):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_vt_worker, args=(r, 2, port, acc, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    (wa, la, keys, ga), (wb, lb, _, gb) = ret[0], ret[1]
    assert len(wa) == 7, keys
    for k in wa:                                    # replicas identical after two optimizer steps: weights and gradients
        assert torch.equal(wa[k], wb[k]), k
        assert torch.equal(ga[k], gb[k]), k
    one, l1, _, g1 = _vt_train(_vt_cfg(acc), [0, 1], 2, 300)                  # rank 0's weights, both ranks' data
    for k in wa:
        # the gradient the first optimizer step consumed == gradient of the (accumulated) big batch
        assert rel_err(ga[k], g1[k]) < 2e-4, k
        # RMSprop's first steps are sign-like (g / sqrt(0.05 g^2)): elements whose gradient is roundoff-sized move by
        # +-lr/0.22 in either direction, so the weights are compared in the l2 sense, not element by element
        assert float((wa[k] - one[k]).norm() / one[k].norm()) < 2e-3, k
    assert abs(0.5 * (la + lb) - l1) < 2e-3 * abs(l1)


def _rccl_single_rank_worker(port, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        from lvt_amd.engine.grad_reducer import BucketedGradReducer
        from lvt_amd.modeling import build_model
        from lvt_amd.utils.events import EventStorage
        cfg = _vt_cfg(2)
        torch.manual_seed(5)
        model = build_model(cfg)
        model.train()
        optimizers, _ = model.configure_optimizers_and_checkpointers()
        # what wrap_parallel installs, forced active in the 1-rank RCCL group, with small buckets (many collectives)
        model._reducers = [BucketedGradReducer(model.model.parameters(), bucket_bytes=4 << 20, reduce_single_rank=True)]
        red = model._reducers[0]
        assert red._avg and len(red.buckets) > 10
        batches = list(_vt_batches(cfg, [0], 2, 4))
        ref = None
        for it, batch in enumerate(batches):
            with EventStorage(it):
                loss = model(batch, mode="supervised")["loss_cross_entropy"]
            loss.backward()
            if it == 0:
                model.finish_gradient_sync()
                ref = {n: p.grad.detach().clone() for n, p in model.model.named_parameters() if p.grad is not None}
            if (it + 1) % 2 == 0:
                for o in optimizers:
                    o["optimizer"].step()          # joins through the pre-step hook
                # what the optimizer just read: every gradient lives in its bucket slot (moved there on the side stream)
                views = sum(int(p.grad is not None and p.grad.data_ptr() == red._slot[p][1].data_ptr()) for p in red.params)
                for o in optimizers:
                    o["optimizer"].zero_grad()
                dropped = sum(int(p.grad is None) for p in red.params)
        torch.cuda.synchronize()
        assert dropped == len(red.params)          # .grad is None between steps: the next backward accumulates nothing
        # the same first backward without any reducer
        torch.manual_seed(5)
        plain = build_model(cfg)
        plain.train()
        with EventStorage(0):
            plain(batches[0], mode="supervised")["loss_cross_entropy"].backward()
        worst = max(float((ref[n] - p.grad).abs().max() / (p.grad.abs().max() + 1e-30))
                    for n, p in plain.model.named_parameters() if p.grad is not None and n in ref)
        ret["out"] = (views, len(red.params), worst, float(loss.detach()))
    finally:
        dist.destroy_process_group()


def test_reducer_on_rccl_backend_single_rank():
    """The RCCL-specific code of the gradient reducer (ReduceOp.AVG, asynchronous all-reduce per bucket on the side stream
    launched from gradient hooks, joins by the optimizer pre-step hook and at the next forward, gradients moved into their
    bucket slots on the side stream) in a one-rank `nccl` group: a single-GPU box cannot host two RCCL ranks, and with one
    rank every collective is an identity, so the averaged gradients must equal the plain ones."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_rccl_single_rank_worker, args=(_free_port(), ret))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    views, nparams, worst, loss = ret["out"]
    assert views > 0.9 * nparams                  # at optimizer.step() the gradients are views of the buckets
    assert worst < 1e-6 and loss == loss
