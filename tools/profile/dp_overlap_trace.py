"""Where the gradient all-reduces of a DSFVT train step are issued and joined: a ONE-rank RCCL group with the reducers active
(LVT_DP_SINGLE_RANK) -- the launches, the side stream and the joins of the data-parallel path on a single-GPU box.  RCCL does not
launch a kernel for a one-rank all-reduce, so a kernel trace shows none; what CAN be timed here is when each bucket's collective
becomes runnable (its event on the communication stream completes once the bucket's last gradient kernel has finished) and
where the optimizer's pre-hook joins.   python tools/profile/dp_overlap_trace.py [steps] > profiles/r05_dp_overlap_timeline.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LVT_DP_SINGLE_RANK"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
import torch
import torch.distributed as dist
import bench
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
leg = bench.DsfvtLeg("cuda:0", 1, 0, 0, 64, 4, dp=True)
r = leg.model._reducers[0]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for i in range(3):
    leg.step(i)
torch.cuda.synchronize()
print("# DSFVT train step (b = 64) in a ONE-rank RCCL group, BucketedGradReducer active (backend %s, ReduceOp.AVG: %s): %d buckets, %d bytes"
      % (dist.get_backend(), r._avg, len(r.buckets), r.bytes_per_backward))
print("# per step: HIP events -- t0 before the forward, fwd_end after the loss, `bucket k runnable` on the communication stream right "
      "after its all_reduce was enqueued (it waits for the stream that produced the gradients), bwd_end after the last backward "
      "launch, `joined` on the compute stream inside the Optimizer.step pre-hook, step_end after the optimizer")
print("# (a one-rank all-reduce is an identity without a kernel: the DURATION of the collectives is not visible here; at 8 GPUs a "
      "ring all-reduce of B bytes moves 2 * 7/8 * B per GPU over xGMI)")
for i in range(steps):
    ev = {k: torch.cuda.Event(enable_timing=True) for k in ("t0", "fwd_end", "bwd_end", "step_end")}
    from lvt_amd.utils.events import EventStorage
    ctx, sl, sidx, ign = leg.batches[i % leg.nbatches]
    r.trace = []
    ev["t0"].record()
    with EventStorage(i):
        loss = leg.model.compute_supervised_loss(ctx, sl, sidx, ign)["loss_cross_entropy"]
    ev["fwd_end"].record()
    import time
    h0 = time.perf_counter()
    loss.backward()
    host_bwd_ms = (time.perf_counter() - h0) * 1e3          # host time of backward(): below the GPU's when the host runs ahead
    ev["bwd_end"].record()
    for o in leg.optimizers:
        o["optimizer"].step()
    for o in leg.optimizers:
        o["optimizer"].zero_grad()
    ev["step_end"].record()
    torch.cuda.synchronize()
    tr, r.trace = r.trace, None
    t = lambda e: ev["t0"].elapsed_time(e)
    print("step %d: forward ends %.2f ms, backward ends %.2f ms (host returned from backward() after %.2f ms), step ends %.2f ms"
          % (i, t(ev["fwd_end"]), t(ev["bwd_end"]), host_bwd_ms, t(ev["step_end"])))
    for kind, bi, nbytes, e in tr:
        if kind == "allreduce_done":
            print("    bucket %2d (%6.1f MB) runnable at %7.2f ms = %6.2f ms BEFORE the backward pass ends" % (bi, nbytes / 1e6, t(e), t(ev["bwd_end"]) - t(e)))
    joins = [t(e) for kind, bi, nbytes, e in tr if kind == "joined"]
    if joins:
        print("    joins (Optimizer.step pre-hook): first at %.2f ms, last at %.2f ms = %.2f ms after the backward pass ends" % (joins[0], joins[-1], joins[-1] - t(ev["bwd_end"])))
dist.destroy_process_group()
