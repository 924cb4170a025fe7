"""Bucketed data-parallel gradient averaging over RCCL (replaces the reference's use of
torch DistributedDataParallel: ae.py:69-73, vt.py:61-63).

One process per GPU.  Parameters are grouped, in reverse registration order (the order their
gradients become available in the backward pass), into flat buckets of >= `bucket_bytes`.  Autograd hands
every parameter a FRESH gradient tensor (`.grad` is None between steps, so nothing is accumulated on the
compute stream); when the last gradient of a bucket has arrived, ONE multi-tensor copy moves the bucket's
gradients into its flat buffer on the communication stream, the buffer is all-reduced there
(ReduceOp.AVG: no separate 1/world pass; RCCL drives all 7 xGMI links) while the remaining backward kernels
run, and `.grad` is re-pointed at the bucket slot -- the compute stream sees no copy, no add and no memset.
(Round 4 kept `.grad` attached to a zeroed bucket view instead: autograd then ADDS every new gradient into
it, one extra kernel per parameter on the compute stream -- 1.2 ms of a 50 ms step at 320 parameters.)

Semantics are those of torch DDP, with no call required from the training loop
(vidgen/engine/trainer.py:79-87 has none):

  * EVERY backward averages the gradients (a DDP without `no_sync`).  With gradient accumulation the
    bucket holds avg(g1) + local g2 before the second reduction and avg(g1) + avg(g2) after it.
  * the reduction is joined automatically (a) before any `torch.optim.Optimizer.step()` (global
    step pre-hook), (b) at the next forward of the owning meta-architecture, and (c) by an explicit
    `wait()` for callers that read `.grad` themselves.  A gradient that arrives for a bucket whose
    previous reduction is still in flight (two `backward()` calls with none of (a)-(c) in between) is an
    ERROR, as it is in torch DDP: autograd has by then accumulated into memory the collective is reading.
  * `optimizer.zero_grad(set_to_none=True)` (torch's default, and what `solver/fused.py` optimizers do) drops `.grad`;
    the next backward's gradient is moved into the slot as above.  A gradient that is ALREADY the bucket view
    (gradient accumulation: a second backward before the step; or `zero_grad(set_to_none=False)`) was accumulated in
    place by autograd and needs no move.

On CPU (gloo, used by the world_size-2 unit tests) the same logic runs with SUM + a division, because
gloo has no AVG.
"""
import weakref

import os

import torch
import torch.distributed as dist
from torch.optim.optimizer import register_optimizer_step_pre_hook


# LVT_DP_SINGLE_RANK=1: every reducer is active in a process group of ONE rank too (bench.py --dp-single-rank, the overlap
# trace of scripts / profiles): the RCCL launches, the side stream and the joins of the 8-GPU run on a single-GPU box.
FORCE_SINGLE_RANK = bool(os.environ.get("LVT_DP_SINGLE_RANK"))


class BucketedGradReducer:
    def __init__(self, params, bucket_bytes=None, group=None, broadcast_params=True, reduce_single_rank=None):
        """reduce_single_rank: run the collectives also in a process group of ONE rank (they are identities there);
        tests use it to drive the RCCL code path -- AVG reduction, asynchronous work on the side stream, the joins -- on
        a single-GPU box."""
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        if bucket_bytes is None:
            bucket_bytes = int(float(os.environ.get("LVT_DP_BUCKET_MB", "16")) * (1 << 20))
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self._div = self.world
        if reduce_single_rank is None:
            reduce_single_rank = FORCE_SINGLE_RANK or bool(os.environ.get("LVT_DP_SINGLE_RANK"))
        if reduce_single_rank and self.world == 1 and dist.is_available() and dist.is_initialized():
            self.world = 2          # "active"; the divisor of the SUM path stays the true group size
        if broadcast_params and self.world > 1:
            # like torch DDP at construction: every replica starts from rank 0's weights
            with torch.no_grad():
                for p in self.params:
                    dist.broadcast(p.data, 0, group=group)
        self.buckets = []          # dict(flat, params, views, pending, work)
        self._slot = {}
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._make_bucket(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._make_bucket(cur)
        self._comm_stream = None
        self.trace = None          # a list: every bucket launch / join appends (kind, bucket index, bytes, event) -- a HIP event on the
                                   # stream the collective / the join was issued on (scratch/dp_overlap_trace.py)
        self.enabled = not os.environ.get("LVT_DP_REDUCERS_OFF")     # False: gradients stay local (bench.py times a step without
                                   # communication; the environment switch does the same from the first step: timing only)
        # gloo (CPU unit tests, and the 2-ranks-on-one-GPU tests) has no AVG: SUM + one division there
        self._avg = self.world > 1 and dist.get_backend(group) == "nccl"
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        ref = weakref.ref(self)

        def _before_step(optimizer, args, kwargs):
            me = ref()
            if me is not None:
                me.wait()
        self._hooks.append(register_optimizer_step_pre_hook(_before_step))

    def remove(self):
        """Detach from the parameters and the optimizers (gradients stay where they are)."""
        self.wait()
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            ref = getattr(p, "_lvt_reducer", None)
            if ref is not None and ref() is self:
                del p._lvt_reducer

    def _make_bucket(self, plist):
        dev, dt = plist[0].device, plist[0].dtype
        total = sum(p.numel() for p in plist)
        flat = torch.zeros(total, dtype=dt, device=dev)
        b = {"flat": flat, "params": list(plist), "views": [], "pending": len(plist), "work": None, "index": len(self.buckets),
             "moves": [], "hold": None}   # (parameter, view, fresh gradient) of this backward, copied by _launch
        off = 0
        for p in plist:
            v = flat[off:off + p.numel()].view_as(p)
            b["views"].append(v)
            self._slot[p] = (len(self.buckets), v)
            p._lvt_reducer = weakref.ref(self)
            off += p.numel()
        self.buckets.append(b)

    def _stream(self, device):
        if device.type != "cuda":
            return None
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=device)
        return self._comm_stream

    def zero_grad(self, only=None):
        """Drop the gradients (of the parameters in `only`, default all): `.grad = None`, torch's set_to_none semantics.
        The next backward then hands autograd-fresh tensors to the hooks (no accumulation kernel on the compute stream)."""
        if self.world == 1:
            return False
        self.wait()
        for b in self.buckets:
            for p in b["params"]:
                if only is None or p in only:
                    p.grad = None
        return True

    @property
    def bytes_per_backward(self):
        """Bytes every rank contributes to the all-reduces of one backward pass."""
        return sum(b["flat"].numel() * b["flat"].element_size() for b in self.buckets)

    def _on_grad(self, p):
        if self.world == 1 or not self.enabled:
            return
        bi, view = self._slot[p]
        b = self.buckets[bi]
        if b["work"] is not None:
            # a second backward without a join in between: autograd has already accumulated the new gradient in place
            # into the bucket view the all-reduce in flight is reading (and, on the SUM path, the join's division would
            # scale the fresh local gradient too) -- the result cannot be repaired here, so fail as torch DDP does
            raise RuntimeError(
                "BucketedGradReducer: a gradient arrived for a bucket whose all-reduce from the previous backward() is still "
                "in flight.  Join it first -- optimizer.step(), the meta-architecture's forward / finish_gradient_sync(), or "
                "reducer.wait() -- before calling backward() again (torch DDP has the same rule)")
        g = p.grad
        if g.data_ptr() != view.data_ptr():
            b["moves"].append((p, view, g))    # a fresh tensor from autograd: moved into the slot by _launch, in one copy per bucket
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def _launch(self, b):
        flat = b["flat"]
        cs = self._stream(flat.device)
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        moves, b["moves"] = b["moves"], []
        if cs is not None:
            cs.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(cs):
                self._move(b, moves)
                b["work"] = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        else:
            self._move(b, moves)
            b["work"] = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        if self.trace is not None and cs is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(cs)               # completes when the collective of this bucket has
            self.trace.append(("allreduce_done", b["index"], flat.numel() * flat.element_size(), ev))

    @staticmethod
    def _move(b, moves):
        """Fresh gradients -> their bucket slots (one multi-tensor copy on the current = communication stream), `.grad`
        re-pointed at the slot.  Anything that reads `.grad` does so after a join, which orders it behind this copy; the
        fresh tensors are kept alive until that join (cheaper than `record_stream`, which makes the caching allocator poll an
        event for every one of them on later allocations)."""
        if not moves:
            return
        with torch.no_grad():
            torch._foreach_copy_([v for _, v, _ in moves], [g for _, _, g in moves])
        for p, v, g in moves:
            p.grad = v
        b["hold"] = [g for _, _, g in moves]

    def _join_all(self, todo):
        """The collectives of `todo` joined with ONE wait of the compute stream: the communication stream waits for each of
        them (c10d runs a collective on its own stream and `Work.wait()` orders the CURRENT stream behind it), the compute
        stream then waits for the communication stream once -- a wait + an event per bucket on the compute stream cost
        0.25 ms of a 12-bucket step."""
        cs = self._comm_stream
        if cs is not None:
            with torch.cuda.stream(cs):
                for b in todo:
                    b["work"].wait()
            main = torch.cuda.current_stream(cs.device)
            main.wait_stream(cs)
            if self.trace is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(main)
                self.trace.append(("joined", todo[-1]["index"], 0, ev))
        else:
            for b in todo:
                b["work"].wait()
        for b in todo:
            if not self._avg:
                b["flat"].div_(self._div)
            b["work"] = None
            b["hold"] = None               # (the compute stream is now ordered behind the copy that read them)
            b["pending"] = len(b["params"])

    def wait(self):
        """Join outstanding all-reduces; afterwards every `.grad` holds the cross-rank mean."""
        if self.world == 1:
            return
        # buckets whose parameters did not all receive a gradient this backward are reduced as they are
        for b in self.buckets:
            if b["work"] is None and 0 < b["pending"] < len(b["params"]):
                self._launch(b)
        todo = [b for b in self.buckets if b["work"] is not None]
        if todo:
            self._join_all(todo)
