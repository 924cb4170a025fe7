"""Bucketed data-parallel gradient averaging over RCCL (replaces the reference's use of
torch DistributedDataParallel: ae.py:69-73, vt.py:61-63).

One process per GPU.  Parameters are grouped into flat buckets of >= `bucket_bytes` in the order their gradients ARRIVE in a
backward pass (measured in the first backward in which every parameter gets one, rank 0's order broadcast to all ranks; before
that: reverse registration order).  Autograd hands every parameter a FRESH gradient tensor (`.grad` is None between steps, so
nothing is accumulated on the compute stream); when the last gradient of a bucket has arrived, ONE multi-tensor copy moves the
bucket's gradients into its flat buffer on the communication stream, the buffer is all-reduced there (ReduceOp.AVG: no separate
1/world pass; RCCL drives all 7 xGMI links) while the remaining backward kernels run, and `.grad` is re-pointed at the bucket
slot -- the compute stream sees no copy, no add and no memset.

Every rank issues the SAME sequence of collectives in every backward, whatever gradients it got (round 6): buckets are launched
strictly in index order -- bucket i when it is complete AND buckets 0 .. i-1 have been launched -- and the join launches whatever
is left, in index order, if the backward produced any gradient at all.  A parameter without a gradient on this rank contributes
zeros and ends up with the mean of the other ranks' gradients (torch DDP raises in that situation unless it is told to search
for unused parameters; launching such a bucket late, as rounds 2-5 did, pairs different collectives on different ranks).  A
parameter that gets no gradient on ANY rank therefore sees a zero gradient instead of None: with the reference's optimizers
(no weight decay) its update is zero either way.  Because the order is fixed, it has to be the arrival order -- a bucket that
closes late would hold back every bucket behind it: the embedding / first-layer gradients, registered first, used to share the
LAST bucket of a DSFVT backward with the tables that arrive first (profiles/r05_dp_overlap_timeline.txt: 18 MB closed at +0.00 ms).

Semantics are otherwise those of torch DDP, with no call required from the training loop
(vidgen/engine/trainer.py:79-87 has none):

  * EVERY backward averages the gradients (a DDP without `no_sync`).  With gradient accumulation the
    bucket holds avg(g1) + local g2 before the second reduction and avg(g1) + avg(g2) after it.
  * the reduction is joined automatically (a) before any `torch.optim.Optimizer.step()` (global
    step pre-hook), (b) at the next forward of the owning meta-architecture, and (c) by an explicit
    `wait()` for callers that read `.grad` themselves.  A gradient that arrives for a bucket whose
    previous reduction is still in flight (two `backward()` calls with none of (a)-(c) in between) is an
    ERROR, as it is in torch DDP: autograd has by then accumulated into memory the collective is reading.
  * `optimizer.zero_grad()` drops `.grad` (set_to_none semantics; `set_to_none=False` is not honoured for bucketed
    parameters: a zeroed view that stays attached makes autograd ADD every new gradient into it, one extra kernel per parameter
    on the compute stream -- 1.2 ms of a 50 ms step at 320 parameters); the next backward's gradient is moved into the slot as
    above.  A gradient that is ALREADY the bucket view (gradient accumulation: a second backward before the step) was
    accumulated in place by autograd and needs no move.
  * memory: the fresh gradients of a backward stay alive next to the flat buckets until the join (one extra gradient copy at the
    peak, ~200 MB for DSFVT): the copy into the bucket runs on the communication stream, and releasing them earlier needs an
    event per bucket on the allocator's path.

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
    CALIBRATION_TRIES = 3          # backward passes that may go by before the bucket order is fixed from an incomplete arrival list

    def __init__(self, params, bucket_bytes=None, group=None, broadcast_params=True, reduce_single_rank=None):
        """reduce_single_rank: run the collectives also in a process group of ONE rank (they are identities there);
        tests use it to drive the RCCL code path -- AVG reduction, asynchronous work on the side stream, the joins -- on
        a single-GPU box."""
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        if bucket_bytes is None:
            bucket_bytes = int(float(os.environ.get("LVT_DP_BUCKET_MB", "16")) * (1 << 20))
        self.bucket_bytes = bucket_bytes
        have_pg = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if have_pg else 1            # the TRUE group size (divisor of the SUM path)
        if reduce_single_rank is None:
            reduce_single_rank = FORCE_SINGLE_RANK or bool(os.environ.get("LVT_DP_SINGLE_RANK"))
        # active: gradients are bucketed and all-reduced (more than one rank, or the one-rank group of the single-GPU drills)
        self.active = self.world > 1 or bool(reduce_single_rank and have_pg)
        if broadcast_params and self.world > 1:
            # like torch DDP at construction: every replica starts from rank 0's weights
            with torch.no_grad():
                for p in self.params:
                    dist.broadcast(p.data, 0, group=group)
        self.buckets = []          # dict(flat, params, views, pending, work, ...)
        self._slot = {}
        self._build(list(reversed(self.params)))     # until measured: the order in which autograd usually finishes them
        self._calibrated = not self.active or os.environ.get("LVT_DP_FIXED_ORDER") is not None
        self._tries = 0
        self._arrivals, self._arrived = [], set()    # this backward: parameters in the order their gradients came
        self._next = 0                                # first bucket not launched yet in this backward
        self._comm_stream = None
        self.trace = None          # a list: every bucket launch / join appends (kind, bucket index, bytes, event) -- a HIP event on the
                                   # stream the collective / the join was issued on (tools/profile/dp_overlap_trace.py)
        self.enabled = not os.environ.get("LVT_DP_REDUCERS_OFF")     # False: gradients stay local (bench.py times a step without
                                   # communication; the environment switch does the same from the first step: timing only)
        # gloo (CPU unit tests, and the 2-ranks-on-one-GPU tests) has no AVG: SUM + one division there
        self._avg = self.active and dist.get_backend(group) == "nccl"
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

    def _build(self, order):
        """(Re)make the flat buckets for the parameters in `order`."""
        self.buckets, self._slot = [], {}
        cur, cur_bytes = [], 0
        for p in order:
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= self.bucket_bytes:
                self._make_bucket(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._make_bucket(cur)

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
        if not self.active:
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
        if not self.active or not self.enabled:
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
        if p not in self._arrived:
            self._arrived.add(p)
            self._arrivals.append(p)
        if not self._calibrated:
            return                               # the first backward passes only measure; everything is launched by the join
        g = p.grad
        if g.data_ptr() != view.data_ptr():
            b["moves"].append((p, view, g))    # a fresh tensor from autograd: moved into the slot by _launch, in one copy per bucket
        b["pending"] -= 1
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _launch(self, b):
        flat = b["flat"]
        cs = self._stream(flat.device)
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        moves, b["moves"] = b["moves"], []
        # parameters of the bucket that got no gradient in this backward and hold none from an earlier one: zeros
        missing = [(p, v) for p, v in zip(b["params"], b["views"]) if p not in self._arrived and
                   (p.grad is None or p.grad.data_ptr() != v.data_ptr())]
        if cs is not None:
            cs.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(cs):
                self._move(b, moves, missing)
                b["work"] = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        else:
            self._move(b, moves, missing)
            b["work"] = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        if self.trace is not None and cs is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(cs)               # completes when the collective of this bucket has
            self.trace.append(("allreduce_done", b["index"], flat.numel() * flat.element_size(), ev))

    @staticmethod
    def _move(b, moves, missing=()):
        """Fresh gradients -> their bucket slots (one multi-tensor copy on the current = communication stream), `.grad`
        re-pointed at the slot.  Anything that reads `.grad` does so after a join, which orders it behind this copy; the
        fresh tensors are kept alive until that join (cheaper than `record_stream`, which makes the caching allocator poll an
        event for every one of them on later allocations)."""
        with torch.no_grad():
            if missing:
                torch._foreach_zero_([v for _, v in missing])
            if moves:
                torch._foreach_copy_([v for _, v, _ in moves], [g for _, _, g in moves])
        for p, v in missing:
            p.grad = v
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
                b["flat"].div_(self.world)
            b["work"] = None
            b["hold"] = None               # (the compute stream is now ordered behind the copy that read them)
            b["pending"] = len(b["params"])

    def _calibrate(self):
        """End of a measuring backward: fix the bucket order.  Every rank must take the same decision and the same order, so
        both come from collectives: MIN over the ranks of "every parameter got a gradient", then rank 0's arrival order."""
        self._tries += 1
        index = {p: i for i, p in enumerate(self.params)}
        dev = self.params[0].device
        complete = torch.tensor([1 if len(self._arrivals) == len(self.params) else 0], dtype=torch.int64, device=dev)
        if self.world > 1:
            dist.all_reduce(complete, op=dist.ReduceOp.MIN, group=self.group)
        if int(complete) == 0 and self._tries < self.CALIBRATION_TRIES:
            return
        rest = [p for b in self.buckets for p in b["params"] if p not in self._arrived]      # (kept in their current order)
        order = torch.tensor([index[p] for p in self._arrivals + rest], dtype=torch.int64, device=dev)
        if self.world > 1:
            dist.broadcast(order, 0, group=self.group)
        held = {p: p.grad for p in self.params if p.grad is not None}
        self._build([self.params[i] for i in order.tolist()])
        for p, g in held.items():            # (a gradient that lived in an old bucket view stays valid: the view keeps its storage)
            p.grad = g
        self._calibrated = True

    def wait(self):
        """Join outstanding all-reduces; afterwards every `.grad` holds the cross-rank mean."""
        if not self.active:
            return
        if self._arrivals and self.enabled:
            if not self._calibrated:
                self._calibrate()
                # the measuring backward launched nothing: queue every gradient there is, in the (possibly new) bucket order
                for p in self._arrivals:
                    bi, view = self._slot[p]
                    if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                        self.buckets[bi]["moves"].append((p, view, p.grad))
            # whatever this backward left unlaunched goes out now, in index order, on every rank alike
            while self._next < len(self.buckets):
                self._launch(self.buckets[self._next])
                self._next += 1
        self._arrivals, self._arrived, self._next = [], set(), 0
        todo = [b for b in self.buckets if b["work"] is not None]
        if todo:
            self._join_all(todo)
