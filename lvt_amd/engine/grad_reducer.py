"""Bucketed data-parallel gradient averaging over RCCL (replaces the reference's use of
torch DistributedDataParallel: ae.py:69-73, vt.py:61-63).

One process per GPU.  Parameters are grouped, in reverse registration order (the order their
gradients become available in the backward pass), into flat buckets of >= `bucket_bytes`.  A
post-accumulate-grad hook copies each finished gradient into its bucket slot; when a bucket is
complete it is all-reduced asynchronously on a dedicated communication stream (RCCL drives all 7
xGMI links), overlapping the remaining backward kernels.  `wait()` (called before the optimizer
step) joins the streams, scales by 1/world and points `.grad` at the averaged bucket views.

On CPU (gloo, used by the world_size-2 unit tests) the same logic runs synchronously.
"""
import torch
import torch.distributed as dist


class BucketedGradReducer:
    def __init__(self, params, bucket_bytes=16 << 20, group=None, broadcast_params=True):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if broadcast_params and self.world > 1:
            # like torch DDP at construction: every replica starts from rank 0's weights
            with torch.no_grad():
                for p in self.params:
                    dist.broadcast(p.data, 0, group=group)
        self.buckets = []          # list of dict(flat, params, offsets, pending, work)
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
        self._handles = []
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    def _make_bucket(self, plist):
        dev, dt = plist[0].device, plist[0].dtype
        total = sum(p.numel() for p in plist)
        b = {"flat": torch.zeros(total, dtype=dt, device=dev), "params": list(plist), "pending": len(plist),
             "offsets": []}
        off = 0
        for p in plist:
            b["offsets"].append(off)
            self._slot[p] = (len(self.buckets), off)
            off += p.numel()
        self.buckets.append(b)

    def _stream(self, device):
        if device.type != "cuda":
            return None
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=device)
        return self._comm_stream

    def _on_grad(self, p):
        if self.world == 1:
            return
        bi, off = self._slot[p]
        b = self.buckets[bi]
        b["flat"][off:off + p.numel()].copy_(p.grad.reshape(-1))
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def _launch(self, b):
        flat = b["flat"]
        cs = self._stream(flat.device)
        if cs is not None:
            cs.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(cs):
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._handles.append((b, work))

    def wait(self):
        """Join outstanding all-reduces, average, and expose the result through `.grad`."""
        if self.world == 1:
            return
        # buckets whose parameters did not all receive a gradient this step are reduced as they are
        for b in self.buckets:
            if 0 < b["pending"] < len(b["params"]):
                self._launch(b)
        for b, work in self._handles:
            work.wait()
        cs = self._comm_stream
        if cs is not None:
            torch.cuda.current_stream(cs.device).wait_stream(cs)
        for b, _ in self._handles:
            b["flat"].div_(self.world)
            for p, off in zip(b["params"], b["offsets"]):
                if p.grad is not None:
                    p.grad = b["flat"][off:off + p.numel()].view_as(p)
        self._handles = []
        for b in self.buckets:
            b["pending"] = len(b["params"])
