"""Host -> device input of a training step, overlapped with the previous step.

The reference's models copy every sample to the device inside `preprocess_data`, one `torch.as_tensor(..., device)` per sample and
step (vidgen/modeling/meta_arch/ae.py:151-168, vt.py:284-299), from the pageable arrays its DataLoader workers return
(vidgen/data/build.py:41-107).  Here a batch crosses PCIe ONCE, from pinned memory, while the previous step computes:

    for data in DevicePrefetcher(loader, device):        # `data` is still a list[dict], the models' input contract
        loss_dict = model(data, mode="supervised")

Per key the per-sample arrays are stacked into a reusable PINNED host buffer (two per key: one being filled while the other is in
flight), copied with one asynchronous `copy_` on a side stream, and handed over as a list of per-sample VIEWS of the batched device
tensor -- `stack_to_device` (modeling/meta_arch/common.py) recognises such views and takes the batch as it is, no second copy.
The consumer's stream waits for the copy (an event), never the host.  25 MB per PR-DVQVAE2 step at 32 clips, 4 MB per DSFVT step.
"""
import numpy as np
import torch


class DevicePrefetcher:
    def __init__(self, loader, device, depth=2):
        self.loader, self.device = loader, torch.device(device)
        self.depth = max(2, int(depth)) + 1     # pinned slots: the batches in the queue + the one being consumed + the one being filled
        self._pinned = {}                     # key -> list of `depth` pinned host buffers (grown on demand)
        self._slot = 0
        self._copied = [None] * self.depth    # per slot: event of the last H2D copy that READ its pinned buffers
        self._stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def _host_buffer(self, key, shape, dtype):
        bufs = self._pinned.setdefault(key, [None] * self.depth)
        b = bufs[self._slot]
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != dtype:
            b = torch.empty(shape, dtype=dtype, pin_memory=self.device.type == "cuda")
            bufs[self._slot] = b
        return b

    def _stage(self, data):
        """list[dict] of host samples -> (list[dict] of device views, event)."""
        keys = [k for k, v in data[0].items() if isinstance(v, (np.ndarray, torch.Tensor))]
        out = [dict(d) for d in data]
        done = None
        if self._copied[self._slot] is not None:
            self._copied[self._slot].synchronize()       # the copy that last read this slot's pinned buffers (`depth` batches ago)
        ctx = torch.cuda.stream(self._stream) if self._stream is not None else _Null()
        with ctx:
            for k in keys:
                first = data[0][k]
                if isinstance(first, torch.Tensor) and first.device.type != "cpu":
                    continue                  # already on a device: left alone
                if isinstance(first, np.ndarray):
                    # numpy's own stack straight into the pinned buffer: one memcpy per sample, no torch thread pool involved (the
                    # same copies through Tensor.copy_ from a worker thread ran 15x slower: tools/profile/h2d_probe.py)
                    arrs = [np.asarray(d[k]) for d in data]
                    host = self._host_buffer(k, (len(arrs),) + tuple(arrs[0].shape), torch.from_numpy(arrs[0][..., :0] if arrs[0].ndim else arrs[0].reshape(1)[:0]).dtype)
                    np.stack(arrs, out=host.numpy())
                else:
                    arrs = [d[k] for d in data]
                    host = self._host_buffer(k, (len(arrs),) + tuple(arrs[0].shape), arrs[0].dtype)
                    torch.stack(arrs, out=host)
                dev = host.to(self.device, non_blocking=True)
                for i, o in enumerate(out):
                    o[k] = dev[i]
            if self._stream is not None:
                done = torch.cuda.Event()
                done.record(self._stream)
                self._copied[self._slot] = done
        self._slot = (self._slot + 1) % self.depth
        return out, done

    def __iter__(self):
        """Staging runs in a worker thread (the copies into pinned memory release the GIL), `depth - 1` batches ahead: the main
        thread spends its time queueing the launches of the current step, which is what bounds a step fed this way."""
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth - 2)
        stop = threading.Event()

        def work():
            try:
                if self._stream is not None:
                    torch.cuda.set_device(self.device)
                for data in self.loader:
                    if stop.is_set():
                        return
                    q.put(("batch", self._stage(data)))
                q.put(("end", None))
            except BaseException as e:        # surfaces in the consumer
                q.put(("error", e))
        t = threading.Thread(target=work, daemon=True)
        t.start()
        try:
            while True:
                kind, item = q.get()
                if kind == "end":
                    return
                if kind == "error":
                    raise item
                cur, ev = item
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)
                yield cur
        finally:
            stop.set()
            while t.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    t.join(0.01)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
