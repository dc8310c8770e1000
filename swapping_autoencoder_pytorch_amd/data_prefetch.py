"""Host-to-device input prefetch for the training loop.

The reference ships a dormant ``DataPrefetcher`` (data/__init__.py:52-82: one batch ahead, a side CUDA stream,
``wait_stream`` on hand-over) and feeds the step from a ``torch.utils.data.DataLoader`` (:96-104).  This is its
MI355X counterpart; the drop-in runner wraps the reference's loader in it whenever the run uses a GPU
(dropin.PrefetchedLoader / wrap_dataloader): batches are staged through PINNED host buffers and
copied on a dedicated HIP stream ``depth`` batches ahead, so the 12.6 MB of a 256x256 B=16 batch (0.2 ms over PCIe
Gen5) never sits on the step's critical path; the consumer stream waits on the copy's event only, and the device
buffers are tied to the consumer stream with ``record_stream`` so the caching allocator cannot recycle them early.

    for batch in DevicePrefetcher(loader, device="cuda:0"):      # batch: dict of device tensors (other values pass through)
        optimizer.train_one_step(batch, step)
"""
import collections

import torch


class DevicePrefetcher:
    def __init__(self, iterable, device="cuda", depth=2):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher stages batches into GPU memory; got device %s" % self.device)
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.source = iter(iterable)
        self.depth = depth
        self.stream = torch.cuda.Stream(self.device)
        self.queue = collections.deque()
        self.exhausted = False
        for _ in range(depth):
            self._stage_one()

    def _to_device(self, value):
        if torch.is_tensor(value):
            host = value if value.is_pinned() else value.contiguous().pin_memory()
            return host.to(self.device, non_blocking=True), host       # keep the pinned source alive until consumed
        if isinstance(value, dict):
            out, keep = {}, []
            for k, v in value.items():
                out[k], h = self._to_device(v)
                keep.append(h)
            return out, keep
        if isinstance(value, (list, tuple)):
            pairs = [self._to_device(v) for v in value]
            return type(value)(p[0] for p in pairs), [p[1] for p in pairs]
        return value, None

    def _stage_one(self):
        if self.exhausted:
            return
        try:
            batch = next(self.source)
        except StopIteration:
            self.exhausted = True
            return
        with torch.cuda.stream(self.stream):
            dev, keep = self._to_device(batch)
            done = torch.cuda.Event()
            done.record(self.stream)
        self.queue.append((dev, keep, done))

    @staticmethod
    def _record(value, stream):
        if torch.is_tensor(value):
            value.record_stream(stream)
        elif isinstance(value, dict):
            for v in value.values():
                DevicePrefetcher._record(v, stream)
        elif isinstance(value, (list, tuple)):
            for v in value:
                DevicePrefetcher._record(v, stream)

    def __iter__(self):
        return self

    def __next__(self):
        if not self.queue:
            raise StopIteration
        dev, keep, done = self.queue.popleft()
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(done)                 # device-side wait only; the host does not block
        self._record(dev, cur)
        self._stage_one()                    # refill: the next copy overlaps the step that consumes `dev`
        return dev
