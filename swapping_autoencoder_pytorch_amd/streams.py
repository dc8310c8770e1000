"""Two HIP streams for the independent branches of a training step.

The step has pairs of sub-graphs that do not depend on each other: the reconstruction and the hybrid pass of the generator
(swapping_autoencoder_model.py:122-124,192-201 of the reference: ``rec = G(sp[:b/2], gl[:b/2])``, ``mix = G(swap(sp), gl)``),
and the image discriminator next to the patch discriminator (:62-114).  Each of them ends in small maps (16 x 16 ... 4 x 4)
whose launches cannot fill 256 CUs; enqueued on two streams the hardware runs a branch's small launches in the slots the
other branch leaves free.  Autograd runs every backward node on the stream its forward ran on and orders the streams where a
gradient crosses, so the backward pass forks and joins the same way without further code.

(Measured and dropped: a third stream for the conv WEIGHT gradients of a backward pass, joined by a final callback of the
autograd engine -- bit-identical values, 281.9 / 280.2 ms per step in place against 281.8 / 282.6 on the extra stream,
profiles/r4_ab_async_wgrad.txt: with two branches in flight the chip has no idle slots left for them.)

Values do not change: every kernel is deterministic and sees the same inputs (tests/test_gpu_determinism.py compares the two
modes bit for bit); the host enqueues the branches in the order the single-stream code ran them, so the random draws (Philox
offsets are advanced at enqueue time) are the same too.  ``SAE_TWO_STREAMS=0`` keeps everything on the current stream (the default
for a rank of a multi-rank job: ``enabled()``)."""
import os

import torch

_SIDE = {}
_SILENCED = [False]
_MAIN = {}      # the stream the step itself runs on, as seen at the last fork


def enabled():
    """``SAE_TWO_STREAMS=1`` / ``0`` decides; unset: two streams in a single-rank process, ONE when the process is a rank of a
    multi-rank job.  There the gradient all-reduce adds a stream whose kernels wait for the buckets: rehearsed on one GPU with
    stand-in kernels for a ring's steps (``bench.py --force-allreduce --ring-rehearsal 8``, profiles/r5_ring_rehearsal.txt), that
    costs the two-stream step 20 ms of its 217 -- more than the second stream gains -- and the one-stream step 1.3 of its 227."""
    v = os.environ.get("SAE_TWO_STREAMS")
    if v is not None:
        return v != "0"
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


def side_stream(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class side_branch:
    """``with side_branch(x, y) as br: out = f(x, y)`` enqueues ``f`` on the device's side stream once everything already
    enqueued on the current stream (the producers of ``x``, ``y``) is ordered before it; ``br.join(out)`` makes the current
    stream wait for the branch, after which ``out`` may be consumed on it.  Tensors that cross are registered with the
    other stream (``record_stream``) so the caching allocator does not recycle their memory under a pending reader.
    With CPU tensors, or ``SAE_TWO_STREAMS=0``, the body simply runs in place."""

    def __init__(self, *inputs):
        self.inputs = [t for t in inputs if torch.is_tensor(t) and t.is_cuda]
        self.on = bool(self.inputs) and enabled()
        self._ctx = None

    def __enter__(self):
        if self.on:
            dev = self.inputs[0].device
            self.main = torch.cuda.current_stream(dev)
            self.side = side_stream(dev)
            _MAIN[(dev.type, dev.index if dev.index is not None else torch.cuda.current_device())] = self.main
            if not _SILENCED[0]:
                # a module used by both branches of a pair (the generator) has its gradients summed on the stream of its first
                # use, the other branch's contribution crossing over: autograd points that out once per process as a possible
                # oversight; here it is the design (the sum needs both anyway)
                _SILENCED[0] = True
                quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
                if quiet is not None:
                    quiet(False)
            self.side.wait_stream(self.main)
            for t in self.inputs:
                t.record_stream(self.side)
            self._ctx = torch.cuda.stream(self.side)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
            self._ctx = None
        return False

    def join(self, *outputs):
        if not self.on:
            return
        self.main.wait_stream(self.side)
        for t in outputs:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(self.main)


_LAUNCH = {}


class collective_launch:
    """``with collective_launch(tensor): work = dist.all_reduce(tensor, async_op=True)`` -- issue a collective from a dedicated
    stream that has waited for BOTH streams of the step.  The gradient all-reduce launches a bucket's collective from the
    grad-ready hook of its LAST gradient, on whichever stream that node runs; the bucket's other gradients may have been
    produced on the other stream (grad_allreduce.py).  RCCL orders the collective after the stream that is current at the call:
    making that a stream of its own keeps the two compute streams from waiting for each other (ordering the CURRENT stream
    after the other one instead cost the single-rank rehearsal 7 ms per step: 15 cross-stream joins in every backward pass).
    On CPU tensors (gloo tests) the body runs in place."""

    def __init__(self, tensor):
        self.tensor = tensor
        self._ctx = None

    def __enter__(self):
        t = self.tensor
        if t.is_cuda:
            dev = t.device
            key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
            if key not in _LAUNCH:
                _LAUNCH[key] = torch.cuda.Stream(device=dev)
            launch = _LAUNCH[key]
            cur = torch.cuda.current_stream(dev)
            launch.wait_stream(cur)
            for other in (_SIDE.get(key), _MAIN.get(key)):
                if other is not None and other != cur:
                    launch.wait_stream(other)
            t.record_stream(launch)
            self._ctx = torch.cuda.stream(launch)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
            self._ctx = None
        return False
