"""Bucketed gradient all-reduce overlapped with backward (one process per GPU, RCCL over xGMI).

Replaces what ``nn.DataParallel`` did implicitly in the reference (models/__init__.py:80-93:
replicate / scatter / gather / reduce_add through GPU 0 on every call).  Design for MI355X:

* every rank holds a full replica and an equal shard of the batch; losses are per-sample means,
  so averaging gradients over ranks reproduces the global-batch gradient;
* the trainable set flips between {E, G} and {D, Dpatch} on every call, so there is one reducer
  per group (not one static DDP bucket list over the whole model);
* gradients are packed into ~32 MB fp32 buckets in REVERSE parameter order (the order backward
  produces them); a post-accumulate-grad hook counts arrivals and launches ``all_reduce`` on
  the bucket (async, on RCCL's own stream) as soon as its last gradient lands, so the 213-223 MB
  of a step fly while the early layers' wgrad kernels still run.  xGMI is point-to-point
  (7 links x ~153 GB/s): a handful of large buckets keeps every ring step bandwidth- rather
  than latency-bound;
* ``finish()`` waits, scales by 1/world and scatters the averaged values back into ``.grad``;
  ``finish_into(optimizer)`` instead hands every completed bucket to the multi-tensor Adam kernel, which reads
  the summed gradients in place (fused_adam.FusedAdam).  Parameters that received no gradient in this pass
  contribute zeros to the collective (its shape stays identical on all ranks) and are skipped by the update;
* gradients as bucket views: while a reducer is armed, the weight-gradient kernels of the convolutions write their
  result STRAIGHT into the parameter's slot of the flat bucket (``claim_destination``, used by
  stylegan2_op/conv2d_gemm.py): autograd then adopts that view as ``.grad`` and the grad-ready hook has nothing to
  copy.  Slots start on 256-byte boundaries so that the kernels' and Adam's 16-byte accesses stay aligned.

With world_size == 1 (or torch.distributed not initialised) every method is a no-op."""
import os

import torch
import torch.distributed as dist

BUCKET_BYTES = int(os.environ.get("SAE_ALLREDUCE_BUCKET_MB", "32")) * 1024 * 1024
SLOT_ALIGN = 64          # floats: every parameter's slot starts on a 256-byte boundary

# data_ptr of a parameter -> its slot in an armed reducer's flat bucket, handed out ONCE per backward pass
_DESTINATIONS = {}
CLAIMED = [0]            # slots handed to weight-gradient producers since import (tests read the difference across a pass)


def claim_destination(weight):
    """For a weight-gradient producer: the bucket slot (shaped like `weight`) the gradient of `weight` should be written
    into, or None.  One-shot: a parameter used twice in a graph gets the slot for its first gradient only, the second is
    accumulated into it by autograd as usual."""
    if not _DESTINATIONS:
        return None
    slot = _DESTINATIONS.pop(weight.data_ptr(), None)
    if slot is None or slot.numel() != weight.numel():
        return None
    CLAIMED[0] += 1
    return slot.view(weight.shape)


class _PassRecord:
    """One backward pass of an instrumented reducer (``GradAllReducer.profile = True``): HIP events, read after the run."""
    __slots__ = ("bytes", "buckets", "first_launch", "last_done", "waits")

    def __init__(self):
        self.bytes, self.buckets, self.first_launch, self.last_done, self.waits = 0, 0, None, None, []


class _Bucket:
    __slots__ = ("params", "offsets", "numel", "flat", "pending", "work")

    def __init__(self):
        self.params, self.offsets, self.numel = [], [], 0
        self.flat, self.pending, self.work = None, 0, None


class GradAllReducer:
    def __init__(self, params, bucket_bytes=BUCKET_BYTES, process_group=None):
        self.params = [p for p in params]
        self.group = process_group
        # SAE_FORCE_ALLREDUCE=1 runs the full bucket / hook / collective path even on a single rank (a
        # one-GPU rehearsal of the multi-GPU code over RCCL; see tests/test_gpu_allreduce.py)
        forced = os.environ.get("SAE_FORCE_ALLREDUCE", "0") == "1"
        self.enabled = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(process_group) > 1 or forced)
        self.world = dist.get_world_size(process_group) if self.enabled else 1
        self.armed = False
        self._arm_pending = False
        self.buckets = []
        self._where = {}
        # instrumentation (bench.py --gpus N): per pass the bytes all-reduced, an event at the first bucket's launch, one after
        # the last bucket's completion was waited for, and an event pair around every wait -- the time a stream that had nothing
        # else left to do stood waiting for a collective is the EXPOSED (non-overlapped) part of the all-reduce
        self.profile = False
        self.records = []
        self._rec = None
        if not self.enabled:
            return
        cur = _Bucket()
        for p in reversed(self.params):
            if cur.numel > 0 and (cur.numel + p.numel()) * 4 > bucket_bytes:
                self.buckets.append(cur)
                cur = _Bucket()
            cur.offsets.append(cur.numel)
            cur.params.append(p)
            cur.numel += (p.numel() + SLOT_ALIGN - 1) // SLOT_ALIGN * SLOT_ALIGN
            self._where[p] = (len(self.buckets), len(cur.params) - 1)
        if cur.numel:
            self.buckets.append(cur)
        for b in self.buckets:
            ref = b.params[0]
            b.flat = torch.zeros(b.numel, dtype=ref.dtype, device=ref.device)
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    # called right before loss.backward()
    def arm(self, hand_out_slots=True):
        if not self.enabled:
            return
        self.armed = True
        self._arm_pending = False
        for b in self.buckets:
            b.pending = len(b.params)
            b.work = None
            lo, hi = b.flat.data_ptr(), b.flat.data_ptr() + b.flat.numel() * b.flat.element_size()
            for p in b.params:
                # a gradient kept alive from the previous pass (zero_grad(set_to_none=False), gradient accumulation) may
                # BE last pass's bucket slot: move it out before the bucket is cleared, or the next producer would write
                # into `.grad` itself and autograd's `grad += new` would then add the slot to itself (2x)
                if p.grad is not None and lo <= p.grad.data_ptr() < hi:
                    p.grad = p.grad.clone()
            b.flat.zero_()
            for p, off in zip(b.params, b.offsets):
                # conv weights: their wgrad kernels can write into the slot -- only when autograd will ADOPT the result as
                # `.grad` (no gradient alive); otherwise it accumulates into the existing one and the hook copies the sum
                # (ModulatedConv2d keeps its weight as [1, O, I, k, k] and convolves a 4-D view of it: same address)
                # ``hand_out_slots=False`` (arming from inside a gradient hook, in the MIDDLE of a backward pass): no slot is
                # registered -- the ``zero_()`` above was enqueued on the hook's stream, and with the step on two streams a
                # weight gradient launched next on the OTHER stream would write its slot with no ordering against it.  The
                # hooks then copy every gradient in (always correct, one copy per conv weight dearer).
                if (hand_out_slots and (p.dim() == 4 or (p.dim() == 5 and p.shape[0] == 1)) and p.is_contiguous()
                        and p.grad is None):
                    _DESTINATIONS[p.data_ptr()] = b.flat[off:off + p.numel()]

    def arm_lazily(self):
        """For a host loop this package does not own (dropin.attach_gradient_allreduce): arm at the first gradient of the next
        pass unless ``arm()`` is called before (the drop-in calls it from the optimizer's ``zero_grad``).  Right after
        ``optimizer.step()`` every conv-weight ``.grad`` still IS its bucket slot: arming there would clone each of them out
        and, with a gradient alive, register no destination -- the write-into-bucket path would be lost from the second
        iteration on.  After ``zero_grad()`` the gradients are gone and the slots can be handed out again."""
        if self.enabled:
            self._arm_pending = True

    def _on_grad(self, p):
        if not self.armed:
            if not self._arm_pending:
                return
            # no zero_grad since the last step (gradient accumulation, model.zero_grad(), p.grad = None): the late,
            # always-correct form -- copy-in-hook only, no bucket slot handed to a producer mid-pass
            self.arm(hand_out_slots=False)
        bi, pi = self._where[p]
        b = self.buckets[bi]
        off = b.offsets[pi]
        slot = b.flat[off:off + p.numel()]
        if p.grad.data_ptr() != slot.data_ptr():          # (a conv weight's gradient was produced in place: nothing to copy)
            slot.copy_(p.grad.reshape(-1))
        b.pending -= 1
        if b.pending == 0:
            from .streams import collective_launch
            with collective_launch(b.flat):     # ordered after BOTH streams of the step: the other gradients may come from either
                self._note_launch(b)
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _release_destinations(self):
        for b in self.buckets:
            for p in b.params:
                _DESTINATIONS.pop(p.data_ptr(), None)

    def _launch_stragglers(self):
        """Buckets whose last gradient never arrived (some parameter got no gradient in this pass) still take part
        in the collective, with zeros in the missing slots, so its shape is identical on every rank."""
        from .streams import collective_launch
        for b in self.buckets:
            if b.work is None:
                # (with the step on two streams the bucket's gradients may come from either: ordered after both)
                with collective_launch(b.flat):
                    self._note_launch(b)
                    b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    # ---- instrumentation ------------------------------------------------------------------------------------------------
    def _note_launch(self, b):
        if not self.profile or not b.flat.is_cuda:
            return
        if self._rec is None:
            self._rec = _PassRecord()
        r = self._rec
        r.bytes += b.flat.numel() * b.flat.element_size()
        r.buckets += 1
        if r.first_launch is None:
            r.first_launch = torch.cuda.Event(enable_timing=True)
            r.first_launch.record()              # on the launch stream, which has waited for the bucket's gradients

    def _wait(self, b):
        """``b.work.wait()`` (the current stream waits for the collective), bracketed by events when instrumented."""
        if not self.profile or not b.flat.is_cuda or self._rec is None:
            b.work.wait()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.work.wait()
        e1.record()
        self._rec.waits.append((e0, e1))
        self._rec.last_done = e1

    def _close_record(self):
        if self._rec is not None:
            self.records.append(self._rec)
            self._rec = None

    def summary(self, skip=0):
        """Means over the recorded passes (after ``torch.cuda.synchronize()``), the first `skip` left out: MB all-reduced per
        pass, buckets, ms from the first bucket's launch to the last bucket's completion, ms the waiting stream stood idle for
        the collectives (exposed)."""
        recs = [r for r in self.records[skip:] if r.first_launch is not None and r.last_done is not None]
        if not recs:
            return None
        span = [r.first_launch.elapsed_time(r.last_done) for r in recs]
        exposed = [sum(a.elapsed_time(b) for a, b in r.waits) for r in recs]
        n = len(recs)
        return {"passes": n, "buckets": recs[0].buckets, "mb_allreduced": round(recs[0].bytes / 1e6, 2),
                "ms_first_launch_to_last_done": round(sum(span) / n, 3), "ms_exposed": round(sum(exposed) / n, 3)}

    # called after loss.backward(), before optimizer.step()
    def finish(self):
        """Wait for the buckets and scatter the averaged gradients back into ``.grad``.  A parameter that received
        no gradient in this pass keeps ``grad = None`` (every rank runs the same graph, so it has none on any
        rank) and the optimizer skips it exactly as it does on one GPU."""
        if not self.enabled or not self.armed:
            return
        self.armed = False
        self._release_destinations()
        inv = 1.0 / self.world
        self._launch_stragglers()
        for b in self.buckets:
            self._wait(b)
            b.flat.mul_(inv)
            for p, off in zip(b.params, b.offsets):
                if p.grad is not None and p.grad.data_ptr() != b.flat[off:off + 1].data_ptr():
                    p.grad.copy_(b.flat[off:off + p.numel()].view_as(p.grad))
            b.work = None
        self._close_record()

    def finish_into(self, optimizer):
        """finish() and optimizer.step() in one: as each bucket's all-reduce completes, the multi-tensor Adam kernel
        (fused_adam.FusedAdam) updates that bucket's parameters reading the SUMMED gradients straight from the flat
        bucket and applying 1 / world itself — no scale pass, no scatter back into ``.grad``, and the update of the
        early buckets overlaps the collectives of the late ones.  Afterwards ``.grad`` is NOT the averaged gradient: a
        gradient that was copied into its slot keeps this rank's local values, one that was produced in place (a conv
        weight's bucket view) holds the rank-SUMMED, unscaled values.  Code that reads ``.grad`` after the step (norm
        logging, clipping) must use ``finish()`` + ``optimizer.step()`` instead."""
        if not self.enabled or not self.armed:
            optimizer.step()
            return
        self.armed = False
        self._release_destinations()
        inv = 1.0 / self.world
        self._launch_stragglers()
        for b in self.buckets:
            self._wait(b)
            views = {p: b.flat[off:off + p.numel()] for p, off in zip(b.params, b.offsets) if p.grad is not None}
            if views:
                optimizer.step(grad_views=views, grad_scale=inv, only=views)
            b.work = None
        self._close_record()


def broadcast_parameters(module, src=0, process_group=None):
    """One-time replica synchronisation at start-up (parameters and buffers)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=process_group)
