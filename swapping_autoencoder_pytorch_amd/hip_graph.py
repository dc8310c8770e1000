"""hipGraph replay of the discriminator call and the generator call of the train step.

The reference's loop (train.py:22-28) enqueues every kernel of an iteration from Python: ~2 600 launches per iteration at the
church preset, ~1 300 of them a few microseconds long (weight re-layouts, split-K reductions, ATen adds, l2 normalisations).
Where such launches follow each other faster than the host can enqueue them the GPU idles; the two-stream step hides part of
that behind the other branch, a one-stream step (the default of a multi-rank job, streams.py) pays it in full, and the
32 x 32 preset is host-bound altogether.  Here each of the two calls of ``SwappingAutoencoderOptimizer.train_one_step`` --
``zero_grad``, the model's loss command, ``backward``, the Adam update -- is captured ONCE into a hipGraph (through
``torch.cuda.CUDAGraph``: PyTorch's allocator, RNG and autograd engine are capture-aware) after a few eager calls, and every
later call is one graph launch.

What makes the step capturable (checked by tests/test_gpu_graph_step.py: graph and eager steps agree bit for bit over several
iterations, parameters and Adam moments included):

  * no host read-back inside the call (the losses are read by ``util.to_numpy`` after it);
  * the random draws (crop windows, noise maps) come from torch's default CUDA generator, whose Philox offset a graph replay
    advances exactly as the eager call does;
  * Adam's update counts live in device memory (``FusedAdam.use_device_steps`` -> ``sae_adam_multi_dev_f32``): as kernel
    arguments the bias corrections would be frozen at their capture-time values;
  * the prepared-weight cache (stylegan2_op/weight_prep.py) is emptied before a capture -- every first use of a weight inside
    the graph re-lays it inside the graph, so a graph never depends on a buffer another graph or an eager call filled -- and
    after every capture / replay, so an eager pass in between (the lazy-R1 call, snapshots, evaluation) never trusts a buffer
    whose parameter a replay has moved on (a replay does not run the Python that bumps autograd's version counters);
  * the lazy-R1 call (every 16th discriminator iteration, double backward) stays eager.

Not captured: a rank of a multi-rank job (the bucketed RCCL all-reduce is driven by grad-ready hooks and per-bucket waits on
the host).  ``SAE_HIP_GRAPH=0`` keeps every call eager; unset or ``1``: graphs on a single-rank GPU process whose optimisers
are FusedAdam."""
import os

import torch

from .stylegan2_op import weight_prep

WARMUP_CALLS = 2         # eager calls of a (call, shape) before its capture: Adam state, constant tables, allocator warm


def wanted(params, optimizers):
    """Graph mode for an optimizer driver with these parameters / optimisers (see the module docstring)."""
    if os.environ.get("SAE_HIP_GRAPH", "1") == "0":
        return False
    if not params or not params[0].is_cuda:
        return False
    from .fused_adam import FusedAdam
    if not all(isinstance(o, FusedAdam) for o in optimizers):
        return False
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


def _fresh_containers(out):
    """The same (static) tensors in new dicts / lists / tuples: the caller adds keys to the loss dictionaries it gets back."""
    if isinstance(out, dict):
        return {k: _fresh_containers(v) for k, v in out.items()}
    if isinstance(out, (list, tuple)):
        return type(out)(_fresh_containers(v) for v in out)
    return out


class _Captured:
    def __init__(self, body, images):
        self.static_in = images.detach().clone()
        weight_prep.invalidate()
        self.graph = torch.cuda.CUDAGraph()
        # capture_begin / capture_end by hand instead of the `torch.cuda.graph` context manager, which empties the caching
        # allocator first: the eager passes that remain (the lazy-R1 call, every 16th discriminator iteration) would then rebuild
        # their gigabytes of blocks from hipMalloc at their first call after the capture -- 100 - 340 ms on the ffhq512 preset
        # (profiles/r6_ffhq512_lazy_r1_graph_vs_eager.txt).  thread_local: a data-loading thread may go on allocating and copying
        # while this thread captures.
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.graph.capture_begin(capture_error_mode="thread_local")
            try:
                self.out = body(self.static_in)
            finally:
                self.graph.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        # nothing has RUN yet: whatever the capture "prepared" holds no data until the first replay
        weight_prep.invalidate()

    def replay(self, images):
        self.static_in.copy_(images)
        self.graph.replay()
        weight_prep.invalidate()
        return _fresh_containers(self.out)


class StepGraphs:
    """``run(key, images, body)``: ``body(images)`` eagerly for the first WARMUP_CALLS calls of (key, shape), then captured and
    replayed.  ``body`` must return tensors (or containers of tensors) only; they are STATIC -- overwritten by the next replay --
    so the caller reads or copies them before it calls again (``train_one_step`` returns them through ``util.to_numpy``)."""

    def __init__(self, warmup=WARMUP_CALLS):
        self.warmup = warmup
        self.calls = {}
        self.graphs = {}

    def run(self, key, images, body):
        k = (key, tuple(images.shape), images.device)
        n = self.calls.get(k, 0)
        self.calls[k] = n + 1
        if n < self.warmup:
            return body(images)
        g = self.graphs.get(k)
        if g is None:
            g = self.graphs[k] = _Captured(body, images)
        return g.replay(images)

    def captured(self):
        return sorted(k[0] for k in self.graphs)
