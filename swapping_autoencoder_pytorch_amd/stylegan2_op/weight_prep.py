"""Prepared conv weights, kept across launches (include/sae_hip.h "Prepared weights").

Every forward / data-gradient launch re-lays its weight for its tile shape (conv_wprep_kernel: 319 launches per iteration of
the church preset).  A parameter changes once per optimiser step (optimizers/swapping_autoencoder_optimizer.py:77,95,107 of the
reference) and is used by 2 - 6 launches in between -- the generator's two passes, the discriminator in the D step and again in
the following G step -- so the re-laid copy is cached here per (parameter, layout, alpha) and handed to the library through
the descriptor.

Freshness is decided by the parameter's autograd VERSION COUNTER: every in-place update PyTorch can see bumps it (the
optimizers, ``copy_`` / ``load_state_dict``, ``state_dict()`` tensors -- they share the counter), and fused_adam.FusedAdam, which
writes through raw pointers, bumps it itself.  What it cannot see is a write through ``.data`` (a separate counter) or through a
foreign raw pointer: such code must call ``invalidate()`` or run with ``SAE_WPREP_CACHE=0``.

Only ``torch.nn.Parameter`` weights (or views of one) are cached: a temporary tensor in the weight position (the double-backward
operators of the R1 penalty convolve with gradients) can be freed and its address and version recur with other contents.
An entry holds a weak reference to its parameter and is dropped -- with its device buffer -- when the parameter dies (the
weak reference's callback).  Memory: one padded copy per (parameter, layout) in use, i.e. the forward and the data-gradient
layouts of every conv weight: about twice the conv weights on top of the parameters themselves for the direct kernels; a layer
on the Winograd route keeps its sixteen transform-domain points instead of nine taps, 16/9 x the weight (padded to 64 output
channels and 8 contraction channels) per direction, i.e. about 3.6 x the weight for both -- 0.9 GB at the church preset."""
import ctypes as C
import os
import weakref

import torch

_ENTRIES = {}        # (data_ptr, layout, alpha, tag) -> _Entry
_QUERIES = {}        # (library id, conv arithmetic, descriptor key, op, modulated activation) -> (floats, layout)


class _Entry:
    __slots__ = ("ref", "version", "buf", "event", "stream")


def enabled():
    return os.environ.get("SAE_WPREP_CACHE", "1") != "0"


def _drop(key, ref):
    """weakref callback: the parameter behind entry `key` died -- free the prepared copy unless the key was re-used since"""
    e = _ENTRIES.get(key)
    if e is not None and e.ref is ref:
        del _ENTRIES[key]


def invalidate():
    """Drop every prepared copy (after writing parameters behind autograd's back)."""
    _ENTRIES.clear()


def cached(w, key, build):
    """A tensor derived from parameter `w` alone (the Winograd-domain weights of stylegan2_op/winograd.py), kept until the
    parameter's version counter moves: build() -> tensor, made on the current stream.  Not a Parameter (or a view of one), or
    the cache switched off: built per call."""
    if not enabled():
        return build()
    base = w._base if w._base is not None else w
    if not isinstance(base, torch.nn.Parameter):
        return build()
    k = (w.data_ptr(),) + tuple(key)
    e = _ENTRIES.get(k)
    cur = torch.cuda.current_stream(w.device) if w.is_cuda else None
    if e is None or e.ref() is not base or e.version != base._version:
        e = _Entry()
        e.ref, e.version = weakref.ref(base, lambda _r, key=k: _drop(key, _r)), base._version
        e.buf = build()
        e.stream = cur
        e.event = cur.record_event() if cur is not None else None
        _ENTRIES[k] = e
    elif cur is not None and cur != e.stream:
        cur.wait_event(e.event)          # built on the step's other stream (streams.py)
        e.buf.record_stream(cur)
    return e.buf


def attach(lib, d, mod, op, w, alpha, tag=(), gkey=None):
    """Point descriptor `d` at prepared weights for the launch (d, mod, op) on weight tensor `w`, preparing them if the cached
    copy is missing or older than the parameter.  `tag`: hashable identity of the weight FACTORS in `mod` as a function of the
    parameter (() = none); a launch whose factors are not a function of the parameter alone must not be cached (tag None).
    Returns the buffer (keep it alive until the launch is enqueued) or None when nothing was attached."""
    if tag is None or not enabled():
        return None
    base = w._base if w._base is not None else w
    if not isinstance(base, torch.nn.Parameter):
        return None
    # (the conv arithmetic is a process-wide switch of the library: bf16x6 lays the weights out as split cells)
    # (gkey: the caller's own hashable form of the descriptor's geometry -- reading 13 fields back out of the ctypes struct costs
    # more than the rest of this function)
    qkey = (id(lib), lib.query("get_conv_math"), gkey if gkey is not None else d.key(), op,
            bool(mod is not None and (mod.x_scale or mod.y_scale)))
    q = _QUERIES.get(qkey)
    if q is None:
        floats, layout = C.c_int64(0), C.c_int64(0)
        lib.call("conv2d_wprep_query", C.byref(d), C.byref(mod) if mod is not None else None, op, C.byref(floats), C.byref(layout))
        q = _QUERIES[qkey] = (floats.value, layout.value)
    floats, layout = q
    if floats <= 0:
        return None
    key = (w.data_ptr(), layout, float(alpha), tag)
    e = _ENTRIES.get(key)
    cur = torch.cuda.current_stream(w.device) if w.is_cuda else None
    if e is None or e.ref() is not base or e.version != base._version or e.buf.numel() != floats:
        e = _Entry()
        e.ref, e.version = weakref.ref(base, lambda _r, key=key: _drop(key, _r)), base._version
        e.buf = torch.empty(floats, dtype=torch.float32, device=w.device)
        lib.call("conv2d_wprep_f32", w.data_ptr(), C.byref(d), C.byref(mod) if mod is not None else None, op, float(alpha),
                 e.buf.data_ptr(), floats, lib.stream(w))
        e.stream = cur
        e.event = cur.record_event() if cur is not None else None
        _ENTRIES[key] = e
    elif cur is not None and cur != e.stream:
        cur.wait_event(e.event)          # prepared on the step's other stream (streams.py)
        e.buf.record_stream(cur)
    d.prepped, d.prepped_floats, d.prepped_layout = e.buf.data_ptr(), floats, layout
    return e.buf
