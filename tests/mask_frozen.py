"""Mask-frozen double reference for whole-network gradient parity (tests/test_gpu_network_parity.py, test_network_parity_cpu.py).

A leaky-ReLU network is piecewise linear in its activations; an fp32 run and a double run of the same network disagree on the
branch of every pre-activation that lies within rounding of zero, and each such flip changes the gradients by a finite amount
(0.8 * sqrt(2) * the incoming gradient of that element), so "fp32 gradient vs double gradient" measures how many flips a run had,
not how accurately it differentiates.  Evaluated with the fp32 run's OWN sign pattern, the double run is the exact gradient of the
function the fp32 run evaluated, and the comparison is box- and library-independent.

The sign pattern of THIS package's run is recovered without touching the product: every activation's output is saved by some
autograd node for its backward pass (the mask is read from it), ``torch.autograd.graph.saved_tensors_hooks`` sees every saved
tensor, and the double restatement (oracle/aten_cpu_path.ActivationMasks.lookup) asks, at each of its activation calls, for the saved
tensor of that shape that equals its own activation to fp32 accuracy -- the sign of that tensor is the pattern the product's backward
used."""
import torch


class SavedActivations:
    """with SavedActivations() as s: y = net(x)   -> s.by_shape: {shape: [tensors saved for backward during the forward pass]}"""

    def __init__(self):
        self.by_shape = {}
        self.matched = 0
        self.elements = 0          # elements of the matched activations (the population the flip counts are out of)
        self.unmatched = []

    def __enter__(self):
        def pack(t):
            if t.is_floating_point() and t.dim() >= 2:
                lst = self.by_shape.setdefault(tuple(t.shape), [])
                if not any(u.data_ptr() == t.data_ptr() and u.stride() == t.stride() for u in lst):
                    lst.append(t.detach())
            return t
        self._hooks = torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t)
        self._hooks.__enter__()
        return self

    def __exit__(self, *exc):
        return self._hooks.__exit__(*exc)

    def sign_pattern(self, pre, act, rtol=1e-4):
        """The restatement's activation ``act`` (double) -> sign pattern of the saved tensor that is the same activation."""
        best, best_err = None, None
        scale = float(act.abs().max()) + 1e-30
        for t in self.by_shape.get(tuple(act.shape), []):
            err = float((t.to(act.dtype) - act).abs().max()) / scale
            if best_err is None or err < best_err:
                best, best_err = t, err
        if best is None or best_err > rtol:
            self.unmatched.append((tuple(act.shape), best_err))
            return None
        self.matched += 1
        self.elements += act.numel()
        return best > 0


def frozen_reference_grads(A, ref64, inputs64, run_ref, loss_weight, masks=None, saved=None):
    """Gradients of the double restatement w.r.t. (inputs, parameters) with the sign patterns of another run imposed:
    ``masks`` = patterns in call order (a recorded restatement run) or ``saved`` = SavedActivations of this package's run.
    -> (output, gradients, flips per activation call)"""
    ins = [t.detach().double().requires_grad_(True) for t in inputs64]
    ctx = A.ActivationMasks.replay(masks) if masks is not None else A.ActivationMasks.lookup(saved.sign_pattern)
    with ctx as m:
        y = run_ref(ref64, ins)
    g = torch.autograd.grad((y * loss_weight.double()).sum(), ins + list(ref64.parameters()))
    return y, g, m.flips
