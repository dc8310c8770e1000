"""Adam on the multi-tensor HIP kernel (csrc/adam.hip, C-ABI ``sae_adam_multi_f32``).

Stands where the reference constructs ``torch.optim.Adam(params, lr=..., betas=(beta1, beta2))``
(optimizers/swapping_autoencoder_optimizer.py:34-42) and steps it (:77,:95,:107): same constructor arguments,
same ``state_dict`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, so optimiser checkpoints move
between the two), same update rule.  Not bit-identical to ATen's: the kernel multiplies sqrt(v) by a host-computed
1 / sqrt(bias_correction2) where torch divides by sqrt(bias_correction2) -- an ulp-level difference per step
(tests/test_adam.py holds the two to 2e-7 relative over several steps).  One call to ``step`` issues a handful of
launches for the whole parameter list (226 tensors per group at the church preset) instead of ATen's per-tensor
elementwise chains.  The per-parameter step counters advance only after the launch was accepted.

``step(grad_views=..., grad_scale=...)`` lets the gradient all-reduce hand over its flat buckets directly: the
kernel reads the summed gradients where RCCL left them and applies the 1 / world_size itself, which removes the
scale and scatter-back passes of ``GradAllReducer.finish`` (grad_allreduce.py)."""
import ctypes as C

import torch

from . import hip_lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise hip_lib.SaeError("FusedAdam covers the reference's configuration: weight_decay = 0, amsgrad = False")
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters lr=%r betas=%r eps=%r" % (lr, betas, eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        self._dev_steps = None       # device-resident update counts (use_device_steps): int64 tensor, {parameter: slot}

    # ---- update counts in device memory (hipGraph replays of the train step: hip_graph.py) ----------------------------------
    def use_device_steps(self):
        """From now on the per-parameter update counts live in ONE int64 device tensor and ``step`` calls
        ``sae_adam_multi_dev_f32``: the kernels read the counts, form the bias corrections themselves and advance the counts, so
        a captured graph of the step replays with the right corrections (as kernel arguments they would be frozen at capture).
        The host-side ``state[p]["step"]`` tensors are refreshed from the device whenever ``state_dict()`` is asked for."""
        if self._dev_steps is not None:
            return
        params = [p for g in self.param_groups for p in g["params"]]
        host = torch.tensor([int(self._state_of(p)["step"].item()) for p in params], dtype=torch.int64)
        self._dev_steps = (host.to(params[0].device), {p: i for i, p in enumerate(params)})

    def _pull_steps(self):
        if self._dev_steps is not None:
            counts, slot = self._dev_steps
            host = counts.cpu()
            for p, i in slot.items():
                self._state_of(p)["step"] = torch.tensor(float(host[i]))

    def state_dict(self):
        self._pull_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self._dev_steps is not None:
            counts, slot = self._dev_steps
            host = torch.tensor([int(self._state_of(p)["step"].item()) for p in slot], dtype=torch.int64)
            counts.copy_(host)

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)                       # host scalar, as torch.optim.Adam keeps it
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None, grad_views=None, grad_scale=1.0, only=None):
        """grad_views: optional {parameter: flat fp32 tensor holding its gradient} (e.g. slices of all-reduced
        buckets); parameters absent from it use ``p.grad``.  only: optional container of parameters to restrict this
        call to (one bucket at a time).  Parameters without a gradient are skipped, as in torch.optim.Adam."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = hip_lib.get()
        for group in self.param_groups:
            ps, gs, ms, vs, ns, steps, keep, updated = [], [], [], [], [], [], [], []
            for p in group["params"]:
                if only is not None and p not in only:
                    continue
                g = grad_views.get(p) if grad_views is not None else None
                if g is None:
                    g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise hip_lib.SaeError("FusedAdam does not support sparse gradients")
                if not p.is_contiguous():
                    raise hip_lib.SaeError("FusedAdam needs contiguous parameters")
                g = g.contiguous()
                st = self._state_of(p)
                lib.check(p, g, st["exp_avg"], st["exp_avg_sq"])
                ps.append(p.data_ptr()); gs.append(g.data_ptr()); ms.append(st["exp_avg"].data_ptr())
                vs.append(st["exp_avg_sq"].data_ptr()); ns.append(p.numel())
                if self._dev_steps is None:
                    steps.append(int(st["step"].item()) + 1)
                keep.append((g, st))
                updated.append(p)
            if not ps:
                continue
            n = len(ps)
            arr_p, arr_g, arr_m, arr_v = ((C.c_void_p * n)(*x) for x in (ps, gs, ms, vs))
            arr_n = (C.c_int64 * n)(*ns)
            beta1, beta2 = group["betas"]
            if self._dev_steps is not None:
                counts, slot = self._dev_steps
                base = counts.data_ptr()
                arr_s = (C.c_void_p * n)(*[base + 8 * slot[p] for p in updated])
                lib.call("adam_multi_dev_f32", arr_p, arr_g, arr_m, arr_v, arr_n, arr_s, n, float(group["lr"]), float(beta1),
                         float(beta2), float(group["eps"]), float(grad_scale), lib.stream(group["params"][0]))
            else:
                arr_s = (C.c_int64 * n)(*steps)
                lib.call("adam_multi_f32", arr_p, arr_g, arr_m, arr_v, arr_n, arr_s, n, float(group["lr"]), float(beta1),
                         float(beta2), float(group["eps"]), float(grad_scale), lib.stream(group["params"][0]))
                for _, st in keep:           # the step counters advance only once the launch was accepted
                    st["step"] += 1
            # the kernel wrote the parameters through raw pointers: tell autograd's version counters, which is what every
            # in-place op of torch.optim does and what stylegan2_op/weight_prep.py (prepared conv weights) goes by
            for p in updated:
                torch.autograd.graph.increment_version(p)
        return loss
