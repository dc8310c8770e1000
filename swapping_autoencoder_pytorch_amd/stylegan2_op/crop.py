"""Random-crop sampler of the patch discriminator (csrc/crop.hip): the bilinear resampling that
util.apply_random_crop delegates to F.grid_sample (reference: util/util.py:323-343), straight from the source
image (no num_crops-fold expansion, no [B*crops, size, size, 2] grid) with a deterministic gather backward."""
import torch
from torch.autograd import Function

from .. import hip_lib


class RandomCropFunction(Function):
    @staticmethod
    def forward(ctx, x, params, size, crops):
        ctx.set_materialize_grads(False)
        lib = hip_lib.get()
        x = x.contiguous()
        params = params.contiguous()
        lin = torch.linspace(-1.0, 1.0, size, device=x.device)
        lib.check(x, params, lin)
        n, c, h, w = x.shape
        if params.shape != (n * crops, 5):
            raise hip_lib.SaeError("params must be [N * crops, 5], got %s" % (tuple(params.shape),))
        out = torch.empty((n * crops, c, size, size), dtype=x.dtype, device=x.device)
        lib.call("random_crop_f32", x.data_ptr(), params.data_ptr(), lin.data_ptr(), out.data_ptr(), n, c, h, w, crops,
                 size, lib.stream(x))
        ctx.save_for_backward(params, lin)
        ctx.geom = (n, c, h, w, crops, size)
        return out

    @staticmethod
    def backward(ctx, gy):
        if gy is None or not ctx.needs_input_grad[0]:
            return None, None, None, None
        params, lin = ctx.saved_tensors
        n, c, h, w, crops, size = ctx.geom
        lib = hip_lib.get()
        gy = gy.contiguous()
        lib.check(gy)
        gx = torch.empty((n, c, h, w), dtype=gy.dtype, device=gy.device)
        lib.call("random_crop_bwd_f32", gy.data_ptr(), params.data_ptr(), lin.data_ptr(), gx.data_ptr(), n, c, h, w, crops,
                 size, lib.stream(gy))
        return gx, None, None, None


def random_crop(x, flip, scale, offset, size, crops):
    """x: [N, C, H, W]; flip [N*crops] (+-1), scale / offset [N*crops, 2] (x, y) in grid units.
    Returns [N*crops, C, size, size] (first-order differentiable w.r.t. x)."""
    params = torch.cat([flip.reshape(-1, 1), scale.reshape(-1, 2), offset.reshape(-1, 2)], dim=1)
    return RandomCropFunction.apply(x, params, size, crops)
