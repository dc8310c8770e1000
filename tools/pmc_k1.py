"""Launch the K1 blur kernels a few times each on the step's largest planes (for rocprofv3 --pmc passes): the LDS-strip kernel and
the streaming kernel on the SAME problem (tuning build, dispatch knob SAE_K1_STREAM), plain and with the forward epilogue."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary  # noqa: E402
from tuning import build_tuning  # noqa: E402

lib = SaeLibrary(build_tuning.build())
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
kk = torch.ones(4, 4, device=dev) / 16


def blur(planes, hw, pad, act):
    x = torch.randn(planes, hw, hw, device=dev)
    o = hw + 2 * pad - 3
    y = torch.empty(planes, o, o, device=dev)
    if act:
        nz, nw, b = torch.randn(planes // 128, o, o, device=dev), torch.full((1,), 0.3, device=dev), torch.randn(128, device=dev)
        return lambda: lib.call("upfirdn2d_noise_bias_act_f32", x.data_ptr(), kk.data_ptr(), y.data_ptr(), planes, hw, hw, 4, 4, pad, pad,
                                pad, pad, nz.data_ptr(), nw.data_ptr(), b.data_ptr(), 128, 0.2, 2 ** 0.5, st), (x, y, nz, nw, b)
    return lambda: lib.call("upfirdn2d_f32", x.data_ptr(), kk.data_ptr(), y.data_ptr(), planes, hw, hw, 1, 4, 4, 1, 1, 1, 1, pad, pad, pad,
                            pad, st), (x, y)


for planes, hw, pad, act in ((5120, 256, 2, False), (2048, 257, 1, True)):
    fn, keep = blur(planes, hw, pad, act)
    for knob in ("0", "2"):
        os.environ["SAE_K1_STREAM"] = knob
        for _ in range(4):
            fn()
    torch.cuda.synchronize()
