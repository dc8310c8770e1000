"""Launch the one-kernel Winograd kernels a few times each on two of the step's shapes (for rocprofv3 --pmc passes, tools/r5_run_d.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swapping_autoencoder_pytorch_amd import hip_lib  # noqa: E402
from swapping_autoencoder_pytorch_amd.stylegan2_op import conv2d_gemm as cg, winograd  # noqa: E402

hip_lib.get()
for n, c, side in [(16, 512, 64), (2, 128, 256)]:          # the first one is tools/pmc_summary.py's reference launch (the larger grid)
    g = cg._Geom(n, c, side, side, c, 3, 1, 1, False, 1.0 / (c * 9) ** 0.5)
    x = torch.randn(n, c, side, side, device="cuda")
    gy = torch.randn(n, c, side, side, device="cuda")
    w = torch.nn.Parameter(torch.randn(c, c, 3, 3, device="cuda"))
    for _ in range(3):
        winograd.conv(x, w, g, kind="fused")
    for _ in range(3):
        winograd.wgrad(x, gy, g, kind="fused")
    torch.cuda.synchronize()
