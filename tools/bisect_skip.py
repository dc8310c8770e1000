"""the skip path of G.UpsamplingResBlock128 alone (generator.py:39-53 of the reference: ConvLayer 1x1 256 -> 128 with
bias + leaky ReLU at 128^2, bilinear x2), this package vs ATen fp32 vs ATen double on the same input: pre-activation accuracy, leaky-ReLU
flips against the double run, and gx / gb / gw against the free and the mask-frozen double run.   python tools/bisect_skip.py > out.json"""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import aten_cpu_path as A
from swapping_autoencoder_pytorch_amd.stylegan2_layers import ConvLayer
from swapping_autoencoder_pytorch_amd.stylegan2_op import upsample2x_add

DEV = "cuda:0"


def l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    g = torch.Generator().manual_seed(5)
    b, cin, cout, hw = 16, 256, 128, 128
    ours = ConvLayer(cin, cout, 1, activate=True, bias=True).to(DEV)
    ref32 = A.GenConvLayerCPU(cin, cout, 1, activate=True, bias=True).to(DEV)
    with torch.no_grad():
        for p, q in zip(ours.parameters(), ref32.parameters()):
            v = (torch.randn(p.shape, generator=g) * (1.0 if p.dim() > 1 else 0.2)).to(DEV)
            p.copy_(v)
            q.copy_(v)
    ref64 = A.GenConvLayerCPU(cin, cout, 1, activate=True, bias=True).to(DEV).double()
    ref64.load_state_dict({k: v.double() for k, v in ref32.state_dict().items()})
    x = torch.randn(b, cin, hw, hw, generator=g).to(DEV)
    res = torch.zeros(b, cout, 2 * hw, 2 * hw, device=DEV)
    t = torch.randn(b, cout, 2 * hw, 2 * hw, generator=g).to(DEV)
    out = {}

    def run_ref(m, xin, masks=None):
        xin = xin.detach().to(next(m.parameters()).dtype).requires_grad_(True)
        ctx = A.ActivationMasks.replay(masks) if masks is not None else A.ActivationMasks.record()
        with ctx as rec:
            a = m(xin)
        y = F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=False) / (2 ** 0.5)
        gr = torch.autograd.grad((y * t.to(y.dtype)).sum(), [xin] + list(m.parameters()))
        return a.detach(), y.detach(), gr, rec.masks

    a64, y64, g64, m64 = run_ref(ref64, x)
    a32, y32, g32, m32 = run_ref(ref32, x)
    xo = x.clone().requires_grad_(True)
    ao = ours(xo)
    yo = upsample2x_add(ao, res, 1.0 / 2 ** 0.5)
    go = torch.autograd.grad((yo * t).sum(), [xo] + list(ours.parameters()))
    mo = [ao.detach() > 0]
    _, _, g64_o, _ = run_ref(ref64, x, mo)
    _, _, g64_c, _ = run_ref(ref64, x, m32)
    names = ["gx"] + ["g " + n for n, _ in ours.named_parameters()]
    out["activation l2 (ours, aten fp32) vs double"] = [l2(ao, a64), l2(a32, a64)]
    out["activation max-norm (ours, aten fp32)"] = [float((ao.double() - a64).abs().max() / a64.abs().max()),
                                                    float((a32.double() - a64).abs().max() / a64.abs().max())]
    out["flips vs double (ours, aten fp32) of %d" % ao.numel()] = [int((mo[0] != m64[0]).sum()), int((m32[0] != m64[0]).sum())]
    out["output l2 (ours, aten fp32)"] = [l2(yo, y64), l2(y32, y64)]
    for i, n in enumerate(names):
        out[n + ": l2 vs free double (ours, aten fp32)"] = [l2(go[i], g64[i]), l2(g32[i], g64[i])]
        out[n + ": l2 vs mask-frozen double (ours, aten fp32)"] = [l2(go[i], g64_o[i]), l2(g32[i], g64_c[i])]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
