"""Shared cases of the style-modulated convolution entry points (sae_modconv2d_*): every kernel family that stages a
modulated operand — the 3x3 / 1x1 gathers, the transposed gather, the three wgrad modes — incl. tiles that span
several images (per-slot factors), channel tails and the split-K path."""
import numpy as np

import abi_harness as H

# (n, c, h, w, m, k, stride, pad, weights stored [C, M, k, k])
MODCONV_CASES = [
    (2, 8, 8, 8, 32, 3, 1, 1, False),        # several images per tile
    (3, 10, 4, 4, 70, 3, 1, 1, False),       # channel tails, 128-row tile
    (1, 9, 20, 33, 130, 3, 1, 1, False),     # one image per tile (uniform factor path), M tail
    (2, 40, 8, 8, 3, 1, 1, 0, False),        # ToRGB: 1x1, 3 output channels
    (2, 22, 32, 32, 3, 1, 1, 0, False),      # ToRGB on a 1024-pixel plane: the streaming weight gradient with factors
    (2, 64, 8, 8, 40, 3, 1, 1, False),       # split-K
    (1, 70, 17, 17, 12, 3, 2, 0, True),      # transposed conv producing 70 channels ([C, M] weights)
    (2, 40, 19, 35, 8, 3, 2, 0, True),       # transposed, strips (2^k + 1 grid), two images
    (2, 12, 9, 9, 20, 3, 2, 0, True),        # transposed, narrow tile
    (2, 40, 17, 33, 24, 3, 2, 0, True),      # transposed, quad-staged (tr2): 8 x 16 input, strips, M / channel tails
    (1, 20, 9, 25, 40, 3, 2, 0, True),       # ... 32-row tile
    (3, 20, 12, 12, 24, 3, 1, 1, False),     # wgrad MODE 1
    (2, 3, 16, 16, 40, 3, 1, 1, False),      # wgrad MODE 2
    (1, 36, 32, 32, 70, 3, 1, 1, False),
    (3, 20, 16, 32, 70, 3, 1, 1, False),     # wgrad: one image per pixel chunk -> factors applied per K-slice in the reduction
]


def run_case(lib, oracle, case, device=None, tol=2e-5):
    n, c, h, w, m, k, s, p, cm = case
    d = H.conv_desc(n, c, h, w, m, k, s, p, cm)
    rng = np.random.default_rng(17)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = rng.standard_normal((c, m, k, k) if cm else (m, c, k, k)).astype(np.float32)
    gy = rng.standard_normal((n, m, d.oh, d.ow)).astype(np.float32)
    xs = (1.0 + 0.5 * rng.standard_normal((n, c))).astype(np.float32)
    ys = (1.0 + 0.5 * rng.standard_normal((n, m))).astype(np.float32)
    wm = rng.uniform(0.5, 2.0, m).astype(np.float32)
    wc = rng.uniform(0.5, 2.0, c).astype(np.float32)
    worst = 0.0
    for op, a, b, shape, kw in [
        (0, x, wt, gy.shape, dict(x_scale=xs, wm_scale=wm)),              # plain modulated forward
        (0, x, wt, gy.shape, dict(wc_scale=wc)),                          # input gradient of the transposed form
        (1, gy, wt, x.shape, dict(wm_scale=wm)),                          # input gradient of the plain form
        (1, gy, wt, x.shape, dict(y_scale=ys, wc_scale=wc)),              # transposed modulated forward
        (2, x, gy, wt.shape, dict(x_scale=xs)),                           # weight gradient, plain form
        (2, x, gy, wt.shape, dict(y_scale=ys)),                           # weight gradient, transposed form
        (2, x, gy, wt.shape, dict()),                                     # no factors: identical to the plain entry point
    ]:
        e = H.modconv(lib, op, d, a, b, shape, alpha=0.37, device=device, **kw)
        o = H.modconv(oracle, op, d, a, b, shape, alpha=0.37, **kw)
        assert not np.isnan(e).any(), (op, kw.keys())
        err = H.rel_err(e, o)
        assert err < tol, (op, sorted(kw), err)
        worst = max(worst, err)
    if s == 1 and not cm and k == 3:
        # StyledConv's plain form in one call (sae_modconv2d_fwd_noise_bias_act_f32): against the oracle's three modules one
        # after the other, and against this library's own two calls (modulated conv, then noise + bias + activation)
        noise = rng.standard_normal((n, 1, d.oh, d.ow)).astype(np.float32)
        nw = np.array([0.37], np.float32)
        bias = rng.standard_normal(m).astype(np.float32)
        for nz, b_ in ((noise, bias), (None, bias), (noise, None)):
            e = H.modconv_noise_bias_act(lib, d, x, wt, xs, wm, nz, nw if nz is not None else None, b_, alpha=0.37, device=device)
            o = H.modconv_noise_bias_act(oracle, d, x, wt, xs, wm, nz, nw if nz is not None else None, b_, alpha=0.37)
            err = H.rel_err(e, o)
            assert err < tol, ("fused noise", nz is not None, b_ is not None, err)
            worst = max(worst, err)
            if (d.oh * d.ow) % 4 == 0:
                two = H.noise_bias_act(lib, H.modconv(lib, 0, d, x, wt, gy.shape, alpha=0.37, device=device, x_scale=xs, wm_scale=wm),
                                       nz, nw, b_, device=device)
                assert H.rel_err(e, two) < 2e-7, ("fused noise vs two calls", H.rel_err(e, two))
    plain = H.conv(lib, 2, d, x, gy, wt.shape, alpha=0.37, device=device)
    assert np.array_equal(plain, H.modconv(lib, 2, d, x, gy, wt.shape, alpha=0.37, device=device))
    return worst
