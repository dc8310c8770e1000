"""Drive the C-ABI (include/sae_hip.h) of any bound library on numpy data.

`lib` is a SaeLibrary: the oracle (host pointers), the emulator build of the kernels (host
pointers) or the real HIP library (device="cuda": buffers are staged through torch tensors on
the GPU and the call goes through exactly the pointers/stream the product passes)."""
import ctypes as C

import numpy as np

from swapping_autoencoder_pytorch_amd.hip_lib import ConvDesc, ConvMod

OPS = ("conv2d_fwd_f32", "conv2d_dgrad_f32", "conv2d_wgrad_f32")


class _Buf:
    def __init__(self, arr, device):
        self.device = device
        if device is None:
            self.np = np.ascontiguousarray(arr, dtype=np.float32)
            self.ptr = self.np.ctypes.data
        else:
            import torch
            self.t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(device)
            self.ptr = self.t.data_ptr()

    def numpy(self):
        if self.device is None:
            return self.np
        return self.t.cpu().numpy()


def _stream(device):
    if device is None:
        return None
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def _out(shape, device):
    return _Buf(np.full(shape, np.nan, np.float32), device)


def upfirdn2d(lib, x, k, up=(1, 1), down=(1, 1), pad=(0, 0, 0, 0), device=None):
    major, ih, iw, minor = x.shape
    kh, kw = k.shape
    oh = (ih * up[1] + pad[2] + pad[3] - kh + down[1]) // down[1]
    ow = (iw * up[0] + pad[0] + pad[1] - kw + down[0]) // down[0]
    bx, bk, by = _Buf(x, device), _Buf(k, device), _out((major, oh, ow, minor), device)
    lib.call("upfirdn2d_f32", bx.ptr, bk.ptr, by.ptr, major, ih, iw, minor, kh, kw, up[0], up[1], down[0], down[1],
             pad[0], pad[1], pad[2], pad[3], _stream(device))
    return by.numpy()


def upfirdn2d_epilogue(lib, x, k, up, pad, y_old=None, act_ref=None, channels=1, slope=0.2, scale=2 ** 0.5, device=None):
    """sae_upfirdn2d_epilogue_f32 on planes x [major, ih, iw]: returns (y, gb or None).  y_old: accumulate into it."""
    major, ih, iw = x.shape
    kh, kw = k.shape
    oh = ih * up + pad[2] + pad[3] - kh + 1
    ow = iw * up + pad[0] + pad[1] - kw + 1
    bx, bk = _Buf(x, device), _Buf(k, device)
    by = _Buf(y_old, device) if y_old is not None else _out((major, oh, ow), device)
    br = _Buf(act_ref, device) if act_ref is not None else None
    n = lib.query("upfirdn2d_epilogue_workspace", major, oh, ow, channels, up) if act_ref is not None else 0
    bgb, ws = _out((channels,), device), _out((max(n, 1),), device)
    lib.call("upfirdn2d_epilogue_f32", bx.ptr, bk.ptr, by.ptr, major, ih, iw, kh, kw, up, pad[0], pad[1], pad[2], pad[3],
             br.ptr if br else None, slope, scale, bgb.ptr if br else None, channels, 1 if y_old is not None else 0,
             ws.ptr if br else None, n, _stream(device))
    return by.numpy(), (bgb.numpy() if br else None)


def upfirdn2d_noise_bias_act(lib, x, k, pad, noise, noise_weight, bias, channels, slope=0.2, scale=2 ** 0.5, device=None):
    """sae_upfirdn2d_noise_bias_act_f32 on planes x [major, ih, iw]; noise [major / channels, oh, ow] or None"""
    major, ih, iw = x.shape
    kh, kw = k.shape
    oh = ih + pad[2] + pad[3] - kh + 1
    ow = iw + pad[0] + pad[1] - kw + 1
    bx, bk, by = _Buf(x, device), _Buf(k, device), _out((major, oh, ow), device)
    keep = [(_Buf(v, device) if v is not None else None) for v in (noise, noise_weight, bias)]
    ptrs = [(b.ptr if b is not None else None) for b in keep]
    lib.call("upfirdn2d_noise_bias_act_f32", bx.ptr, bk.ptr, by.ptr, major, ih, iw, kh, kw, pad[0], pad[1], pad[2], pad[3],
             ptrs[0], ptrs[1], ptrs[2], channels, slope, scale, _stream(device))
    return by.numpy()


def bias_act(lib, x, b, ref, act=3, grad=0, alpha=0.2, scale=2 ** 0.5, device=None):
    step_b = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    bx = _Buf(x, device)
    bb = _Buf(b, device) if b is not None else None
    br = _Buf(ref, device) if ref is not None else None
    by = _out(x.shape, device)
    lib.call("bias_act_f32", bx.ptr, bb.ptr if bb else None, br.ptr if br else None, by.ptr, x.size, step_b,
             b.size if b is not None else 1, act, grad, alpha, scale, _stream(device))
    return by.numpy()


def bias_act_bwd(lib, gy, y, alpha=0.2, scale=2 ** 0.5, device=None):
    step_b = int(np.prod(gy.shape[2:])) if gy.ndim > 2 else 1
    size_b = gy.shape[1]
    n = lib.query("bias_act_bwd_workspace", gy.size, step_b, size_b)
    bg, by = _Buf(gy, device), _Buf(y, device)
    bgx, bgb, ws = _out(gy.shape, device), _out((size_b,), device), _out((max(n, 1),), device)
    lib.call("bias_act_bwd_f32", bg.ptr, by.ptr, bgx.ptr, bgb.ptr, ws.ptr, n, gy.size, step_b, size_b, alpha, scale,
             _stream(device))
    return bgx.numpy(), bgb.numpy()


def conv_desc(n, c, h, w, m, k, stride, pad, w_cm_layout=False):
    d = ConvDesc()
    d.n, d.c, d.h, d.w, d.m = n, c, h, w, m
    d.kh = d.kw = k
    d.stride, d.pad = stride, pad
    d.oh = (h + 2 * pad - k) // stride + 1
    d.ow = (w + 2 * pad - k) // stride + 1
    if w_cm_layout:   # parameter stored [C, M, k, k]
        d.w_stride_m, d.w_stride_c = k * k, m * k * k
    else:             # parameter stored [M, C, k, k]
        d.w_stride_m, d.w_stride_c = c * k * k, k * k
    return d


def conv(lib, op, d, a, b, out_shape, alpha=1.0, device=None):
    n = lib.query("conv2d_workspace", C.byref(d), op)
    ba, bb, bo, ws = _Buf(a, device), _Buf(b, device), _out(out_shape, device), _out((max(n, 1),), device)
    lib.call(OPS[op], ba.ptr, bb.ptr, bo.ptr, C.byref(d), alpha, ws.ptr, n, _stream(device))
    return bo.numpy()


def conv_prepped(lib, op, d, a, b, out_shape, alpha=1.0, device=None, poison_workspace=True, wrong_layout=False):
    """The same conv through PREPARED weights (include/sae_hip.h: sae_conv2d_wprep_query / sae_conv2d_wprep_f32 and the
    descriptor's prepped fields).  Returns (result, floats of the layout).  poison_workspace: the workspace is filled with
    NaN first, so a launch that re-laid the weights there anyway would still be right, but one that READ the workspace
    instead of the prepared buffer would not.  wrong_layout: hand the buffer over under a foreign layout identity -- the
    library must ignore it and re-lay."""
    floats, layout = C.c_int64(0), C.c_int64(0)
    lib.call("conv2d_wprep_query", C.byref(d), None, op, C.byref(floats), C.byref(layout))
    n = lib.query("conv2d_workspace", C.byref(d), op)
    ba, bb, bo = _Buf(a, device), _Buf(b, device), _out(out_shape, device)
    wsn = np.full((max(n, 1),), np.nan if (poison_workspace and not wrong_layout) else 0.0, np.float32)
    ws = _Buf(wsn, device)
    if floats.value == 0:
        lib.call(OPS[op], ba.ptr, bb.ptr, bo.ptr, C.byref(d), alpha, ws.ptr, n, _stream(device))
        return bo.numpy(), 0
    wp = _out((floats.value,), device)
    lib.call("conv2d_wprep_f32", bb.ptr, C.byref(d), None, op, alpha, wp.ptr, floats.value, _stream(device))
    d2 = type(d)()
    C.memmove(C.byref(d2), C.byref(d), C.sizeof(d))
    d2.prepped, d2.prepped_floats, d2.prepped_layout = wp.ptr, floats.value, layout.value + (2 if wrong_layout else 0)
    lib.call(OPS[op], ba.ptr, bb.ptr, bo.ptr, C.byref(d2), alpha, ws.ptr, n, _stream(device))
    return bo.numpy(), floats.value


def conv_bias_act(lib, d, x, w, bias, alpha=1.0, slope=0.2, scale=2 ** 0.5, device=None):
    n = lib.query("conv2d_workspace", C.byref(d), 0)
    bx, bw = _Buf(x, device), _Buf(w, device)
    bb = _Buf(bias, device) if bias is not None else None
    bo, ws = _out((d.n, d.m, d.oh, d.ow), device), _out((max(n, 1),), device)
    lib.call("conv2d_fwd_bias_act_f32", bx.ptr, bw.ptr, bb.ptr if bb else None, bo.ptr, C.byref(d), alpha, slope, scale,
             ws.ptr, n, _stream(device))
    return bo.numpy()


def conv_residual(lib, d, x, w, residual, alpha=1.0, res_scale=0.5, device=None):
    nws = lib.query("conv2d_workspace", C.byref(d), 0)
    bx, bw, br = _Buf(x, device), _Buf(w, device), _Buf(residual, device)
    by, ws = _out(residual.shape, device), _out((max(nws, 1),), device)
    lib.call("conv2d_fwd_residual_f32", bx.ptr, bw.ptr, br.ptr, by.ptr, C.byref(d), alpha, res_scale, ws.ptr, nws, _stream(device))
    return by.numpy()


def gemm(lib, a, b, bias, m, n, k, a_si, a_sk, b_sk, b_sj, alpha=1.0, device=None, split=False):
    """sae_gemm_f32, or (split=True) sae_gemm_ws_f32 with the workspace sae_gemm_workspace asks for -> (result, slices' floats)"""
    ba, bb = _Buf(a, device), _Buf(b, device)
    bbias = _Buf(bias, device) if bias is not None else None
    bc = _out((m, n), device)
    if split:
        n_ws = lib.query("gemm_workspace", m, n, k)
        ws = _out((max(n_ws, 1),), device)
        lib.call("gemm_ws_f32", ba.ptr, bb.ptr, bbias.ptr if bbias else None, bc.ptr, m, n, k, a_si, a_sk, b_sk, b_sj, n,
                 alpha, ws.ptr if n_ws else None, n_ws, _stream(device))
        return bc.numpy(), n_ws
    lib.call("gemm_f32", ba.ptr, bb.ptr, bbias.ptr if bbias else None, bc.ptr, m, n, k, a_si, a_sk, b_sk, b_sj, n,
             alpha, _stream(device))
    return bc.numpy()


def upsample2x_add(lib, x, res, alpha=1.0, device=None):
    planes, h, w = x.shape
    bx = _Buf(x, device)
    br = _Buf(res, device) if res is not None else None
    by = _out((planes, 2 * h, 2 * w), device)
    lib.call("upsample2x_bilinear_add_f32", bx.ptr, br.ptr if br else None, by.ptr, planes, h, w, alpha, _stream(device))
    return by.numpy()


def upsample2x_bwd(lib, gy, alpha=1.0, device=None):
    planes, oh, ow = gy.shape
    bg, bo = _Buf(gy, device), _out((planes, oh // 2, ow // 2), device)
    lib.call("upsample2x_bilinear_bwd_f32", bg.ptr, bo.ptr, planes, oh // 2, ow // 2, alpha, _stream(device))
    return bo.numpy()


def noise_bias_act(lib, x, noise, nw, bias, alpha=0.2, scale=2 ** 0.5, device=None):
    n, c = x.shape[:2]
    hw = int(np.prod(x.shape[2:]))
    bx, bn, bw, bb, bo = (_Buf(x, device), _Buf(noise, device) if noise is not None else None, _Buf(nw, device),
                          _Buf(bias, device) if bias is not None else None, _out(x.shape, device))
    lib.call("noise_bias_act_f32", bx.ptr, bn.ptr if bn else None, bw.ptr, bb.ptr if bb else None, bo.ptr, n, c, hw,
             alpha, scale, _stream(device))
    return bo.numpy()


def noise_bias_act_bwd(lib, gy, y, noise, alpha=0.2, scale=2 ** 0.5, device=None):
    n, c = gy.shape[:2]
    hw = int(np.prod(gy.shape[2:]))
    nws = lib.query("noise_bias_act_bwd_workspace", n, c, hw)
    bg, by, bn = _Buf(gy, device), _Buf(y, device), _Buf(noise, device) if noise is not None else None
    bgx, bgb, bgw, bws = _out(gy.shape, device), _out((c,), device), _out((1,), device), _out((max(nws, 1),), device)
    lib.call("noise_bias_act_bwd_f32", bg.ptr, by.ptr, bn.ptr if bn else None, bgx.ptr, bgb.ptr, bgw.ptr, bws.ptr, nws,
             n, c, hw, alpha, scale, _stream(device))
    return bgx.numpy(), bgb.numpy(), bgw.numpy()


def plane_scale_dot(lib, g, x, s, device=None):
    planes = int(np.prod(g.shape[:2]))
    hw = int(np.prod(g.shape[2:]))
    bg, bx, bs = _Buf(g, device), _Buf(x, device), _Buf(s, device)
    bgx, bgs = _out(g.shape, device), _out(s.shape, device)
    lib.call("plane_scale_dot_f32", bg.ptr, bx.ptr, bs.ptr, bgx.ptr, bgs.ptr, planes, hw, _stream(device))
    return bgx.numpy(), bgs.numpy()


def plane_scale_dot_act(lib, g, x, s, noise, alpha=0.2, scale=2 ** 0.5, device=None):
    n, c = g.shape[:2]
    hw = int(np.prod(g.shape[2:]))
    nws = lib.query("plane_scale_dot_act_workspace", n, c)
    bg, bx, bs = _Buf(g, device), _Buf(x, device), _Buf(s, device)
    bn = _Buf(noise, device) if noise is not None else None
    bgx, bgs, bgb, bgw, bws = (_out(g.shape, device), _out(s.shape, device), _out((c,), device), _out((1,), device),
                               _out((max(nws, 1),), device))
    lib.call("plane_scale_dot_act_f32", bg.ptr, bx.ptr, bs.ptr, bn.ptr if bn else None, bgx.ptr, bgs.ptr, bgb.ptr, bgw.ptr, bws.ptr,
             nws, n, c, hw, alpha, scale, _stream(device))
    return bgx.numpy(), bgs.numpy(), bgb.numpy(), bgw.numpy()


def weight_demod(lib, w, alpha, eps=1e-8, device=None):
    rows, cols = w.shape[0], int(np.prod(w.shape[1:]))
    bw, bd = _Buf(w, device), _out((rows,), device)
    lib.call("weight_demod_f32", bw.ptr, bd.ptr, rows, cols, alpha, eps, _stream(device))
    return bd.numpy()


def weight_demod_bwd(lib, geff, w, d, alpha, device=None):
    rows, cols = w.shape[0], int(np.prod(w.shape[1:]))
    bg, bw, bd, bo = _Buf(geff, device), _Buf(w, device), _Buf(d, device), _out(w.shape, device)
    lib.call("weight_demod_bwd_f32", bg.ptr, bw.ptr, bd.ptr, bo.ptr, rows, cols, alpha, _stream(device))
    return bo.numpy()


def random_crop(lib, x, params, size, crops, device=None):
    n, c, h, w = x.shape
    import torch
    lin = torch.linspace(-1.0, 1.0, size).numpy()      # the grid the reference builds (util/util.py:330)
    bx, bp, bl, bo = _Buf(x, device), _Buf(params, device), _Buf(lin, device), _out((n * crops, c, size, size), device)
    lib.call("random_crop_f32", bx.ptr, bp.ptr, bl.ptr, bo.ptr, n, c, h, w, crops, size, _stream(device))
    return bo.numpy()


def random_crop_bwd(lib, gy, params, shape, crops, device=None):
    n, c, h, w = shape
    size = gy.shape[-1]
    import torch
    lin = torch.linspace(-1.0, 1.0, size).numpy()
    bg, bp, bl, bo = _Buf(gy, device), _Buf(params, device), _Buf(lin, device), _out(shape, device)
    lib.call("random_crop_bwd_f32", bg.ptr, bp.ptr, bl.ptr, bo.ptr, n, c, h, w, crops, size, _stream(device))
    return bo.numpy()


def reflect_pad(lib, x, pads, device=None):
    l, r, t, b = pads
    planes = int(np.prod(x.shape[:-2]))
    h, w = x.shape[-2:]
    bx, bo = _Buf(x, device), _out(x.shape[:-2] + (h + t + b, w + l + r), device)
    lib.call("reflect_pad_f32", bx.ptr, bo.ptr, planes, h, w, l, r, t, b, _stream(device))
    return bo.numpy()


def reflect_pad_adj(lib, gy, pads, device=None):
    l, r, t, b = pads
    planes = int(np.prod(gy.shape[:-2]))
    h, w = gy.shape[-2] - t - b, gy.shape[-1] - l - r
    bg, bo = _Buf(gy, device), _out(gy.shape[:-2] + (h, w), device)
    lib.call("reflect_pad_adj_f32", bg.ptr, bo.ptr, planes, h, w, l, r, t, b, _stream(device))
    return bo.numpy()


def rel_err(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(float(np.abs(b).max()), 1e-30))


def adam_multi(lib, params, grads, ms, vs, steps, lr, beta1, beta2, eps, grad_scale=1.0, device=None, offset_elems=0,
               dev_steps=False):
    """Runs sae_adam_multi_f32 on lists of 1-D numpy arrays; returns updated (params, ms, vs).  offset_elems > 0
    places every tensor that many floats into its buffer (pointers not 16-byte aligned).  dev_steps: the same update through
    sae_adam_multi_dev_f32 -- the counts (steps - 1, i.e. BEFORE the update) staged in device memory; returns the counts after
    the call as a fourth value."""
    n = len(params)

    def stage(arrs):
        bufs = [_Buf(np.concatenate([np.zeros(offset_elems, np.float32), a]), device) for a in arrs]
        return bufs, [b.ptr + 4 * offset_elems for b in bufs]

    bp, pp = stage(params)
    bg, pg = stage(grads)
    bm, pm = stage(ms)
    bv, pv = stage(vs)
    arr = lambda xs: (C.c_void_p * n)(*xs)
    numel = (C.c_int64 * n)(*[a.size for a in params])
    cut = lambda bufs: [b.numpy()[offset_elems:] for b in bufs]
    if dev_steps:
        before = np.asarray([t - 1 for t in steps], np.int64)
        if device is None:
            counts = before.copy()
            base = counts.ctypes.data
        else:
            import torch
            counts = torch.from_numpy(before).to(device)
            base = counts.data_ptr()
        lib.call("adam_multi_dev_f32", arr(pp), arr(pg), arr(pm), arr(pv), numel, arr([base + 8 * i for i in range(n)]), n, lr,
                 beta1, beta2, eps, grad_scale, _stream(device))
        after = counts if device is None else counts.cpu().numpy()
        return cut(bp), cut(bm), cut(bv), [int(v) for v in after]
    st = (C.c_int64 * n)(*steps)
    lib.call("adam_multi_f32", arr(pp), arr(pg), arr(pm), arr(pv), numel, st, n, lr, beta1, beta2, eps, grad_scale,
             _stream(device))
    return cut(bp), cut(bm), cut(bv)


MOD_OPS = ("modconv2d_fwd_f32", "modconv2d_dgrad_f32", "modconv2d_wgrad_f32")


def modconv(lib, op, d, a, b, out_shape, x_scale=None, y_scale=None, wm_scale=None, wc_scale=None, alpha=1.0, device=None):
    """sae_modconv2d_{fwd,dgrad,wgrad}_f32 on numpy operands; the factor arrays are optional."""
    n = lib.query("conv2d_workspace", C.byref(d), op)
    ba, bb, bo, ws = _Buf(a, device), _Buf(b, device), _out(out_shape, device), _out((max(n, 1),), device)
    keep = [(_Buf(v, device) if v is not None else None) for v in (x_scale, y_scale, wm_scale, wc_scale)]
    mod = ConvMod(*[(k.ptr if k is not None else None) for k in keep])
    lib.call(MOD_OPS[op], ba.ptr, bb.ptr, bo.ptr, C.byref(d), C.byref(mod), alpha, ws.ptr, n, _stream(device))
    return bo.numpy()


def modconv_noise_bias_act(lib, d, x, w, x_scale, wm_scale, noise, noise_weight, bias, alpha=1.0, slope=0.2, scale=2 ** 0.5,
                           device=None):
    """sae_modconv2d_fwd_noise_bias_act_f32: StyledConv's plain form in one call"""
    n = lib.query("conv2d_workspace", C.byref(d), 0)
    out_shape = (d.n, d.m, d.oh, d.ow)
    bx, bw, bo, ws = _Buf(x, device), _Buf(w, device), _out(out_shape, device), _out((max(n, 1),), device)
    keep = [(_Buf(v, device) if v is not None else None) for v in (x_scale, None, wm_scale, None, noise, noise_weight, bias)]
    mod = ConvMod(*[(k.ptr if k is not None else None) for k in keep[:4]])
    opt = [(k.ptr if k is not None else None) for k in keep[4:]]
    lib.call("modconv2d_fwd_noise_bias_act_f32", bx.ptr, bw.ptr, opt[0], opt[1], opt[2], bo.ptr, C.byref(d), C.byref(mod), alpha,
             slope, scale, ws.ptr, n, _stream(device))
    return bo.numpy()


def l2_normalize(lib, x, eps=1e-8, device=None):
    outer, ch = x.shape[:2]
    inner = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    bx, by = _Buf(x, device), _out(x.shape, device)
    lib.call("l2_normalize_f32", bx.ptr, by.ptr, outer, ch, inner, eps, _stream(device))
    return by.numpy()


def l2_normalize_bwd(lib, gy, x, eps=1e-8, device=None):
    outer, ch = x.shape[:2]
    inner = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    bg, bx, bo = _Buf(gy, device), _Buf(x, device), _out(x.shape, device)
    lib.call("l2_normalize_bwd_f32", bg.ptr, bx.ptr, bo.ptr, outer, ch, inner, eps, _stream(device))
    return bo.numpy()


def plane_affine(lib, x, a, b, device=None):
    planes, hw = int(np.prod(x.shape[:2])), int(np.prod(x.shape[2:]))
    bx, ba, bb, by = _Buf(x, device), _Buf(a, device), _Buf(b, device), _out(x.shape, device)
    lib.call("plane_affine_f32", bx.ptr, ba.ptr, bb.ptr, by.ptr, planes, hw, _stream(device))
    return by.numpy()


def plane_affine_bwd(lib, g, x, a, device=None):
    planes, hw = int(np.prod(x.shape[:2])), int(np.prod(x.shape[2:]))
    bg, bx, ba = _Buf(g, device), _Buf(x, device), _Buf(a, device)
    bgx, bga, bgb = _out(x.shape, device), _out(a.shape, device), _out(a.shape, device)
    lib.call("plane_affine_bwd_f32", bg.ptr, bx.ptr, ba.ptr, bgx.ptr, bga.ptr, bgb.ptr, planes, hw, _stream(device))
    return bgx.numpy(), bga.numpy(), bgb.numpy()


def softplus_mean(lib, x, sign, device=None):
    batch, inner = x.shape[0], int(np.prod(x.shape[1:]))
    bx, by = _Buf(x, device), _out((batch,), device)
    lib.call("softplus_mean_f32", bx.ptr, by.ptr, batch, inner, sign, _stream(device))
    return by.numpy()


def softplus_mean_bwd(lib, gy, x, sign, device=None):
    batch, inner = x.shape[0], int(np.prod(x.shape[1:]))
    bg, bx, bo = _Buf(gy, device), _Buf(x, device), _out(x.shape, device)
    lib.call("softplus_mean_bwd_f32", bg.ptr, bx.ptr, bo.ptr, batch, inner, sign, _stream(device))
    return bo.numpy()


# ---- Winograd F(2x2, 3x3) transforms (sae_wino_*)
def wino_weights(lib, w, m, c, sm, sc, flip=False, alpha=1.0, row_scale=None, col_scale=None, device=None):
    bw, bu = _Buf(w, device), _out((16, m, c), device)
    br = _Buf(row_scale, device) if row_scale is not None else None
    bc = _Buf(col_scale, device) if col_scale is not None else None
    lib.call("wino_weights_f32", bw.ptr, br.ptr if br else None, bc.ptr if bc else None, bu.ptr, m, c, sm, sc, 1 if flip else 0,
             alpha, _stream(device))
    return bu.numpy()


def wino_input(lib, x, plane_scale=None, pad=1, device=None):
    planes, h, w = x.shape
    bx, bv = _Buf(x, device), _out((16, planes, (h + 2 * pad - 2) // 2, (w + 2 * pad - 2) // 2), device)
    bs = _Buf(plane_scale, device) if plane_scale is not None else None
    lib.call("wino_input_f32", bx.ptr, bs.ptr if bs else None, bv.ptr, planes, h, w, pad, _stream(device))
    return bv.numpy()


def wino_output(lib, md, h, w, channels, bias=None, act=None, plane_scale=None, noise=None, noise_weight=None, device=None):
    planes = md.shape[1]
    bm, by = _Buf(md, device), _out((planes, h, w), device)
    opt = [(_Buf(t, device) if t is not None else None) for t in (plane_scale, noise, noise_weight, bias)]
    slope, scale = act if act is not None else (0.0, 1.0)
    lib.call("wino_output_f32", bm.ptr, *[(b.ptr if b else None) for b in opt], by.ptr, planes, channels, h, w,
             1 if act is not None else 0, slope, scale, _stream(device))
    return by.numpy()


def wino_conv(lib, x, wt, alpha=1.0, transpose=False, bias=None, act=None, x_scale=None, cm_layout=False, row_scale=None,
              col_scale=None, out_scale=None, noise=None, noise_weight=None, pad=1, device=None):
    """The whole route on numpy data: alpha * conv3x3(x * x_scale, wt * row_scale[m] * col_scale[c]) (pad 1) or, transpose=True,
    its data gradient for x = gy (row_scale / col_scale then name the axes of THAT product: rows = its outputs), times
    out_scale per output plane, then the optional noise + bias + leaky-ReLU epilogue."""
    n, cin, ih, iw = x.shape
    ipad = 2 - pad if transpose else pad           # `pad` is the LAYER's; its data gradient pads the output gradient by 2 - pad
    h, w = ih + 2 * ipad - 2, iw + 2 * ipad - 2
    if cm_layout:
        c_, m_ = wt.shape[0], wt.shape[1]
        sm, sc = 9, m_ * 9
    else:
        m_, c_ = wt.shape[0], wt.shape[1]
        sm, sc = c_ * 9, 9
    if transpose:
        cout, sm, sc = c_, sc, sm
        assert cin == m_
    else:
        cout = m_
        assert cin == c_
    u = wino_weights(lib, wt, cout, cin, sm, sc, flip=transpose, alpha=alpha, row_scale=row_scale, col_scale=col_scale, device=device)
    v = wino_input(lib, x.reshape(n * cin, ih, iw), None if x_scale is None else x_scale.reshape(-1), pad=ipad, device=device)
    th, tw = h // 2, w // 2
    n_ws = lib.query("wino_gemm_workspace", n, cin, cout, th, tw)
    bv, bu, bm, ws = _Buf(v, device), _Buf(u, device), _out((16, n * cout, th, tw), device), _out((max(n_ws, 1),), device)
    lib.call("wino_gemm_f32", bv.ptr, bu.ptr, bm.ptr, n, cin, cout, th, tw, ws.ptr, n_ws, _stream(device))
    md = bm.numpy()
    return wino_output(lib, md, h, w, cout, bias=bias, act=act, plane_scale=None if out_scale is None else out_scale.reshape(-1),
                       noise=noise, noise_weight=noise_weight, device=device).reshape(n, cout, h, w)


def wino_wgrad(lib, x, gy, alpha=1.0, cm_layout=False, x_scale=None, y_scale=None, pad=1, device=None):
    """The weight gradient on the route: alpha * sum (gy * y_scale) (x) (x * x_scale) -> [m, c, 3, 3] ([c, m, 3, 3] if cm_layout)."""
    n, c, ih, iw = x.shape
    m, h, w = gy.shape[1], gy.shape[2], gy.shape[3]
    th, tw = h // 2, w // 2
    v = wino_input(lib, x.reshape(n * c, ih, iw), None if x_scale is None else x_scale.reshape(-1), pad=pad, device=device)
    bg, be = _Buf(gy.reshape(n * m, h, w), device), _out((16, n * m, th, tw), device)
    bs = _Buf(y_scale.reshape(-1), device) if y_scale is not None else None
    lib.call("wino_gy_f32", bg.ptr, bs.ptr if bs else None, be.ptr, n * m, h, w, _stream(device))
    n_ws = lib.query("wino_wgrad_gemm_workspace", n, c, m, th, tw)
    bv, bu, ws = _Buf(v, device), _out((16, m, c), device), _out((max(n_ws, 1),), device)
    lib.call("wino_wgrad_gemm_f32", bv.ptr, be.ptr, bu.ptr, n, c, m, th, tw, ws.ptr, n_ws, _stream(device))
    shape, sm, sc = ((c, m, 3, 3), 9, m * 9) if cm_layout else ((m, c, 3, 3), c * 9, 9)
    bw = _out(shape, device)
    lib.call("wino_wgrad_output_f32", bu.ptr, bw.ptr, m, c, sm, sc, alpha, _stream(device))
    return bw.numpy(), be.numpy(), bu.numpy()


def wino_fused_conv(lib, x, wt, alpha=1.0, transpose=False, bias=None, act=None, x_scale=None, cm_layout=False, row_scale=None,
                    col_scale=None, out_scale=None, noise=None, noise_weight=None, pad=1, device=None):
    """wino_conv's contract through the ONE-kernel route (sae_wino_fused_weights_f32 + sae_wino_fused_conv_f32)."""
    n, cin, ih, iw = x.shape
    ipad = 2 - pad if transpose else pad
    h, w = ih + 2 * ipad - 2, iw + 2 * ipad - 2
    if cm_layout:
        c_, m_ = wt.shape[0], wt.shape[1]
        sm, sc = 9, m_ * 9
    else:
        m_, c_ = wt.shape[0], wt.shape[1]
        sm, sc = c_ * 9, 9
    if transpose:
        cout, sm, sc = c_, sc, sm
        assert cin == m_
    else:
        cout = m_
        assert cin == c_
    floats = lib.query("wino_fused_weights_floats", cout, cin)
    bw, bu = _Buf(wt, device), _out((floats,), device)
    opt = [None if t is None else _Buf(np.asarray(t, np.float32).reshape(-1), device)
           for t in (row_scale, col_scale, x_scale, out_scale, noise, noise_weight, bias)]
    prs, pcs, pxs, pos, pnz, pnw, pb = [b.ptr if b is not None else None for b in opt]
    lib.call("wino_fused_weights_f32", bw.ptr, prs, pcs, bu.ptr, cout, cin, sm, sc, 1 if transpose else 0, alpha, _stream(device))
    bx, by = _Buf(x, device), _out((n, cout, h, w), device)
    slope, scale = act if act is not None else (0.0, 1.0)
    lib.call("wino_fused_conv_f32", bx.ptr, pxs, bu.ptr, pos, pnz, pnw, pb, by.ptr, n, cin, cout, ih, iw, ipad,
             1 if act is not None else 0, float(slope), float(scale), _stream(device))
    return by.numpy()


def wino_fused_wgrad(lib, x, gy, alpha=1.0, cm_layout=False, x_scale=None, y_scale=None, pad=1, device=None):
    """wino_wgrad's contract through sae_wino_fused_wgrad_f32 (one kernel + the slice reduction)."""
    n, c, ih, iw = x.shape
    m = gy.shape[1]
    shape, sm, sc = ((c, m, 3, 3), 9, m * 9) if cm_layout else ((m, c, 3, 3), c * 9, 9)
    n_ws = lib.query("wino_fused_wgrad_workspace", n, c, m, ih, iw, pad)
    bx, bg, bw, ws = _Buf(x, device), _Buf(gy, device), _out(shape, device), _out((max(n_ws, 1),), device)
    bxs = _Buf(x_scale.reshape(-1), device) if x_scale is not None else None
    bys = _Buf(y_scale.reshape(-1), device) if y_scale is not None else None
    lib.call("wino_fused_wgrad_f32", bx.ptr, bxs.ptr if bxs else None, bg.ptr, bys.ptr if bys else None, bw.ptr, n, c, m, ih, iw, pad,
             sm, sc, alpha, ws.ptr, n_ws, _stream(device))
    return bw.numpy()


def s2wino_dgrad(lib, gy, wt, alpha=1.0, cm_layout=False, g_scale=None, row_scale=None, col_scale=None, out_scale=None, device=None):
    """The data gradient / transposed convolution of a 3x3 stride-2 pad-0 layer through sae_s2wino_weights_f32 (flip = 1) +
    sae_s2wino_dgrad_f32: gy [n][m][h][w] -> dx [n][c][2h+1][2w+1]; wt is the layer's weight, [m][c][3][3] or ([C, M] layout)
    [c][m][3][3]; row_scale [c] / col_scale [m] follow the product's output / contraction axes."""
    n, m, h, w = gy.shape
    if cm_layout:
        c, m2 = wt.shape[0], wt.shape[1]
        s_out, s_in = m2 * 9, 9
    else:
        m2, c = wt.shape[0], wt.shape[1]
        s_out, s_in = 9, c * 9
    assert m2 == m
    floats = lib.query("s2wino_weights_floats", c, m)
    bw, bu = _Buf(wt, device), _out((floats,), device)
    opt = [None if t is None else _Buf(np.asarray(t, np.float32).reshape(-1), device) for t in (row_scale, col_scale, g_scale, out_scale)]
    prs, pcs, pgs, pos = [b.ptr if b is not None else None for b in opt]
    lib.call("s2wino_weights_f32", bw.ptr, prs, pcs, bu.ptr, c, m, s_out, s_in, 1, alpha, _stream(device))
    bg, bo = _Buf(gy, device), _out((n, c, 2 * h + 1, 2 * w + 1), device)
    lib.call("s2wino_dgrad_f32", bg.ptr, pgs, bu.ptr, pos, bo.ptr, n, m, c, h, w, _stream(device))
    return bo.numpy()
