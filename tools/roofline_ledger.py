"""Per-kernel-class roofline table of the train step, measured live.

Every C-ABI call of `--steps` timed iterations is bracketed with HIP events on the launch stream and its
ALGORITHMIC work is computed from its own arguments (convs / linears 2*MAC FLOPs; K1
4*(numel_in + numel_out) bytes; K2 8*numel forward, 12*numel backward; other elementwise ops their tensors once).
Rows are grouped by kernel class (entry point x kernel size x stride, i.e. the kernel template the dispatcher
picks), with the time, work and achieved rate per iteration and the fraction of the bound's peak
(fp32 MFMA 157.3 TFLOP/s, HBM 8 TB/s).  "outside the C-ABI" = wall time of the iteration minus the bracketed
time: ATen glue, Adam, allocator, launch gaps.

    python tools/roofline_ledger.py --preset church256 [--conv-math f32] > profiles/r2_roofline_by_kernel_church256.txt
"""
import argparse
import collections
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from swapping_autoencoder_pytorch_amd import hip_lib  # noqa: E402

if os.environ.get("KB_LIBRARY"):        # A/B of dispatch knobs: a tuning build (tests/tuning, tools/variants) instead of the product library
    hip_lib._LIB = hip_lib.SaeLibrary(os.path.abspath(os.environ["KB_LIBRARY"]))

MFMA_PEAK, HBM_PEAK = 157.3e12, 8.0e12
BY_SHAPE = False     # --by-shape: one row per conv geometry, with the time above 0.85 of the MFMA peak


def _desc(arg):
    return arg._obj


def work_of(name, a):
    """(class label, bound, algorithmic work: FLOPs for 'mfma', bytes for 'hbm')"""
    if name == "conv2d_wprep_query":
        return name, "hbm", 0.0
    if name == "conv2d_wprep_f32":        # prepared weights (stylegan2_op/weight_prep.py): the layout is written once, the parameter read once
        return "weight re-layout kept across launches (sae_conv2d_wprep_f32)", "hbm", 8.0 * a[6]
    if (name.startswith("conv2d_") and name != "conv2d_workspace") or name.startswith("modconv2d_"):
        d = _desc(a[6] if name == "modconv2d_fwd_noise_bias_act_f32" else
                  a[4] if name in ("conv2d_fwd_bias_act_f32", "conv2d_fwd_residual_f32") else a[3])
        op = {"conv2d_fwd_f32": "fwd", "conv2d_fwd_bias_act_f32": "fwd+bias+lrelu", "conv2d_fwd_residual_f32": "fwd+residual",
              "conv2d_dgrad_f32": "dgrad",
              "conv2d_wgrad_f32": "wgrad", "modconv2d_fwd_f32": "fwd (modulated)", "modconv2d_fwd_noise_bias_act_f32": "fwd (modulated)", "modconv2d_dgrad_f32": "dgrad (modulated)",
              "modconv2d_wgrad_f32": "wgrad (modulated)"}[name]
        width = "narrow(<=64ch)" if max(d.m if op.startswith("fwd") else d.c, 1) <= 64 else "wide"
        if op.startswith("wgrad"):
            width = "narrow(<=32ch)" if (d.m <= 32 and d.c <= 32) else "wide"
        label = "conv %dx%d s%d %-17s %s" % (d.kh, d.kw, d.stride, op, width)
        if BY_SHAPE:
            label = "conv %dx%d s%d %-17s n%-3d %4d->%-4d %3dx%-3d" % (d.kh, d.kw, d.stride, op, d.n, d.c, d.m, d.h, d.w)
        return label, "mfma", 2.0 * d.n * d.m * d.oh * d.ow * d.c * d.kh * d.kw
    if name == "adam_multi_f32":
        n = a[6]
        numel = sum(a[4][i] for i in range(n))
        return "adam (multi-tensor)", "hbm", 28.0 * numel
    if name in ("gemm_f32", "gemm_ws_f32"):
        m, n, k = a[4], a[5], a[6]
        if BY_SHAPE:
            return "gemm (linear layers) m%-5d n%-5d k%-5d" % (m, n, k), "mfma", 2.0 * m * n * k
        return "gemm (linear layers)", "mfma", 2.0 * m * n * k
    if name == "upfirdn2d_epilogue_f32":
        major, ih, iw, kh, kw, up, px0, px1, py0, py1 = a[3:13]
        oh, ow = ih * up + py0 + py1 - kh + 1, iw * up + px0 + px1 - kw + 1
        act, acc = bool(a[13]), bool(a[18])
        kind = ("blur" if up == 1 else "zero-insert x2") + (" + accumulate" if acc else "") + (" + act backward" if act else "")
        # algorithmic bytes: the FIR's input and output once, plus one read per fused operand (old y, activation reference)
        return "upfirdn2d %s (fused epilogue)" % kind, "hbm", 4.0 * major * (ih * iw + oh * ow * (1 + int(act) + int(acc)))
    if name == "upfirdn2d_noise_bias_act_f32":
        major, ih, iw, kh, kw, px0, px1, py0, py1 = a[3:12]
        oh, ow = ih + py0 + py1 - kh + 1, iw + px0 + px1 - kw + 1
        return "upfirdn2d blur + noise + bias + lrelu forward (fused epilogue)", "hbm", 4.0 * major * (ih * iw + oh * ow)
    if name == "plane_scale_dot_act_f32":
        return "style modulation backward + activation backward (fused)", "hbm", 4.0 * a[10] * a[12] * (3 * a[11] + 1)
    if name == "upfirdn2d_f32":
        major, ih, iw, minor, kh, kw, ux, uy, dx, dy, px0, px1, py0, py1 = a[3:17]
        oh = (ih * uy + py0 + py1 - kh + dy) // dy
        ow = (iw * ux + px0 + px1 - kw + dx) // dx
        kind = "blur" if (ux == 1 and dx == 1) else ("decimate x2" if dx == 2 else "zero-insert x2")
        return "upfirdn2d %s %dx%d taps" % (kind, kh, kw), "hbm", 4.0 * major * minor * (ih * iw + oh * ow)
    if name == "bias_act_f32":
        return "bias_act forward", "hbm", 8.0 * a[4] + (4.0 * a[4] if a[2] else 0.0)
    if name == "bias_act_bwd_f32":
        return "bias_act backward (+grad_bias)", "hbm", 12.0 * a[6]
    if name == "noise_bias_act_f32":
        return "noise+bias+lrelu forward", "hbm", 4.0 * a[5] * a[7] * (2 * a[6] + 1)
    if name == "noise_bias_act_bwd_f32":
        return "noise+bias+lrelu backward", "hbm", 4.0 * a[8] * a[10] * (3 * a[9] + 1)
    if name == "plane_scale_dot_f32":
        return "style modulation backward", "hbm", 12.0 * a[5] * a[6]
    if name == "add_scale_f32":
        return "residual merge (a+b)/sqrt2", "hbm", 12.0 * a[3]
    if name == "upsample2x_bilinear_add_f32":
        return "upsample x2 + residual", "hbm", 4.0 * a[3] * a[4] * a[5] * (1 + 4 + (4 if a[1] else 0))
    if name == "upsample2x_bilinear_bwd_f32":
        return "upsample x2 backward", "hbm", 4.0 * a[2] * a[3] * a[4] * 5
    if name == "weight_demod_f32":
        return "demodulation factor (+bwd)", "hbm", 4.0 * a[2] * a[3]
    if name == "weight_demod_bwd_f32":
        return "demodulation factor (+bwd)", "hbm", 12.0 * a[4] * a[5]
    if name in ("l2_normalize_f32", "l2_normalize_bwd_f32"):
        n = a[2 if name == "l2_normalize_f32" else 3] * a[3 if name == "l2_normalize_f32" else 4] * a[4 if name == "l2_normalize_f32" else 5]
        return "l2 normalize (+bwd)", "hbm", 4.0 * n * (2 if name == "l2_normalize_f32" else 3)
    if name in ("plane_affine_f32", "plane_affine_bwd_f32"):
        n = a[4] * a[5] if name == "plane_affine_f32" else a[6] * a[7]
        return "generator modulation (+bwd)", "hbm", 4.0 * n * (2 if name == "plane_affine_f32" else 3)
    if name in ("softplus_mean_f32", "softplus_mean_bwd_f32"):
        n = a[2] * a[3] if name == "softplus_mean_f32" else a[3] * a[4]
        return "gan loss (+bwd)", "hbm", 8.0 * n
    if name in ("random_crop_f32", "random_crop_bwd_f32"):
        images, ch, h, w, crops, size = a[4:10]
        return name[:-4], "hbm", 4.0 * ch * (images * h * w + images * crops * size * size)
    # Winograd route (SAE_WINOGRAD=1): the products are counted with the FLOPs they EXECUTE (16 per 2x2 tile and channel pair,
    # 4/9 of the layer's algorithmic count), the transforms with the bytes they move (x or y once, the transform domain 4x)
    if name in ("wino_gemm_f32", "wino_wgrad_gemm_f32"):
        n, c, m, th, tw = a[3:8]
        if BY_SHAPE:
            return "winograd products %-5s n%-3d %4d->%-4d %3dx%-3d tiles" % ("wgrad" if "wgrad" in name else "", n, c, m, th, tw), "mfma", 32.0 * n * c * m * th * tw
        return "winograd products (16 x 1x1%s)" % (" weight gradient" if "wgrad" in name else ""), "mfma", 32.0 * n * c * m * th * tw
    if name == "wino_fused_conv_f32":
        # the one-kernel route, counted with the FLOPs it EXECUTES: 16 multiply-adds per 2x2 output tile and channel pair
        n, c, m, h, w, pad = a[8], a[9], a[10], a[11], a[12], a[13]
        th, tw = (h + 2 * pad - 2) // 2, (w + 2 * pad - 2) // 2
        if BY_SHAPE:
            return "winograd fused conv n%-3d %4d->%-4d %3dx%-3d tiles" % (n, c, m, th, tw), "mfma", 32.0 * n * c * m * th * tw
        return "winograd fused conv (one kernel; executed FLOPs = direct / 2.25)", "mfma", 32.0 * n * c * m * th * tw
    if name == "wino_fused_wgrad_f32":
        n, c, m, h, w, pad = a[5], a[6], a[7], a[8], a[9], a[10]
        th, tw = (h + 2 * pad - 2) // 2, (w + 2 * pad - 2) // 2
        if BY_SHAPE:
            return "winograd fused wgrad n%-3d %4d->%-4d %3dx%-3d tiles" % (n, c, m, th, tw), "mfma", 32.0 * n * c * m * th * tw
        return "winograd fused weight gradient (one kernel + slice reduction; executed FLOPs = direct / 2.25)", "mfma", 32.0 * n * c * m * th * tw
    if name == "s2wino_dgrad_f32":
        # the polyphase stride-2 data gradient, counted with the FLOPs it EXECUTES: 25 multiply-adds per 2x2 of small-side positions
        n, cin, cout, h, w = a[5], a[6], a[7], a[8], a[9]
        if BY_SHAPE:
            return "conv 3x3 s2 dgrad polyphase n%-3d %4d->%-4d %3dx%-3d" % (n, cin, cout, 2 * h + 1, 2 * w + 1), "mfma", 2.0 * 25 / 4 * n * cin * cout * h * w
        return "conv 3x3 s2 dgrad on the polyphase form (executed FLOPs = direct x 25 / 36)", "mfma", 2.0 * 25 / 4 * n * cin * cout * h * w
    if name == "s2wino_weights_f32":
        return "winograd weight transform", "hbm", 140.0 * a[4] * a[5]
    if name == "wino_fused_weights_f32":
        return "winograd weight transform", "hbm", 100.0 * a[4] * a[5]
    if name in ("wino_input_f32", "wino_gy_f32"):
        return "winograd input / gradient transform", "hbm", 20.0 * a[3] * a[4] * a[5]
    if name == "wino_output_f32":
        return "winograd output transform (+ epilogue)", "hbm", 20.0 * a[6] * a[8] * a[9]
    if name == "wino_weights_f32":
        return "winograd weight transform", "hbm", 100.0 * a[4] * a[5]
    if name == "wino_wgrad_output_f32":
        return "winograd weight transform", "hbm", 100.0 * a[2] * a[3]
    if name in ("reflect_pad_f32", "reflect_pad_adj_f32"):
        planes, h, w, l, r, t, b = a[2:9]
        return name[:-4], "hbm", 4.0 * planes * (h * w + (h + t + b) * (w + l + r))
    return name, "hbm", 0.0


class Ledger:
    def __init__(self, lib):
        self.lib, self.records, self.active = lib, [], False
        orig = lib.call

        def call(name, *args):
            if not self.active or name in ("set_conv_math", "conv2d_wprep_query", "wino_gemm_workspace", "wino_wgrad_gemm_workspace",
                                               "wino_fused_weights_floats", "s2wino_weights_floats", "gemm_workspace"):
                return orig(name, *args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(name, *args)
            e1.record()
            label, bound, work = work_of(name, args)
            if BY_SHAPE and bound == "hbm":       # one row per argument tuple: the integer arguments that are not addresses
                dims = [str(v) for v in args if isinstance(v, int) and not isinstance(v, bool) and 0 <= v < (1 << 31)]
                label = "%s [%s]" % (label, ",".join(dims))
            self.records.append(((label, bound, work), e0, e1))

        lib.call = call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="church256")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--conv-math", default="f32")
    ap.add_argument("--with-r1", action="store_true", help="place the lazy-R1 iteration inside the window")
    ap.add_argument("--by-shape", action="store_true", help="one row per conv geometry, sorted by time above 0.85 of peak")
    args = ap.parse_args()
    global BY_SHAPE
    BY_SHAPE = args.by_shape
    sys.path.insert(0, ROOT)
    # ONE stream: the step's two branches otherwise overlap and the bracketed times of concurrent calls add up to more than the wall
    os.environ["SAE_TWO_STREAMS"] = "0"
    os.environ["SAE_HIP_GRAPH"] = "0"       # eager calls: the brackets sit around individual launches (hip_graph.py replays whole calls)
    import bench
    from swapping_autoencoder_pytorch_amd.options import make_options
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_model import create_model
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_optimizer import create_optimizer
    lib = hip_lib.get()
    hip_lib.set_conv_math(args.conv_math)
    ledger = Ledger(lib)
    batch = args.batch or bench.DEFAULT_BATCH[args.preset]
    opt = make_options(args.preset, batch_size=batch, num_gpus=1)
    torch.manual_seed(0)
    model = create_model(opt)
    optimizer = create_optimizer(opt, model)
    if args.with_r1:
        optimizer.discriminator_iter_counter = opt.R1_once_every - args.warmup - 1
    size = opt.crop_size
    pool = [torch.rand(batch, 3, size, size, device="cuda") * 2 - 1 for _ in range(4)]
    for i in range(args.warmup):
        optimizer.train_one_step({"real_A": pool[(2 * i) % 4]}, i)
        optimizer.train_one_step({"real_A": pool[(2 * i + 1) % 4]}, i)
    torch.cuda.synchronize()
    ledger.active = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        optimizer.train_one_step({"real_A": pool[(2 * i) % 4]}, i)
        optimizer.train_one_step({"real_A": pool[(2 * i + 1) % 4]}, i)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / args.steps
    ledger.active = False

    rows = collections.OrderedDict()
    for (label, bound, work), e0, e1 in ledger.records:
        r = rows.setdefault(label, [bound, 0, 0.0, 0.0])
        r[1] += 1
        r[2] += work
        r[3] += e0.elapsed_time(e1)
    k = args.steps
    print("# per-kernel-class roofline, %s preset, %dx%d, B=%d, conv math %s, %d timed iterations%s (event-bracketed C-ABI calls, the step on ONE stream: SAE_TWO_STREAMS=0)"
          % (args.preset, size, size, batch, args.conv_math, k, " incl. one lazy-R1 call" if args.with_r1 else ", no R1 call"))
    print("# bound peaks: fp32 MFMA 157.3 TFLOP/s, HBM 8.0 TB/s (spec; ~6.3 achievable).  Work is ALGORITHMIC (SURVEY 8d).")
    print("%-52s %5s %8s %12s %10s %9s %6s" % ("kernel class", "bound", "calls/it", "work/it", "ms/it", "achieved", "frac"))
    tot_ms = tot_fl = tot_by = 0.0
    def excess(r):          # time above 0.85 of the MFMA peak / above 0.6 of the HBM peak (4.8 TB/s, what the streaming kernels reach)
        return r[3] - r[2] / ((0.85 * MFMA_PEAK) if r[0] == "mfma" else (0.6 * HBM_PEAK)) * 1e3
    order = (lambda kv: -excess(kv[1])) if BY_SHAPE else (lambda kv: -kv[1][3])
    for label, (bound, calls, work, ms) in sorted(rows.items(), key=order):
        rate = work / (ms * 1e-3) if ms > 0 else 0.0
        if bound == "mfma":
            ws, rs, frac = "%9.1f GF" % (work / k / 1e9), "%6.1f TF/s" % (rate / 1e12), rate / MFMA_PEAK
            tot_fl += work
        else:
            ws, rs, frac = "%9.2f GB" % (work / k / 1e9), "%6.2f TB/s" % (rate / 1e12), rate / HBM_PEAK
            tot_by += work
        tot_ms += ms
        extra = ""
        if BY_SHAPE:
            extra = "   +%.2f ms over %s" % (excess((bound, calls, work, ms)) / k, "0.85" if bound == "mfma" else "4.8 TB/s")
        print("%-52s %5s %8.1f %12s %10.3f %9s %6.3f%s" % (label, bound, calls / k, ws, ms / k, rs, frac, extra))
    print("%-52s %5s %8s %12s %10.3f" % ("sum of bracketed C-ABI calls", "", "", "", tot_ms / k))
    print("%-52s %5s %8s %12s %10.3f" % ("outside the C-ABI (ATen glue, Adam, gaps)", "", "", "", wall - tot_ms / k))
    print("%-52s %5s %8s %12s %10.3f   -> %.2f images/s" % ("wall per iteration", "", "", "", wall, batch / wall * 1e3))
    print("# totals per iteration: %.2f TFLOP on the matrix cores (%.1f TF/s over the whole wall time = %.3f of peak), "
          "%.1f GB algorithmic through the HBM-bound kernels" % (tot_fl / k / 1e12, tot_fl / k / wall / 1e9,
                                                                 tot_fl / k / wall / 1e9 / 157.3e3 * 1e3, tot_by / k / 1e9))


if __name__ == "__main__":
    main()
