#!/usr/bin/env python
"""bench.py — images/sec of the full Swapping-Autoencoder train iteration on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
  (one rank per GPU, RCCL; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).

One "step" = one full training iteration of the reference's driver on one synthetic batch:
a discriminator call and a generator call of SwappingAutoencoderOptimizer.train_one_step
(optimizers/swapping_autoencoder_optimizer.py:59-65), each with its Adam update, with the lazy R1
penalty on every 16th discriminator iteration (the default K = 16 contains exactly one).
value = N * B * K / t   (whole-job images per second; B images per GPU -> weak scaling).

Also reported on the same JSON line:
  roofline     – dominant kernel (the one-kernel Winograd F(2x2,3x3) convolution wino_fused_kernel<*> of
                 csrc/winograd_fused.hip: forward and data gradient of the wide 3x3 stride-1 layers): FLOPs per launch
                 -- EXECUTED (direct / 2.25) for `achieved` / `frac`, algorithmic (the direct convolution's) alongside --
                 / mean launch duration measured with HIP events on the launch stream in the kernel pass that follows
                 the timed region, against the fp32 MFMA peak (157.3 TFLOP/s); `traffic` from the committed counter
                 record PMC_DOMINANT_FILE (refused if it names another kernel), scaled by FLOPs to the average launch;
  roofline_by_kernel – the same event-timed fraction for the other MFMA classes (stride-2 family, weight gradients, 1x1);
  hbm_by_kernel – the HBM-bound classes (upfirdn2d, bias / activation, modulation) in TB/s against 8 TB/s;
  cpu_baseline – the reference's CPU path (ATen on the host cores, restated in oracle/aten_cpu_path.py and pinned
                 to the reference's own modules) MEASURED on one whole iteration of the preset at B = 2 (discriminator
                 call + generator call with their Adam updates; rank 0, N = 1 only); the D-only FLOP-scaled sample that
                 picks the thread count and the C oracle's rate ride along;
  other_presets – BASELINE configs 3 and 5 (ffhq512 B = 8, ffhq1024 B = 4) measured in their own processes.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# before the HIP runtime starts (see swapping_autoencoder_pytorch_amd/__init__.py: streams that share a hardware queue serialise)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

MFMA_F32_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix
# HBM-side traffic of the dominant kernel: read from the committed counter summary (tools/run_pmc.sh -> tools/pmc_summary.py
# --dominant-json): separate rocprofv3 --pmc passes over tools/pmc_kernels.py; reads = TCC_EA0_RDREQ x 128 B (FETCH_SIZE tallies
# those 128-byte requests at 64 B, the x2 correction of the guide, confirmed on a 1 GiB copy in the same passes), writes =
# TCC_EA0_WRREQ_64B x 64 B + the remaining write requests x 32 B, on the kernel's reference launch (128 -> 128 3x3 @256x256
# B=16).  Counters cannot be read inside a timed run; the figure is scaled to the average launch of the timed region by FLOPs.
PMC_DOMINANT_FILE = os.path.join(ROOT, "profiles", "r6_pmc_dominant.json")
# Round 5: the wide 3x3 stride-1 layers (forward and data gradient) run as ONE-kernel Winograd F(2x2,3x3) convolutions
# (csrc/winograd_fused.hip); that kernel is now the largest single consumer of the step.  (Rounds 2-4: the direct gather
# conv_igemm_kernel<3,1,2,2,1,4,8,false,true>, still reported as a class of its own.)
DOMINANT_KERNEL = "wino_fused_kernel"


def load_pmc_dominant():
    """The committed PMC record of the dominant kernel; refuses a record of another kernel (a stale file after a kernel
    change must not silently describe the new one)."""
    with open(PMC_DOMINANT_FILE) as f:
        rec = json.load(f)
    name = rec["kernel"].replace(" ", "")
    if not name.startswith(DOMINANT_KERNEL.replace(" ", "")):
        raise SystemExit("bench.py: %s describes kernel %r, the timed dominant kernel is %r: re-run tools/run_pmc.sh and "
                         "tools/pmc_summary.py --dominant-json" % (PMC_DOMINANT_FILE, rec["kernel"], DOMINANT_KERNEL))
    return rec


HBM_PEAK_TBS = 8.0                    # same table: HBM3E
MFMA_BF16_PEAK_TFLOPS = 2500.0        # same table, dense bf16 matrix; the bf16x6 arithmetic spends 6 bf16 products
                                      # (20/3 with the zero-padded ninth tap) per fp32 product
FLOPS_PER_IMAGE = {"church256": 1.815e12, "bedroom256": 1.815e12, "ffhq512": 3.91e12, "ffhq1024": 6.08e12}
DEFAULT_BATCH = {"church256": 16, "bedroom256": 16, "ffhq512": 8, "ffhq1024": 4, "tiny32": 4}
ALT_STREAMS_WATCHDOG_S = 180          # multi-rank runs: the extra leg in the other stream mode may take this long at most
CPU_BASELINE_BATCH = 2                # images of the measured whole CPU iteration of the default run (cpu_baseline)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preset", default="church256", choices=sorted(DEFAULT_BATCH))
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: the preset's)")
    ap.add_argument("--conv-math", default="f32", choices=["f32", "bf16x6"],
                    help="arithmetic of the 3x3 conv kernels for the reported value (include/sae_hip.h: sae_set_conv_math)")
    ap.add_argument("--alt-steps", type=int, default=8,
                    help="steps of the extra measurement with the OTHER conv arithmetic (0 = skip); reported as alt_conv_math")
    ap.add_argument("--force-allreduce", action="store_true",
                    help="with --gpus 1: run the multi-GPU gradient path (buckets, grad-ready hooks, asynchronous RCCL "
                         "all-reduce, per-bucket Adam) on the single rank -- the cost of that machinery, NOT a scaling number")
    ap.add_argument("--ring-rehearsal", type=int, default=0, metavar="N",
                    help="with --force-allreduce: behind every bucket's (empty, one-rank) all-reduce, put on a side stream the memory "
                         "traffic and kernel launches a RING all-reduce over N GPUs costs THIS GPU -- 2 (N - 1) steps of a 1/N slice, "
                         "read + reduce + write -- concurrent with the backward pass.  A rehearsal of the contention, NOT a scaling number")
    ap.add_argument("--same-device", action="store_true",
                    help="with --gpus N > 1: every rank on cuda:0 over the gloo backend -- the multi-rank code path (buckets, grad-ready "
                         "hooks, collectives, per-bucket Adam, the line's per-rank / all-reduce fields) rehearsed on a one-GPU box.  "
                         "NOT a scaling number: the ranks share one GPU and the collective goes through host memory")
    ap.add_argument("--alt-streams-steps", type=int, default=8,
                    help="multi-rank runs: steps of an extra measurement with the OTHER stream mode (streams.enabled(): one stream "
                         "is the default for a rank of a multi-rank job, SAE_TWO_STREAMS overrides), reported as `alt_streams` so that "
                         "one run on real hardware decides the default (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-cpu-baseline", action="store_true",
                    help="cpu_baseline.value from ONE whole iteration of the preset on the host at the preset's FULL batch (a "
                         "discriminator call + a generator call of the reference's driver through "
                         "oracle/aten_cpu_path.TrainIterationCPU; several minutes at 256 x 256, B = 16) instead of the same whole "
                         "iteration at B = 2 (the default: 10 - 30 s of CPU work)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-graph", action="store_true",
                    help="A/B only: SAE_HIP_GRAPH=0 -- every call enqueued from Python (the default on one GPU replays the "
                         "discriminator call and the generator call as hipGraphs: swapping_autoencoder_pytorch_amd/hip_graph.py)")
    ap.add_argument("--no-winograd", action="store_true",
                    help="A/B only: SAE_WINOGRAD=0 -- every 3x3 stride-1 layer on the direct MFMA kernels (stylegan2_op/winograd.py "
                         "routes the wide ones through Winograd F(2x2,3x3) by default); `config.winograd` records it")
    ap.add_argument("--no-winograd-fused", action="store_true",
                    help="A/B only: SAE_WINOGRAD_FUSED=0 -- the three-kernel form of the route everywhere (no csrc/winograd_fused.hip)")
    ap.add_argument("--kernel-steps", type=int, default=8,
                    help="steps of the kernel pass that follows the timed region: the same iterations with the step's branches on "
                         "ONE stream (SAE_TWO_STREAMS=0) and a HIP-event bracket around every launch of the tracked kernels -> "
                         "roofline / roofline_by_kernel.  In the timed region two kernels share the chip (streams.py) and a "
                         "bracket there would measure the sharing, not the kernel")
    ap.add_argument("--other-presets", default="ffhq512,ffhq1024",
                    help="default run, one GPU, church256 only: BASELINE configs 3 and 5 (the presets' own batch sizes per GPU) measured "
                         "in their own processes after the main measurement and reported as `other_presets` ('' = skip)")
    ap.add_argument("--other-steps", type=int, default=16, help="timed steps of each --other-presets leg (16 holds one lazy-R1 call)")
    ap.add_argument("--via-dropin", action="store_true",
                    help="time the SAME iteration through the drop-in runner's pre-seeding (swapping_autoencoder_pytorch_amd.dropin, "
                         "SAE_DROPIN_LEVEL, default full) under a reference-style tree: option parser, models.create_model -> "
                         "MultiGPUModelWrapper / nn.DataParallel, optimizers.create_optimizer -> train_one_step.  The tree is "
                         "--reference-root if given, else the framework stand-in tests/standin_tree.py writes (the reference "
                         "checkout does not exist on the GPU box).  One GPU; prints its own JSON line.")
    ap.add_argument("--reference-root", default=None, help="with --via-dropin: a checkout of the reference to run under")
    ap.add_argument("--dropin-steps", type=int, default=8,
                    help="default run, one GPU: steps of the extra --via-dropin measurement (own process) reported as "
                         "`via_dropin` on the same line (0 = skip)")
    return ap.parse_args()


def _quad_gather_tile(mout, in_w, out_h, out_w, pad):
    """Which quad-staged instantiation csrc/conv2d.hip dispatches a wide exact-fp32 3x3 stride-1 gather to -- "64x256"
    (conv_igemm_kernel<3,1,2,2,1,4,8,false,true>), "128x128" (<3,1,2,2,2,2,8,false,true>) or None -- for a launch producing
    `mout` channels on an out_h x out_w grid from rows of in_w floats.  Mirrors fwd_shape() (maps at least 128 wide take
    the 64-row tile; otherwise the 128-row tile unless a 64-row tile pads the channels 8 % less), pick_tile() and the quad
    conditions of launch_igemm() (rows a multiple of four floats, the widened patch within the tile's quad budget)."""
    def up(v, q):
        return (v + q - 1) // q * q

    def pow2(v):
        p = 1
        while p < v:
            p *= 2
        return p
    if mout <= 64:
        return None
    tile64 = out_w >= 128 or up(mout, 64) * 100 < up(mout, 128) * 92
    bn = 256 if tile64 else 128
    tw = min(max(pow2(out_w), 4), 32)
    th = max(pow2(out_h), 4)
    while tw * th > bn:
        th //= 2
    th = max(th, 1)
    tn = bn // (tw * th)
    if in_w % 4 != 0 or pad > 4 or tn * (th + 2) * ((tw + 8) // 4) > bn // 2:
        return None
    return "64x256" if tile64 else "128x128"


# MFMA kernel classes of roofline_by_kernel: the template each class runs.  FLOPs are ALGORITHMIC (the direct convolution's, SURVEY
# 8d); the Winograd classes also carry the FLOPs they EXECUTE (16 multiply-adds per 2x2 output tile and channel pair = direct / 2.25)
KERNEL_CLASSES = {
    "dominant": ("conv 3x3 s1 forward / data gradient on the one-kernel Winograd F(2x2,3x3) route", "wino_fused_kernel<*>"),
    "wino_wgrad": ("conv 3x3 s1 weight gradient on the one-kernel Winograd route (+ slice reduction)", "wino_fused_wgrad_kernel<*>"),
    "wino_unfused": ("conv 3x3 s1 on the three-kernel Winograd route (transform / sixteen 1x1 products / transform)",
                     "wino_input* / conv_igemm_kernel<1,1,*> batched / wino_output*"),
    "s1_gather_256": ("conv 3x3 s1 direct gather on maps >= 128 wide: 64 x 256 tile, quad staging", "conv_igemm_kernel<3,1,2,2,1,4,8,false,true>"),
    "s1_gather_128": ("conv 3x3 s1 direct gather on smaller maps: 128 x 128 tile, quad staging", "conv_igemm_kernel<3,1,2,2,2,2,8,false,true>"),
    "s2_dgrad": ("conv 3x3 s2 data gradient / transposed conv (plain + modulated)", "conv_igemm_tr2_kernel<2,16,*> / conv_igemm_tr_kernel"),
    "s2_dgrad_poly": ("conv 3x3 s2 data gradient / transposed conv on the polyphase minimal-filtering form (2^k + 1 maps up to 65 wide, "
                      ">= 256 contraction channels; 25 of the direct form's 36 multiplications)", "s2w_dgrad_kernel<*>"),
    "s2_fwd": ("conv 3x3 s2 forward gather (plain + modulated)", "conv_igemm_kernel<3,2,*>"),
    "s1_wgrad": ("conv 3x3 s1 weight gradient, direct (plain + modulated)", "conv_wgrad16_kernel<1> (fallback conv_wgrad_kernel<3,1,*>)"),
    "s2_wgrad": ("conv 3x3 s2 weight gradient (plain + modulated)", "conv_wgrad_kernel<3,2,*> (<= 64 gradient channels: conv_wgrad16_kernel<2>)"),
}
WINO_CLASSES = ("dominant", "wino_wgrad", "wino_unfused")
EXECUTED_NOTE = {"s2_dgrad_poly": "achieved = executed FLOPs (algorithmic x 25 / 36); *_algorithmic = the direct convolution's"}


class DominantKernelTimer:
    """Brackets, with HIP events on the launch stream, every launch of the 3x3 conv classes of KERNEL_CLASSES: the Winograd
    routes at stylegan2_op.winograd.conv / wgrad (which the conv wrappers reach before any direct launch), the direct kernels at
    conv2d_gemm._launch / _launch_fused / _launch_mod for the launches winograd.route() leaves to them (_quad_gather_tile repeats
    the library's dispatch rule so that the two gather tiles are told apart).  Records (class, algorithmic FLOPs, executed FLOPs,
    events); durations are read after the final synchronise."""

    def __init__(self):
        self.records = []
        self.active = False

    def classify(self, cg, wino, op, geom, activation_factor):
        """Class key of a DIRECT conv launch, or None (also None when the launch belongs to the Winograd route, which is
        bracketed where it is taken)."""
        if not self.active or geom.k != 3:
            return None
        if geom.stride == 1:
            if wino.route(geom, {cg.SAE_CONV_FWD: wino.FWD, cg.SAE_CONV_DGRAD: wino.DGRAD, cg.SAE_CONV_WGRAD: wino.WGRAD}[op]) is not None:
                return None
            if op == cg.SAE_CONV_WGRAD:
                return "s1_wgrad" if max(geom.m, geom.c) > 32 else None
            if activation_factor:
                return None
            tile = None
            if op == cg.SAE_CONV_FWD:
                tile = _quad_gather_tile(geom.m, geom.w, geom.oh, geom.ow, geom.pad)
            elif op == cg.SAE_CONV_DGRAD:
                tile = _quad_gather_tile(geom.c, geom.ow, geom.h, geom.w, 2 - geom.pad)
            return {"64x256": "s1_gather_256", "128x128": "s1_gather_128"}.get(tile)
        if op == cg.SAE_CONV_DGRAD and wino.route(geom, wino.DGRAD) is not None:
            return None                 # (the polyphase form: bracketed in winograd.conv)
        return {cg.SAE_CONV_FWD: "s2_fwd", cg.SAE_CONV_DGRAD: "s2_dgrad", cg.SAE_CONV_WGRAD: "s2_wgrad"}.get(op)

    def install(self):
        import torch
        from swapping_autoencoder_pytorch_amd.stylegan2_op import conv2d_gemm as cg
        from swapping_autoencoder_pytorch_amd.stylegan2_op import winograd as wino
        timer = self

        def bracket(key, geom, call, executed_ratio=1.0):
            if key is None:
                return call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = call()
            e1.record()
            fl = 2.0 * geom.n * geom.m * geom.oh * geom.ow * geom.c * 9
            timer.records.append((key, fl, fl * executed_ratio, e0, e1))
            return out

        orig_launch = cg._launch

        def launch(name, op, geom, a, b, out_shape, out=None):
            return bracket(timer.classify(cg, wino, op, geom, False), geom, lambda: orig_launch(name, op, geom, a, b, out_shape, out=out))

        cg._launch = launch
        orig_fused = cg._launch_fused

        def launch_fused(geom, x, w, bias, slope, scale):     # forward with the fused bias + leaky-ReLU epilogue
            return bracket(timer.classify(cg, wino, cg.SAE_CONV_FWD, geom, False), geom,
                           lambda: orig_fused(geom, x, w, bias, slope, scale))

        cg._launch_fused = launch_fused
        orig_mod = cg._launch_mod

        def launch_mod(name, op, geom, a, b, out_shape, x_scale=None, y_scale=None, wm_scale=None, wc_scale=None, **kw):
            # a modulated-conv call whose ACTIVATION carries no factor runs the un-modulated kernel instantiation (weight
            # factors ride in the weight re-layout): the data gradient of the generator's plain modulated convs
            key = timer.classify(cg, wino, op, geom, x_scale is not None or y_scale is not None)
            return bracket(key, geom, lambda: orig_mod(name, op, geom, a, b, out_shape, x_scale, y_scale, wm_scale, wc_scale, **kw))

        cg._launch_mod = launch_mod
        # the Winograd routes: 16 multiply-adds per 2x2 output tile and channel pair instead of 36
        orig_conv, orig_wgrad = wino.conv, wino.wgrad

        def wino_conv(x, w, geom, transpose=False, **kw):
            kind = kw.get("kind") or wino.route(geom, wino.DGRAD if transpose else wino.FWD) or "unfused"
            key = {"fused": "dominant", "s2poly": "s2_dgrad_poly"}.get(kind, "wino_unfused") if timer.active else None
            return bracket(key, geom, lambda: orig_conv(x, w, geom, transpose=transpose, **kw), 25.0 / 36.0 if kind == "s2poly" else 4.0 / 9.0)

        def wino_wgrad(x, gy, geom, **kw):
            kind = kw.get("kind") or wino.route(geom, wino.WGRAD) or "unfused"
            key = ("wino_wgrad" if kind == "fused" else "wino_unfused") if timer.active else None
            return bracket(key, geom, lambda: orig_wgrad(x, gy, geom, **kw), 4.0 / 9.0)

        wino.conv, wino.wgrad = wino_conv, wino_wgrad

    def summary(self, conv_math="f32", steps=1):
        """(roofline of the dominant kernel, roofline_by_kernel list) from the event brackets of the kernel pass."""
        dom = [r for r in self.records if r[0] == "dominant"]
        if not dom or conv_math != "f32":
            # (bf16x6 runs no Winograd route: its line carries the by-class table only)
            dom = []
        peak = MFMA_F32_PEAK_TFLOPS if conv_math == "f32" else MFMA_BF16_PEAK_TFLOPS / 6.0
        roof = None
        if dom:
            ms = sum(e0.elapsed_time(e1) for _, _, _, e0, e1 in dom)
            fl = sum(f for _, f, _, _, _ in dom)
            fx = sum(f for _, _, f, _, _ in dom)
            n = len(dom)
            pmc = load_pmc_dominant()
            scale = (fx / n / 1e9) / pmc["gflop"]
            roof = {"bound": "mfma", "kernel": DOMINANT_KERNEL + "<*> (csrc/winograd_fused.hip)",
                    "achieved": round(fx / (ms * 1e-3) / 1e12, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(fx / (ms * 1e-3) / 1e12 / peak, 4),
                    "achieved_algorithmic": round(fl / (ms * 1e-3) / 1e12, 2),
                    "frac_algorithmic": round(fl / (ms * 1e-3) / 1e12 / peak, 4),
                    "traffic": round((pmc["read_bytes"] + pmc["write_bytes"]) * scale), "launches": n,
                    "avg_launch_ms": round(ms / n, 4), "avg_launch_gflop": round(fx / n / 1e9, 2),
                    "avg_launch_gflop_algorithmic": round(fl / n / 1e9, 2),
                    "note": ("achieved / frac = the FLOPs the kernel EXECUTES (Winograd F(2x2,3x3): 16 multiply-adds per 2x2 output tile "
                             "and channel pair) over the mean launch duration, against the fp32 MFMA peak: the utilisation of the "
                             "matrix pipe.  achieved_algorithmic / frac_algorithmic = the ALGORITHMIC FLOPs of the direct convolution "
                             "(SURVEY 8d: 2 N M OH OW C 9), Winograd route -- 2.25x the executed ones, so it can exceed the peak.  The "
                             "event bracket includes the transform-domain weight preparation when the parameter changed since the "
                             "last launch (once per optimiser step).  traffic = HBM-side bytes per average launch from the PMC "
                             "passes (%s: %.0f MB read + %.0f MB written per %.0f GFLOP executed on the reference launch against "
                             "%.0f MB algorithmic), scaled by FLOPs"
                             % (os.path.relpath(PMC_DOMINANT_FILE, ROOT), pmc["read_bytes"] / 1e6, pmc["write_bytes"] / 1e6,
                                pmc["gflop"], pmc["algorithmic_bytes"] / 1e6))}
        by_kernel = []
        for key, (what, template) in KERNEL_CLASSES.items():
            rec = [r for r in self.records if r[0] == key]
            if not rec:
                continue
            kms = sum(e0.elapsed_time(e1) for _, _, _, e0, e1 in rec)
            kfl = sum(f for _, f, _, _, _ in rec)
            kfx = sum(f for _, _, f, _, _ in rec)
            row = {"class": what, "kernel": template if conv_math == "f32" else "bf16x6 counterpart of " + template,
                   "launches": len(rec), "ms_per_step": round(kms / steps, 3),
                   "achieved": round(kfx / (kms * 1e-3) / 1e12, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                   "frac": round(kfx / (kms * 1e-3) / 1e12 / peak, 4)}
            if key in WINO_CLASSES or key in EXECUTED_NOTE:
                row["achieved_algorithmic"] = round(kfl / (kms * 1e-3) / 1e12, 2)
                row["frac_algorithmic"] = round(kfl / (kms * 1e-3) / 1e12 / peak, 4)
                row["flops"] = EXECUTED_NOTE.get(key, "achieved = executed FLOPs (algorithmic / 2.25); *_algorithmic = the direct convolution's")
            by_kernel.append(row)
        return roof, (by_kernel or None)


class HbmKernelTimer:
    """Event brackets around the K1 / K2 C-ABI calls (sae_upfirdn2d*_f32, sae_bias_act*_f32, sae_noise_bias_act*_f32) of the
    one-stream kernel pass: ALGORITHMIC bytes (tools/roofline_ledger.work_of: every operand and the result once) per class
    over the summed launch durations, against the 8 TB/s HBM peak -- BASELINE config 3's "HBM GB/s vs peak"."""
    PREFIXES = ("upfirdn2d", "bias_act", "noise_bias_act")

    def __init__(self):
        self.records, self.active = [], False

    def install(self):
        import torch
        from swapping_autoencoder_pytorch_amd import hip_lib
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import roofline_ledger
        lib = hip_lib.get()
        orig = lib.call

        def call(name, *a):
            if not self.active or not name.startswith(self.PREFIXES) or "workspace" in name:
                return orig(name, *a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(name, *a)
            e1.record()
            label, _, work = roofline_ledger.work_of(name, a)
            self.records.append((label, work, e0, e1))
            return out

        lib.call = call

    def summary(self, steps):
        rows = {}
        for label, work, e0, e1 in self.records:
            r = rows.setdefault(label, [0, 0.0, 0.0])
            r[0] += 1
            r[1] += work
            r[2] += e0.elapsed_time(e1)
        out = [{"class": k, "launches": n, "ms_per_step": round(ms / steps, 3), "gb_per_step": round(by / steps / 1e9, 3),
                "achieved": round(by / (ms * 1e-3) / 1e12, 3), "peak": HBM_PEAK_TBS, "unit": "TB/s",
                "frac": round(by / (ms * 1e-3) / 1e12 / HBM_PEAK_TBS, 4)} for k, (n, by, ms) in rows.items() if ms > 0]
        out.sort(key=lambda r: -r["ms_per_step"])
        tot_by, tot_ms = sum(r[1] for r in rows.values()), sum(r[2] for r in rows.values())
        total = None
        if tot_ms > 0:
            total = {"class": "all K1 / K2 launches", "ms_per_step": round(tot_ms / steps, 3), "gb_per_step": round(tot_by / steps / 1e9, 3),
                     "achieved": round(tot_by / (tot_ms * 1e-3) / 1e12, 3), "peak": HBM_PEAK_TBS, "unit": "TB/s",
                     "frac": round(tot_by / (tot_ms * 1e-3) / 1e12 / HBM_PEAK_TBS, 4)}
        return total, out


def other_presets_leg(args):
    """BASELINE configs 3 and 5 on the driver's line: each preset at its own batch size in its own process (a fresh
    allocator and module table), `other_steps` timed iterations holding one lazy-R1 call + a short one-stream kernel pass."""
    import subprocess
    legs = []
    for preset in [p for p in args.other_presets.split(",") if p]:
        cmd = [sys.executable, os.path.abspath(__file__), "--preset", preset, "--steps", str(args.other_steps), "--warmup", "3",
               "--conv-math", args.conv_math, "--alt-steps", "0", "--dropin-steps", "0", "--no-cpu-baseline", "--kernel-steps", "3",
               "--other-presets", ""]
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        except Exception as e:      # noqa: BLE001 -- the main measurement stands on its own
            legs.append({"preset": preset, "error": "%s: %s" % (type(e).__name__, str(e)[:300])})
            continue
        leg = {"preset": preset, "workload": rec["config"]["workload"]}
        for k in ("value", "unit", "steps", "warmup", "ms_per_step", "value_r1_every_16", "ms_r1_extra", "r1_iterations_in_window", "ms_d_call_median",
                  "ms_g_call_median", "model_tflops_per_gpu", "frac_of_mfma_f32_roofline", "hbm_k1_k2", "hbm_by_kernel"):
            if k in rec:
                leg[k] = rec[k]
        if "roofline" in rec:
            leg["roofline_dominant"] = {k: rec["roofline"][k] for k in ("kernel", "achieved", "peak", "unit", "frac", "launches")}
        if "roofline_by_kernel" in rec:
            leg["roofline_by_kernel"] = [{k: r[k] for k in ("class", "ms_per_step", "achieved", "frac")} for r in rec["roofline_by_kernel"]]
        legs.append(leg)
    return legs


def cpu_baseline_config1(A, threads_hint):
    """BASELINE.json configs[0], the reference's own CPU-runnable case, measured whole: 32 x 32, B = 4, one reconstruction
    forward + backward (E -> G -> L1, gradients of every E and G parameter) through the ATen-CPU restatement of the
    reference's StyleGAN2ResnetEncoder / StyleGAN2ResnetGenerator (oracle/aten_cpu_path.py, pinned to the unmodified
    reference modules by tests/dropin_ref_worker.py::aten_cpu_path_pin).  No scaling: images per second of that pass."""
    import torch
    from swapping_autoencoder_pytorch_amd.options import make_options
    opt = make_options("tiny32", batch_size=4, num_gpus=0)
    torch.manual_seed(0)
    enc, gen = A.EncoderCPU(opt), A.GeneratorCPU(opt)
    params = list(enc.parameters()) + list(gen.parameters())
    img = torch.rand(4, 3, 32, 32) * 2 - 1

    def one_pass():
        for p in params:
            p.grad = None
        (gen(*enc(img)) - img).abs().mean().backward()

    best, sweep = None, []
    for threads in sorted({1, 8, max(1, min(threads_hint, 32))}):
        torch.set_num_threads(threads)
        one_pass()
        t0, reps = time.time(), 0
        while reps < 3 or (time.time() - t0 < 2.0 and reps < 50):
            one_pass()
            reps += 1
        dt = (time.time() - t0) / reps
        sweep.append({"threads": threads, "ms": round(dt * 1e3, 1)})
        if best is None or dt < best[1]:
            best = (threads, dt)
    threads, dt = best
    return {"value": round(4 / dt, 3), "unit": "images/s", "cores": threads, "kind": "port", "ms_per_pass": round(dt * 1e3, 1),
            "sample": "BASELINE.json configs[0]: E+G reconstruction forward+backward, 4 images 32x32, default channel widths "
                      "(%d parameters), ATen CPU, best of %s" % (sum(p.numel() for p in params), sweep)}


def cpu_baseline_full(A, preset, batch, threads):
    """SURVEY 8d's CPU figure measured, not scaled: one discriminator call + one generator call of the reference's driver
    (swapping_autoencoder_model.py:116-136,187-231, Adam updates included, no lazy-R1 call) on the ATen CPU path
    (oracle/aten_cpu_path.TrainIterationCPU, pinned to the reference's golden loss dictionaries in
    tests/test_network_parity_cpu.py), the preset's networks and batch, `threads` host threads."""
    import torch
    from swapping_autoencoder_pytorch_amd.options import make_options
    opt = make_options(preset, batch_size=batch, num_gpus=0)
    torch.manual_seed(0)
    torch.set_num_threads(threads)
    it = A.TrainIterationCPU(opt)
    g = torch.Generator().manual_seed(1)
    t0 = time.time()
    it.discriminator_call(torch.rand(batch, 3, opt.crop_size, opt.crop_size, generator=g) * 2 - 1)
    t1 = time.time()
    it.generator_call(torch.rand(batch, 3, opt.crop_size, opt.crop_size, generator=g) * 2 - 1)
    t2 = time.time()
    return {"value": round(batch / (t2 - t0), 5), "unit": "images/s", "cores": threads, "kind": "port", "extrapolated": False,
            "s_d_call": round(t1 - t0, 2), "s_g_call": round(t2 - t1, 2),
            "sample": "ONE whole iteration of the %s preset, B = %d: discriminator call + generator call with their Adam updates, "
                      "no lazy-R1 call, first and only pass (thread-pool and primitive start-up included)" % (preset, batch)}


def cpu_baseline(preset, size, batch, full=False):
    """The reference's CPU code path timed on this host (rank 0, N = 1): the image discriminator forward + backward
    (weights trainable, input without gradient = the D(real) pass of a discriminator step) through ATen on all
    host cores — F.conv2d / its two backward kernels (MKLDNN), upfirdn2d_native (F.pad + F.conv2d), F.leaky_relu —
    restated in oracle/aten_cpu_path.py and pinned to the reference's own Discriminator in
    tests/test_dropin_train.py.  The sample's conv FLOPs per second are converted to images/s with the FLOPs per
    image of the full iteration.  The C oracle's rate on one conv layer is kept as `oracle_port`."""
    import subprocess
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import abi_harness as H
    import aten_cpu_path as A
    from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary
    cores = os.cpu_count() or 1
    per_image = FLOPS_PER_IMAGE.get(preset, 1.815e12)

    torch.manual_seed(0)
    disc = A.DiscriminatorCPU(size, 2)
    n = 2                                      # bounded sample: 2 images of the church preset = 0.56 TFLOP of conv work
    x = torch.rand(n, 3, size, size) * 2 - 1
    flops = disc.train_flops(n, size)

    def one_pass():
        for p in disc.parameters():
            p.grad = None
        torch.nn.functional.softplus(-disc(x)).mean().backward()

    # ATen's CPU kernels do not scale to every hardware thread of a large host (256 threads measured 4x SLOWER than
    # 8 on this workload): sweep a few thread counts inside a ~30 s budget and report the best one, with the thread
    # count actually used in `cores`
    candidates = sorted({max(8, min(cores, t)) for t in (cores // 8, cores // 4, cores // 2)})
    best, sweep, t_begin = None, [], time.time()
    for threads in candidates:
        torch.set_num_threads(threads)
        one_pass()                             # MKLDNN primitive creation, thread pool start-up
        t0 = time.time()
        one_pass()
        dt = time.time() - t0
        sweep.append({"threads": threads, "s": round(dt, 2)})
        if best is None or dt < best[1]:
            best = (threads, dt)
        if time.time() - t_begin > 30.0:
            break
    threads, dt = best
    out = {"value": round(flops / dt / per_image, 5), "unit": "images/s", "cores": threads, "host_cores": cores, "kind": "port",
           "extrapolated": True,
           "port_of": "aten-cpu restatement of the reference's CPU path (oracle/aten_cpu_path.py: F.conv2d + autograd, "
                      "upfirdn2d_native, F.leaky_relu; torch.set_num_threads(%d), best of %s)" % (threads, sweep),
           "sample": "image discriminator forward + backward, %d images %dx%d: %.2f TFLOP of conv work in %.2f s = %.2f "
                     "TFLOP/s, scaled by %.3f TFLOP/image of the full iteration" % (n, size, size, flops / 1e12, dt,
                                                                                  flops / dt / 1e12, per_image / 1e12)}

    # cpu_baseline.value is MEASURED: one whole iteration of the preset (discriminator call + generator call, Adam updates
    # included) on the ATen CPU path -- at the preset's batch with --full-cpu-baseline (minutes), by default at B = 2, which
    # bounds the CPU work to the 10 - 30 s SURVEY 8d asks for (the swap pairs images, so 2 is the smallest batch the step
    # accepts).  The D-only, FLOP-scaled sample above chose the thread count and rides along as `flop_scaled_sample`.
    scaled = {k: out[k] for k in ("value", "sample")}
    out.update(cpu_baseline_full(A, preset, batch if full else CPU_BASELINE_BATCH, threads))
    out["flop_scaled_sample"] = scaled
    out["config1"] = cpu_baseline_config1(A, threads)

    so = os.path.join(ROOT, "oracle", "libsae_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    ora = SaeLibrary(so, prefix="oracle_", device_only=False)
    rng = np.random.default_rng(0)
    c, hw, m = 128, 256, 128
    d = H.conv_desc(1, c, hw, hw, m, 3, 1, 1)
    xo = rng.standard_normal((1, c, hw, hw)).astype(np.float32)
    w = rng.standard_normal((m, c, 3, 3)).astype(np.float32)
    gy = rng.standard_normal((1, m, hw, hw)).astype(np.float32)
    H.conv(ora, 0, H.conv_desc(1, 8, 32, 32, 8, 3, 1, 1), xo[:, :8, :32, :32], w[:8, :8], (1, 8, 32, 32))   # OpenMP start-up
    oflops = 3 * 2.0 * m * hw * hw * c * 9
    t0 = time.time()
    H.conv(ora, 0, d, xo, w, gy.shape)
    H.conv(ora, 1, d, gy, w, xo.shape)
    H.conv(ora, 2, d, xo, gy, w.shape)
    odt = time.time() - t0
    out["oracle_port"] = {"value": round(oflops / odt / per_image, 5), "unit": "images/s",
                          "sample": "oracle/sae_oracle.c (double accumulation, OpenMP) conv fwd+dgrad+wgrad 128->128 3x3 "
                                    "@256x256, 1 image: %.0f GFLOP in %.2f s" % (oflops / 1e9, odt)}
    return out


def preset_argv(preset, batch):
    """The preset's flags (swapping_autoencoder_pytorch_amd/options.py: PRESETS, the reference launchers' values) as a
    command line for the reference-style option parser."""
    from swapping_autoencoder_pytorch_amd.options import PRESETS
    argv = ["--name", "bench_via_dropin", "--dataset_mode", "synthetic", "--num_gpus", "1", "--batch_size", str(batch)]
    for k, v in PRESETS[preset].items():
        argv += ["--" + k, str(v)]
    return argv


def main_via_dropin(args):
    """`--via-dropin`: the church256 iteration driven through the drop-in runner's pre-seeded modules by a reference-style
    tree's OWN train_one_step (north_star: "drops into train.py unchanged").  Same synthetic batches, same fences and
    clock as the main measurement; one rank."""
    import tempfile
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
    if args.gpus != 1:
        raise SystemExit("bench.py --via-dropin measures one GPU")
    torch.cuda.set_device(0)
    from swapping_autoencoder_pytorch_amd import dropin, hip_lib
    from swapping_autoencoder_pytorch_amd.fused_adam import FusedAdam
    hip_lib.get()
    hip_lib.set_conv_math(args.conv_math)
    if args.reference_root:
        ref_root, tree = os.path.abspath(args.reference_root), "reference checkout"
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import standin_tree
        ref_root = standin_tree.write_framework(tempfile.mkdtemp(prefix="sae_standin_"))
        tree = "framework stand-in (tests/standin_tree.py: FRAMEWORK_FILES)"
    sys.path.insert(0, ref_root)
    dropin.install_missing_dependency_stubs()
    level = dropin.preseed()
    dropin.patch_util()
    dropin.inject_synthetic_dataset()
    torch.optim.Adam = FusedAdam                 # as dropin.main does
    import models                               # the tree's packages from here on
    import optimizers
    from options import TrainOptions
    batch = args.batch or DEFAULT_BATCH[args.preset]
    workdir = tempfile.mkdtemp(prefix="sae_via_dropin_")
    argv = sys.argv
    sys.argv = ["train.py", "--checkpoints_dir", workdir] + preset_argv(args.preset, batch)
    try:
        opt = TrainOptions().parse()
    finally:
        sys.argv = argv
    torch.manual_seed(0)
    model = models.create_model(opt)
    dropin.wrap_reference_r1()
    optimizer = optimizers.create_optimizer(opt, model)
    torch.manual_seed(1234)
    size = opt.crop_size
    pool = [torch.rand(batch, 3, size, size, device="cuda:0") * 2 - 1 for _ in range(4)]
    call_ms = {"d": [], "g": []}

    def iteration(i):
        t0 = time.perf_counter()
        optimizer.train_one_step({"real_A": pool[(2 * i) % 4]}, i)
        t1 = time.perf_counter()
        optimizer.train_one_step({"real_A": pool[(2 * i + 1) % 4]}, i)
        t2 = time.perf_counter()
        call_ms["d"].append((t1 - t0) * 1e3)
        call_ms["g"].append((t2 - t1) * 1e3)

    for i in range(args.warmup):
        iteration(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        iteration(args.warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    net = model.singlegpu_model
    every = opt.R1_once_every
    d_calls, g_calls = sorted(call_ms["d"][args.warmup:]), sorted(call_ms["g"][args.warmup:])
    line = {
        "metric": "images/sec (G+D+Dpatch train step)", "value": round(batch * args.steps / dt, 3), "unit": "images/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s preset, %dx%d, B=%d/GPU: full E/G/D/Dpatch D-step + G-step with Adam, lazy R1 every %dth D iteration, "
                               "driven through swapping_autoencoder_pytorch_amd.dropin (level %s) by the %s's own option parser, "
                               "create_model / MultiGPUModelWrapper (nn.DataParallel) and optimizer.train_one_step"
                               % (args.preset, size, size, batch, every, level, tree),
                   "global_batch": batch, "parallelism": "dp1", "conv_math": args.conv_math, "dropin_level": level},
        "r1_iterations_in_window": sum(1 for j in range(args.warmup + 1, args.warmup + args.steps + 1) if j % every == 0),
        "ms_d_call_median": round(d_calls[len(d_calls) // 2], 2), "ms_g_call_median": round(g_calls[len(g_calls) // 2], 2),
        "wrapper": type(model).__name__ + " / " + type(model.parallelized_model).__name__,
        "model_class": [c.__module__ + "." + c.__name__ for c in type(net).__mro__[:3]],
        "optimizer_class": type(optimizer).__module__ + "." + type(optimizer).__name__,
        "adam": type(optimizer.optimizer_D).__module__ + "." + type(optimizer.optimizer_D).__name__,
    }
    print(json.dumps(line), flush=True)


def via_dropin_leg(args, line):
    """The default one-GPU run: the --via-dropin measurement in its own process (fresh module table), summarised next to
    this run's own D / G call medians -- the like-for-like figures, neither contains a lazy-R1 call."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--via-dropin", "--steps", str(args.dropin_steps), "--warmup", "3",
           "--preset", args.preset, "--conv-math", args.conv_math] + (["--batch", str(args.batch)] if args.batch else [])
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as e:      # noqa: BLE001 -- the main measurement stands on its own
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    leg = {k: rec[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "ms_d_call_median", "ms_g_call_median",
                               "r1_iterations_in_window", "wrapper", "model_class", "optimizer_class", "adam")}
    leg["dropin_level"] = rec["config"]["dropin_level"]
    leg["workload"] = rec["config"]["workload"]
    if "ms_d_call_median" in line:
        mine = line["ms_d_call_median"] + line["ms_g_call_median"]
        leg["ms_d_plus_g_median"] = round(rec["ms_d_call_median"] + rec["ms_g_call_median"], 2)
        leg["ms_d_plus_g_median_direct"] = round(mine, 2)
        leg["dropin_over_direct"] = round((rec["ms_d_call_median"] + rec["ms_g_call_median"]) / mine, 4)
    return leg


def _install_ring_rehearsal(n_gpus):
    """RCCL launches nothing for an all-reduce over ONE rank, so the single-rank rehearsal of the gradient path says nothing about what
    the collective's kernels take from the backward pass.  This stands in for them: behind every bucket's all-reduce a side stream
    runs what a ring over `n_gpus` costs the local GPU -- N - 1 reduce-scatter steps (read the local 1/N slice and the received
    one, write the sum) and N - 1 all-gather steps (write the received slice, read it to send it on) -- as elementwise kernels over
    slices of the bucket itself, concurrent with whatever the step runs.  The link time is NOT modelled (nothing leaves the GPU)."""
    import torch
    import torch.distributed as dist
    real = dist.all_reduce
    # SAE_RING_REHEARSAL_STREAM: "own" (default: a stream of its own, as RCCL's kernels have), "high" (the same, high priority),
    # "launch" (no further stream: behind the collective on the stream it was launched from)
    where = os.environ.get("SAE_RING_REHEARSAL_STREAM", "own")
    # "nowait": a stream of its own that does NOT wait for the bucket (is it the dependency or the kernels?)
    side = None if where == "launch" else torch.cuda.Stream(priority=-1 if where == "high" else 0)
    scratch = {}
    # SAE_RING_REHEARSAL_KERNEL=persistent:CHANNELS:HOLD_US -- ONE kernel per bucket, CHANNELS persistent workgroups that walk the
    # 2 (N - 1) steps and take at least HOLD_US per step (tools/probe/ring_standin.hip): the shape of RCCL's ring kernel.  Default:
    # a chain of 2 (N - 1) ATen elementwise kernels over the slices (whole-GPU grids, a few microseconds each)
    form = os.environ.get("SAE_RING_REHEARSAL_KERNEL", "")
    host = [0.0, 0.0, 0]
    standin = None
    if form.startswith("persistent"):
        import ctypes
        _, channels, hold_us = (form.split(":") + ["32", "0"])[:3]
        lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "probe", "libring_standin.so"))
        lib.ring_standin_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p]
        standin = (lib, int(channels), int(hold_us))

    def rehearsed(tensor, *a, **kw):
        work = real(tensor, *a, **kw)
        flat = tensor.view(-1)
        n = flat.numel() // n_gpus
        if n == 0:
            return work
        if standin is not None:
            t0 = time.perf_counter()
            if side is not None and where != "nowait":
                side.wait_stream(torch.cuda.current_stream())
            t1 = time.perf_counter()
            st = side if side is not None else torch.cuda.current_stream()
            with torch.cuda.stream(st):
                buf = scratch.get(n)
                if buf is None:
                    buf = scratch[n] = torch.zeros(n, device=tensor.device)
                rc = standin[0].ring_standin_launch(flat.data_ptr(), buf.data_ptr(), flat.numel(), n_gpus, standin[1], standin[2],
                                                    st.cuda_stream)
                assert rc == 0, rc
            t2 = time.perf_counter()
            host[0] += t1 - t0; host[1] += t2 - t1; host[2] += 1
            if host[2] % 8 == 0 and os.environ.get("SAE_RING_REHEARSAL_HOST_TIMES"):
                print("ring rehearsal, host side: wait_stream %.3f ms, launch %.3f ms per collective (%d collectives)"
                      % (host[0] / host[2] * 1e3, host[1] / host[2] * 1e3, host[2]), file=sys.stderr, flush=True)
            return work
        if side is not None and where != "nowait":
            side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side if side is not None else torch.cuda.current_stream()):
            buf = scratch.get(n)
            if buf is None:
                buf = scratch[n] = torch.zeros(n, device=tensor.device)
            for s in range(2 * (n_gpus - 1)):
                seg = flat[(s % n_gpus) * n:(s % n_gpus + 1) * n]
                if s < n_gpus - 1:
                    torch.add(buf, seg, out=buf)     # reduce-scatter step
                else:
                    buf.copy_(seg)                   # all-gather step
        return work

    dist.all_reduce = rehearsed


def _stage_all_reduce_through_host():
    """--same-device only: gloo's own handling of device tensors stalls on the 32 MB gradient buckets when both ranks share one GPU
    (both ranks stuck in work.wait(), round 6 session E; the 3 MB buckets of tests/test_ddp_fullmodel.py pass).  The rehearsal
    stages every all-reduce through host memory instead: device -> host, gloo on the host tensor, host -> device, synchronously.
    Slower, and nothing overlaps -- it is a rehearsal of the code path and of the line's fields, not of the timing."""
    import torch.distributed as dist
    orig = dist.all_reduce

    class _Done:
        def wait(self):
            return True

    def all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not tensor.is_cuda:
            return orig(tensor, op=op, group=group, async_op=async_op)
        host = tensor.detach().cpu()
        orig(host, op=op, group=group)
        tensor.copy_(host)
        return _Done() if async_op else None

    dist.all_reduce = all_reduce


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch_command(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: the command that re-runs this
    script as N ranks (one per GPU) under torch.distributed.run, or None when this process is the worker."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)


def main():
    args = parse()
    if os.environ.get("SAE_BENCH_STACKS_AFTER_S"):      # diagnosis of a stuck multi-rank run: every thread's stack to stderr
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["SAE_BENCH_STACKS_AFTER_S"]), repeat=False, exit=False)
    if args.via_dropin:
        return main_via_dropin(args)
    relaunch = self_launch_command(args, sys.argv[1:])
    if relaunch is not None:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # RCCL needs dmabuf IPC on this host driver
        if args.same_device and "GPU_MAX_HW_QUEUES" not in os.environ.get("SAE_BENCH_KEEP_ENV", ""):
            # N processes x 8 hardware queues on ONE GPU oversubscribe the device's queues and the driver time-slices them
            # (HISTORY.md section 6, round 4: a two-rank one-GPU test took 14 minutes instead of 45 s): 4 per process here
            os.environ["GPU_MAX_HW_QUEUES"] = "4"
        os.execv(relaunch[0], relaunch)                              # never returns
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); one rank per GPU is the "
                         "contract" % (args.gpus, world))
    if args.same_device:
        local_rank = 0                       # every rank on the one GPU of the box (rehearsal of the multi-rank path)
    if torch.cuda.device_count() < min(args.gpus, local_rank + 1):
        raise SystemExit("bench.py: --gpus %d but only %d device(s) are visible" % (args.gpus, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    launched = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)    # torch.distributed.run, any N
    if args.force_allreduce:
        if world != 1:
            raise SystemExit("bench.py: --force-allreduce is the single-rank rehearsal of the multi-GPU gradient path")
        os.environ["SAE_FORCE_ALLREDUCE"] = "1"         # grad_allreduce.GradAllReducer: stay enabled at world size 1
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not launched:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            launched = True
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.same_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            if os.environ.get("SAE_BENCH_SAME_DEVICE_STAGE", "1") != "0":
                _stage_all_reduce_through_host()
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus
        world = dist.get_world_size()
    if args.ring_rehearsal:
        if not args.force_allreduce or args.ring_rehearsal < 2:
            raise SystemExit("bench.py: --ring-rehearsal N (N >= 2) goes with --force-allreduce")
        _install_ring_rehearsal(args.ring_rehearsal)
    dev = torch.device("cuda", local_rank)

    from swapping_autoencoder_pytorch_amd import hip_lib
    from swapping_autoencoder_pytorch_amd.grad_allreduce import broadcast_parameters
    from swapping_autoencoder_pytorch_amd.options import make_options
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_model import create_model
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_optimizer import create_optimizer
    hip_lib.get()           # fail loudly here if the HIP library is missing
    hip_lib.set_conv_math(args.conv_math)
    if args.no_graph:
        os.environ["SAE_HIP_GRAPH"] = "0"
    from swapping_autoencoder_pytorch_amd.stylegan2_op import winograd
    if args.no_winograd:
        winograd.configure(enabled=False)
    if args.no_winograd_fused:
        winograd.configure(fused=False)

    batch = args.batch or DEFAULT_BATCH[args.preset]
    opt = make_options(args.preset, batch_size=batch, num_gpus=1)
    torch.manual_seed(0)                        # identical replicas
    model = create_model(opt)
    broadcast_parameters(model.singlegpu_model)
    optimizer = create_optimizer(opt, model)
    torch.manual_seed(1234 + rank)              # per-rank crops / noise / data
    if world > 1 or args.force_allreduce:       # the all-reduce fields of the line (grad_allreduce.GradAllReducer.summary)
        optimizer.reducer_D.profile = optimizer.reducer_G.profile = True

    size = opt.crop_size
    pool = [torch.rand(batch, 3, size, size, device=dev) * 2 - 1 for _ in range(4)]

    timer = DominantKernelTimer()
    hbm_timer = HbmKernelTimer()
    if not args.no_kernel_timing:
        timer.install()
        hbm_timer.install()

    call_ms = {"d": [], "g": []}

    def iteration(i):
        # every train_one_step ends with a device->host copy of the losses (util.to_numpy, as in the
        # reference), so host clocks around a call measure that call
        t0 = time.perf_counter()
        optimizer.train_one_step({"real_A": pool[(2 * i) % 4]}, i)       # discriminator call
        t1 = time.perf_counter()
        optimizer.train_one_step({"real_A": pool[(2 * i + 1) % 4]}, i)   # generator call
        t2 = time.perf_counter()
        call_ms["d"].append((t1 - t0) * 1e3)
        call_ms["g"].append((t2 - t1) * 1e3)

    # hipGraph mode: the third call of each kind is the capture (hip_graph.WARMUP_CALLS eager ones before it).  With fewer
    # than three warm-up steps the capture would land in the timed region: it is set-up, like building the library, and is
    # brought forward by as many extra untimed iterations as are missing (reported as `graph_setup_iterations`).
    graph_setup = 0
    if optimizer.graphs is not None:
        from swapping_autoencoder_pytorch_amd import hip_graph
        graph_setup = max(0, hip_graph.WARMUP_CALLS + 1 - args.warmup)
    skew = graph_setup          # keeps the iteration numbers (batch pool index, R1 schedule bookkeeping below) in step
    for i in range(graph_setup):
        iteration(i)
    for i in range(args.warmup):
        iteration(graph_setup + i)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    records_before = (len(optimizer.reducer_D.records), len(optimizer.reducer_G.records))     # passes of the warm-up: left out
    t0 = time.perf_counter()
    for i in range(args.steps):
        iteration(skew + args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    dt_by_rank = None
    allreduce_fields = None
    if world > 1:
        every_rank = torch.zeros(world, device=dev, dtype=torch.float64)      # (a sum of one-hot vectors: all_reduce only)
        every_rank[rank] = dt
        dist.all_reduce(every_rank, op=dist.ReduceOp.SUM)
        dt_by_rank = [float(v) for v in every_rank.tolist()]
        dt = max(dt_by_rank)
    if world > 1 or args.force_allreduce:
        allreduce_fields = {"D": optimizer.reducer_D.summary(records_before[0]), "G": optimizer.reducer_G.summary(records_before[1])}
        if world > 1:
            ex = sum(v["ms_exposed"] for v in allreduce_fields.values() if v)
            worst = torch.tensor([ex], device=dev, dtype=torch.float64)
            dist.all_reduce(worst, op=dist.ReduceOp.MAX)
            allreduce_fields["ms_exposed_per_step_max_over_ranks"] = round(float(worst.item()), 3)
        allreduce_fields["note"] = ("per backward pass of rank 0, timed steps only: MB all-reduced, buckets, HIP-event time from the first "
                                    "bucket's launch to the last bucket's completion, and the time the waiting stream stood idle for "
                                    "collectives (exposed = not overlapped with backward / Adam)")
    done = skew + args.warmup + args.steps
    # kernel pass (not `value`): the same iterations with everything on ONE stream and an event bracket around every launch
    # of the tracked kernels.  The timed region above runs the step's independent branches on two streams: two kernels then
    # share the CUs and the duration of either says how the chip was shared, not how good the kernel is.
    kernel_ms_per_step = None
    if not args.no_kernel_timing and args.kernel_steps > 0:
        # (the parameters' AccumulateGrad nodes remember the stream of their first use; moving a branch back to the main stream
        # for this pass makes autograd point that out once per parameter -- it is the intent here)
        _quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if _quiet is not None:
            _quiet(False)
        prev = os.environ.get("SAE_TWO_STREAMS")
        os.environ["SAE_TWO_STREAMS"] = "0"
        graphs, optimizer.graphs = optimizer.graphs, None      # eager: the brackets sit around individual launches
        try:
            iteration(done)
            fence()
            timer.active = hbm_timer.active = True
            t0 = time.perf_counter()
            for i in range(args.kernel_steps):
                iteration(done + 1 + i)
            fence()
            kernel_ms_per_step = (time.perf_counter() - t0) / args.kernel_steps * 1e3
            timer.active = hbm_timer.active = False
        finally:
            optimizer.graphs = graphs
            if prev is None:
                os.environ.pop("SAE_TWO_STREAMS", None)
            else:
                os.environ["SAE_TWO_STREAMS"] = prev
        done += 1 + args.kernel_steps

    # the same iterations with the other conv arithmetic (not `value`)
    alt = None
    if args.alt_steps > 0:
        other = "bf16x6" if args.conv_math == "f32" else "f32"
        hip_lib.set_conv_math(other)
        main_graphs = optimizer.graphs
        settle = 2                               # workspaces change size with the arithmetic: let the allocator settle
        if main_graphs is not None:              # the captured graphs hold the OTHER arithmetic's kernels: capture this one's
            optimizer.graphs = hip_graph.StepGraphs(warmup=1)
            settle = 3                           # one eager call, the capture, one replay
        for i in range(settle):
            iteration(done + i)
        fence()
        t0 = time.perf_counter()
        for i in range(args.alt_steps):
            iteration(done + settle + i)
        fence()
        adt = time.perf_counter() - t0
        hip_lib.set_conv_math(args.conv_math)
        optimizer.graphs = main_graphs
        if world > 1:
            t = torch.tensor([adt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            adt = float(t.item())
        alt = {"conv_math": other, "steps": args.alt_steps, "value": round(world * batch * args.alt_steps / adt, 3),
               "unit": "images/s", "ms_per_step": round(adt / args.alt_steps * 1e3, 3),
               "note": "no lazy-R1 iteration in this window unless it spans a multiple of 16"}

    if rank == 0:
        images = world * batch * args.steps
        value = images / dt
        per_image = FLOPS_PER_IMAGE.get(args.preset)
        line = {
            "metric": "images/sec (G+D+Dpatch train step)", "value": round(value, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s preset, %dx%d, B=%d/GPU: full E/G/D/Dpatch D-step + G-step with Adam, lazy R1 "
                                   "every 16th D iteration" % (args.preset, size, size, batch),
                       "global_batch": world * batch, "parallelism": "dp%d" % world, "conv_math": args.conv_math},
        }
        from swapping_autoencoder_pytorch_amd import streams
        line["config"]["streams"] = ("the step's independent branches on two HIP streams" if streams.enabled() else
                                     "the step on one HIP stream (streams.enabled(): the default for a rank of a multi-rank job, or SAE_TWO_STREAMS=0)")
        line["config"]["launch"] = (
            "the discriminator call and the generator call replayed as hipGraphs (hip_graph.py: captured at the third call of each; "
            "the lazy-R1 call is enqueued eagerly)" if optimizer.graphs is not None else
            "every kernel enqueued from Python (SAE_HIP_GRAPH=0 / --no-graph, or a rank of a multi-rank job)")
        if graph_setup:
            line["graph_setup_iterations"] = graph_setup
        line["config"]["winograd"] = (
            "off: every 3x3 stride-1 layer on the direct MFMA kernels" if not winograd.enabled() else
            "3x3 stride-1 launches selected by stylegan2_op/winograd.route() run as Winograd F(2x2,3x3) (%s): 2.25x fewer "
            "multiplications than the ALGORITHMIC FLOP count (direct convolution, SURVEY 8d) this line's roofline figures use"
            % ("one-kernel form where it pays, three-kernel form elsewhere" if winograd._CFG.fused else "three-kernel form"))
        if args.force_allreduce:
            line["config"]["force_allreduce"] = ("single-rank rehearsal of the multi-GPU gradient path: %d + %d buckets all-reduced over "
                                                 "RCCL per iteration" % (len(optimizer.reducer_D.buckets), len(optimizer.reducer_G.buckets)))
            if args.ring_rehearsal:
                line["config"]["ring_rehearsal"] = ("behind every bucket: the reads / writes and launches of a ring all-reduce over %d GPUs "
                                                    "(2 x %d steps of a 1/%d slice) on a side stream, concurrent with the step; no link time"
                                                    % (args.ring_rehearsal, args.ring_rehearsal - 1, args.ring_rehearsal))
        if alt:
            line["alt_conv_math"] = alt
        if dt_by_rank is not None:
            line["ms_per_step_by_rank"] = [round(t / args.steps * 1e3, 3) for t in dt_by_rank]
        if allreduce_fields is not None:
            line["allreduce"] = allreduce_fields
        if args.same_device and world > 1:
            line["config"]["same_device"] = ("REHEARSAL: all %d ranks on cuda:0 over gloo -- the ranks share one GPU, nothing here is a "
                                             "scaling number" % world)
        if per_image:
            line["model_tflops_per_gpu"] = round(value / world * per_image / 1e12, 2)
            line["frac_of_mfma_f32_roofline"] = round(value / world * per_image / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)
        # lazy R1 runs on every R1_once_every-th discriminator iteration (the optimizer's own counter, 1-based)
        every = opt.R1_once_every
        first = skew + args.warmup      # iterations before the timed window (the optimizer's counter is 1-based)
        r1_in_window = sum(1 for j in range(first + 1, first + args.steps + 1) if j % every == 0)
        line["r1_iterations_in_window"] = r1_in_window
        d_window = call_ms["d"][first:first + args.steps]
        d_calls = sorted(d_window)
        g_calls = sorted(call_ms["g"][first:first + args.steps])
        if d_calls and g_calls:
            # median D call (without the lazy R1 extra), median G call, and the R1 surcharge: the D calls KNOWN to carry the R1
            # penalty (the optimizer's 1-based counter is a multiple of `every`) minus the median D call
            d_med = d_calls[len(d_calls) // 2]
            line["ms_d_call_median"] = round(d_med, 2)
            line["ms_g_call_median"] = round(g_calls[len(g_calls) // 2], 2)
            r1_calls = [d_window[j - first - 1] for j in range(first + 1, first + args.steps + 1) if j % every == 0]
            line["metric_version"] = 2      # 2: `value` is the plain wall-clock quotient of the K timed steps (rounds 1-3 and 5); round 4's
                                            # line carried the R1-normalised figure there
            if r1_calls:
                # SURVEY 8d defines the metric as B / (t_D + t_G + t_R1 / 16).  `value` is the wall-clock quotient of exactly the K
                # timed steps (which hold r1_in_window R1 calls, whatever K is); the same steps re-weighted to one lazy-R1 call
                # per `every` iterations ride along under their own keys.
                x = (sum(r1_calls) / len(r1_calls) - d_med) * 1e-3
                line["ms_r1_extra"] = round(x * 1e3, 2)
                t_norm = (dt - r1_in_window * x) / args.steps + x / every
                line["value_r1_every_%d" % every] = round(world * batch / t_norm, 3)
                line["ms_per_step_r1_every_%d" % every] = round(t_norm * 1e3, 3)
                line["value_note"] = ("value = N * B * steps / wall time of exactly the %d timed steps, which hold %d lazy-R1 call(s); "
                                      "value_r1_every_%d = B / (t_D + t_G + t_R1 / %d) (SURVEY 8d): the same steps with the R1 surcharge "
                                      "(the R1 iterations' D calls minus the median D call) re-weighted to one per %d"
                                      % (args.steps, r1_in_window, every, every, every))
            else:
                line["value_note"] = ("no lazy-R1 call fell into the timed window: value EXCLUDES the R1 surcharge of SURVEY 8d's "
                                      "metric (use --steps 16 or more)")
        roof, by_kernel = timer.summary(args.conv_math, max(args.kernel_steps, 1))
        if roof:
            roof["note"] = ("launch durations from the kernel pass: %d further iterations with the step on ONE stream "
                            "(SAE_TWO_STREAMS=0, %.1f ms per step) -- in the timed region the step's branches run on two streams and "
                            "kernels overlap; " % (args.kernel_steps, kernel_ms_per_step)) + roof["note"]
            line["roofline"] = roof
        if by_kernel:
            line["roofline_by_kernel"] = by_kernel
        if kernel_ms_per_step is not None:
            line["ms_per_step_one_stream"] = round(kernel_ms_per_step, 3)
            hbm_total, hbm_rows = hbm_timer.summary(max(args.kernel_steps, 1))
            if hbm_total:
                line["hbm_k1_k2"] = hbm_total
                line["hbm_by_kernel"] = hbm_rows
        if world == 1 and args.dropin_steps > 0 and not args.force_allreduce:
            line["via_dropin"] = via_dropin_leg(args, line)
        if world == 1 and args.preset == "church256" and args.other_presets and not launched and not args.no_winograd:
            line["other_presets"] = other_presets_leg(args)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.preset, size, batch, full=args.full_cpu_baseline)
    else:
        line = None
    # Multi-rank runs, LAST of all legs: a few steps in the OTHER stream mode on the same ranks and batches (decides
    # streams.enabled()'s multi-rank default on hardware).  Guarded: this combination (two streams + a multi-rank RCCL communicator)
    # has never run on more than one GPU; should it stall, a watchdog on every rank lets rank 0 print the line without it and
    # ends the process, so the run's headline figures are never lost to the extra leg.
    if world > 1 and args.alt_streams_steps > 0:
        import threading
        from swapping_autoencoder_pytorch_amd import streams as _streams

        def _give_up():
            if rank == 0:
                line["alt_streams"] = {"error": "the leg did not finish within %d s: left out" % ALT_STREAMS_WATCHDOG_S}
                print(json.dumps(line), flush=True)
            os._exit(0)

        watchdog = threading.Timer(ALT_STREAMS_WATCHDOG_S, _give_up)
        watchdog.daemon = True
        watchdog.start()
        was = os.environ.get("SAE_TWO_STREAMS")
        other_two = not _streams.enabled()
        os.environ["SAE_TWO_STREAMS"] = "1" if other_two else "0"
        try:
            for i in range(2):
                iteration(done + i)
            fence()
            t0 = time.perf_counter()
            for i in range(args.alt_streams_steps):
                iteration(done + 2 + i)
            fence()
            sdt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(sdt, op=dist.ReduceOp.MAX)
            sdt = float(sdt.item())
        finally:
            if was is None:
                os.environ.pop("SAE_TWO_STREAMS", None)
            else:
                os.environ["SAE_TWO_STREAMS"] = was
        watchdog.cancel()
        if rank == 0:
            line["alt_streams"] = {"streams": "two" if other_two else "one", "steps": args.alt_streams_steps,
                                   "value": round(world * batch * args.alt_streams_steps / sdt, 3), "unit": "images/s",
                                   "ms_per_step": round(sdt / args.alt_streams_steps * 1e3, 3),
                                   "note": "no lazy-R1 iteration in this window unless it spans a multiple of 16"}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
