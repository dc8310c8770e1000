"""Per-layer A/B of the Winograd F(2x2,3x3) route against the direct MFMA kernels (HISTORY.md 4.0f): every 3x3 stride-1 geometry one
training iteration of a preset launches (forward, data gradient, weight gradient, with their call counts), timed on both routes
with HIP events, and the per-iteration saving of an eligibility rule read off the table.
    python tools/wino_ab.py --preset church256 > gpurun_out/wino_ab_church256.json"""
import argparse
import collections
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="church256")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--min-c", type=int, default=64)
    args = ap.parse_args()
    import bench
    from swapping_autoencoder_pytorch_amd import hip_lib
    from swapping_autoencoder_pytorch_amd.options import make_options
    from swapping_autoencoder_pytorch_amd.stylegan2_op import conv2d_gemm as cg
    from swapping_autoencoder_pytorch_amd.stylegan2_op import winograd
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_model import create_model
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_optimizer import create_optimizer
    hip_lib.get()
    winograd.configure(enabled=False)
    seen = collections.OrderedDict()
    recording = [False]
    orig = winograd.eligible

    def spy(geom, op="fwd"):
        if recording[0] and geom.k == 3 and geom.stride == 1 and geom.pad in (0, 1) and not (geom.h & 1) and not (geom.w & 1) \
                and min(geom.c, geom.m) >= args.min_c:
            key = (op,) + geom.key
            seen[key] = seen.get(key, 0) + 1
        return False

    winograd.eligible = spy
    batch = args.batch or bench.DEFAULT_BATCH[args.preset]
    opt = make_options(args.preset, batch_size=batch, num_gpus=1)
    torch.manual_seed(0)
    model = create_model(opt)
    optimizer = create_optimizer(opt, model)
    size = opt.crop_size
    pool = [torch.rand(batch, 3, size, size, device="cuda") * 2 - 1 for _ in range(2)]
    every = opt.R1_once_every
    for i in range(every):          # one full R1 period: counts are per `every` iterations
        recording[0] = i >= 0
        optimizer.train_one_step({"real_A": pool[0]}, i)
        optimizer.train_one_step({"real_A": pool[1]}, i)
    torch.cuda.synchronize()
    winograd.eligible = orig
    del model, optimizer
    torch.cuda.empty_cache()

    def timed(fn):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps

    rows = []
    for (op, n, c, h, w, m, k, s, p, cm), count in seen.items():
        g = cg._Geom(n, c, h, w, m, k, s, p, cm, 1.0 / (c * 9) ** 0.5)
        x = torch.randn(n, c, h, w, device="cuda")
        wt = torch.nn.Parameter(torch.randn(*g.weight_shape(), device="cuda"))     # a Parameter: the transform-domain weights are kept per version, as in the step
        gy = torch.randn(n, m, g.oh, g.ow, device="cuda")
        if op == "fwd":
            direct = lambda: cg._launch("conv2d_fwd_f32", cg.SAE_CONV_FWD, g, x, wt, (n, m, g.oh, g.ow))
            wino = lambda: winograd.conv(x, wt, g, kind="unfused")
            fused = (lambda: winograd.conv(x, wt, g, kind="fused")) if w >= 4 else None
        elif op == "dgrad":
            direct = lambda: cg._launch("conv2d_dgrad_f32", cg.SAE_CONV_DGRAD, g, gy, wt, (n, c, h, w))
            wino = lambda: winograd.conv(gy, wt, g, transpose=True, kind="unfused")
            fused = (lambda: winograd.conv(gy, wt, g, transpose=True, kind="fused")) if g.ow >= 4 else None
        else:
            direct = lambda: cg._launch("conv2d_wgrad_f32", cg.SAE_CONV_WGRAD, g, x, gy, g.weight_shape())
            wino = lambda: winograd.wgrad(x, gy, g, kind="unfused")
            fused = (lambda: winograd.wgrad(x, gy, g, kind="fused")) if g.ow % 16 == 0 else None
        td, tw = timed(direct), timed(wino)
        tf = timed(fused) if fused is not None else None
        gf = 2.0 * n * m * g.oh * g.ow * c * 9 / 1e9
        rows.append({"op": op, "n": n, "c": c, "m": m, "h": h, "w": w, "pad": p, "calls_per_%d_iterations" % every: count,
                     "direct_ms": round(td, 4), "winograd_ms": round(tw, 4), "direct_tflops": round(gf / td, 1),
                     "winograd_equiv_tflops": round(gf / tw, 1), "saving_ms_per_iteration": round((td - tw) * count / every, 4),
                     "fused_ms": None if tf is None else round(tf, 4),
                     "fused_equiv_tflops": None if tf is None else round(gf / tf, 1),
                     "fused_mfma_frac": None if tf is None else round(gf / 2.25 / tf / 157.3, 3),
                     "best_saving_ms_per_iteration": round((td - min(t for t in (td, tw, tf) if t is not None)) * count / every, 4)})
        del x, wt, gy
    rows.sort(key=lambda r: -r["saving_ms_per_iteration"])
    best = sum(r["saving_ms_per_iteration"] for r in rows if r["saving_ms_per_iteration"] > 0)
    allon = sum(r["saving_ms_per_iteration"] for r in rows if min(r["c"], r["m"]) >= 256)
    best3 = sum(r["best_saving_ms_per_iteration"] for r in rows)
    print(json.dumps({"preset": args.preset, "batch": batch, "rows": rows,
                      "saving_ms_per_iteration_if_only_winning_rows_take_the_route": round(best, 3),
                      "saving_ms_per_iteration_all_rows_with_256_channels": round(allon, 3),
                      "saving_ms_per_iteration_best_of_direct_unfused_fused": round(best3, 3)}, indent=1))


if __name__ == "__main__":
    main()
