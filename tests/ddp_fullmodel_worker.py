"""Worker of tests/test_ddp_fullmodel.py: the FULL E / G / D / Dpatch model of the micro preset through four calls of the
package's SwappingAutoencoderOptimizer.train_one_step (D, G, D + lazy R1, G) on one rank of a `gloo` process group --
arm() -> weight gradients written into bucket slots -> grad-ready hooks -> asynchronous all-reduce ->
finish_into(FusedAdam) -- or, with --mode single, the same four updates in ONE process that walks the ranks' shards in
turn and accumulates their gradients (what data parallelism must reproduce: the mean over ranks of the per-rank losses.
NOT the same as one call on the concatenated batch: compute_discriminator_losses / compute_generator_losses reconstruct
`b // 2` images of THEIR batch, swapping_autoencoder_model.py:121-124,192-194, exactly as each nn.DataParallel replica of
the reference does with its scattered shard, models/__init__.py:80-93).

    python tests/ddp_fullmodel_worker.py --mode rank --rank R --world W --port P --device cuda:0 --lib hip --out F

Both ranks may sit on the same GPU (gloo moves the buckets; RCCL refuses two ranks per device): the only multi-rank run of
the real library that one MI355X allows.  --lib oracle binds the CPU oracle behind the C-ABI (CPU suite)."""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

PER_RANK_BATCH = 4
CALLS = 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["rank", "single"], required=True)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--port", type=int, default=29611)
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--lib", choices=["hip", "oracle"], default="oracle")
    ap.add_argument("--out", required=True)
    ap.add_argument("--keep-grad", action="store_true", help="zero_grad(set_to_none=False) between the calls")
    return ap.parse_args()


def bind_library(kind):
    from swapping_autoencoder_pytorch_amd import hip_lib
    if kind == "oracle":
        hip_lib._LIB = hip_lib.SaeLibrary(os.path.join(ROOT, "oracle", "libsae_oracle.so"), prefix="oracle_", device_only=False)
    lib = hip_lib.get()
    return lib


def rng_state(device):
    import torch
    return (torch.get_rng_state(), torch.cuda.get_rng_state(device) if device.startswith("cuda") else None)


def set_rng_state(state, device):
    import torch
    torch.set_rng_state(state[0])
    if state[1] is not None:
        torch.cuda.set_rng_state(state[1], device)


def seed_of(call, rank):
    return 5000 + 100 * call + rank


def images_of(call, world, device):
    from param_recipe import uniform_images
    return uniform_images(PER_RANK_BATCH * world, 32, 600 + call).to(device)


def gloo_moves_device_tensors(device):
    """Whether this build's gloo all-reduces a device tensor; if not, all_reduce is staged through the host (test only)."""
    import torch
    import torch.distributed as dist
    if not device.startswith("cuda"):
        return True
    try:
        t = torch.ones(4, device=device)
        dist.all_reduce(t)
        return bool(t[0].item() == dist.get_world_size())
    except Exception:      # noqa: BLE001 -- any refusal means "stage it"
        return False


class _Done:
    def wait(self):
        return True


def stage_all_reduce_through_host():
    import torch.distributed as dist
    orig = dist.all_reduce

    def all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
        host = tensor.detach().cpu()
        orig(host, op=op, group=group)
        tensor.copy_(host)
        return _Done() if async_op else None

    dist.all_reduce = all_reduce


def build(device):
    import parity_common as P
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_optimizer import SwappingAutoencoderOptimizer
    opt, model, net = P.build_micro(device, batch_size=PER_RANK_BATCH)
    optimizer = SwappingAutoencoderOptimizer(model, fused_adam=True)
    return opt, model, net, optimizer


def run_rank(args):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(args.port)
    dist.init_process_group("gloo", rank=args.rank, world_size=args.world)
    staged = not gloo_moves_device_tensors(args.device)
    if staged:
        stage_all_reduce_through_host()
    lib = bind_library(args.lib)
    from swapping_autoencoder_pytorch_amd.grad_allreduce import GradAllReducer, broadcast_parameters
    opt, model, net, optimizer = build(args.device)
    broadcast_parameters(net)
    # small buckets: several collectives in flight per pass (the micro model fits one 32 MB bucket)
    optimizer.reducer_G = GradAllReducer(optimizer.Gparams, bucket_bytes=96 * 1024)
    optimizer.reducer_D = GradAllReducer(optimizer.Dparams, bucket_bytes=96 * 1024)
    assert optimizer.reducer_G.enabled and optimizer.reducer_D.enabled
    assert len(optimizer.reducer_G.buckets) > 2 and len(optimizer.reducer_D.buckets) > 2
    optimizer.reducer_G.profile = optimizer.reducer_D.profile = True          # bench.py --gpus N reports these per pass
    if args.keep_grad:
        for o in (optimizer.optimizer_G, optimizer.optimizer_D):
            zero = o.zero_grad
            o.zero_grad = lambda set_to_none=False, _z=zero: _z(set_to_none=False)
    losses, in_place = [], []
    for call in range(CALLS):
        torch.manual_seed(seed_of(call, args.rank))
        shard = images_of(call, args.world, args.device)[args.rank * PER_RANK_BATCH:(args.rank + 1) * PER_RANK_BATCH]
        out = optimizer.train_one_step({"real_A": shard}, call)
        losses.append({k: float(v) for k, v in out.items()})
        red = optimizer.reducer_D if call % 2 == 0 else optimizer.reducer_G
        spans = [(b.flat.data_ptr(), b.flat.data_ptr() + b.flat.numel() * 4) for b in red.buckets]
        views = sum(1 for p in red.params if p.grad is not None and any(lo <= p.grad.data_ptr() < hi for lo, hi in spans))
        in_place.append([views, sum(1 for p in red.params if p.grad is not None)])
    if args.device.startswith("cuda"):
        torch.cuda.synchronize()
    maps = sorted({l.split("/")[-1].strip() for l in open("/proc/self/maps") if "libsae" in l})
    torch.save({"state": {k: v.detach().cpu() for k, v in net.state_dict().items()}, "losses": losses, "in_place": in_place,
                "library": lib.path, "maps": maps, "staged_all_reduce": staged,
                "buckets": [len(optimizer.reducer_D.buckets), len(optimizer.reducer_G.buckets)],
                "allreduce": {"D": optimizer.reducer_D.summary(), "G": optimizer.reducer_G.summary()}}, args.out)
    dist.barrier()
    dist.destroy_process_group()


def run_single(args):
    import torch
    lib = bind_library(args.lib)
    opt, model, net, optimizer = build(args.device)
    assert not optimizer.reducer_D.enabled
    world, dev = args.world, args.device

    def accumulate(command, adam, factor, states, shards, extra=()):
        adam.zero_grad()
        for r in range(world):
            set_rng_state(states[r], dev)
            out = model(shards[r], *extra, command=command)
            losses = out[0] if isinstance(out, tuple) else out
            (sum(v.mean() for v in losses.values()) * (factor / world)).backward()
            states[r] = rng_state(dev)
        adam.step()

    d_iters = 0
    for call in range(CALLS):
        full = images_of(call, world, dev)
        shards = [full[r * PER_RANK_BATCH:(r + 1) * PER_RANK_BATCH] for r in range(world)]
        states = []
        for r in range(world):
            torch.manual_seed(seed_of(call, r))
            states.append(rng_state(dev))
        if call % 2 == 0:      # the first call of the driver is a discriminator step (optimizer.toggle_training_mode)
            optimizer.set_requires_grad(optimizer.Dparams, True)
            optimizer.set_requires_grad(optimizer.Gparams, False)
            d_iters += 1
            accumulate("compute_discriminator_losses", optimizer.optimizer_D, 1.0, states, shards)
            if d_iters % opt.R1_once_every == 0:
                accumulate("compute_R1_loss", optimizer.optimizer_D, float(opt.R1_once_every), states, shards)
        else:
            optimizer.set_requires_grad(optimizer.Dparams, False)
            optimizer.set_requires_grad(optimizer.Gparams, True)
            accumulate("compute_generator_losses", optimizer.optimizer_G, 1.0, states, shards, extra=(None, None))
    if dev.startswith("cuda"):
        torch.cuda.synchronize()
    torch.save({"state": {k: v.detach().cpu() for k, v in net.state_dict().items()}, "library": lib.path}, args.out)


if __name__ == "__main__":
    a = parse()
    (run_rank if a.mode == "rank" else run_single)(a)
