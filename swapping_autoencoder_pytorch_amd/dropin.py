"""Run the reference's UNMODIFIED ``train.py`` (or ``test.py``) on the MI355X hot path.

    python -m swapping_autoencoder_pytorch_amd.dropin /path/to/swapping-autoencoder-pytorch train.py --name ... 

How: every import of the reference is an absolute import with the repo root on
``sys.path`` and Python consults ``sys.modules`` before the file system, so pre-seeding

    models.networks.stylegan2_op            (+ .upfirdn2d, .fused_act)   <- stylegan2_op of this package
    models.networks.stylegan2_layers                                     <- stylegan2_layers of this package

puts the gfx950 kernels underneath the reference's own networks, model, optimizer and training
loop.  ``util.is_custom_kernel_supported`` (util/util.py:432-436, which raises on ROCm) is never
reached because the two modules that called it are replaced.

Multi-GPU: the reference drives ``nn.DataParallel`` from one process (models/__init__.py:80).
Launched under ``python -m torch.distributed.run --nproc-per-node N``, this runner instead gives
every rank one GPU (``--num_gpus 1`` semantics: DataParallel over a single device is a pass-through),
broadcasts rank 0's initial weights and attaches the bucketed RCCL gradient all-reduce
(grad_allreduce.GradAllReducer) to the reference optimizer's two Adam instances through optimizer
step hooks — train.py and the optimizer source stay byte-identical.
"""
import os
import runpy
import sys


def preseed():
    """Install this package's operator and layer modules under the reference's import names."""
    from . import stylegan2_layers, stylegan2_op
    from .stylegan2_op import fused_act, upfirdn2d
    sys.modules["models.networks.stylegan2_op"] = stylegan2_op
    sys.modules["models.networks.stylegan2_op.upfirdn2d"] = upfirdn2d
    sys.modules["models.networks.stylegan2_op.fused_act"] = fused_act
    sys.modules["models.networks.stylegan2_layers"] = stylegan2_layers


def attach_gradient_allreduce(optimizer):
    """Give a reference SwappingAutoencoderOptimizer data-parallel semantics across ranks."""
    import torch.distributed as dist
    from .grad_allreduce import GradAllReducer, broadcast_parameters
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return optimizer
    broadcast_parameters(optimizer.model.singlegpu_model)
    for params, opt in ((optimizer.Gparams, optimizer.optimizer_G), (optimizer.Dparams, optimizer.optimizer_D)):
        reducer = GradAllReducer(params)
        reducer.arm()
        opt.register_step_pre_hook(lambda o, a, k, r=reducer: r.finish())
        opt.register_step_post_hook(lambda o, a, k, r=reducer: r.arm())
    return optimizer


def _init_distributed():
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return
    import torch.distributed as dist
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    # the reference addresses "cuda:0" literally: make that this rank's GPU
    os.environ["HIP_VISIBLE_DEVICES"] = os.environ.get("HIP_VISIBLE_DEVICES", str(local_rank))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 2:
        raise SystemExit("usage: python -m swapping_autoencoder_pytorch_amd.dropin REFERENCE_ROOT SCRIPT.py [args...]")
    ref_root, script = os.path.abspath(argv[0]), argv[1]
    sys.path.insert(0, ref_root)
    from . import hip_lib
    hip_lib.get()               # fail loudly before anything else if the HIP library is not built
    preseed()
    _init_distributed()
    import optimizers           # the reference's package (imports its models on the pre-seeded layers)
    _create = optimizers.create_optimizer
    optimizers.create_optimizer = lambda opt, model: attach_gradient_allreduce(_create(opt, model))
    sys.argv = [script] + argv[2:]
    runpy.run_path(os.path.join(ref_root, script), run_name="__main__")


if __name__ == "__main__":
    main()
