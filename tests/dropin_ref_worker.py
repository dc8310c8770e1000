"""Subprocess bodies of tests/test_dropin_train.py (kept out of the pytest process so the reference's packages
never leak into other tests).  Both run the reference's UNMODIFIED code from /root/reference on top of
dropin's pre-seeded modules, with the CPU oracle bound behind the C-ABI (test-only back end: the product
itself refuses CPU tensors)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

MICRO_ARGV = ["--dataset_mode", "synthetic", "--num_gpus", "0", "--batch_size", "4", "--crop_size", "32", "--load_size", "32",
              "--netE_num_downsampling_sp", "2", "--patch_size", "32", "--patch_num_crops", "2", "--global_code_ch", "64",
              "--netE_scale_capacity", "0.25", "--netG_scale_capacity", "0.125", "--netD_scale_capacity", "0.03125",
              "--netPatchD_scale_capacity", "0.5", "--netPatchD_max_nc", "32", "--R1_once_every", "2", "--dataroot", "."]


def _prepare():
    import torch
    from swapping_autoencoder_pytorch_amd import dropin, hip_lib
    from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary
    hip_lib._LIB = SaeLibrary(os.path.join(ROOT, "oracle", "libsae_oracle.so"), prefix="oracle_", device_only=False)
    sys.path.insert(0, REF)
    dropin.install_missing_dependency_stubs()
    dropin.preseed()
    dropin.inject_synthetic_dataset()
    torch.cuda.synchronize = lambda *a, **k: None      # util/iter_counter.py calls it unconditionally; no GPU here
    return dropin


def train_end_to_end(ckpt_dir):
    """python train.py ... : option parser, data loader, model, optimizer, loss log, checkpoint."""
    import runpy
    dropin = _prepare()
    import data
    data.create_dataset = dropin.wrap_dataloader(data.create_dataset)    # as dropin.main does; a pass-through without a GPU
    sys.argv = ["train.py", "--name", "dropin_e2e", "--total_nimgs", "24", "--checkpoints_dir", ckpt_dir, "--print_freq", "8",
                "--display_freq", "100000", "--save_freq", "100000", "--evaluation_freq", "100000"] + MICRO_ARGV
    runpy.run_path(os.path.join(REF, "train.py"), run_name="__main__")


def _ddp_rank(rank, world, port, out_dir, adam="torch"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dropin = _prepare()
    if adam == "fused":       # as dropin.main does: the all-reduce then hands its buckets to the multi-tensor Adam (finish_into)
        from swapping_autoencoder_pytorch_amd.fused_adam import FusedAdam
        torch.optim.Adam = FusedAdam
    import models
    import optimizers
    from options import TrainOptions
    from param_recipe import fill_params, uniform_images
    sys.argv = ["train.py", "--name", "dropin_ddp", "--checkpoints_dir", out_dir] + MICRO_ARGV
    opt = TrainOptions().parse(save=False) if "save" in TrainOptions.parse.__code__.co_varnames else TrainOptions().parse()
    model = models.create_model(opt)
    fill_params(model.singlegpu_model, seed=10 + rank)            # deliberately different replicas
    optimizer = dropin.attach_gradient_allreduce(optimizers.create_optimizer(opt, model))
    assert type(optimizer.optimizer_D).__name__ == ("FusedAdam" if adam == "fused" else "Adam")
    state0 = {k: v.clone() for k, v in model.singlegpu_model.state_dict().items()}
    for it in range(4):                                            # D, G, D (+R1), G with DIFFERENT data per rank
        torch.manual_seed(500 + 10 * it + rank)
        optimizer.train_one_step({"real_A": uniform_images(4, 32, 900 + 10 * it + rank)}, it)
    optimizer.save(0)
    torch.save({"start": state0, "end": model.singlegpu_model.state_dict()}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def ddp(out_dir, port, adam="torch"):
    import torch.multiprocessing as mp
    mp.spawn(_ddp_rank, args=(2, int(port), out_dir, adam), nprocs=2, join=True)




def aten_cpu_path_pin():
    """oracle/aten_cpu_path.py against the reference's own Discriminator, UpsamplingResnetBlock, StyleGAN2ResnetEncoder and
    StyleGAN2ResnetGenerator on their CPU path (unmodified modules)."""
    import torch
    import ref_shims
    ref_shims.install()                                  # fake torch.version.cuda: the reference takes its native path
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import aten_cpu_path as A
    from models.networks.stylegan2_layers import Discriminator
    torch.manual_seed(0)
    ref = Discriminator(64, 2)
    mine = A.DiscriminatorCPU(64, 2)
    rp, mp_ = list(ref.parameters()), list(mine.parameters())
    assert len(rp) == len(mp_), (len(rp), len(mp_))
    with torch.no_grad():
        for a, b in zip(rp, mp_):
            assert a.shape == b.shape, (a.shape, b.shape)
            a.copy_(torch.randn(a.shape) * (1.0 if a.dim() > 1 else 0.1))
            b.copy_(a)
    x = torch.rand(3, 3, 64, 64) * 2 - 1
    yr, ym = ref(x), mine(x)
    err = float((yr - ym).abs().max() / yr.abs().max())
    assert err < 1e-5, err
    gr = torch.autograd.grad(yr.sum(), rp)
    gm = torch.autograd.grad(ym.sum(), mp_)
    for a, b in zip(gr, gm):
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max() + 1e-12)
    # one upsampling block of the generator, explicit noise maps
    from models.networks.generator import UpsamplingResnetBlock
    refb = UpsamplingResnetBlock(12, 8, 16, use_noise=True)
    mineb = A.UpsamplingResnetBlockCPU(12, 8, 16)
    rp, mp_ = list(refb.parameters()), list(mineb.parameters())
    assert [tuple(a.shape) for a in rp] == [tuple(b.shape) for b in mp_], ([a.shape for a in rp], [b.shape for b in mp_])
    with torch.no_grad():
        for a, b in zip(rp, mp_):
            a.copy_(torch.randn(a.shape) * (1.0 if a.dim() > 1 else 0.3))
            b.copy_(a)
    xb, st = torch.randn(3, 12, 8, 8), torch.randn(3, 16)
    z1, z2 = torch.randn(3, 1, 16, 16), torch.randn(3, 1, 16, 16)
    refb.conv1.noise.fixed_noise, refb.conv2.noise.fixed_noise = z1, z2
    yr, ym = refb(xb, st), mineb(xb, st, z1, z2)
    errb = float((yr - ym).abs().max() / yr.abs().max())
    assert errb < 1e-5, errb
    gr = torch.autograd.grad(yr.square().sum(), rp)
    gm = torch.autograd.grad(ym.square().sum(), mp_)
    for a, b in zip(gr, gm):
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max() + 1e-12)
    # the encoder and the generator of BASELINE.json's config 1 (tiny32: 32 x 32, B = 4), the reconstruction loss through both
    from types import SimpleNamespace
    from models.networks.encoder import StyleGAN2ResnetEncoder
    from models.networks.generator import StyleGAN2ResnetGenerator
    erre = 0.0
    for n_sp in (2, 3):        # 3: the first upsampling block keeps its channel count, its skip is the Identity (generator.py:48-49)
        opt = SimpleNamespace(netE_num_downsampling_sp=n_sp, netE_num_downsampling_gl=1, netE_nc_steepness=2.0, netE_scale_capacity=0.25,
                              spatial_code_ch=8, global_code_ch=64, use_antialias=True, num_classes=0, netG_scale_capacity=0.125,
                              netG_num_base_resnet_layers=2, netG_use_noise=True, netG_resnet_ch=256, lambda_patchD=1.0)
        rE, rG = StyleGAN2ResnetEncoder(opt), StyleGAN2ResnetGenerator(opt)
        mE, mG = A.EncoderCPU(opt), A.GeneratorCPU(opt)
        for r, m in ((rE, mE), (rG, mG)):
            rp, mp_ = list(r.parameters()), list(m.parameters())
            assert [tuple(a.shape) for a in rp] == [tuple(b.shape) for b in mp_]
            with torch.no_grad():
                for a, b in zip(rp, mp_):
                    if a.dim() == 1 or tuple(a.shape) == (1, 3, 1, 1):
                        a.normal_(0.0, 0.3)
                    b.copy_(a)

        def reconstruction(E, G, img):
            torch.manual_seed(9)                       # the noise maps are drawn in the same order by both
            out = G(*E(img))
            loss = (out - img).abs().mean()
            return out, loss, torch.autograd.grad(loss, list(E.parameters()) + list(G.parameters()))

        img = torch.rand(4, 3, 32, 32) * 2 - 1
        yr, lr_, gr = reconstruction(rE, rG, img)
        ym, lm, gm = reconstruction(mE, mG, img)
        erre = max(erre, float((yr - ym).abs().max() / yr.abs().max()))
        assert erre < 1e-5 and abs(float(lr_) - float(lm)) < 1e-6, (erre, float(lr_), float(lm))
        for a, b in zip(gr, gm):
            assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max() + 1e-12)
    print("aten-cpu-path-pinned", err, errb, erre)


def preseed_level(level):
    """The reference's driver (options, models.create_model, optimizers.create_optimizer, train_one_step) on top of
    dropin.preseed(level) + patch_util(), oracle behind the C-ABI: deviations from the golden loss dictionaries."""
    import json
    import torch
    import ref_shims
    ref_shims.install()
    from swapping_autoencoder_pytorch_amd import dropin, hip_lib
    from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary
    hip_lib._LIB = SaeLibrary(os.path.join(ROOT, "oracle", "libsae_oracle.so"), prefix="oracle_", device_only=False)
    dropin.preseed(level)
    dropin.patch_util()
    import models
    import optimizers
    import util
    import parity_common as P
    from options import TrainOptions
    from param_recipe import MICRO, fill_params, uniform_images
    sys.argv = ["train.py", "--name", "level", "--dataset_mode", "imagefolder"]
    opt = TrainOptions().gather_options()
    opt.isTrain = True
    for k, v in MICRO.items():
        setattr(opt, k, v)
    model = models.create_model(opt)
    wrapped = dropin.wrap_reference_r1()
    net = model.singlegpu_model
    fill_params(net, seed=3)
    optimizer = optimizers.create_optimizer(opt, model)
    _, info = P.golden()
    worst = 0.0
    with P.cpu_random_stream("cpu"):
        for it in range(4):
            torch.manual_seed(1000 + it)
            losses = optimizer.train_one_step({"real_A": uniform_images(4, 32, 600 + it)}, it)
            want = info["micro_steps"]["step%d" % it]
            assert set(losses) == set(want), (sorted(losses), sorted(want))
            worst = max(worst, max(abs(float(losses[k]) - v) / max(1.0, abs(v)) for k, v in want.items()))
    from models.base_model import BaseModel
    import swapping_autoencoder_pytorch_amd.swapping_autoencoder_model as mirror
    print("LEVEL-REPORT " + json.dumps({
        "level": level, "worst_loss_dev": worst, "layer_module": type(net.E.FromRGB).__module__,
        "encoder_module": type(net.E).__module__, "model_is_ours": isinstance(net, mirror.SwappingAutoencoderModel),
        "model_is_basemodel": isinstance(net, BaseModel), "normalize_module": util.normalize.__module__,
        "crop_module": util.apply_random_crop.__module__, "r1_wrapped": bool(wrapped),
        "optimizer_module": type(optimizer).__module__, "wrapper": type(model).__name__}))


if __name__ == "__main__":
    {"train": train_end_to_end, "ddp": ddp, "aten_pin": aten_cpu_path_pin, "preseed_level": preseed_level}[sys.argv[1]](*sys.argv[2:])
