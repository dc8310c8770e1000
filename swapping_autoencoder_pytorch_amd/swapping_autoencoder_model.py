"""SwappingAutoencoderModel: E -> G -> D / Dpatch graphs and losses of one training step.

Host-side mirror of the reference's models/swapping_autoencoder_model.py (command names, loss
keys, loss weights, RNG consumption order and checkpoint layout are the contract; cited per
method) plus the command dispatch of models/base_model.py:114-123.  All arithmetic below the
module boundary runs on the MI355X kernels."""
import os

import torch

from . import loss, util
from . import networks
from .streams import side_branch
from .stylegan2_op import input_grads_only


class SwappingAutoencoderModel(torch.nn.Module):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        # swapping_autoencoder_model.py:10-24
        parser.add_argument("--spatial_code_ch", default=8, type=int)
        parser.add_argument("--global_code_ch", default=2048, type=int)
        parser.add_argument("--lambda_R1", default=10.0, type=float)
        parser.add_argument("--lambda_patch_R1", default=1.0, type=float)
        parser.add_argument("--lambda_L1", default=1.0, type=float)
        parser.add_argument("--lambda_GAN", default=1.0, type=float)
        parser.add_argument("--lambda_PatchGAN", default=1.0, type=float)
        parser.add_argument("--patch_min_scale", default=1 / 8, type=float)
        parser.add_argument("--patch_max_scale", default=1 / 4, type=float)
        parser.add_argument("--patch_num_crops", default=8, type=int)
        parser.add_argument("--patch_use_aggregation", type=util.str2bool, default=True)
        return parser

    def __init__(self, opt):
        # (not super().__init__(): under the drop-in runner this class is combined with the reference's BaseModel, whose
        # constructor takes `opt` and pins 'cuda:0', base_model.py:11-13)
        torch.nn.Module.__init__(self)
        self.opt = opt
        # the reference hard-codes "cuda:0" (base_model.py:13); with one process per GPU that is
        # this rank's current device
        if opt.num_gpus > 0:
            self.device = torch.device("cuda", torch.cuda.current_device())
        else:
            self.device = torch.device("cpu")

    def initialize(self):
        """swapping_autoencoder_model.py:26-48"""
        opt = self.opt
        self.E = networks.create_network(opt, opt.netE, "encoder")
        self.G = networks.create_network(opt, opt.netG, "generator")
        if opt.lambda_GAN > 0.0:
            self.D = networks.create_network(opt, opt.netD, "discriminator")
        if opt.lambda_PatchGAN > 0.0:
            self.Dpatch = networks.create_network(opt, opt.netPatchD, "patch_discriminator")
        # iteration count of the discriminator (lazy R1); part of the checkpoint
        self.register_buffer("num_discriminator_iters", torch.zeros(1, dtype=torch.long))
        self.l1_loss = torch.nn.L1Loss()
        if (not opt.isTrain) or opt.continue_train:
            self.load()
        if opt.num_gpus > 0:
            self.to(self.device)

    def per_gpu_initialize(self):
        pass

    # ---- command dispatch (base_model.py:114-123) --------------------------------------------
    def forward(self, *args, command=None, **kwargs):
        if command is None:
            raise ValueError(command)
        method = getattr(self, command)
        assert callable(method), "[%s] is not a method of %s" % (command, type(self).__name__)
        return method(*args, **kwargs)

    # ---- pieces -------------------------------------------------------------------------------
    def swap(self, x):
        """Exchange neighbours (0<->1, 2<->3, ...) of the minibatch (:53-60)."""
        assert x.shape[0] % 2 == 0, "Minibatch size must be a multiple of 2"
        paired = x.view(x.shape[0] // 2, 2, *x.shape[1:])
        return torch.flip(paired, [1]).view(x.shape)

    def get_random_crops(self, x, crop_window=None):
        """:84-93"""
        return util.apply_random_crop(x, self.opt.patch_size, (self.opt.patch_min_scale, self.opt.patch_max_scale),
                                      num_crops=self.opt.patch_num_crops)

    def compute_image_discriminator_losses(self, real, rec, mix):
        """:62-82"""
        if self.opt.lambda_GAN == 0.0:
            return {}
        lam = self.opt.lambda_GAN
        # one pass over the concatenated batch instead of three (D has no cross-sample statistics,
        # so the per-sample predictions are unchanged): fewer, larger launches, and the three
        # weight-gradient accumulations of every layer become one
        pred_real, pred_rec, pred_mix = torch.split(self.D(torch.cat([real, rec, mix], 0)),
                                                    [real.size(0), rec.size(0), mix.size(0)])
        return {
            "D_real": loss.gan_loss(pred_real, should_be_classified_as_real=True) * lam,
            "D_rec": loss.gan_loss(pred_rec, should_be_classified_as_real=False) * (0.5 * lam),
            "D_mix": loss.gan_loss(pred_mix, should_be_classified_as_real=False) * (0.5 * lam),
        }

    def compute_patch_discriminator_losses(self, real, mix):
        """:95-114 (crop order: reference patches, target patches, mix patches)"""
        agg = self.opt.patch_use_aggregation
        # crops are drawn in the reference's order (reference patches, target patches, mix patches),
        # then the three feature extractions run as one pass over the concatenated crops
        crops = [self.get_random_crops(real), self.get_random_crops(real), self.get_random_crops(mix)]
        b, t = crops[0].shape[:2]
        feats = self.Dpatch.extract_features(torch.cat(crops, 0))
        real_feat, target_feat, mix_feat = torch.split(feats, [c.size(0) * t for c in crops])
        if agg:   # extract_features(aggregate=True): mean over the crops of an image (patch_discriminator.py:155-157)
            real_feat = real_feat.view(b, t, *real_feat.shape[1:]).mean(1, keepdim=True)
            real_feat = real_feat.expand(-1, t, -1, -1, -1).flatten(0, 1)
        lam = self.opt.lambda_PatchGAN
        return {
            "PatchD_real": loss.gan_loss(self.Dpatch.discriminate_features(real_feat, target_feat), True) * lam,
            "PatchD_mix": loss.gan_loss(self.Dpatch.discriminate_features(real_feat, mix_feat), False) * lam,
        }

    # ---- commands -----------------------------------------------------------------------------
    def compute_discriminator_losses(self, real):
        """:116-136"""
        self.num_discriminator_iters.add_(1)
        sp, gl = self.E(real)
        b = real.size(0)
        assert b % 2 == 0, "Batch size must be even on each GPU."
        # the two generator passes, then the two discriminators, are independent pairs: the first of each pair is enqueued on
        # the side stream (streams.py), in the order the reference runs them (the random draws stay in its order)
        with side_branch(sp, gl) as br:
            rec = self.G(sp[:b // 2], gl[:b // 2])     # GAN loss on half of the reconstructions
        mix = self.G(self.swap(sp), gl)
        br.join(rec)
        if self.opt.lambda_PatchGAN > 0.0:
            with side_branch(real, mix) as br:
                patch_losses = self.compute_patch_discriminator_losses(real, mix)
        losses = self.compute_image_discriminator_losses(real, rec, mix)
        if self.opt.lambda_PatchGAN > 0.0:
            br.join(*patch_losses.values())
            losses.update(patch_losses)
        return losses, {}, sp.detach(), gl.detach()

    def compute_R1_loss(self, real):
        """:138-185 — gradient penalties on D (w.r.t. the image) and Dpatch (w.r.t. both crops)."""
        opt = self.opt
        losses = {}
        br = None
        if opt.lambda_R1 > 0.0:
            # (the image penalty on the side stream, the patch penalty on the current one: streams.py; only the patch branch
            # draws random numbers, so their order is the reference's)
            with side_branch(real) as br:
                real.requires_grad_()
                pred_real = self.D(real).sum()
                with input_grads_only():     # only d(pred)/d(image) is asked for: no weight-gradient kernels
                    grad_real, = torch.autograd.grad(outputs=pred_real, inputs=[real], create_graph=True, retain_graph=True)
                grad_penalty = grad_real.pow(2).sum(list(range(1, grad_real.ndim))) * (opt.lambda_R1 * 0.5)
        else:
            grad_penalty = 0.0

        if opt.lambda_patch_R1 > 0.0:
            real_crop = self.get_random_crops(real).detach().requires_grad_()
            target_crop = self.get_random_crops(real).detach().requires_grad_()
            real_feat = self.Dpatch.extract_features(real_crop, aggregate=opt.patch_use_aggregation)
            target_feat = self.Dpatch.extract_features(target_crop)
            pred = self.Dpatch.discriminate_features(real_feat, target_feat).sum()
            with input_grads_only():
                g_real, g_target = torch.autograd.grad(outputs=pred, inputs=[real_crop, target_crop], create_graph=True,
                                                       retain_graph=True)
            dims = list(range(1, g_real.ndim))
            grad_crop_penalty = (g_real.pow(2).sum(dims) + g_target.pow(2).sum(dims)) * (0.5 * opt.lambda_patch_R1 * 0.5)
        else:
            grad_crop_penalty = 0.0

        if br is not None:
            br.join(grad_penalty)
        losses["D_R1"] = grad_penalty + grad_crop_penalty
        return losses

    def compute_generator_losses(self, real, sp_ma, gl_ma):
        """:187-231"""
        opt = self.opt
        losses, metrics = {}, {}
        b = real.size(0)
        sp, gl = self.E(real)
        with side_branch(sp, gl, real) as br:        # the reconstruction pass next to the hybrid pass (streams.py)
            rec = self.G(sp[:b // 2], gl[:b // 2])
            metrics["L1_dist"] = self.l1_loss(rec, real[:b // 2])
            if opt.lambda_L1 > 0.0:
                losses["G_L1"] = metrics["L1_dist"] * opt.lambda_L1
        sp_mix = self.swap(sp)

        if opt.crop_size >= 1024:   # memory saving of the reference: half the mix batch
            real, gl, sp_mix = real[b // 2:], gl[b // 2:], sp_mix[b // 2:]

        mix = self.G(sp_mix, gl)
        br.join(rec, metrics["L1_dist"], losses.get("G_L1"))

        if opt.lambda_PatchGAN > 0.0:                # the patch discriminator next to the image discriminator
            with side_branch(real, mix) as br:
                real_feat = self.Dpatch.extract_features(self.get_random_crops(real),
                                                         aggregate=opt.patch_use_aggregation).detach()
                mix_feat = self.Dpatch.extract_features(self.get_random_crops(mix))
                g_mix = loss.gan_loss(self.Dpatch.discriminate_features(real_feat, mix_feat), True) * opt.lambda_PatchGAN

        if opt.lambda_GAN > 0.0:
            pred_rec, pred_mix = torch.split(self.D(torch.cat([rec, mix], 0)), [rec.size(0), mix.size(0)])
            losses["G_GAN_rec"] = loss.gan_loss(pred_rec, True) * (opt.lambda_GAN * 0.5)
            losses["G_GAN_mix"] = loss.gan_loss(pred_mix, True) * (opt.lambda_GAN * 1.0)

        if opt.lambda_PatchGAN > 0.0:
            br.join(g_mix)
            losses["G_mix"] = g_mix

        return losses, metrics

    def encode(self, image, extract_features=False):
        return self.E(image, extract_features=extract_features)

    def decode(self, spatial_code, global_code):
        return self.G(spatial_code, global_code)

    def get_visuals_for_snapshot(self, real):
        """:233-243 — reconstruction and swap of a few images for the training snapshot.  The reference paints the
        structure code with a PCA to three channels (util/util.py:231-253, sklearn); here the layout is the first
        three principal directions computed with torch.pca_lowrank — same picture up to sign/rotation of the PCA."""
        if self.opt.isTrain:
            real = real[:2] if self.opt.num_gpus > 1 else real[:4]
        sp, gl = self.E(real)
        rec = self.G(sp, gl)
        mix = self.G(sp, self.swap(gl))
        return {"real": real, "layout": self._visualize_spatial_code(sp, real.shape[-2:]), "rec": rec, "mix": mix}

    @staticmethod
    def _visualize_spatial_code(sp, size):
        b, c, h, w = sp.shape
        if c <= 2:
            img = sp.repeat(1, 3, 1, 1)[:, :3]
        elif c == 3:
            img = sp
        else:
            flat = sp.detach().permute(0, 2, 3, 1).reshape(-1, c).float().cpu()
            flat = flat - flat.mean(0, keepdim=True)
            _, _, v = torch.pca_lowrank(flat, q=3, center=False)
            z = (flat @ v[:, :3]).reshape(b, h, w, 3).permute(0, 3, 1, 2)
            img = ((z - z.min()) / (z.max() - z.min() + 1e-12) * 2 - 1).to(sp.device)
        return torch.nn.functional.interpolate(img, size=tuple(size), mode="bilinear", align_corners=False)

    def fix_noise(self, sample_image=None):
        """:245-257"""
        if sample_image is not None:
            sp, gl = self.E(sample_image)
            self.G(sp, gl)
        return self.G.fix_and_gather_noise_parameters()

    def get_parameters_for_mode(self, mode):
        """:266-275 — "generator" = G + E, "discriminator" = D + Dpatch."""
        if mode == "generator":
            return list(self.G.parameters()) + list(self.E.parameters())
        if mode == "discriminator":
            params = []
            if self.opt.lambda_GAN > 0.0:
                params += list(self.D.parameters())
            if self.opt.lambda_PatchGAN > 0.0:
                params += list(self.Dpatch.parameters())
            return params
        raise ValueError(mode)

    # ---- checkpoints (base_model.py:33-112): model weights only, same file names -------------
    def _ckpt_dir(self, name):
        return os.path.join(self.opt.checkpoints_dir, name)

    def save(self, total_steps_so_far):
        """base_model.py:33-48.  One process per GPU: every rank holds the same weights, so rank 0 alone writes
        (N ranks racing on the file and on the remove/symlink of latest_checkpoint.pth can tear it); the file is
        written under a temporary name and renamed, the others wait at a barrier."""
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not multi or dist.get_rank() == 0:
            savedir = self._ckpt_dir(self.opt.name)
            os.makedirs(savedir, exist_ok=True)
            checkpoint_name = "%dk_checkpoint.pth" % (total_steps_so_far // 1000)
            final = os.path.join(savedir, checkpoint_name)
            torch.save(self.state_dict(), final + ".tmp")
            os.replace(final + ".tmp", final)
            sympath = os.path.join(savedir, "latest_checkpoint.pth")
            tmplink = sympath + ".tmp"
            if os.path.lexists(tmplink):
                os.remove(tmplink)
            os.symlink(checkpoint_name, tmplink)
            os.replace(tmplink, sympath)
        if multi:
            dist.barrier()

    def load(self):
        opt = self.opt
        name = opt.pretrained_name if (opt.isTrain and opt.pretrained_name is not None) else opt.name
        path = os.path.join(self._ckpt_dir(name), "%s_checkpoint.pth" % opt.resume_iter)
        if not os.path.exists(path):
            assert opt.isTrain, "In test mode, the checkpoint file %s must exist" % path
            print("checkpoint %s does not exist: training starts from scratch" % path)
            return
        state = torch.load(path, map_location=str(self.device))
        own = self.state_dict()
        for key, dst in own.items():
            if not opt.isTrain and (key.startswith("D.") or key.startswith("Dpatch.")):
                continue
            if key not in state:
                print("Key %s does not exist in checkpoint. Skipping..." % key)
                continue
            src = state[key]
            if src.shape != dst.shape:   # the reference asks interactively (:76-108); never block a job
                print("Shape mismatch for %s: checkpoint %s vs model %s. Skipping..." % (key, tuple(src.shape), tuple(dst.shape)))
                continue
            dst.copy_(src)
        print("checkpoint loaded from %s" % path)


class ModelWrapper:
    """What ``models.create_model`` returns (models/__init__.py:57-93).  The reference wraps the
    model in single-process ``nn.DataParallel``; here there is one process per GPU, the wrapper
    is a plain pass-through and gradient exchange is RCCL all-reduce (grad_allreduce.py)."""

    def __init__(self, opt, model):
        self.opt = opt
        self.singlegpu_model = model
        model(command="per_gpu_initialize")

    def get_parameters_for_mode(self, mode):
        return self.singlegpu_model.get_parameters_for_mode(mode)

    def save(self, total_steps_so_far):
        self.singlegpu_model.save(total_steps_so_far)

    def __call__(self, *args, **kwargs):
        return self.singlegpu_model(*args, **kwargs)


def create_model(opt):
    model = SwappingAutoencoderModel(opt)
    model.initialize()
    return ModelWrapper(opt, model)
