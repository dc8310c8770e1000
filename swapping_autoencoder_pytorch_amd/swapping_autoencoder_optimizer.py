"""SwappingAutoencoderOptimizer: the alternating D / G step driver.

Host-side mirror of optimizers/swapping_autoencoder_optimizer.py (cited per method): two Adam
optimisers (lr 0.002, betas (0, 0.99)), the lazy-regularisation correction c = 16/17 on the
discriminator's, D and G steps alternating call by call starting with D, the R1 penalty every
``R1_once_every`` discriminator iterations scaled by that interval.

Multi-GPU: one process per GPU, each with its own ``opt.batch_size`` images per call (the reference's
``batch_size`` is the GLOBAL batch that nn.DataParallel scatters; here it is the per-rank batch and the global
batch is world_size times it); when torch.distributed is initialised the gradients of the group
being trained are averaged with RCCL all-reduce, bucketed and launched from grad-ready hooks so
they overlap the rest of backward (grad_allreduce.GradAllReducer) — the MI355X replacement of the
reference's nn.DataParallel replicate/scatter/gather/reduce (models/__init__.py:75-93)."""
import torch

from . import hip_graph, util
from .fused_adam import FusedAdam
from .grad_allreduce import GradAllReducer


class SwappingAutoencoderOptimizer:
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--lr", default=0.002, type=float)
        parser.add_argument("--beta1", default=0.0, type=float)
        parser.add_argument("--beta2", default=0.99, type=float)
        parser.add_argument("--R1_once_every", default=16, type=int,
                            help="lazy R1 regularization: the R1 loss is computed once every this many D iterations")
        return parser

    def __init__(self, model, fused_adam=None):
        self.opt = opt = model.opt
        self.model = model
        self.train_mode_counter = 0
        self.discriminator_iter_counter = 0
        self.Gparams = model.get_parameters_for_mode("generator")
        self.Dparams = model.get_parameters_for_mode("discriminator")
        # the multi-tensor HIP Adam (fused_adam.FusedAdam) wherever the parameters live on a GPU; torch.optim.Adam
        # (what the reference constructs, :34-42) otherwise, i.e. in the CPU test-suite on the oracle back end
        on_gpu = len(self.Gparams) > 0 and self.Gparams[0].is_cuda
        Adam = FusedAdam if (fused_adam if fused_adam is not None else on_gpu) else torch.optim.Adam
        self.optimizer_G = Adam(self.Gparams, lr=opt.lr, betas=(opt.beta1, opt.beta2))
        # StyleGAN2 appendix B: compensate for regularising only every k-th iteration (:36-42)
        c = opt.R1_once_every / (1 + opt.R1_once_every)
        self.optimizer_D = Adam(self.Dparams, lr=opt.lr * c, betas=(opt.beta1 ** c, opt.beta2 ** c))
        # one reducer per parameter group: the trainable set flips every call
        self.reducer_G = GradAllReducer(self.Gparams)
        self.reducer_D = GradAllReducer(self.Dparams)
        # hipGraph replay of the two calls (hip_graph.py): single-rank GPU process with the fused Adam; SAE_HIP_GRAPH=0: eager
        self.graphs = None
        if hip_graph.wanted(self.Gparams + self.Dparams, [self.optimizer_G, self.optimizer_D]):
            self.graphs = hip_graph.StepGraphs()
            self.optimizer_G.use_device_steps()
            self.optimizer_D.use_device_steps()

    def set_requires_grad(self, params, requires_grad):
        for p in params:
            p.requires_grad_(requires_grad)

    def prepare_images(self, data_i):
        return data_i["real_A"]

    def toggle_training_mode(self):
        """:54-57 — note the reference's naming: the call that returns "generator" runs the D step."""
        modes = ["discriminator", "generator"]
        self.train_mode_counter = (self.train_mode_counter + 1) % len(modes)
        return modes[self.train_mode_counter]

    def train_one_step(self, data_i, total_steps_so_far):
        images = self.prepare_images(data_i)
        if self.toggle_training_mode() == "generator":
            losses = self.train_discriminator_one_step(images)
        else:
            losses = self.train_generator_one_step(images)
        return util.to_numpy(losses)

    def _backward_and_step(self, total_loss, optimizer, reducer):
        reducer.arm()
        total_loss.backward()
        if isinstance(optimizer, FusedAdam):
            reducer.finish_into(optimizer)   # per-bucket: wait for its all-reduce, Adam reads the bucket in place
        else:
            reducer.finish()                 # waits for the in-flight buckets, grads now hold the global mean
            optimizer.step()

    def _generator_call(self, images):
        """zero_grad -> losses -> backward -> Adam of :67-79: what a hipGraph of the generator call holds (hip_graph.py)"""
        self.optimizer_G.zero_grad()
        g_losses, g_metrics = self.model(images, None, None, command="compute_generator_losses")
        g_loss = sum(v.mean() for v in g_losses.values())
        self._backward_and_step(g_loss, self.optimizer_G, self.reducer_G)
        return g_losses, g_metrics

    def _discriminator_call(self, images):
        """the same for :81-95 (without the lazy-R1 call, which stays eager)"""
        self.optimizer_D.zero_grad()
        d_losses, d_metrics, sp, gl = self.model(images, command="compute_discriminator_losses")
        d_loss = sum(v.mean() for v in d_losses.values())
        self._backward_and_step(d_loss, self.optimizer_D, self.reducer_D)
        return d_losses, d_metrics, sp.detach(), gl.detach()

    def _run(self, key, body, images):
        if self.graphs is not None:
            return self.graphs.run(key, images, body)
        return body(images)

    def train_generator_one_step(self, images):
        """:67-79"""
        self.set_requires_grad(self.Dparams, False)
        self.set_requires_grad(self.Gparams, True)
        g_losses, g_metrics = self._run("generator", self._generator_call, images)
        g_losses.update(g_metrics)
        return g_losses

    def train_discriminator_one_step(self, images):
        """:81-111"""
        opt = self.opt
        if opt.lambda_GAN == 0.0 and opt.lambda_PatchGAN == 0.0:
            return {}
        self.set_requires_grad(self.Dparams, True)
        self.set_requires_grad(self.Gparams, False)
        self.discriminator_iter_counter += 1
        d_losses, d_metrics, self.previous_sp, self.previous_gl = self._run("discriminator", self._discriminator_call, images)

        needs_R1 = opt.lambda_R1 > 0.0 or opt.lambda_patch_R1 > 0.0
        if needs_R1 and self.discriminator_iter_counter % opt.R1_once_every == 0:
            self.optimizer_D.zero_grad()
            r1_losses = self.model(images, command="compute_R1_loss")
            d_losses.update(r1_losses)
            r1_loss = sum(v.mean() for v in r1_losses.values()) * opt.R1_once_every
            self._backward_and_step(r1_loss, self.optimizer_D, self.reducer_D)

        d_losses["D_total"] = sum(v.mean() for v in d_losses.values())
        d_losses.update(d_metrics)
        return d_losses

    def get_visuals_for_snapshot(self, data_i):
        """:113-116"""
        images = self.prepare_images(data_i)
        with torch.no_grad():
            return self.model(images, command="get_visuals_for_snapshot")

    def save(self, total_steps_so_far):
        """:118-119 saves the model; the reference never stores the optimiser, so a resumed run restarts Adam's second
        moments from zero.  Here the two Adam states and the driver's counters are written next to the weights
        (``<N>k_optimizer.pth`` + ``latest_optimizer.pth``, rank 0 only, atomic rename) and picked up by ``load``."""
        import os
        import torch.distributed as dist
        self.model.save(total_steps_so_far)
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not multi or dist.get_rank() == 0:
            savedir = os.path.join(self.opt.checkpoints_dir, self.opt.name)
            os.makedirs(savedir, exist_ok=True)
            name = "%dk_optimizer.pth" % (total_steps_so_far // 1000)
            final = os.path.join(savedir, name)
            torch.save(self.state_dict(), final + ".tmp")
            os.replace(final + ".tmp", final)
            link, tmplink = os.path.join(savedir, "latest_optimizer.pth"), os.path.join(savedir, "latest_optimizer.pth.tmp")
            if os.path.lexists(tmplink):
                os.remove(tmplink)
            os.symlink(name, tmplink)
            os.replace(tmplink, link)
        if multi:
            dist.barrier()

    def state_dict(self):
        return {"optimizer_G": self.optimizer_G.state_dict(), "optimizer_D": self.optimizer_D.state_dict(),
                "train_mode_counter": self.train_mode_counter,
                "discriminator_iter_counter": self.discriminator_iter_counter}

    def load_state_dict(self, state):
        self.optimizer_G.load_state_dict(state["optimizer_G"])
        self.optimizer_D.load_state_dict(state["optimizer_D"])
        self.train_mode_counter = int(state["train_mode_counter"])
        self.discriminator_iter_counter = int(state["discriminator_iter_counter"])
        if self.graphs is not None:
            # torch's load_state_dict REPLACES the moment tensors: graphs captured before hold the old ones' addresses
            self.graphs = hip_graph.StepGraphs()

    def load(self, resume_iter="latest"):
        """Restore what ``save`` wrote (optimiser moments + counters); returns False when there is nothing to load."""
        import os
        path = os.path.join(self.opt.checkpoints_dir, self.opt.name, "%s_optimizer.pth" % resume_iter)
        if not os.path.exists(path):
            return False
        self.load_state_dict(torch.load(path, map_location="cpu"))
        return True


def create_optimizer(opt, model):
    optimizer = SwappingAutoencoderOptimizer(model)
    if getattr(opt, "continue_train", False):
        optimizer.load(getattr(opt, "resume_iter", "latest"))
    return optimizer
