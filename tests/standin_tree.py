"""A minimal stand-in for the reference checkout, WRITTEN BY THIS FILE (nothing is copied from the reference): a few
lines per module that do the same ABSOLUTE imports and use the same literals the reference does at the points where the
drop-in runner binds: ``import data / models / optimizers``, ``from options import TrainOptions``,
``from models.networks.stylegan2_layers import ...``, ``from models.networks.stylegan2_op import ...``, third-party
imports that are absent from a bare image, ``torch.optim.Adam(params, lr=, betas=)``, the ``'cuda:0'`` literal, a
``ConfigurableDataLoader`` built on ``torch.utils.data.DataLoader`` whose dataset class is looked up by name, and a
train loop that calls ``next(dataset)`` / ``optimizer.train_one_step``.

The reference itself does not exist on the GPU box, so this is how ``python -m swapping_autoencoder_pytorch_amd.dropin
ROOT train.py ...`` is exercised there with the real HIP library (tests/test_gpu_dropin.py); the CPU suite runs the same
tree on the oracle back end so the harness itself is tested where the reference's byte-identical train.py also runs."""
import os

FILES = {
    "train.py": '''
import json
import sys

import data
import models
import optimizers
from options import TrainOptions

opt = TrainOptions().parse()
dataset = data.create_dataset(opt)
model = models.create_model(opt)
optimizer = optimizers.create_optimizer(opt, model)
before = [p.detach().clone() for p in model.parameters()]
batch_devices, losses = [], []
for step in range(opt.steps):
    cur = next(dataset)
    batch_devices.append(str(cur["real_A"].device))
    losses.append(optimizer.train_one_step(cur, step))
import torch
moved = sum(int(not torch.equal(a, b.detach())) for a, b in zip(before, model.parameters()))
report = {
    "losses": losses, "batch_devices": batch_devices, "params": len(before), "params_moved": moved,
    "adam": type(optimizer.optimizer_D).__module__ + "." + type(optimizer.optimizer_D).__name__,
    "convlayer": models.ConvLayer.__module__, "upfirdn2d": models.upfirdn2d.__module__,
    "loader": type(dataset).__name__, "dataset": type(dataset.underlying_dataset).__name__,
    "stubbed": sorted(n for n in ("dominate", "visdom", "lmdb") if type(sys.modules.get(n)).__name__ in ("_Stub", "_MissingDataPackage")),
    "param_device": str(next(model.parameters()).device),
    "maps": sorted({l.split("/")[-1].strip() for l in open("/proc/self/maps") if "libsae" in l}),
}
print("STANDIN-REPORT " + json.dumps(report))
print("Training finished.")
''',
    "options/__init__.py": '''
import argparse

import data


class TrainOptions:
    def parse(self):
        p = argparse.ArgumentParser()
        p.add_argument("--dataset_mode", default="imagefolder")
        p.add_argument("--num_gpus", type=int, default=1)
        p.add_argument("--batch_size", type=int, default=2)
        p.add_argument("--crop_size", type=int, default=32)
        p.add_argument("--steps", type=int, default=3)
        p.add_argument("--phase", default="train")
        known, _ = p.parse_known_args()
        p = data.get_option_setter(known.dataset_mode)(p, True)      # the dataset class adds its own flags
        opt = p.parse_args()
        opt.isTrain = True
        return opt
''',
    "data/__init__.py": '''
import importlib

import torch.utils.data
import lmdb          # absent from a bare image; only the LSUN dataset would use it
from data.base_dataset import BaseDataset


def find_dataset_using_name(name):
    lib = importlib.import_module("data." + name + "_dataset")
    for k, cls in lib.__dict__.items():
        if k.lower() == name.replace("_", "") + "dataset" and issubclass(cls, BaseDataset):
            return cls
    raise ValueError(name)


def get_option_setter(name):
    return find_dataset_using_name(name).modify_commandline_options


class ConfigurableDataLoader:
    def __init__(self, opt):
        self.opt = opt
        self.phase = opt.phase
        self.underlying_dataset = find_dataset_using_name(opt.dataset_mode)(opt)
        self.dataloader = torch.utils.data.DataLoader(self.underlying_dataset, batch_size=opt.batch_size, shuffle=False,
                                                      num_workers=0, drop_last=True)
        self.dataloader_iterator = iter(self.dataloader)

    def set_phase(self, phase):
        self.phase = phase

    def __iter__(self):
        self.dataloader_iterator = iter(self.dataloader)
        return self

    def __len__(self):
        return len(self.underlying_dataset)

    def __next__(self):
        try:
            return next(self.dataloader_iterator)
        except StopIteration:
            self.dataloader_iterator = iter(self.dataloader)
            return next(self.dataloader_iterator)


def create_dataset(opt):
    return ConfigurableDataLoader(opt)
''',
    "data/base_dataset.py": '''
import torch.utils.data


class BaseDataset(torch.utils.data.Dataset):
    def __init__(self, opt):
        self.opt = opt

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser
''',
    "models/__init__.py": '''
import torch
import dominate                    # absent from a bare image; never used on the training path
from visdom import Visdom         # likewise
from models.networks.stylegan2_layers import ConvLayer, EqualLinear, ResBlock
from models.networks.stylegan2_op import fused_leaky_relu, upfirdn2d


class TinyD(torch.nn.Module):
    def __init__(self, size):
        super().__init__()
        self.stem = ConvLayer(3, 16, 1)
        self.block = ResBlock(16, 32)
        self.head = ConvLayer(32, 32, 3)
        self.bias = torch.nn.Parameter(torch.zeros(32))
        self.linear = EqualLinear(32 * (size // 2) ** 2, 1)
        self.register_buffer("taps", torch.ones(2, 2) / 4)

    def forward(self, x):
        x = self.head(self.block(self.stem(x)))
        x = fused_leaky_relu(upfirdn2d(x, self.taps, pad=(1, 0)), self.bias)
        return self.linear(x.flatten(1))


def create_model(opt):
    model = TinyD(opt.crop_size)
    return model.to("cuda:0") if opt.num_gpus > 0 else model
''',
    "models/networks/__init__.py": "",
    "optimizers/__init__.py": '''
import torch


class TinyOptimizer:
    def __init__(self, opt, model):
        self.opt, self.model = opt, model
        self.Dparams = list(model.parameters())
        self.optimizer_D = torch.optim.Adam(self.Dparams, lr=0.002, betas=(0.0, 0.99))

    def train_one_step(self, data_i, step):
        real = data_i["real_A"]
        real = real.to("cuda:0") if self.opt.num_gpus > 0 else real
        real.requires_grad_(step % 2 == 1)
        self.optimizer_D.zero_grad()
        pred = self.model(real)
        loss = torch.nn.functional.softplus(-pred).mean()
        out = {"D_real": float(loss)}
        if step % 2 == 1:              # the lazy-R1 pattern: a gradient penalty differentiated again
            grad, = torch.autograd.grad(pred.sum(), [real], create_graph=True)
            r1 = grad.pow(2).sum(dim=(1, 2, 3)).mean()
            loss = loss + 5.0 * r1
            out["D_R1"] = float(r1)
        loss.backward()
        self.optimizer_D.step()
        return out


def create_optimizer(opt, model):
    return TinyOptimizer(opt, model)
''',
}


# ----------------------------------------------------------------------------------------------------------------------
# The FRAMEWORK stand-in: what surrounds the model in the reference -- the option parser that collects flags from the model,
# the four networks and the optimizer; the reflection loaders (importlib + class-name scan with their issubclass checks);
# create_model -> MultiGPUModelWrapper over nn.DataParallel on the 'cuda:0' literal; the alternating D / G driver with lazy
# R1 and the per-call device->host copy of the losses -- restated tersely, so that `dropin.preseed("full")` can be exercised
# and TIMED on the GPU box, where the reference checkout does not exist (bench.py --via-dropin, tests/test_dropin_standin.py).
# The model, network and util hot helpers of this tree refuse to run: they must have been pre-seeded / rebound.
FRAMEWORK_FILES = {
    "data/__init__.py": FILES["data/__init__.py"],
    "data/base_dataset.py": FILES["data/base_dataset.py"],
    "train.py": '''
import json
import time

import torch

import data
import models
import optimizers
from options import TrainOptions

opt = TrainOptions().parse()
dataset = data.create_dataset(opt)
model = models.create_model(opt)
optimizer = optimizers.create_optimizer(opt, model)
losses, t0 = [], time.time()
for step in range(opt.steps):
    losses.append({k: float(v) for k, v in optimizer.train_one_step(next(dataset), step).items()})
if opt.num_gpus > 0:
    torch.cuda.synchronize()
net = model.singlegpu_model
print("STANDIN-REPORT " + json.dumps({
    "losses": losses, "seconds": time.time() - t0, "wrapper": type(model).__name__,
    "parallel": type(model.parallelized_model).__name__, "model_mro": [c.__module__ + "." + c.__name__ for c in type(net).__mro__[:4]],
    "encoder": type(net.E).__module__, "optimizer": type(optimizer).__module__,
    "adam": type(optimizer.optimizer_D).__module__ + "." + type(optimizer.optimizer_D).__name__,
    "param_device": str(next(net.parameters()).device),
    "maps": sorted({l.split("/")[-1].strip() for l in open("/proc/self/maps") if "libsae" in l})}))
print("Training finished.")
''',
    "options/__init__.py": '''
import argparse

import data
import models
import models.networks as networks
import optimizers
import util


class TrainOptions:
    isTrain = True

    def parse(self):
        p = argparse.ArgumentParser()
        p.add_argument("--name", default="standin")
        p.add_argument("--num_gpus", type=int, default=1)
        p.add_argument("--checkpoints_dir", default="./checkpoints/")
        p.add_argument("--model", default="swapping_autoencoder")
        p.add_argument("--optimizer", default="swapping_autoencoder")
        p.add_argument("--phase", default="train")
        p.add_argument("--resume_iter", default="latest")
        p.add_argument("--num_classes", type=int, default=0)
        p.add_argument("--batch_size", type=int, default=1)
        p.add_argument("--load_size", type=int, default=256)
        p.add_argument("--crop_size", type=int, default=256)
        p.add_argument("--dataset_mode", default="lmdb")
        p.add_argument("--netG", default="StyleGAN2Resnet")
        p.add_argument("--netD", default="StyleGAN2")
        p.add_argument("--netE", default="StyleGAN2Resnet")
        p.add_argument("--netPatchD", default="StyleGAN2")
        p.add_argument("--use_antialias", type=util.str2bool, default=True)
        p.add_argument("--continue_train", type=util.str2bool, default=False)
        p.add_argument("--pretrained_name", default=None)
        p.add_argument("--steps", type=int, default=4)
        known, _ = p.parse_known_args()
        p = models.get_option_setter(known.model)(p, self.isTrain)
        p = networks.modify_commandline_options(p, self.isTrain)
        p = optimizers.get_option_setter(known.optimizer)(p, self.isTrain)
        p = data.get_option_setter(known.dataset_mode)(p, self.isTrain)
        opt = p.parse_args()
        opt.isTrain = self.isTrain
        assert opt.num_gpus <= opt.batch_size
        return opt
''',
    "util/__init__.py": '''
import dominate                    # absent from a bare image; never used on the training path
from visdom import Visdom         # likewise
from .util import *
''',
    "util/util.py": '''
import argparse
import importlib


def normalize(v):
    raise RuntimeError("stand-in util.normalize reached: dropin.patch_util() should have rebound it")


def apply_random_crop(x, target_size, scale_range, num_crops=1, return_rect=False):
    raise RuntimeError("stand-in util.apply_random_crop reached: dropin.patch_util() should have rebound it")


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def find_class_in_module(target_cls_name, module):
    wanted = target_cls_name.replace("_", "").lower()
    lib = importlib.import_module(module)
    found = [obj for name, obj in lib.__dict__.items() if name.lower() == wanted]
    assert found, "no class %s in %s" % (target_cls_name, module)
    return found[-1]


def to_numpy(metric_dict):
    out = {}
    for k, v in metric_dict.items():
        if "numpy" not in str(type(v)):
            v = v.detach().cpu().mean().numpy()
        out[k] = v
    return out
''',
    "models/__init__.py": '''
import importlib

import torch
from models.base_model import BaseModel


def find_model_using_name(model_name):
    lib = importlib.import_module("models." + model_name + "_model")
    wanted = model_name.replace("_", "") + "model"
    found = None
    for name, cls in lib.__dict__.items():
        if name.lower() == wanted.lower() and issubclass(cls, BaseModel):
            found = cls
    if found is None:
        raise SystemExit("no BaseModel subclass named %s in models.%s_model" % (wanted, model_name))
    return found


def get_option_setter(model_name):
    return find_model_using_name(model_name).modify_commandline_options


def create_model(opt):
    instance = find_model_using_name(opt.model)(opt)
    instance.initialize()
    wrapped = MultiGPUModelWrapper(opt, instance)
    print("model [%s] was created" % type(instance).__name__)
    return wrapped


class MultiGPUModelWrapper:
    def __init__(self, opt, model):
        self.opt = opt
        if opt.num_gpus > 0:
            model = model.to("cuda:0")
        self.parallelized_model = torch.nn.parallel.DataParallel(model)
        self.parallelized_model(command="per_gpu_initialize")
        self.singlegpu_model = self.parallelized_model.module
        self.singlegpu_model(command="per_gpu_initialize")

    def get_parameters_for_mode(self, mode):
        return self.singlegpu_model.get_parameters_for_mode(mode)

    def save(self, total_steps_so_far):
        self.singlegpu_model.save(total_steps_so_far)

    def __call__(self, *args, **kwargs):
        return self.parallelized_model(*args, **kwargs)
''',
    "models/base_model.py": '''
import torch


class BaseModel(torch.nn.Module):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.device = torch.device("cuda:0") if opt.num_gpus > 0 else torch.device("cpu")

    def initialize(self):
        pass

    def per_gpu_initialize(self):
        pass

    def forward(self, *args, command=None, **kwargs):
        return getattr(self, command)(*args, **kwargs)
''',
    "models/swapping_autoencoder_model.py": '''
raise ImportError("the stand-in tree has no model of its own: models.swapping_autoencoder_model must be pre-seeded (SAE_DROPIN_LEVEL=full)")
''',
    "models/networks/__init__.py": '''
import util
from .base_network import BaseNetwork


def find_network_using_name(target_network_name, filename):
    network = util.find_class_in_module(target_network_name + filename, "models.networks." + filename)
    assert issubclass(network, BaseNetwork), "%s is not a BaseNetwork" % network
    return network


def modify_commandline_options(parser, is_train):
    opt, _ = parser.parse_known_args()
    for name, kind in ((opt.netE, "encoder"), (opt.netG, "generator"), (opt.netD, "discriminator"), (opt.netPatchD, "patch_discriminator")):
        if name is not None:
            parser = find_network_using_name(name, kind).modify_commandline_options(parser, is_train)
    return parser


def create_network(opt, network_name, mode, verbose=True):
    if network_name is None:
        return None
    net = find_network_using_name(network_name, mode)(opt)
    if verbose:
        net.print_architecture(verbose=True)
    return net
''',
    "models/networks/base_network.py": '''
raise ImportError("the stand-in tree has no networks of its own: models.networks.base_network must be pre-seeded")
''',
    "optimizers/__init__.py": '''
import importlib

from optimizers.base_optimizer import BaseOptimizer


def find_optimizer_using_name(optimizer_name):
    lib = importlib.import_module("optimizers." + optimizer_name + "_optimizer")
    wanted = optimizer_name.replace("_", "") + "optimizer"
    for name, cls in lib.__dict__.items():
        if name.lower() == wanted.lower() and issubclass(cls, BaseOptimizer):
            return cls
    raise SystemExit("no BaseOptimizer subclass named %s" % wanted)


def get_option_setter(optimizer_name):
    return find_optimizer_using_name(optimizer_name).modify_commandline_options


def create_optimizer(opt, model):
    return find_optimizer_using_name(opt.optimizer)(model)
''',
    "optimizers/base_optimizer.py": '''
class BaseOptimizer:
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def train_one_step(self, data_i, total_steps_so_far):
        pass
''',
    "optimizers/swapping_autoencoder_optimizer.py": '''
import torch

import util
from models import MultiGPUModelWrapper
from optimizers.base_optimizer import BaseOptimizer


class SwappingAutoencoderOptimizer(BaseOptimizer):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--lr", default=0.002, type=float)
        parser.add_argument("--beta1", default=0.0, type=float)
        parser.add_argument("--beta2", default=0.99, type=float)
        parser.add_argument("--R1_once_every", default=16, type=int)
        return parser

    def __init__(self, model: MultiGPUModelWrapper):
        self.opt = opt = model.opt
        self.model = model
        self.train_mode_counter = 0
        self.discriminator_iter_counter = 0
        self.Gparams = model.get_parameters_for_mode("generator")
        self.Dparams = model.get_parameters_for_mode("discriminator")
        self.optimizer_G = torch.optim.Adam(self.Gparams, lr=opt.lr, betas=(opt.beta1, opt.beta2))
        c = opt.R1_once_every / (1 + opt.R1_once_every)
        self.optimizer_D = torch.optim.Adam(self.Dparams, lr=opt.lr * c, betas=(opt.beta1 ** c, opt.beta2 ** c))

    def set_requires_grad(self, params, requires_grad):
        for p in params:
            p.requires_grad_(requires_grad)

    def train_one_step(self, data_i, total_steps_so_far):
        images = data_i["real_A"]
        self.train_mode_counter = (self.train_mode_counter + 1) % 2
        if self.train_mode_counter == 1:
            losses = self.train_discriminator_one_step(images)
        else:
            losses = self.train_generator_one_step(images)
        return util.to_numpy(losses)

    def train_generator_one_step(self, images):
        self.set_requires_grad(self.Dparams, False)
        self.set_requires_grad(self.Gparams, True)
        self.optimizer_G.zero_grad()
        g_losses, g_metrics = self.model(images, None, None, command="compute_generator_losses")
        sum([v.mean() for v in g_losses.values()]).backward()
        self.optimizer_G.step()
        g_losses.update(g_metrics)
        return g_losses

    def train_discriminator_one_step(self, images):
        opt = self.opt
        self.set_requires_grad(self.Dparams, True)
        self.set_requires_grad(self.Gparams, False)
        self.discriminator_iter_counter += 1
        self.optimizer_D.zero_grad()
        d_losses, d_metrics, sp, gl = self.model(images, command="compute_discriminator_losses")
        self.previous_sp, self.previous_gl = sp.detach(), gl.detach()
        sum([v.mean() for v in d_losses.values()]).backward()
        self.optimizer_D.step()
        if (opt.lambda_R1 > 0.0 or opt.lambda_patch_R1 > 0.0) and self.discriminator_iter_counter % opt.R1_once_every == 0:
            self.optimizer_D.zero_grad()
            r1_losses = self.model(images, command="compute_R1_loss")
            d_losses.update(r1_losses)
            (sum([v.mean() for v in r1_losses.values()]) * opt.R1_once_every).backward()
            self.optimizer_D.step()
        d_losses["D_total"] = sum([v.mean() for v in d_losses.values()])
        d_losses.update(d_metrics)
        return d_losses

    def save(self, total_steps_so_far):
        self.model.save(total_steps_so_far)
''',
}


def write_framework(root):
    """The framework stand-in (FRAMEWORK_FILES) under `root`."""
    return write(root, FRAMEWORK_FILES)


def write(root, files=None):
    for rel, text in (FILES if files is None else files).items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text.lstrip("\n"))
    return root
