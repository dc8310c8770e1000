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


def write(root):
    for rel, text in FILES.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text.lstrip("\n"))
    return root
