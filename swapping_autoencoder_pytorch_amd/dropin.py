"""Run the reference's UNMODIFIED ``train.py`` (or ``test.py``) on the MI355X hot path.

    python -m swapping_autoencoder_pytorch_amd.dropin /path/to/swapping-autoencoder-pytorch train.py --name ...

How: every import of the reference is an absolute import with the repo root on
``sys.path`` and Python consults ``sys.modules`` before the file system, so pre-seeding

    models.networks.stylegan2_op            (+ .upfirdn2d, .fused_act)   <- stylegan2_op of this package
    models.networks.stylegan2_layers                                     <- stylegan2_layers of this package

puts the gfx950 kernels underneath the reference's own networks, model, optimizer and training
loop.  ``util.is_custom_kernel_supported`` (util/util.py:432-436, which raises on ROCm) is never
reached because the two modules that called it are replaced.

The reflection loaders of the networks and of the model (models/networks/__init__.py:6-14 -> util/util.py:61-71,
models/__init__.py:28-48) are ``importlib.import_module`` calls too, so the same mechanism reaches one level up.
``SAE_DROPIN_LEVEL`` (or ``preseed(level=...)``) chooses how far:

    layers    operators + layer library only: the reference's own networks / model files run on them, un-fused at the
              block level (three separate D passes, per-module generator blocks, ATen ``normalize``)
    networks  + models.networks.{base_network,encoder,generator,discriminator,patch_discriminator} of this package
              (ticketed generator blocks, ResBlock nodes, stems evaluated inside the first block)
    full      (default) + models.swapping_autoencoder_model: this package's model class, made a subclass of the
              reference's ``models.base_model.BaseModel`` so that ``find_model_using_name`` accepts it (batched D / Dpatch
              passes, HIP crop sampler, R1 without weight-gradient kernels) -- the configuration ``bench.py`` times.

At every level ``util.normalize`` / ``util.apply_random_crop`` (util/util.py:18-22,323-343) are rebound to the kernels
(``SAE_DROPIN_UTIL=0`` keeps the reference's ATen versions).  train.py, options/, optimizers/, data/, models/__init__.py
(create_model, MultiGPUModelWrapper) and models/base_model.py stay the reference's, byte-identical, at every level.

Besides the pre-seeding the runner
  * stands in for the reference's third-party imports that are absent from a bare PyTorch-ROCm image and never
    executed on the training path (``install_missing_dependency_stubs``: torchvision, dominate, visdom, ...),
  * offers ``--dataset_mode synthetic``: ``data.synthetic_dataset.SyntheticDataset`` (uniform [-1, 1] images of
    ``--crop_size``, the data contract of data/base_dataset.py:136-141) so the loop runs without a dataset on disk,
  * substitutes fused_adam.FusedAdam for ``torch.optim.Adam`` inside this process (same constructor, update rule and
    state_dict; ``SAE_DROPIN_ADAM=0`` keeps ATen's),
  * wraps the loader ``data.create_dataset`` returns (data/__init__.py:81-129) in ``PrefetchedLoader`` when the run uses
    a GPU: batches are staged into device memory two steps ahead on a side stream (data_prefetch.DevicePrefetcher, the
    counterpart of the DataPrefetcher the reference keeps commented out; ``SAE_DROPIN_PREFETCH=0`` turns it off).

Multi-GPU: the reference drives ``nn.DataParallel`` from one process (models/__init__.py:80) and addresses its
device as the literal ``'cuda:0'`` (models/__init__.py:79, base_model.py:13, swapping_autoencoder_model.py:48).
Launched under ``python -m torch.distributed.run --nproc-per-node N``, this runner gives every rank ONE visible
GPU — ``HIP_VISIBLE_DEVICES`` is narrowed to the rank's device BEFORE the HIP runtime initialises
(``pin_rank_device``), so ``'cuda:0'`` is that GPU in every rank — pass ``--num_gpus 1`` (DataParallel over a
single device is a pass-through; ``--batch_size`` is then the PER-RANK batch, the global batch is N times it),
broadcasts rank 0's initial weights and attaches the bucketed RCCL gradient all-reduce
(grad_allreduce.GradAllReducer) to the reference optimizer's two Adam instances through optimizer step hooks;
checkpoints are written by rank 0 only.  train.py and the optimizer source stay byte-identical.
"""
import os
import runpy
import sys
import types

_PINNED_FLAG = "SAE_DROPIN_DEVICE_PINNED"


def pin_rank_device(environ=None, reexec=True):
    """Narrow HIP_VISIBLE_DEVICES to this rank's GPU.  Must run before the HIP runtime initialises (the runtime
    reads the variable once): it is the first thing ``main`` does, before ``torch`` is imported.  If some earlier
    import already initialised torch.cuda the process re-executes itself with the variable set.
    Returns the device string chosen, or None when not launched with WORLD_SIZE > 1."""
    env = os.environ if environ is None else environ
    if int(env.get("WORLD_SIZE", "1")) <= 1:
        return None
    if env.get(_PINNED_FLAG) == "1":
        return env.get("HIP_VISIBLE_DEVICES")
    local_rank = int(env.get("LOCAL_RANK", "0"))
    # the user's own restriction of the node's GPUs: HIP_VISIBLE_DEVICES, else CUDA_VISIBLE_DEVICES (the HIP runtime
    # honours both, in that order).  ROCR_VISIBLE_DEVICES acts one level below (it renumbers what HIP sees), so HIP
    # indices are relative to it and it is left alone.
    spec = env.get("HIP_VISIBLE_DEVICES")
    if spec is None or spec == "":
        spec = env.get("CUDA_VISIBLE_DEVICES", "")
    visible = [d for d in spec.split(",") if d != ""]
    if visible and local_rank >= len(visible):
        raise RuntimeError("LOCAL_RANK %d but only %d visible device(s) (%s)" % (local_rank, len(visible), spec))
    mine = visible[local_rank] if visible else str(local_rank)
    env["HIP_VISIBLE_DEVICES"] = mine
    env.pop("CUDA_VISIBLE_DEVICES", None)      # one source of truth: a stale list here would contradict the pin
    env[_PINNED_FLAG] = "1"
    torch = sys.modules.get("torch")
    if reexec and torch is not None and torch.cuda.is_initialized():
        os.execv(sys.executable, list(getattr(sys, "orig_argv", [sys.executable] + sys.argv)))
    return mine


class _Inert:
    """Instance of a stubbed class: construction and every method call succeed and do nothing (methods return
    True, so ``visdom.Visdom(...).check_connection()`` reports a live connection and nothing is spawned)."""

    def __init__(self, *args, **kwargs):
        pass

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return lambda *args, **kwargs: True


class _MissingDataPackage(types.ModuleType):
    """Stand-in for a package the reference's real datasets need (lmdb, cv2): importing it succeeds, so that
    ``data/*`` modules load, but USING it fails with the package's name instead of an AttributeError deep in a loader."""

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        raise ImportError("the reference's dataset code needs the `%s` package, which is not installed in this environment "
                          "(the drop-in runner only stood in for its import); install it or use --dataset_mode synthetic"
                          % self.__name__)


class _Stub(types.ModuleType):
    """Import-only stand-in: any attribute is an inert class (enough for ``import x`` / ``from x import Y`` /
    ``class Z(x.Y)`` at module top level and for objects that are constructed but never do real work)."""

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return type(item, (_Inert,), {})


OPTIONAL_DEPENDENCIES = ("torchvision", "torchvision.transforms", "torchvision.transforms.functional",
                         "torchvision.models", "torchvision.datasets", "dominate", "dominate.tags", "func_timeout",
                         "visdom", "GPUtil", "cv2", "lmdb")


DATA_PATH_DEPENDENCIES = ("cv2", "lmdb")     # used by the real LSUN / FFHQ datasets: stubbed for import only


def install_missing_dependency_stubs():
    """The reference imports these at module top level (util/visualizer.py, util/html.py, data/*, evaluation/*) but
    the training step never executes them.  Only what is NOT importable is stubbed."""
    import importlib.util
    stubbed = []
    for name in OPTIONAL_DEPENDENCIES:
        if name in sys.modules:
            continue
        root = name.split(".")[0]
        if root not in stubbed and not isinstance(sys.modules.get(root), _Stub):
            try:
                if importlib.util.find_spec(name) is not None:
                    continue
            except (ImportError, ValueError):
                pass
        m = (_MissingDataPackage if root in DATA_PATH_DEPENDENCIES else _Stub)(name)
        m.__path__ = []
        sys.modules[name] = m
        stubbed.append(name)
    return stubbed


LEVELS = ("layers", "networks", "full")


def preseed(level=None):
    """Install this package's modules under the reference's import names, up to `level` (module docstring; default: the
    ``SAE_DROPIN_LEVEL`` environment variable, else "full").  "full" imports the reference's ``models.base_model`` (two
    imports: os, torch), so the reference root must be on ``sys.path`` by then.  Returns the level installed."""
    level = level or os.environ.get("SAE_DROPIN_LEVEL", "full")
    if level not in LEVELS:
        raise ValueError("SAE_DROPIN_LEVEL must be one of %s, got %r" % (LEVELS, level))
    from . import stylegan2_layers, stylegan2_op
    from .stylegan2_op import fused_act, upfirdn2d
    sys.modules["models.networks.stylegan2_op"] = stylegan2_op
    sys.modules["models.networks.stylegan2_op.upfirdn2d"] = upfirdn2d
    sys.modules["models.networks.stylegan2_op.fused_act"] = fused_act
    sys.modules["models.networks.stylegan2_layers"] = stylegan2_layers
    if level in ("networks", "full"):
        # models/networks/__init__.py:3 binds BaseNetwork from .base_network and asserts issubclass against it (:11-12):
        # the base class comes along so that the check holds for the four network classes of this package
        from .networks import base_network, discriminator, encoder, generator, patch_discriminator
        for name, mod in (("base_network", base_network), ("encoder", encoder), ("generator", generator),
                          ("discriminator", discriminator), ("patch_discriminator", patch_discriminator)):
            sys.modules["models.networks." + name] = mod
    if level == "full":
        _preseed_model()
    return level


def _preseed_model():
    """``models.swapping_autoencoder_model`` <- this package's model class under the reference's BaseModel.
    models/__init__.py:28-48 looks the class up by name in that module and requires ``issubclass(cls, BaseModel)``."""
    import importlib
    base = importlib.import_module("models.base_model")          # the reference's own file
    from . import swapping_autoencoder_model as mirror

    class SwappingAutoencoderModel(mirror.SwappingAutoencoderModel, base.BaseModel):
        __doc__ = mirror.SwappingAutoencoderModel.__doc__

    SwappingAutoencoderModel.__module__ = "models.swapping_autoencoder_model"
    mod = types.ModuleType("models.swapping_autoencoder_model")
    mod.__doc__ = "pre-seeded by swapping_autoencoder_pytorch_amd.dropin (level full)"
    mod.SwappingAutoencoderModel = SwappingAutoencoderModel
    sys.modules["models.swapping_autoencoder_model"] = mod
    return mod


def patch_util():
    """Rebind the two hot helpers of the reference's util/util.py -- ``normalize`` (:18-22, called by the encoder and the
    generator) and ``apply_random_crop`` (:323-343, called by the model) -- to this package's kernels.  Callers write
    ``util.normalize(...)`` against the package namespace that ``from .util import *`` filled (util/__init__.py:4), so both
    the package and the module are patched.  Needs the reference root on sys.path and the third-party stubs installed."""
    if os.environ.get("SAE_DROPIN_UTIL", "1") == "0":
        return False
    import importlib
    import importlib.util
    from . import util as mine
    if importlib.util.find_spec("util") is None:       # a tree without the helper package has nothing to rebind
        return False
    pkg = importlib.import_module("util")
    for target in (pkg, sys.modules.get("util.util")):
        if target is None:
            continue
        for name in ("normalize", "apply_random_crop"):
            if hasattr(target, name):
                setattr(target, name, getattr(mine, name))
    return True


def inject_synthetic_dataset():
    """Register ``data.synthetic_dataset`` (found by data/__init__.py:19-39 through importlib, i.e. sys.modules
    first): uniform [-1, 1] images of crop_size, the tensor contract of the reference's datasets."""
    import torch
    from data.base_dataset import BaseDataset      # the reference's own base class (reference root is on sys.path)

    class SyntheticDataset(BaseDataset):
        @staticmethod
        def modify_commandline_options(parser, is_train):
            parser.add_argument("--synthetic_dataset_size", type=int, default=1 << 20)
            return parser

        def __init__(self, opt):
            BaseDataset.__init__(self, opt)
            self.size = int(getattr(opt, "synthetic_dataset_size", 1 << 20))
            self.res = int(opt.crop_size)

        def __len__(self):
            return self.size

        def __getitem__(self, index):
            g = torch.Generator().manual_seed(int(index))
            return {"real_A": torch.rand(3, self.res, self.res, generator=g) * 2 - 1, "path_A": "synthetic/%09d" % index}

    mod = types.ModuleType("data.synthetic_dataset")
    mod.SyntheticDataset = SyntheticDataset
    sys.modules["data.synthetic_dataset"] = mod
    return mod


class PrefetchedLoader:
    """The reference's ``ConfigurableDataLoader`` (data/__init__.py:81-129) with its batches staged into GPU memory
    ``depth`` steps ahead by data_prefetch.DevicePrefetcher -- the working counterpart of the ``DataPrefetcher`` the
    reference keeps commented out (data/__init__.py:52-78,97).  train.py:22-28 then receives ``real_A`` already
    device-resident (``.cuda()`` / ``.to(device)`` in the model are no-ops).  Everything else (len, set_phase,
    underlying_dataset, ...) is the wrapped loader's; a change of phase or a new ``iter()`` restarts the staging."""

    def __init__(self, inner, device="cuda:0", depth=2):
        self.__dict__["_inner"] = inner
        self.__dict__["_device"] = device
        self.__dict__["_depth"] = depth
        self.__dict__["_prefetcher"] = None

    def __getattr__(self, name):
        return getattr(self.__dict__["_inner"], name)

    def __setattr__(self, name, value):
        setattr(self.__dict__["_inner"], name, value)

    def __len__(self):
        return len(self._inner)

    def _pull(self):
        inner = self._inner
        while True:
            try:
                yield next(inner)
            except StopIteration:
                return

    def set_phase(self, target_phase):
        if self._inner.phase != target_phase:
            self.__dict__["_prefetcher"] = None       # staged batches of the old phase are dropped
        self._inner.set_phase(target_phase)

    def __iter__(self):
        iter(self._inner)
        self.__dict__["_prefetcher"] = None
        return self

    def __next__(self):
        if self.__dict__["_prefetcher"] is None:
            from .data_prefetch import DevicePrefetcher
            self.__dict__["_prefetcher"] = DevicePrefetcher(self._pull(), device=self._device, depth=self._depth)
        return next(self.__dict__["_prefetcher"])


def wrap_dataloader(create_dataset):
    """``data.create_dataset`` -> the same loader behind a PrefetchedLoader when the run uses a GPU."""
    def create(opt):
        loader = create_dataset(opt)
        import torch
        if int(getattr(opt, "num_gpus", 0)) > 0 and torch.cuda.is_available() and os.environ.get("SAE_DROPIN_PREFETCH", "1") != "0":
            return PrefetchedLoader(loader, device="cuda:0")
        return loader
    return create


def wrap_reference_r1():
    """Levels "layers" / "networks" keep the reference's model, whose ``compute_R1_loss`` (swapping_autoencoder_model.py:
    138-185) asks ``autograd.grad`` for IMAGE gradients only; a custom autograd node cannot see that and would run its
    weight-gradient kernels (and, for the ResBlock node, rebuild differentiable weight-gradient graphs) for nothing.  The
    method only builds the penalty -- its ``backward()`` happens later in the optimizer -- so it runs as a whole under
    ``input_grads_only()``."""
    mod = sys.modules.get("models.swapping_autoencoder_model")
    cls = getattr(mod, "SwappingAutoencoderModel", None)
    if cls is None or getattr(mod, "__file__", None) is None:        # absent, or this package's own (level full)
        return False
    if getattr(cls.compute_R1_loss, "_sae_wrapped", False):
        return True
    from .stylegan2_op import input_grads_only
    inner = cls.compute_R1_loss

    def compute_R1_loss(self, *args, **kwargs):
        with input_grads_only():
            return inner(self, *args, **kwargs)

    compute_R1_loss._sae_wrapped = True
    cls.compute_R1_loss = compute_R1_loss
    return True


def _bucketwise_step(opt, reducer):
    """``opt.step`` for a FusedAdam behind a GradAllReducer.  (A function of its own: the replacement re-installs ITSELF after
    every call, and a closure defined in attach_gradient_allreduce's loop would look that name up late and install the last
    group's wrapper on every optimizer -- the first group would then never be stepped again.)"""
    plain = opt.step

    def step(closure=None, **kw):
        if kw or closure is not None or not (reducer.enabled and reducer.armed):
            return plain(closure, **kw)
        opt.step = plain                 # finish_into drives the optimizer's own step, bucket by bucket
        try:
            reducer.finish_into(opt)
        finally:
            opt.step = step
        reducer.arm_lazily()

    return step


def attach_gradient_allreduce(optimizer):
    """Give a reference SwappingAutoencoderOptimizer data-parallel semantics across ranks."""
    import torch.distributed as dist
    from .grad_allreduce import GradAllReducer, broadcast_parameters
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return optimizer
    broadcast_parameters(optimizer.model.singlegpu_model)
    from .fused_adam import FusedAdam
    for params, opt in ((optimizer.Gparams, optimizer.optimizer_G), (optimizer.Dparams, optimizer.optimizer_D)):
        reducer = GradAllReducer(params)
        reducer.arm()
        if isinstance(opt, FusedAdam):
            # the bench's path: every bucket goes to the multi-tensor Adam as its collective completes, summed gradients
            # read in place with 1 / world applied by the kernel (GradAllReducer.finish_into)
            opt.step = _bucketwise_step(opt, reducer)
        else:
            opt.register_step_pre_hook(lambda o, a, k, r=reducer: r.finish())
            opt.register_step_post_hook(lambda o, a, k, r=reducer: r.arm_lazily())
        # the reference's loop is zero_grad() -> backward() -> step() (optimizers/swapping_autoencoder_optimizer.py:69-77,
        # 84-107): re-arm once the old gradients are gone, so that the conv weights' slots are handed out again
        # (GradAllReducer.arm_lazily covers a loop that never calls zero_grad)
        plain_zero = opt.zero_grad

        def zero_grad(*a, _z=plain_zero, _r=reducer, **kw):
            out = _z(*a, **kw)
            if _r.enabled and (_r.armed or _r._arm_pending):
                _r.arm()
            return out

        opt.zero_grad = zero_grad
    # one writer: every rank holds the same weights, and N ranks racing on the same checkpoint file and on the
    # remove/symlink of latest_checkpoint.pth (models/base_model.py:33-48) can tear it
    from .swapping_autoencoder_model import SwappingAutoencoderModel as _Mirror
    if not isinstance(optimizer.model.singlegpu_model, _Mirror):      # (this package's model class does exactly this itself)
        save = optimizer.save

        def save_on_rank0(*args, **kwargs):
            if dist.get_rank() == 0:
                save(*args, **kwargs)
            dist.barrier()

        optimizer.save = save_on_rank0
    return optimizer


def _init_distributed():
    """One rank = one visible GPU = cuda:0 (pin_rank_device ran first)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return
    import torch
    import torch.distributed as dist
    assert os.environ.get(_PINNED_FLAG) == "1", "pin_rank_device() must run before torch.cuda is touched"
    if torch.cuda.device_count() != 1:
        raise RuntimeError("rank %s sees %d devices after pinning HIP_VISIBLE_DEVICES=%s: the HIP runtime was initialised "
                           "before the runner could narrow it" % (os.environ.get("RANK"), torch.cuda.device_count(),
                                                                 os.environ.get("HIP_VISIBLE_DEVICES")))
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))


def main(argv=None):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # (package __init__: the step's two streams need queues of their own)
    pin_rank_device()           # first: before anything can initialise the HIP runtime
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 2:
        raise SystemExit("usage: python -m swapping_autoencoder_pytorch_amd.dropin REFERENCE_ROOT SCRIPT.py [args...]")
    ref_root, script = os.path.abspath(argv[0]), argv[1]
    sys.path.insert(0, ref_root)
    from . import hip_lib
    hip_lib.get()               # fail loudly before anything else if the HIP library is not built
    install_missing_dependency_stubs()
    level = preseed()
    patch_util()
    if int(os.environ.get("RANK", "0")) == 0:
        # say it once: at the higher levels edits to the checkout's own network / model files are NOT what runs
        print("[sae dropin] pre-seed level %r: %s come from swapping_autoencoder_pytorch_amd (SAE_DROPIN_LEVEL=layers keeps the "
              "checkout's own networks and model on the MI355X layer library)"
              % (level, {"layers": "models.networks.stylegan2_op / stylegan2_layers",
                         "networks": "the operator / layer library and models.networks.{encoder,generator,discriminator,"
                                     "patch_discriminator}",
                         "full": "the operator / layer library, the four networks and models.swapping_autoencoder_model"}[level]),
              file=sys.stderr)
    inject_synthetic_dataset()
    if os.environ.get("SAE_DROPIN_ADAM", "1") != "0":
        # the reference constructs torch.optim.Adam(params, lr=, betas=) (optimizers/swapping_autoencoder_optimizer.py:
        # 34-42): the same constructor, update and state_dict on the multi-tensor HIP kernel
        import torch
        from .fused_adam import FusedAdam
        torch.optim.Adam = FusedAdam
    _init_distributed()
    import data                 # the reference's package
    data.create_dataset = wrap_dataloader(data.create_dataset)
    import optimizers           # the reference's package (imports its models on the pre-seeded layers)
    import models               # (already imported by optimizers; the reflection loader fills it lazily)
    _create_model = models.create_model

    def create_model(opt):
        model = _create_model(opt)   # find_model_using_name has imported models.swapping_autoencoder_model by now
        wrap_reference_r1()
        return model

    models.create_model = create_model
    _create = optimizers.create_optimizer
    optimizers.create_optimizer = lambda opt, model: attach_gradient_allreduce(_create(opt, model))
    sys.argv = [script] + argv[2:]
    runpy.run_path(os.path.join(ref_root, script), run_name="__main__")


if __name__ == "__main__":
    main()
