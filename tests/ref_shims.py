"""Import shims that let the read-only reference checkout import in the build container
(SURVEY.md §7.1): a fake torch.version.cuda so util.is_custom_kernel_supported returns False
instead of raising, and import-only stand-ins for packages the reference imports at module top
level but never executes on the hot path."""
import sys
import types

REF = "/root/reference"


class _Stub(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return type(item, (), {"__init__": lambda self, *a, **k: None})


def install():
    import torch
    if torch.version.cuda is None:
        torch.version.cuda = "0.0"
    for name in ["torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.models",
                 "torchvision.datasets", "dominate", "dominate.tags", "func_timeout", "visdom", "GPUtil", "cv2", "lmdb"]:
        if name not in sys.modules:
            m = _Stub(name)
            m.__path__ = []
            sys.modules[name] = m
    if REF not in sys.path:
        sys.path.insert(0, REF)
