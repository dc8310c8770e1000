"""`python -m swapping_autoencoder_pytorch_amd.dropin ROOT train.py ...` on a stand-in reference tree written by
tests/standin_tree.py (the reference checkout does not exist on the GPU box): pre-seeding of the reference's import names,
stubs for absent third-party packages, FusedAdam substitution, the synthetic dataset, the device prefetcher around the
reference-style loader, optimiser steps incl. the R1 double backward.  GPU: the real library on cuda:0 (what north_star's
"drops into train.py unchanged" means on hardware); CPU: the same tree on the oracle back end (pass-through loader)."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

import standin_tree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--dataset_mode", "synthetic", "--batch_size", "4", "--crop_size", "32", "--steps", "4"]


def _report(out):
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert "Training finished." in out.stdout
    line = [l for l in out.stdout.splitlines() if l.startswith("STANDIN-REPORT ")][-1]
    return json.loads(line[len("STANDIN-REPORT "):])


def _common_checks(rep):
    assert rep["convlayer"] == "swapping_autoencoder_pytorch_amd.stylegan2_layers"          # pre-seeded layer library
    assert rep["upfirdn2d"].startswith("swapping_autoencoder_pytorch_amd.stylegan2_op")       # ... and operators
    assert rep["adam"] == "swapping_autoencoder_pytorch_amd.fused_adam.FusedAdam"             # torch.optim.Adam substituted
    assert rep["dataset"] == "SyntheticDataset"
    def natively_importable(name):
        # (another test of this process may have left the runner's stand-in module behind: that is not the real package)
        if type(sys.modules.get(name)).__name__ in ("_Stub", "_MissingDataPackage"):
            return False
        try:
            return importlib.util.find_spec(name) is not None
        except (ValueError, ImportError):
            return False
    expected = sorted(n for n in ("dominate", "visdom", "lmdb") if not natively_importable(n))
    assert rep["stubbed"] == expected
    assert rep["params_moved"] == rep["params"] and rep["params"] >= 10                       # every parameter was trained
    assert len(rep["losses"]) == 4 and "D_R1" in rep["losses"][1] and "D_R1" in rep["losses"][3]
    for l in rep["losses"]:
        assert all(v == v and abs(v) < 1e6 for v in l.values()), l


def test_dropin_runner_on_the_standin_tree_oracle_backend(tmp_path):
    standin_tree.write(str(tmp_path / "ref"))
    code = ("import os, sys\n"
            "sys.path.insert(0, %r)\n"
            "from swapping_autoencoder_pytorch_amd import dropin, hip_lib\n"
            "from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary\n"
            "hip_lib._LIB = SaeLibrary(os.path.join(%r, 'oracle', 'libsae_oracle.so'), prefix='oracle_', device_only=False)\n"
            "dropin.main([%r, 'train.py', '--num_gpus', '0'] + %r)\n" % (ROOT, ROOT, str(tmp_path / "ref"), ARGS))
    out = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, SAE_DROPIN_LEVEL="layers"))       # this small tree brings its own model
    rep = _report(out)
    _common_checks(rep)
    assert rep["loader"] == "ConfigurableDataLoader"          # no GPU: the prefetch wrapper is a pass-through
    assert set(rep["batch_devices"]) == {"cpu"}


@pytest.mark.gpu
def test_dropin_runner_on_the_standin_tree_real_library_gpu(tmp_path):
    standin_tree.write(str(tmp_path / "ref"))
    out = subprocess.run([sys.executable, "-m", "swapping_autoencoder_pytorch_amd.dropin", str(tmp_path / "ref"), "train.py",
                          "--num_gpus", "1"] + ARGS, cwd=ROOT, capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, PYTHONPATH=ROOT, SAE_DROPIN_LEVEL="layers"))
    rep = _report(out)
    _common_checks(rep)
    assert rep["maps"] == ["libsae_hip.so"]                   # the hipcc-built library is what ran; no oracle in the process
    assert rep["param_device"] == "cuda:0"
    assert rep["loader"] == "PrefetchedLoader"                # the reference-style loader behind the device prefetcher
    assert set(rep["batch_devices"]) == {"cuda:0"}            # train.py received device-resident batches


# ---- the framework stand-in: reflection loaders, MultiGPUModelWrapper / DataParallel, the D / G driver (level full) ---------
MICRO_ARGS = ["--dataset_mode", "synthetic", "--batch_size", "4", "--crop_size", "32", "--load_size", "32",
              "--netE_num_downsampling_sp", "2", "--patch_size", "32", "--patch_num_crops", "2", "--global_code_ch", "64",
              "--netE_scale_capacity", "0.25", "--netG_scale_capacity", "0.125", "--netD_scale_capacity", "0.03125",
              "--netPatchD_scale_capacity", "0.5", "--netPatchD_max_nc", "32", "--R1_once_every", "2", "--steps", "4"]


def _framework_checks(rep):
    ours = "swapping_autoencoder_pytorch_amd"
    assert rep["wrapper"] == "MultiGPUModelWrapper" and rep["parallel"] == "DataParallel"       # the tree's own wrapper
    assert rep["optimizer"] == "optimizers.swapping_autoencoder_optimizer"                      # ... and its own driver
    assert rep["model_mro"][0] == "models.swapping_autoencoder_model.SwappingAutoencoderModel"
    assert rep["model_mro"][1] == ours + ".swapping_autoencoder_model.SwappingAutoencoderModel"  # pre-seeded model class
    assert rep["model_mro"][2] == "models.base_model.BaseModel"                                 # under the tree's BaseModel
    assert rep["encoder"] == ours + ".networks.encoder"
    assert rep["adam"] == ours + ".fused_adam.FusedAdam"
    assert len(rep["losses"]) == 4 and "D_R1" in rep["losses"][2] and "G_L1" in rep["losses"][1]
    for l in rep["losses"]:
        assert all(v == v and abs(v) < 1e6 for v in l.values()), l


def test_dropin_full_level_on_the_framework_standin_oracle_backend(tmp_path):
    standin_tree.write_framework(str(tmp_path / "ref"))
    code = ("import os, sys\n"
            "sys.path.insert(0, %r)\n"
            "from swapping_autoencoder_pytorch_amd import dropin, hip_lib\n"
            "from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary\n"
            "hip_lib._LIB = SaeLibrary(os.path.join(%r, 'oracle', 'libsae_oracle.so'), prefix='oracle_', device_only=False)\n"
            "dropin.main([%r, 'train.py', '--num_gpus', '0'] + %r)\n" % (ROOT, ROOT, str(tmp_path / "ref"), MICRO_ARGS))
    out = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, SAE_DROPIN_LEVEL="full"))
    rep = _report(out)
    _framework_checks(rep)
    assert rep["param_device"] == "cpu"


def test_framework_standin_refuses_lower_levels(tmp_path):
    """The stand-in has no networks / model of its own: anything below level full must fail loudly, not fall back."""
    standin_tree.write_framework(str(tmp_path / "ref"))
    code = ("import os, sys\n"
            "sys.path.insert(0, %r)\n"
            "from swapping_autoencoder_pytorch_amd import dropin, hip_lib\n"
            "from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary\n"
            "hip_lib._LIB = SaeLibrary(os.path.join(%r, 'oracle', 'libsae_oracle.so'), prefix='oracle_', device_only=False)\n"
            "dropin.main([%r, 'train.py', '--num_gpus', '0'] + %r)\n" % (ROOT, ROOT, str(tmp_path / "ref"), MICRO_ARGS))
    out = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, SAE_DROPIN_LEVEL="networks"))
    assert out.returncode != 0 and "must be pre-seeded" in out.stderr, out.stderr[-1500:]


@pytest.mark.gpu
def test_dropin_full_level_on_the_framework_standin_real_library_gpu(tmp_path):
    standin_tree.write_framework(str(tmp_path / "ref"))
    out = subprocess.run([sys.executable, "-m", "swapping_autoencoder_pytorch_amd.dropin", str(tmp_path / "ref"), "train.py",
                          "--num_gpus", "1"] + MICRO_ARGS, cwd=ROOT, capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, PYTHONPATH=ROOT, SAE_DROPIN_LEVEL="full"))
    rep = _report(out)
    _framework_checks(rep)
    assert rep["maps"] == ["libsae_hip.so"] and rep["param_device"] == "cuda:0"
