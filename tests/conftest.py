import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _hip_library_is_current():
    """Rebuild csrc/libsae_hip.so when a kernel source is newer (hipcc cross-compiles for gfx950
    without a GPU); a missing hipcc leaves whatever prebuilt library travelled with the snapshot."""
    import shutil
    from swapping_autoencoder_pytorch_amd.csrc import build as hip_build
    if not hip_build.up_to_date() and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        hip_build.build()
    # ... and the -DSAE_TUNING build of the same sources the GPU bit-identity tests load (tests/tuning)
    from tuning import build_tuning
    if not build_tuning.up_to_date() and build_tuning.hipcc():
        build_tuning.build()
    yield


@pytest.fixture(scope="session")
def oracle_lib():
    """CPU oracle (oracle/sae_oracle.c) bound with the product's ctypes signatures."""
    from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary
    so = os.path.join(ROOT, "oracle", "libsae_oracle.so")
    src = os.path.join(ROOT, "oracle", "sae_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return SaeLibrary(so, prefix="oracle_", device_only=False)


@pytest.fixture(scope="session")
def emu_lib():
    """The product's kernel sources compiled for the host against the hipemu header."""
    from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary
    from emu import build_emu
    return SaeLibrary(build_emu.build(), prefix="sae_", device_only=False)
