import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# The whole-network parity tests run stock ATen / MIOpen convolutions as their fp32 control: MIOpen's default "find" benchmarks
# every solver for every new conv geometry (a minute or two per network at full size).  The immediate mode is enough for a control.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order: with ``-x`` one red test hides everything collected after it, so the strongest evidence goes first --
# kernels against the C oracle, the HIP path against the reference's own golden outputs, the style-modulated conv, fused nodes
# and the prepared-weight cache -- and the statistical whole-network comparisons go last.  Files not named keep their
# alphabetical place in between.
_FIRST = ["test_gpu_kernels", "test_gpu_parity", "test_modconv", "test_adam", "test_weight_prep", "test_resblock_fused",
          "test_styled_fused", "test_quad_paths", "test_winograd", "test_ws_gather", "test_glue", "test_f8_gather"]
_LAST = ["test_gpu_fullsize_oracle", "test_gpu_fullsize_properties", "test_gpu_step_parity", "test_gpu_network_parity"]


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _FIRST:
            return (0, _FIRST.index(name))
        if name in _LAST:
            return (2, _LAST.index(name))
        return (1, 0)
    items.sort(key=key)          # stable: order within a file and among unnamed files is kept


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
