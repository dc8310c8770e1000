"""Build tests/tuning/libsae_hip_tuning.so: the product's kernel sources for gfx950 with -DSAE_TUNING, i.e. WITH the dispatch
knobs (SAE_IGEMM_QUAD, SAE_TR2, SAE_F8, ... read from the environment) and the recorded-experiment kernels that the product
library is built without.  Test infrastructure: the bit-identity tests (tests/test_quad_paths.py, tests/test_f8_gather.py)
need to force one kernel or the other inside one process each; nothing in the product loads this file."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "swapping_autoencoder_pytorch_amd", "csrc")
OUT = os.path.join(HERE, "libsae_hip_tuning.so")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def up_to_date():
    incs = [os.path.join(d, f) for d in (CSRC, os.path.join(CSRC, "tuning")) if os.path.isdir(d) for f in os.listdir(d) if f.endswith(".inc")]
    deps = sources() + incs + [os.path.join(CSRC, "sae_common.h"), os.path.join(ROOT, "include", "sae_hip.h")]
    return os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps)


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return None


def build(force=False):
    """Returns the library path; builds when stale and hipcc is here, keeps a prebuilt file that travelled otherwise."""
    if not force and up_to_date():
        return OUT
    cc = hipcc()
    if cc is None:
        if os.path.exists(OUT):
            return OUT
        raise RuntimeError("hipcc not found and no prebuilt %s" % OUT)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from swapping_autoencoder_pytorch_amd.csrc.build import compile_and_link
    compile_and_link(OUT, os.path.join(CSRC, "build", "tuning"), extra_flags=["-DSAE_TUNING"], force=force)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
