"""Build csrc/libsae_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libsae_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hip"))


def includes():
    """The parts csrc/conv2d.hip is #included from (csrc/*.inc; csrc/tuning/*.inc only exist in -DSAE_TUNING builds)."""
    out = []
    for d in (HERE, os.path.join(HERE, "tuning")):
        if os.path.isdir(d):
            out += sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".inc"))
    return out


def deps():
    return sources() + includes() + [os.path.join(HERE, "sae_common.h"), os.path.join(ROOT, "include", "sae_hip.h")]


def up_to_date():
    return os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps())


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build the gfx950 hot-path library")


def build(force=False, verbose=False):
    if not force and up_to_date():
        return OUT
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-I", os.path.join(ROOT, "include"), "-I", HERE] + sources() + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    verify_loads(OUT)
    return OUT


def verify_loads(path):
    """dlopen the fresh library in a child process (no GPU needed): hipcc's host pass can drop a kernel's launch stub without a
    diagnostic (seen with an LDS-DMA builtin whose operand came from a template-sized array), which only shows as an
    undefined symbol when the library is loaded."""
    code = "import ctypes, sys; ctypes.CDLL(sys.argv[1])"
    r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True)
    if r.returncode != 0:
        os.replace(path, path + ".broken")
        raise RuntimeError("%s was built but does not load: %s" % (path, r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "?"))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
