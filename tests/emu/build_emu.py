"""Build tests/emu/libsae_emu.so: the UNMODIFIED kernel sources of the product compiled for the
host against the hipemu header (tests/emu/hip/hip_runtime.h).  Test infrastructure only."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "swapping_autoencoder_pytorch_amd", "csrc")
OUT = os.path.join(HERE, "libsae_emu.so")


def host_clang():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", "clang++"):
        if os.path.exists(c) or c == "clang++":
            return c
    return "clang++"


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    incs = [os.path.join(d, f) for d in (CSRC, os.path.join(CSRC, "tuning")) if os.path.isdir(d) for f in os.listdir(d) if f.endswith(".inc")]
    deps = sources() + incs + [os.path.join(CSRC, "sae_common.h"), os.path.join(ROOT, "include", "sae_hip.h"),
                               os.path.join(HERE, "hip", "hip_runtime.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False):
    if not force and not needs_build():
        return OUT
    cmd = [host_clang(), "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-nogpulib", "-Wno-unused-value", "-Wno-psabi",
           "-DSAE_TUNING", "-I", HERE, "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    for s in sources():
        cmd += [s]
    cmd += ["-o", OUT]
    # every source must be treated as C++ (the .hip suffix would select the HIP language)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
