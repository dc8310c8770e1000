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


def _source_deps(src):
    """What one source is rebuilt for: itself, the two headers, and -- for the file that #includes them -- the .inc parts."""
    d = [src, os.path.join(HERE, "sae_common.h"), os.path.join(ROOT, "include", "sae_hip.h")]
    with open(src) as f:
        if ".inc\"" in f.read():
            d += includes()
    return d


def compile_and_link(out, objdir, extra_flags=(), force=False, verbose=False, jobs=None):
    """One object per source (stale ones only, compiled side by side), then one link.  Shared with tests/tuning/build_tuning.py.
    Objects live under a git-ignored build/ directory; only the linked library travels to the GPU box."""
    from concurrent.futures import ThreadPoolExecutor
    cc = hipcc()
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-I", os.path.join(ROOT, "include"),
             "-I", HERE] + list(extra_flags)
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in _source_deps(src)):
            todo.append((src, obj))

    def one(job):
        cmd = [cc] + flags + ["-c", job[0], "-o", job[1]]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(6, os.cpu_count() or 1)) as ex:
            list(ex.map(one, todo))
    subprocess.check_call([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", out])
    verify_loads(out)
    return out


def build(force=False, verbose=False):
    if not force and up_to_date():
        return OUT
    return compile_and_link(OUT, os.path.join(HERE, "build", "product"), force=force, verbose=verbose)


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
