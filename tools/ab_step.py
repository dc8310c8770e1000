"""In-step A/B of dispatch knobs: bench.py's train-step measurement on a TUNING build of the library (tests/tuning or
tools/variants/*.so, built with -DSAE_TUNING), so that an environment knob (SAE_IGEMM_PREFER64=0, SAE_TR2=0, ...) selects
the kernel for the whole process.  The product library has no knobs; this is measurement tooling.

    SAE_IGEMM_PREFER64=0 python tools/ab_step.py tests/tuning/libsae_hip_tuning.so --steps 12 --warmup 4 --no-cpu-baseline --alt-steps 0"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from swapping_autoencoder_pytorch_amd import hip_lib  # noqa: E402

lib_path = os.path.abspath(sys.argv[1])
hip_lib._LIB = hip_lib.SaeLibrary(lib_path)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
