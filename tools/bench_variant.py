"""bench.py on a variant of the hot-path library (tools/build_variant.sh):  python tools/bench_variant.py NAME [bench.py arguments]
Same-box A/B of whole-step effects; the product's own loader takes no path from the environment."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1]
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
from swapping_autoencoder_pytorch_amd import hip_lib  # noqa: E402

hip_lib._LIB = hip_lib.SaeLibrary(os.path.join(ROOT, "tools", "variants", name + ".so"))
import bench  # noqa: E402

bench.main()
