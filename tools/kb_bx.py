"""Kernel-bench subset for the bf16x6 arithmetic: 3x3 layers of the church preset (see SAE_CONV_MATH)."""
import sys, os
sel = sys.argv[1:] or ["s1", "tr"]
sys.argv = [sys.argv[0]]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_bench as K
if "s1" in sel:
    K.bench_conv(16, 128, 256, 256, 128, 3, 1, 1, "s1 128@256")
    K.bench_conv(16, 512, 64, 64, 512, 3, 1, 1, "D 512@64")
if "fused" in sel:
    K.bench_conv_fused(16, 128, 256, 256, 128, 3, 1, 1, "s1 128@256")
    K.bench_conv_fused(16, 512, 64, 64, 512, 3, 1, 1, "D 512@64")
    K.bench_conv_fused(16, 128, 257, 257, 256, 3, 2, 0, "s2 128->256@257")
    K.bench_conv_fused(16, 128, 256, 256, 256, 1, 1, 0, "1x1 128->256@256")
if "tr" in sel or "s2" in sel:
    K.bench_conv(16, 128, 257, 257, 256, 3, 2, 0, "s2 128->256@257")
    K.bench_conv(16, 256, 129, 129, 512, 3, 2, 0, "s2 256->512@129")
