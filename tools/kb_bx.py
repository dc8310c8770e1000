"""Kernel-bench subset for the bf16x6 arithmetic: 3x3 stride-1 layers of the church preset."""
import sys, os
sys.argv = [sys.argv[0]]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_bench as K
K.bench_conv(16, 128, 256, 256, 128, 3, 1, 1, "s1 128@256")
K.bench_conv(24, 256, 128, 128, 256, 3, 1, 1, "s1 256@128 B24")
K.bench_conv(16, 512, 64, 64, 512, 3, 1, 1, "D 512@64")
K.bench_conv(16, 512, 16, 16, 512, 3, 1, 1, "tail 512@16")
