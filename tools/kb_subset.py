"""Small kernel-bench subset for tuning knobs: stride-2 family + big 3x3."""
import sys, os
sys.argv = [sys.argv[0]]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_bench as K
K.bench_conv(16, 128, 257, 257, 256, 3, 2, 0, "s2 128->256@257")
K.bench_conv(16, 256, 129, 129, 512, 3, 2, 0, "s2 256->512@129")
K.bench_conv(40, 512, 65, 65, 512, 3, 2, 0, "s2 512->512@65 B40")
K.bench_conv(16, 128, 256, 256, 128, 3, 1, 1, "s1 128@256")
K.bench_conv(128, 32, 128, 128, 32, 3, 1, 1, "Dp 32@128")
K.bench_conv(128, 64, 64, 64, 64, 3, 1, 1, "Dp 64@64")
K.bench_conv(16, 512, 64, 64, 512, 3, 1, 1, "D 512@64")
