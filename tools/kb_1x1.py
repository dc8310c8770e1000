"""Kernel-bench subset: 1x1 convolutions of the church preset."""
import sys, os
sys.argv = [sys.argv[0]]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_bench as K
K.bench_conv(16, 128, 256, 256, 256, 1, 1, 0, "1x1 128->256@256")
K.bench_conv(40, 256, 64, 64, 512, 1, 1, 0, "1x1 256->512@64 B40")
K.bench_conv(16, 512, 64, 64, 256, 1, 1, 0, "G 1x1 512->256@64")
K.bench_conv(16, 128, 257, 257, 256, 1, 2, 0, "1x1 s2 128->256")
