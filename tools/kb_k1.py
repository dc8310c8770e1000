"""K1 micro-benchmark subset (church256 shapes + config 3)."""
import os, sys
sys.argv = [sys.argv[0]]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_bench as K
K.bench_blur(16 * 128, 256, 256, 4, 2)
K.bench_blur(16 * 128, 256, 256, 4, 1)
K.bench_blur(16 * 128, 257, 257, 4, 1)
K.bench_blur(128 * 32, 128, 128, 4, 2)
K.bench_blur(16 * 32, 259, 259, 3, 0)
K.bench_blur(16 * 512, 32, 32, 4, 2)
K.bench_blur(128 * 384, 8, 8, 4, 2)
K.bench_blur(8 * 128, 513, 513, 4, 1)
K.bench_blur(8 * 64, 512, 512, 4, 2)
