"""K1 / K2 micro-benchmark at the 512x512-class shapes SURVEY.md 8(d) names for BASELINE config 3 (ffhq512, B=8):
(8,128,513,513) -> 512^2 k4, (8,64,512,512) -> 513^2 / 511^2, (8,32,515,515) k3, leaky-ReLU (8,128,512,512).
Reports GB/s of algorithmic bytes.   python tools/kb_config3.py > gpurun_out/kb_config3.jsonl"""
import os
import sys
sys.argv = [sys.argv[0]]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_bench as K
K.bench_blur(8 * 128, 513, 513, 4, 1)      # after the transposed conv: 513 -> 512
K.bench_blur(8 * 64, 512, 512, 4, 2)       # before the 3x3 stride-2 conv: 512 -> 513
K.bench_blur(8 * 64, 512, 512, 4, 1)       # before the 1x1 stride-2 skip: 512 -> 511
K.bench_blur(8 * 32, 515, 515, 3, 0)       # encoder [1,2,1] after the reflection pad
K.bench_bias_act((8, 128, 512, 512))
K.bench_bias_act((8, 64, 512, 512))
K.bench_blur(4 * 32, 1024, 1024, 4, 2)     # ffhq1024 D: 1024 -> 1025
K.bench_blur(2 * 102, 1025, 1025, 4, 1)    # ffhq1024 G: 1025 -> 1024
K.bench_bias_act((4, 32, 1024, 1024))
