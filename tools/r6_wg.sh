#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6p
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6p/upsample_residual_loads.txt
import os, sys, torch
sys.path.insert(0, '.')
from swapping_autoencoder_pytorch_amd import hip_lib as L
dev = torch.device('cuda:0'); st = torch.cuda.current_stream(dev).cuda_stream
libs = [('up_head', L.SaeLibrary('tools/variants/up_head.so')), ('product', L.SaeLibrary(L.DEFAULT_LIBRARY))]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for planes, h in [(16 * 128, 128), (16 * 256, 64), (16 * 512, 32), (16 * 512, 16)]:
    x = torch.randn(planes, h, h, device=dev); res = torch.randn(planes, 2 * h, 2 * h, device=dev)
    gb = 4.0 * (x.numel() + 2 * res.numel()) / 1e9
    row = []
    outs = []
    for rnd in range(2):
        for name, lib in libs:
            y = torch.empty_like(res)
            fn = lambda: lib.call('upsample2x_bilinear_add_f32', x.data_ptr(), res.data_ptr(), y.data_ptr(), planes, h, h, 2 ** -0.5, st)
            fn(); outs.append(y.clone()); row.append('%s %.2f TB/s' % (name, gb / timeit(fn)))
    print('upsample x2 + residual %5d planes %3d^2: ' % (planes, h) + '  '.join(row) + ('  identical' if all(torch.equal(o, outs[0]) for o in outs) else '  DIFFERS'), flush=True)
PY
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize_properties.py -m gpu -x -q -k "upsample" 2>&1 | tail -2
