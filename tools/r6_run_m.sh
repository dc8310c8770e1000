#!/bin/bash
# round 6 session M: deep-prefetch schedule of the one-kernel Winograd kernels as the product; A/B record; GPU tests; bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6m
timeout 900 python tools/wf_variants.py wf_before product 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6m/ab_wino_deep_prefetch.txt
timeout 1200 python -m pytest tests/test_winograd.py tests/test_s2wino.py tests/test_gpu_fullsize_oracle.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r6m/tests.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r6m/bench.json 2> gpurun_out/r6m/bench.err
tail -c 1500 gpurun_out/r6m/bench.json
echo SESSION_M_DONE
