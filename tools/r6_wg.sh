#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6p
for r in 1 2; do
echo "== head"; timeout 300 python tools/ab_s2wino.py dgrad --mod --only "G up" --lib tools/variants/s2w_head.so 2>&1 | grep -v amdgpu.ids
echo "== new";  timeout 300 python tools/ab_s2wino.py dgrad --mod --only "G up" 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r6p/s2w_epilogue_mod.txt
timeout 900 python -m pytest tests/test_s2wino.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r6p/s2w_tests.txt
