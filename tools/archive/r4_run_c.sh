#!/bin/bash
# round-4 GPU session C: conv kernel tests on the product library with wg16, then the step
o=gpurun_out/r4C; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py tests/test_modconv.py tests/test_gpu_fullsize_oracle.py tests/test_quad_paths.py tests/test_gpu_parity.py tests/test_gpu_determinism.py -m gpu -q -x > $o/gputests.log 2>&1; tail -4 $o/gputests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 > $o/bench.json 2> $o/bench.err; cut -c1-400 $o/bench.json; tail -2 $o/bench.err
python - <<'PY'
import json
l = json.load(open("gpurun_out/r4C/bench.json"))
print("BY_KERNEL", [(k["class"][:28], k["ms_per_step"], k["frac"]) for k in l.get("roofline_by_kernel", [])])
PY
echo DONE
