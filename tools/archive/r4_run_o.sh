#!/bin/bash
# round-4 session O: kernel tests + the driver's bench command after the plan models and the 1x1 occupancy change
o=gpurun_out/r4_o; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py tests/test_quad_paths.py tests/test_gpu_fullsize_oracle.py -m gpu -x -q 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; cut -c1-200 $o/bench_default.json
python - <<'PY'
import json
l = json.load(open("gpurun_out/r4_o/bench_default.json"))
print(l["value"], l["ms_per_step"], l["frac_of_mfma_f32_roofline"], l["ms_per_step_one_stream"], l["roofline"]["frac"])
print([(k["class"][:22], k["ms_per_step"], k["frac"]) for k in l["roofline_by_kernel"]])
PY
echo DONE
