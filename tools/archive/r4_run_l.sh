#!/bin/bash
o=gpurun_out/r4L; mkdir -p $o
run() { env "$@" python bench.py --steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --no-kernel-timing $EXTRA 2>$o/tmp.err | grep '^{' > $o/tmp.json
  python - <<PY
import json
try:
    l = json.load(open("$o/tmp.json"))
    print("$* $EXTRA", l["value"], l["ms_per_step"], l["ms_d_call_median"], l["ms_g_call_median"])
except Exception as e:
    print("$*", "FAILED", open("$o/tmp.err").read()[-300:])
PY
}
EXTRA=""; run SAE_X=0
EXTRA="--force-allreduce"; run SAE_X=0
EXTRA="--force-allreduce"; run GPU_MAX_HW_QUEUES=4
EXTRA=""; run SAE_X=0
EXTRA="--force-allreduce"; run SAE_X=0
python -m pytest tests/test_ddp_fullmodel.py tests/test_gpu_allreduce.py -m gpu -q -x 2>&1 | tail -2
echo DONE
