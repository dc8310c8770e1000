#!/bin/bash
# round-4 GPU session A: the new GPU tests (two ranks x full model on one GPU, framework stand-in under the drop-in runner),
# the default bench line with its via_dropin leg
o=gpurun_out/r4A; mkdir -p $o
python -m pytest tests/test_ddp_fullmodel.py tests/test_dropin_standin.py -m gpu -q > $o/gputests.log 2>&1; tail -15 $o/gputests.log
cp gpurun_out/ddp_fullmodel_one_gpu.txt $o/ 2>/dev/null; cat $o/ddp_fullmodel_one_gpu.txt
python bench.py --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; cut -c1-900 $o/bench_default.json; tail -3 $o/bench_default.err
python - <<'PY'
import json
l = json.load(open("gpurun_out/r4A/bench_default.json"))
print("VIA_DROPIN", json.dumps(l.get("via_dropin")))
print("BY_KERNEL", [(k["class"][:28], k["ms_per_step"], k["frac"]) for k in l.get("roofline_by_kernel", [])])
PY
echo DONE
