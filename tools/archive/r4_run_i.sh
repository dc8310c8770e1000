#!/bin/bash
# round-4 GPU session I: the GPU suite on the current tree, the other presets
o=gpurun_out/r4I; mkdir -p $o
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_default.json 2> $o/bench_default.err; cut -c1-330 $o/bench_default.json; tail -2 $o/bench_default.err
python bench.py --preset ffhq512 --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 > $o/bench_ffhq512.json 2> $o/bench_ffhq512.err; cut -c1-330 $o/bench_ffhq512.json
python bench.py --preset ffhq1024 --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 > $o/bench_ffhq1024.json 2> $o/bench_ffhq1024.err; cut -c1-330 $o/bench_ffhq1024.json
python bench.py --preset tiny32 --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 > $o/bench_tiny32.json 2> $o/bench_tiny32.err; cut -c1-330 $o/bench_tiny32.json
echo DONE
python - <<'PY'
import json
for n in ("default", "ffhq512", "ffhq1024", "tiny32"):
    try:
        l = json.load(open("gpurun_out/r4I/bench_%s.json" % n))
    except Exception as e:
        print(n, "FAILED", e); continue
    print(n, l["value"], l["ms_per_step"], l.get("frac_of_mfma_f32_roofline"), l.get("ms_per_step_one_stream"), l.get("roofline", {}).get("frac"))
    print("  ", [(k["class"][:24], k["ms_per_step"], k["frac"]) for k in l.get("roofline_by_kernel", [])])
    if "via_dropin" in l: print("   via_dropin", l["via_dropin"].get("dropin_over_direct"), l["via_dropin"].get("ms_d_plus_g_median"))
PY
