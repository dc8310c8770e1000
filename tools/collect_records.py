"""Copy a final-records session (tools/r6_run_*.sh -> gpurun_out/<dir>) into profiles/ under the round's names.
   python tools/collect_records.py gpurun_out/r6q r6 "session Q" """
import csv
import json
import os
import re
import shutil
import sys

src, rnd, tag = sys.argv[1], sys.argv[2], sys.argv[3]
P = "profiles"


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("sae::", "")
    return re.sub(r"^void ", "", name)[:160]


def trace(csv_path, log_path, out, cmd, what):
    log = open(log_path).read()
    v = re.search(r'"value": ([0-9.]+)', log)
    ms = re.search(r'"ms_per_step": ([0-9.]+)', log)
    rows = list(csv.DictReader(open(csv_path)))[:60]
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats of `%s`\n" % cmd)
        f.write("# (round 6 %s, church256 B=16: %s; %s images/s, %s ms per step in this run; the 60 heaviest kernels)\n" % (
            tag, what, v.group(1) if v else "?", ms.group(1) if ms else "?"))
        f.write("#    pct  calls   total_ms     avg_us  kernel\n")
        for r in rows:
            f.write("%8.2f %6d %10.3f %10.1f  %s\n" % (float(r["Percentage"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                    float(r["AverageNs"]) / 1e3, short(r["Name"])))


B = "python bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
trace(os.path.join(src, "trace_graph_kernel_stats_top70.csv"), os.path.join(src, "trace_graph.log"),
      os.path.join(P, rnd + "_step_church256_b16_f32_kernel_trace.txt"), B,
      "hipGraph replay + two streams = the mode bench.py times `value` in")
trace(os.path.join(src, "trace_one_kernel_stats_top70.csv"), os.path.join(src, "trace_one.log"),
      os.path.join(P, rnd + "_step_church256_b16_f32_kernel_trace_one_stream.txt"), "SAE_TWO_STREAMS=0 " + B + " --no-graph",
      "eager calls, the step on ONE stream -- the mode bench.py's kernel pass measures `roofline.avg_launch_ms` in")

line = [x for x in open(os.path.join(src, "bench_default.json")).read().splitlines() if x.startswith("{")][-1]
json.loads(line)
open(os.path.join(P, rnd + "_bench_default.json"), "w").write(line + "\n")
two = [x for x in open(os.path.join(src, "two_ranks_church256.json")).read().splitlines() if x.startswith("{")]
if two:
    open(os.path.join(P, rnd + "_bench_two_ranks_one_gpu_staged.json"), "w").write(two[-1] + "\n")
for a, b in [("gpu_tests.log", "_gpu_tests.log"), ("fullsize_parity.jsonl", "_fullsize_parity.jsonl"),
             ("network_parity.jsonl", "_network_parity.jsonl"), ("network_parity_tensors.jsonl", "_network_parity_tensors.jsonl"),
             ("step_parity_fullsize.jsonl", "_step_parity_fullsize.jsonl"), ("pmc_dominant.json", "_pmc_dominant.json"),
             ("pmc_summary.txt", "_pmc_wino_fused.txt"), ("roofline_by_kernel_church256.txt", "_roofline_by_kernel_church256.txt"),
             ("roofline_by_kernel_ffhq512.txt", "_roofline_by_kernel_ffhq512.txt"),
             ("roofline_by_kernel_ffhq1024.txt", "_roofline_by_kernel_ffhq1024.txt"),
             ("roofline_by_shape_church256.txt", "_roofline_by_shape_church256.txt")]:
    if os.path.exists(os.path.join(src, a)) and os.path.getsize(os.path.join(src, a)) > 0:
        shutil.copy(os.path.join(src, a), os.path.join(P, rnd + b))
    else:
        print("missing:", a)
print("collected", src, "->", P)
