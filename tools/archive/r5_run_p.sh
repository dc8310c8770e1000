#!/bin/bash
# Round 5, GPU session P: ring rehearsal on its own stream WITHOUT the wait for the bucket; and a kernel trace of the costly form.
o=gpurun_out/r5p; mkdir -p $o
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
SAE_RING_REHEARSAL_STREAM=nowait python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "ring8 on its own stream, not waiting for the bucket" | tee -a $o/ring_variants2.txt
root=$(pwd); export TMPDIR=/tmp; cd /tmp
T="--steps 4 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $root/$o/trace_own -- python $root/bench.py $T --force-allreduce --ring-rehearsal 8 > $root/$o/trace_own.log 2>&1
SAE_RING_REHEARSAL_STREAM=launch timeout 300 rocprofv3 --kernel-trace --output-format csv -d $root/$o/trace_launch -- python $root/bench.py $T --force-allreduce --ring-rehearsal 8 > $root/$o/trace_launch.log 2>&1
cd $root
python - <<'PY'
import csv, glob, collections
for tag in ("own", "launch"):
    fs = glob.glob("gpurun_out/r5p/trace_%s/**/*kernel_trace.csv" % tag, recursive=True)
    rows = []
    for f in fs:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    if not rows:
        print(tag, "no rows"); continue
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    # last third of the trace = steady state
    lo = t0 + (t1 - t0) * 2 // 3
    sel = [r for r in rows if r[0] >= lo]
    byq = collections.Counter(); busy = collections.Counter()
    for s, e, n, q, st in sel:
        byq[q] += 1; busy[q] += e - s
    span = (max(r[1] for r in sel) - sel[0][0]) / 1e6
    # union of busy time and time with >= 2 kernels in flight
    ev = []
    for s, e, n, q, st in sel:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth = 0; last = ev[0][0]; t_any = 0; t_two = 0
    for t, d in ev:
        if depth >= 1: t_any += t - last
        if depth >= 2: t_two += t - last
        depth += d; last = t
    elem = [r for r in sel if "elementwise" in r[2] or "CUDAFunctor_add" in r[2]]
    print("%s: last third of the trace %.1f ms: kernels %d, queues %s" % (tag, span, len(sel), {q: (byq[q], round(busy[q] / 1e6, 1)) for q in byq}))
    print("   device busy %.1f ms, two or more kernels in flight %.1f ms; ATen elementwise kernels: %d, %.2f ms, on queues %s" % (
        t_any / 1e6, t_two / 1e6, len(elem), sum(e - s for s, e, *_ in elem) / 1e6, sorted({r[3] for r in elem})))
PY
find $o -name "*.csv" -size +1M -delete; find $o -name "*.db" -delete
echo SESSION_P_DONE
