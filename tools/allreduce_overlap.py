"""Do the RCCL kernels of the gradient all-reduce run concurrently with the backward pass's kernels, or serialised behind
them?  Reads a `rocprofv3 --kernel-trace --output-format csv` directory of `bench.py --force-allreduce` and prints, for the
collective kernels (names containing nccl / rccl): count, total time, the part of that time during which at least one
non-collective kernel was executing on the device, and which kernels they overlapped most.

    python tools/allreduce_overlap.py gpurun_out/ar_trace > profiles/r3_allreduce_1rank_trace.txt"""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("sae::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:70]


def main(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    if not rows:
        raise SystemExit("no kernel_trace.csv under %s" % d)
    rows.sort()
    coll = [r for r in rows if re.search(r"nccl|rccl", r[2], re.I)]
    other = [r for r in rows if not re.search(r"nccl|rccl", r[2], re.I)]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print("# kernels: %d in %.1f ms of trace; collective kernels: %d, other: %d" % (len(rows), (t1 - t0) / 1e6, len(coll), len(other)))
    print("# queues: collective %s, other %s" % (sorted({r[3] for r in coll}), sorted({r[3] for r in other})))
    if not coll:
        print("no RCCL kernels in the trace")
        return
    # sweep: for every collective kernel interval, the time covered by at least one other kernel
    starts = [r[0] for r in other]
    import bisect
    total = covered = 0
    partners = collections.Counter()
    for s, e, name, _ in coll:
        total += e - s
        i = bisect.bisect_left(starts, s)
        j = max(0, i - 64)                      # kernels that started before s may still run
        segs = []
        for os_, oe, oname, _ in other[j:]:
            if os_ >= e:
                break
            lo, hi = max(s, os_), min(e, oe)
            if hi > lo:
                segs.append((lo, hi))
                partners[short(oname)] += hi - lo
        segs.sort()
        cur_lo = cur_hi = None
        for lo, hi in segs:
            if cur_hi is None or lo > cur_hi:
                if cur_hi is not None:
                    covered += cur_hi - cur_lo
                cur_lo, cur_hi = lo, hi
            else:
                cur_hi = max(cur_hi, hi)
        if cur_hi is not None:
            covered += cur_hi - cur_lo
    print("collective kernel time: %.3f ms total, %.3f ms (%.1f %%) concurrent with another kernel, %.3f ms alone on the device"
          % (total / 1e6, covered / 1e6, 100.0 * covered / max(total, 1), (total - covered) / 1e6))
    names = collections.Counter()
    for s, e, name, _ in coll:
        names[short(name)] += e - s
    for n, t in names.most_common(4):
        print("  collective kernel %-70s %.3f ms" % (n, t / 1e6))
    for n, t in partners.most_common(8):
        print("  overlapped with  %-70s %.3f ms" % (n, t / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
