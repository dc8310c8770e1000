"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite database) into the per-kernel table that
is committed under profiles/:  python tools/prof_summary.py gpurun_out/prof2 > profiles/r1_xxx.txt"""
import glob
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = name.replace("sae::", "").replace("at::native::", "aten::")
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:100]


def main(path, top=40):
    dbs = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    if not dbs:
        raise SystemExit("no rocpd database under %s" % path)
    cur = sqlite3.connect(dbs[0]).cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            "from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows)
    span = list(cur.execute("select min(start), max(end) from kernels"))[0]
    print("# rocprofv3 --kernel-trace summary of %s" % os.path.basename(dbs[0]))
    print("# kernels: %d dispatches, %.2f ms busy, %.2f ms first-start..last-end" %
          (sum(r[1] for r in rows), total / 1e6, (span[1] - span[0]) / 1e6))
    # with more than one stream kernels overlap: time with at least one kernel resident (union of the intervals), and the
    # time during which two or more were
    ev = []
    for st, en in cur.execute("select start, end from kernels"):
        ev.append((st, 1)); ev.append((en, -1))
    ev.sort()
    depth, last, any_ns, multi_ns = 0, None, 0, 0
    for t, d in ev:
        if last is not None and depth > 0:
            any_ns += t - last
            if depth > 1:
                multi_ns += t - last
        depth += d
        last = t
    print("# at least one kernel resident %.2f ms (%.1f %% of the span), two or more %.2f ms; sum of durations / resident time = %.3f"
          % (any_ns / 1e6, 100.0 * any_ns / max(span[1] - span[0], 1), multi_ns / 1e6, total / max(any_ns, 1)))
    print("%10s %6s %7s %11s %11s %11s  %s" % ("total_ms", "pct", "calls", "avg_us", "min_us", "max_us", "kernel"))
    for name, cnt, s, avg, mn, mx in rows[:top]:
        print("%10.3f %6.2f %7d %11.1f %11.1f %11.1f  %s" % (s / 1e6, 100.0 * s / total, cnt, avg / 1e3, mn / 1e3, mx / 1e3,
                                                             short(name)))
    rest = rows[top:]
    if rest:
        print("%10.3f %6.2f %7d %11s %11s %11s  (%d other kernels)" % (sum(r[2] for r in rest) / 1e6,
              100.0 * sum(r[2] for r in rest) / total, sum(r[1] for r in rest), "", "", "", len(rest)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
