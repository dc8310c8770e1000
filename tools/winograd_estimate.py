"""Projection of the unfused Winograd F(2x2,3x3) route (csrc/winograd.hip) over the step's 3x3 stride-1 forward / data-gradient /
weight-gradient launches, from the per-shape ledger: for each launch with at least 128 contraction channels and even sides,
    t = (FLOP / 2.25) / rate_1x1(K)  (the sixteen 1x1 products; the 1x1 gather's measured TFLOP/s by K loop length, and at least
        the time to move V and M through HBM)  +  bytes(x + V + M + y) / 4.5 TB/s  (the two transform passes)
and the launch counts as a candidate when t is below its measured direct time.   python tools/winograd_estimate.py [ledger]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ledger = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r4_roofline_by_shape_church256.txt")
rows = []
for line in open(ledger):
    m = re.match(r'conv 3x3 s1 (fwd\S*|dgrad|wgrad)\s*(\(modulated\))?\s+n(\d+)\s+(\d+)->(\d+)\s+(\d+)x(\d+)\s+mfma\s+([\d.]+)\s+([\d.]+) GF\s+([\d.]+)', line)
    if m:
        op, mod, n, ci, co, h, w, calls, gf, ms = m.groups()
        rows.append((op[:5], bool(mod), int(n), int(ci), int(co), int(h), float(calls), float(gf), float(ms)))
total = cand = new = 0.0
print("%-6s %-4s %-5s %-11s %-5s %9s %9s" % ("op", "mod", "n", "channels", "map", "direct ms", "route ms"))
for op, mod, n, ci, co, h, calls, gf, ms in sorted(rows, key=lambda r: -r[8]):
    total += ms
    k = ci if op.startswith("fwd") else co
    m_ = co if op.startswith("fwd") else ci
    if op == "wgrad":            # the map side is the OUTPUT's (pad 1: the same); both activations are transformed (5x each)
        opix = round(gf * 1e9 / (18.0 * ci * co * n * calls))
        h = int(round(opix ** 0.5))
        k, m_ = ci, co
    pix = n * h * h
    if min(ci, co) < 128 or h < 8 or h % 2:
        continue
    if op == "wgrad":
        gemm = max((gf / 2.25) / 105.0, 4.0 * (4 * k * pix + 4 * m_ * pix) * calls / 4.5e9)   # the 1x1 weight-gradient kernel's rate
        t = gemm + 4.0 * (5 * k * pix + 5 * m_ * pix) * calls / 4.5e9
    else:
        rate = 100.0 if k >= 512 else 92.0 if k >= 256 else 80.0        # TFLOP/s of the 1x1 gather (profiles/r4_ab_1x1_occupancy.txt)
        gemm = max((gf / 2.25) / rate, 4.0 * (4 * k * pix + 4 * m_ * pix) * calls / 4.5e9)
        t = gemm + 4.0 * (5 * k * pix + 5 * m_ * pix) * calls / 4.5e9
    if t < ms:
        cand += ms
        new += t
        print("%-6s %-4d n%-4d %4d->%-5d @%-4d %9.3f %9.3f" % (op, mod, n, ci, co, h, ms, t))
print("3x3 stride-1 forward + data gradient + weight gradient: %.1f ms per iteration; candidates %.1f ms -> %.1f ms (%.1f ms less)" % (total, cand, new, cand - new))
