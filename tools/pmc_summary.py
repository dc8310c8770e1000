"""Merge rocprofv3 --pmc passes (--output-format csv) into one per-kernel table.
   python tools/pmc_summary.py gpurun_out/pmcA gpurun_out/pmcB ...
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); GRBM_GUI_ACTIVE is summed over
the 8 XCDs, so clock = GRBM_GUI_ACTIVE / 8 / duration."""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("sae::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:60]


# Round 5: the dominant kernel is the one-kernel Winograd convolution; tools/pmc_wino_fused.py's reference launch of it is
# 512 -> 512 3x3 @64x64, B = 16 (grid 2048 workgroups).  gflop = the FLOPs it EXECUTES (16 multiply-adds per 2x2 tile and channel
# pair); algorithmic bytes = x and y once + the prepared weights once.
DOMINANT = "wino_fused_kernel<false,false,0>"      # (third argument: 0 = no input scale; before round 6's last session a bool)
DOMINANT_REF = {"gflop": 32.0 * 16 * 512 * 512 * 32 * 32 / 1e9,
                "algorithmic_bytes": 2.0 * 16 * 512 * 64 * 64 * 4 + 16.0 * 512 * 512 * 4,
                "launch": "512 -> 512 3x3 stride 1 @64x64, B = 16, forward (tools/pmc_wino_fused.py)"}
# (rounds 2-4: conv_igemm_kernel<3,1,2,2,1,4,8,false,true> on 128 -> 128 @256x256, B = 16: profiles/r4_pmc_dominant.json)


def dominant_record(agg, source):
    """bench.py's roofline.traffic record: request-level HBM-side traffic of the dominant kernel's reference launch."""
    keys = [k for k in agg if k[0].replace(" ", "").startswith(DOMINANT)]
    if not keys:
        raise SystemExit("pmc_summary: no %s launch in the passes" % DOMINANT)
    biggest = max(int(k[1]) for k in keys)          # the reference launch (the split-K tail runs the same template on a small grid)
    rows = [agg[k] for k in keys if int(k[1]) == biggest]
    need = ("TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum")
    rows = [{c: sum(v) / len(v) for c, v in m.items()} for m in rows]
    rows = [m for m in rows if all(c in m for c in need)]
    if not rows:
        raise SystemExit("pmc_summary: the passes lack %s" % (need,))
    m = {c: sum(r[c] for r in rows) / len(rows) for c in need + ("TCC_EA0_RDREQ_32B_sum",) if all(c in r for r in rows)}
    rd32 = m.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    rec = dict(DOMINANT_REF)
    rec.update({"kernel": DOMINANT, "read_requests": m["TCC_EA0_RDREQ_sum"], "read_requests_32B": rd32,
                "read_bytes": (m["TCC_EA0_RDREQ_sum"] - rd32) * 128 + rd32 * 32,
                "write_requests": m["TCC_EA0_WRREQ_sum"], "write_requests_64B": m["TCC_EA0_WRREQ_64B_sum"],
                "write_bytes": m["TCC_EA0_WRREQ_64B_sum"] * 64 + (m["TCC_EA0_WRREQ_sum"] - m["TCC_EA0_WRREQ_64B_sum"]) * 32,
                "source": source,
                "method": "rocprofv3 --pmc, one counter set per run, kernel trace only; TCC_EA0_RDREQ x 128 B (32-byte requests "
                          "x 32 B), TCC_EA0_WRREQ_64B x 64 B + other write requests x 32 B; mean over the launches of the passes"})
    rec["traffic_over_algorithmic"] = round((rec["read_bytes"] + rec["write_bytes"]) / rec["algorithmic_bytes"], 3)
    return rec


def main(dirs):
    out_json = None
    if "--dominant-json" in dirs:
        i = dirs.index("--dominant-json")
        out_json = dirs[i + 1]
        dirs = dirs[:i] + dirs[i + 2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = (short(r["Kernel_Name"]), r["Grid_Size"])
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if "Start_Timestamp" in r and r.get("End_Timestamp"):
                    dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, cs in agg.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        if m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) == 0 and not any(t in k[0] for t in ("blur", "elementwise", "copy", "reduce", "conv_")):
            continue
        us = sum(dur[k]) / len(dur[k]) / 1e3 if dur[k] else 0.0
        gui = m.get("GRBM_GUI_ACTIVE", 0.0)
        line = "%-58s grid=%-9s avg_us=%8.1f" % (k[0], k[1], us)
        if gui and us:
            line += "  clock~%.2fGHz  mfma_util=%.3f" % (gui / 8 / us / 1e3, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * gui / 8))
        if "FETCH_SIZE" in m or "WRITE_SIZE" in m:      # reported in KiB
            line += "  FETCH=%.1f MB (as reported)  WRITE=%.1f MB" % (m.get("FETCH_SIZE", 0) * 1024 / 1e6, m.get("WRITE_SIZE", 0) * 1024 / 1e6)
        print(line)
        print("   " + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(m.items())))
    if out_json:
        import json
        with open(out_json, "w") as f:
            json.dump(dominant_record(agg, "tools/run_pmc.sh passes " + " ".join(dirs)), f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main(sys.argv[1:])
