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


def main(dirs):
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


if __name__ == "__main__":
    main(sys.argv[1:])
