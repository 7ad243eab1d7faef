#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc pass (counter_collection.csv): one row per kernel, one column per counter,
plus derived per-wave figures when SQ_WAVES is present.  Usage: pmc_summary.py <counter_collection.csv> [out.md]"""
import csv
import sys
from collections import defaultdict


def main(path, out=None):
    agg = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    counters = sorted({c for v in agg.values() for c in v})
    lines = ["| kernel | dispatches | " + " | ".join(counters) + " |", "|---|---:|" + "---:|" * len(counters)]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get(counters[0], 0))):
        lines.append("| %s | %d | " % (k, len(disp[k])) + " | ".join("%.4g" % v.get(c, 0) for c in counters) + " |")
    if "SQ_WAVES" in counters:
        lines += ["", "| kernel | waves | VALU inst/wave | wave cycles/wave (x4 clk) | busy cycles/wave | WAIT_ANY % | WAIT_INST_ANY % | LDS bank conflict % |",
                  "|---|---:|---:|---:|---:|---:|---:|---:|"]
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
            w = v.get("SQ_WAVES", 0) or 1
            wc = v.get("SQ_WAVE_CYCLES", 0) or 1
            lines.append("| %s | %d | %.0f | %.0f | %.0f | %.1f | %.1f | %.1f |" % (
                k, w, v.get("SQ_INSTS_VALU", 0) / w, wc / w, v.get("SQ_BUSY_CYCLES", 0) / w, 100 * v.get("SQ_WAIT_ANY", 0) / wc,
                100 * v.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / (v.get("SQ_LDS_ACTIVE", 0) or 1) if "SQ_LDS_ACTIVE" in v else 0))
    text = "\n".join(lines)
    if out:
        with open(out, "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
