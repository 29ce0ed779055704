#!/usr/bin/env python3
"""Fold one rocprofv3 --pmc SQ pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE) into a per-kernel-family JSON (families as tools/pmc_summary.py names them).

  python tools/pmc_sq_summary.py gpurun_out/r2i/pmc_sq profiles/r02_pmc_sq.json "comment"
"""
import csv
import glob
import json
import sys
from collections import defaultdict

from pmc_summary import family


def main():
    root, out, comment = sys.argv[1], sys.argv[2], sys.argv[3]
    tot = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            fam = family(row["Kernel_Name"])
            if fam is None:
                continue
            tot[fam][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[fam].add(row["Dispatch_Id"])
    rec = {"_comment": comment}
    for fam in sorted(tot):
        c = tot[fam]
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        rec[fam] = {"launches": len(launches[fam]),
                    "mfma_util": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(c.get("GRBM_GUI_ACTIVE", 0.0) * 128, 1.0), 4),
                    "wait_any": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 4), "wait_inst": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
                    "active_inst": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4)}
    json.dump(rec, open(out, "w"), indent=1)
    for k, v in rec.items():
        print(k, v)


if __name__ == "__main__":
    main()
