#!/usr/bin/env python3
"""Fold a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE pass into per-kernel-family LDS-array cycles and conflict share.

  python tools/dev/lds_pmc_summary.py gpurun_out/r3ad/pmc_lds_old
"""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pmc_summary import family  # noqa: E402

tot = defaultdict(lambda: defaultdict(float))
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        fam = family(row["Kernel_Name"])
        if fam is not None:
            tot[fam][row["Counter_Name"]] += float(row["Counter_Value"])
for fam in sorted(tot):
    c = tot[fam]
    act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    print(f"{fam:36s} lds_active={act:.4g} bank_conflict={c.get('SQ_LDS_BANK_CONFLICT', 0.0):.4g} share={c.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(act, 1.0):.3f}")
