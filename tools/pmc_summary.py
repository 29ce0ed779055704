#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter CSVs (one pass per counter, as MI355X_MICROARCH.md prescribes) into
profiles/pmc_traffic.json: per kernel FAMILY (the names bench.py / dc_igemm_variant use) the per-launch
averages of FETCH_SIZE and WRITE_SIZE in KB.  bench.py applies the gfx950 correction (FETCH_SIZE x2) itself.

  python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_traffic.json "comment" [workload]

With a fifth argument the record is MERGED into the existing file under "workloads" / <workload> (bench.py reads the other BASELINE
configurations' traffic from there); without it the file is rewritten with the headline workload's families at the top level.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def family(kernel):
    """kernel name as rocprofv3 prints it (Itanium-mangled, occasionally demangled) -> the family name bench.py
    reports (None: not one of ours)."""
    k = kernel.replace(" ", "")
    dt = "bf16" if ("DF16b" in k or "__bf16" in k) else ("f16" if ("DF16_" in k or "_Float16" in k) else "f32")
    ints = [int(v.replace("n", "-")) for v in re.findall(r"Li(n?\d+)E", k)] or [int(v) for v in re.findall(r"[<,](-?\d+)(?=[,>])", k)]
    if "conv3_halo_kernel" in k or "conv3_halo_pers_kernel" in k:     # <T, NW, (GN,) NTAP, ...>: the persistent form is the same family
        pn = bool(re.search(r"Lb[01]ELb1EE", k) or re.search(r"(true|false),true>", k))      # <..., STG, PN = true>: producer-side GroupNorm
        if len(ints) > 1 and ints[1] == 4:                      # NTAP = 4: the four-phase upsample conv
            return f"conv3_up4<{dt},{ints[0]}w{',pn' if pn else ''}>"
        if pn:
            return f"conv3_halo<{dt},{ints[0]}w,pn>"
        return f"conv3_halo<{dt},{ints[0]}w>"
    if "conv3_ws_kernel" in k or "conv3_wsp_kernel" in k:      # <T, GN>: wave-specialised halo conv (the persistent form is the same family)
        gn = "Lb1E" in k or "true" in k
        if "_Accum" in k:          # rocprofv3's demangler garbles <__bf16, bool>: the bench runs it in bf16 only
            dt, gn = "bf16", True
        return f"conv3_ws<{dt},gn>" if gn else f"conv3_ws<{dt}>"
    if "tblock_front_kernel" in k:
        return "tblock_front"
    if "igemm_wide8_kernel" in k:
        return f"igemm_wide8<{dt},256x256>"
    if "conv3_thin_kernel" in k:
        if "_Accum" in k:          # rocprofv3's demangler garbles <__bf16, bool> into "<bool _Accum, bool, E>": the bench runs it in bf16 only
            dt = "bf16"
        return f"conv3_thin<{dt}>"
    if "igemm_pipe_kernel" in k:
        bm, st, nh = ints[0], ints[1], ints[2]
        if len(ints) > 3 and ints[3] == 5:                      # EV = 5: four-phase upsample conv on the tap-gather kernel
            return f"igemm_pipe_up4<{dt},{bm}x{128 * nh},{st}st>"
        return f"igemm_pipe<{dt},{bm}x{128 * nh},{st}st>"
    if "igemm_xreg_kernel" in k:
        return "igemm_xreg<bf16,96xN>" if dt == "f32" else f"igemm_xreg<{dt},96xN>"   # demangler drops the type: 16-bit only kernel
    if "igemm_kernel" in k:
        return f"igemm<{dt},{ints[0]}x{ints[1]}>"
    if "gn_" in k:
        return "groupnorm"
    return None


def collect(root, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    files = glob.glob(f"{root}/**/*counter_collection.csv", recursive=True)
    assert files, f"no counter_collection.csv under {root}"
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            fam = family(row["Kernel_Name"])
            if fam is None:
                continue
            tot[fam] += float(row["Counter_Value"])
            cnt[fam] += 1
    return tot, cnt


def main():
    fetch_dir, write_dir, out, comment = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    workload = sys.argv[5] if len(sys.argv) > 5 else None
    ft, fc = collect(fetch_dir, "FETCH_SIZE")
    wt, wc = collect(write_dir, "WRITE_SIZE")
    rec = {"_comment": comment}
    for fam in sorted(ft):
        rec[fam] = {"fetch_kb_per_launch": round(ft[fam] / fc[fam], 1),
                    "write_kb_per_launch": round(wt.get(fam, 0.0) / max(wc.get(fam, 0), 1), 1), "launches": fc[fam]}
    if workload:
        import os
        full = json.load(open(out)) if os.path.exists(out) else {}
        full.setdefault("workloads", {})[workload] = rec
        rec = full
    else:
        import os
        if os.path.exists(out):            # keep the other workloads' records
            old = json.load(open(out)).get("workloads")
            if old:
                rec["workloads"] = old
    json.dump(rec, open(out, "w"), indent=1)
    for k, v in rec.items():
        print(k, v)


if __name__ == "__main__":
    main()
