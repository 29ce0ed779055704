#!/bin/bash
set -o pipefail
o=gpurun_out/r3u; mkdir -p $o
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $root/$o/avail.txt 2>&1; echo "list rc=$?"
grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*" $root/$o/avail.txt | sort -u | tr '\n' ' '; echo
for arm in stag old; do
  unset DCAMD_HALO_NO_STAG
  [ $arm = old ] && export DCAMD_HALO_NO_STAG=1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $root/$o/pmc_lds_$arm -- python3 $root/tools/bench_igemm.py --shapes c8_512_256,c16_256_128,c32_128_128 --reps 5 > $root/$o/pmc_lds_$arm.log 2>&1; echo "pmc lds $arm rc=$?"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $root/$o/pmc_sq_$arm -- python3 $root/tools/bench_igemm.py --shapes c8_512_256,c16_256_128,c32_128_128 --reps 5 > $root/$o/pmc_sq_$arm.log 2>&1; echo "pmc sq $arm rc=$?"
done
unset DCAMD_HALO_NO_STAG
cd $root
python3 - <<'PY'
import csv, glob, collections
for arm in ("stag","old"):
    for kind in ("lds","sq"):
        tot=collections.defaultdict(lambda: collections.defaultdict(float))
        for f in glob.glob(f"gpurun_out/r3u/pmc_{kind}_{arm}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "conv3" not in r["Kernel_Name"]: continue
                key=r["Kernel_Name"][:70]+" grid="+r.get("Grid_Size","")
                tot[key][r["Counter_Name"]]+=float(r["Counter_Value"])
        for k,c in tot.items():
            wc=c.get("SQ_WAVE_CYCLES",1)
            print(arm,kind,k,{n:round(v/wc,4) if n.startswith("SQ_") and n!="SQ_WAVE_CYCLES" else v for n,v in c.items()})
PY
