#!/bin/bash
# SQ counter passes (two separate --pmc runs) over one driver script, summarised for the kernels whose name contains $2:
#   tools/pmc_sq_kernel.sh <out name> <kernel needle> <driver.py> [driver args]
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; NEEDLE=$2; DRIVER=$3; shift 3; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $OUT/sq -- python3 $ROOT/$DRIVER "$@" > $OUT/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/sq2 -- python3 $ROOT/$DRIVER "$@" > $OUT/sq2.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, json, collections
out = {}
for f in glob.glob("$OUT/sq*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "$NEEDLE" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = {"launches": len(v), "mean": sum(v) / len(v)}
json.dump(out, open("$OUT/sq_raw.json", "w"), indent=1)
w = out.get("SQ_WAVE_CYCLES", {}).get("mean", 0)
for k, v in sorted(out.items()):
    print("%-28s %14.0f %s" % (k, v["mean"], ("%.3f of wave cycles" % (v["mean"] / w)) if w else ""))
b = out.get("SQ_BUSY_CYCLES", {}).get("mean", 0) / 32.0
if b:
    print("kernel cycles per SE %.0f; MFMA busy %.3f; VALU per MFMA %.2f; LDS conflict share %.3f" % (
        b, out["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024 / b, out["SQ_INSTS_VALU"]["mean"] / max(out["SQ_INSTS_MFMA"]["mean"], 1),
        out["SQ_LDS_BANK_CONFLICT"]["mean"] / max(out["SQ_LDS_IDX_ACTIVE"]["mean"], 1)))
PY
