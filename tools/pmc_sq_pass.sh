#!/bin/bash
# SQ counter passes over the two-block forward kernel (wave cycles, stall buckets, MFMA busy, LDS conflicts, instruction mix)
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-pmc_sq}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $OUT/sq -- python3 $ROOT/tools/pmc_block2_fwd.py > $OUT/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/sq2 -- python3 $ROOT/tools/pmc_block2_fwd.py > $OUT/sq2.log 2>&1 || true
cd $ROOT
python3 - <<PY
import csv, glob, json, collections
out = {}
for f in glob.glob("$OUT/sq*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "fwd_rs" in r["Kernel_Name"] or "block2_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = {"launches": len(v), "mean": sum(v) / len(v)}
json.dump(out, open("$OUT/sq_raw.json", "w"), indent=1)
w = out.get("SQ_WAVE_CYCLES", {}).get("mean", 0)
for k, v in sorted(out.items()):
    print("%-28s %14.0f %s" % (k, v["mean"], ("%.3f of wave cycles" % (v["mean"] / w)) if w else ""))
PY
