#!/bin/bash
# Per-kernel SQ counters (two passes) over a short bench run, for the kernels matching $2 (default: all of ours)
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-pmc_kernel}; NEEDLE=${2:-wdsr}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \
  --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU \
  --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $OUT/b.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/[ab]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$NEEDLE" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    g = {n: sum(v) / len(v) for n, v in c.items()}
    print(k)
    w = g.get("SQ_WAVE_CYCLES", 1); nw = g.get("SQ_WAVES", 1)
    for n in sorted(g):
        print(f"   {n:28s} {g[n]:14.0f}   per wave {g[n] / nw:10.1f}   of wave cycles {g[n] / w:6.3f}")
PY
