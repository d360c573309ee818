#!/bin/bash
# kernel-trace of the single-rank DDP/RCCL bench (what DDP adds to the step)
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-prof_ddp}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --force-ddp > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
python3 tools/summarize_rocprof.py $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.txt
python3 - <<PY
import csv, glob, re
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "FusedAdam" in r["Kernel_Name"]]
a, b = idx[40], idx[41]
prev = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"gap {(s - prev) / 1e3:6.2f} dur {(e - s) / 1e3:7.2f} {re.sub('^void ', '', r['Kernel_Name'])[:70]}")
    prev = e
PY
