#!/bin/bash
# Round profile pass on the GPU box: kernel-trace stats of the default bench, then three separate PMC passes
# over the two-block forward kernel.  Output under gpurun_out/<tag>/ (copy the summaries into profiles/).
set -e -o pipefail
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$OUT/pmc_$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 $ROOT/tools/pmc_block2_fwd.py > $d.log 2>&1
done
cd $ROOT
python3 tools/summarize_rocprof.py $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.txt $(find $OUT/trace -name '*kernel_trace.csv' | head -1) fwd_rs
python3 - <<PY
import csv, glob, json, collections
out = {}
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "fwd_rs" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
json.dump(out, open("$OUT/pmc_fwd_rs2_raw.json", "w"), indent=1)
print(json.dumps(out))
PY
cat $OUT/bench.json
