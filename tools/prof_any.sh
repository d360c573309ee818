#!/bin/bash
# rocprofv3 kernel-trace stats of an arbitrary python tool: tools/prof_any.sh <tag> <script.py>
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/$2 > $OUT/run.log 2> $OUT/run.err
cd $ROOT
python3 tools/summarize_rocprof.py $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.txt
cat $OUT/run.log
