#!/bin/bash
# SQ counters per kernel over a short bench run (which kernels lose LDS cycles to bank conflicts, which are parked)
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-pmc_sq_bench}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \
  --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $OUT/sq.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.match(r"(void )?(_Z\d+)?(\w+?)(I|\(|<)", n)
        key = (m.group(3) if m else n)[:34] + ("/R0" if "ELi0EEv" in n else "/R1" if "ELi1EEv" in n else "")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':40s} {'launches':>8s} {'lds_conf%':>9s} {'lds_act/busy':>12s} {'mfma_busy%':>10s} {'issue%':>7s} {'stall%':>7s} {'parked%':>8s}")
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0]))):
    g = lambda n: sum(c.get(n, [0])) / max(len(c.get(n, [1])), 1)
    if g("SQ_WAVE_CYCLES") == 0: continue
    busy = g("SQ_BUSY_CYCLES") / 32.0                     # per shader engine
    print(f"{k:40s} {len(c['SQ_WAVE_CYCLES']):8d} {100 * g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):9.1f} "
          f"{g('SQ_LDS_IDX_ACTIVE') / 256 / max(busy, 1):12.2f} {100 * g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / max(busy, 1):10.1f} "
          f"{100 * g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES'):7.1f} {100 * g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):7.1f} "
          f"{100 * g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):8.1f}")
PY
