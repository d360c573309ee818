#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` into a short table (kernel names shortened) for profiles/."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_Z\d+(\w+?)I(DF16b|f)Li(\d+)", name)
    if m:
        return f"{m.group(1)}<{'bf16' if m.group(2) == 'DF16b' else 'fp32'},{m.group(3)}>"
    name = re.sub(r"at::native::(\(anonymous namespace\)::)?", "", name)
    return name[:90]


def main(path, out):
    rows = list(csv.DictReader(open(path)))
    with open(out, "w") as f:
        f.write(f"# source: {path}\n")
        f.write(f"{'kernel':92s} {'calls':>7s} {'avg_us':>10s} {'total_ms':>10s} {'pct':>6s}\n")
        for r in rows:
            f.write(f"{short(r['Name']):92s} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:10.2f} "
                    f"{float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['Percentage']):6.2f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
