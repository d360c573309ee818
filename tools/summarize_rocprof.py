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


def by_grid(trace, out, needle):
    """append, from the `*_kernel_trace.csv` of the same run, the launches of kernels matching `needle` split by grid
    size (the bench launches the graded kernel at batch 32 and, for one extra figure, at batch 512)"""
    import collections
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if needle in r["Kernel_Name"]:
            wgs = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            acc[(short(r["Kernel_Name"])[:60], wgs)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(out, "a") as f:
        f.write(f"\n# `{needle}` launches split by workgroup count (from the kernel trace of the same run)\n")
        f.write(f"{'kernel':62s} {'workgroups':>10s} {'calls':>7s} {'avg_us':>10s} {'median_us':>10s}\n")
        for (k, wgs), v in sorted(acc.items()):
            v.sort()
            f.write(f"{k:62s} {wgs:10d} {len(v):7d} {sum(v) / len(v) / 1e3:10.2f} {v[len(v) // 2] / 1e3:10.2f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
    if len(sys.argv) > 4:
        by_grid(sys.argv[3], sys.argv[2], sys.argv[4])
