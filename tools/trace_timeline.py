"""Diagnostic: print the kernel timeline (start offset, duration, gap to previous end) of the last full step found in a
rocprofv3 kernel-trace csv.  usage: trace_timeline.py <kernel_trace.csv> <first-kernel-substring> [n_steps_back]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2]; back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
# step boundaries = first occurrence of key after a different kernel
starts = [i for n, i in enumerate(idx) if n == 0 or idx[n - 1] != i - 1]
a, b = starts[-back - 1], starts[-back]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0; busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3} {r['Kernel_Name'][:90]}")
    prev_end = max(prev_end, e); busy += e - s
print(f"step span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, {b - a} launches")
