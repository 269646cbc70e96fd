#!/usr/bin/env python3
"""GPU-busy time of a rocprofv3 kernel trace: union of the kernel intervals over the last `steps` occurrences of a marker kernel (default: any adamw kernel).
    python tools/busy_union.py <..._kernel_trace.csv> [steps] [marker substring]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mark = sys.argv[3] if len(sys.argv) > 3 else "adamw"
idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
a, b = idx[-steps - 1], idx[-1]
t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["End_Timestamp"])
busy, cur_s, cur_e, n, total = 0, None, None, 0, 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n += 1; total += e - s
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"{steps} steps: wall {1e-6 * (t1 - t0) / steps:.3f} ms/step, GPU busy (union) {1e-6 * busy / steps:.3f} ms/step, sum of kernel durations {1e-6 * total / steps:.3f} ms/step, {n / steps:.0f} launches/step")
