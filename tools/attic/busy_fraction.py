"""GPU busy fraction of the timed steps from a rocprofv3 kernel trace (union of the kernels' [start, end) intervals over all streams ÷ wall time), per step window.

    python tools/busy_fraction.py gpurun_out/prof_overlap/step_kernel_trace.csv [adamw]

The last argument is a substring of the kernel that ends a step (default: adamw)."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marker = sys.argv[2] if len(sys.argv) > 2 else "adamw"
ends = [e for s, e, n in rows if marker in n.lower()]
print(f"{len(rows)} launches, {len(ends)} step ends")
for a, b in zip(ends[:-1], ends[1:]):
    win = [(s, e) for s, e, n in rows if s >= a and e <= b]
    busy, cur_s, cur_e = 0, None, None
    for s, e in win:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    gaps = sorted(((s2 - e1) for (s1, e1), (s2, e2) in zip(win[:-1], win[1:]) if s2 > e1), reverse=True)
    print(f"step {1e-6 * (b - a):7.3f} ms: {len(win)} launches, busy {busy / (b - a):.4f}, sum of kernel time {1e-6 * sum(e - s for s, e in win):7.3f} ms")
