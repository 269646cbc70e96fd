#!/usr/bin/env python3
"""Where is the GPU idle or thinly used inside ONE steady-state step of a multi-stream kernel trace (rocprofv3 …_kernel_trace.csv)?
Prints: step wall time, time with no kernel running, time with exactly one kernel running (by kernel), and the longest gaps with their neighbours."""
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw_guarded" in r["Kernel_Name"]]
w = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = marks[-w], marks[-w + 1]
win = rows[a + 1:b + 1]
t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["End_Timestamp"])
ev = []
for r in win:
    s, e = max(int(r["Start_Timestamp"]), t0), min(int(r["End_Timestamp"]), t1)
    if e > s:
        ev.append((s, 1, r["Kernel_Name"])); ev.append((e, -1, r["Kernel_Name"]))
ev.sort()
active = collections.Counter(); depth_time = collections.Counter(); solo = collections.Counter(); gaps = []
prev = t0; last_end_name = rows[a]["Kernel_Name"]
for t, d, n in ev:
    k = sum(active.values())
    depth_time[min(k, 4)] += t - prev
    if k == 0 and t - prev > 0:
        gaps.append((t - prev, last_end_name, n))
    if k == 1:
        solo[next(iter(x for x, c in active.items() if c > 0))[:70]] += t - prev
    prev = t
    active[n] += d
    if d < 0: last_end_name = n
print(f"step wall {1e-6 * (t1 - t0):.3f} ms; kernels {len(win)}; sum of durations {1e-6 * sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in win):.3f} ms")
for k in sorted(depth_time): print(f"  {k}{'+' if k == 4 else ''} kernels running: {1e-6 * depth_time[k]:.3f} ms")
print("longest idle gaps (us, after -> before):")
for g, p, n in sorted(gaps, reverse=True)[:12]: print(f"  {g / 1e3:7.1f}  {p[:60]} -> {n[:60]}")
print("time alone on the chip, by kernel (ms):")
for n, t in solo.most_common(14): print(f"  {1e-6 * t:6.3f}  {n}")
