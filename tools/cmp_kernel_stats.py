"""Side-by-side of two rocprofv3 kernel_stats CSVs (e.g. tools/prof_step_ab.sh: LayerNorm folded / stand-alone): calls, total ms, average us."""
import csv
import re
import sys


def load(p):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3) for r in csv.DictReader(open(p))}


def short(n):
    m = re.search(r'gemm_tn_ring_kernelIDF16bLi(\d+)ELi256ELi2ELi4ELi64ELi(\d)ELi(n?\d+)E', n)
    return f"ring<{m.group(1)},{m.group(2)},mask {m.group(3)}>" if m else n[:60]


a, b = load(sys.argv[1]), load(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
print("total ms", round(sum(v[1] for v in a.values()), 2), round(sum(v[1] for v in b.values()), 2))
for n in sorted(set(a) | set(b), key=lambda n: -(a.get(n, (0, 0, 0))[1] + b.get(n, (0, 0, 0))[1]))[:top]:
    fa, fb = a.get(n, (0, 0, 0)), b.get(n, (0, 0, 0))
    print(f"{short(n):62s} A {fa[0]:4d} {fa[1]:7.2f} ms {fa[2]:7.1f} us   B {fb[0]:4d} {fb[1]:7.2f} ms {fb[2]:7.1f} us")
