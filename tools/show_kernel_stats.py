"""Top rows of a rocprofv3 kernel_stats.csv: python tools/show_kernel_stats.py file.csv [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print("total ms", round(sum(float(r["TotalDurationNs"]) for r in rows) / 1e6, 2))
for r in rows[:n]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>5s} {float(r['AverageNs']) / 1e3:8.1f}us {float(r['TotalDurationNs']) / 1e6:8.2f}ms {float(r['Percentage']):5.1f}%")
