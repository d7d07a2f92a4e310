"""Print a per-step kernel time table from a rocprofv3 --stats kernel_stats.csv."""
import csv, glob, sys
path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
f = path if path.endswith(".csv") else glob.glob(path + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time: {tot / steps / 1e6:.2f} ms/step over {steps:g} steps")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 34]:
    n = r["Name"].replace("(anonymous namespace)::", "")[:100]
    print(f"{float(r['TotalDurationNs']) / steps / 1e6:8.3f} ms {int(r['Calls']) / steps:7.1f}x {float(r['AverageNs']) / 1e3:9.1f} us  {n}")
