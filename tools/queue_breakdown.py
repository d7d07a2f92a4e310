"""From a rocprofv3 kernel trace csv: per queue, the time of one step by kernel name (sum of durations and launches)."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "embed_fwd_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sub = rows[marks[k]:marks[k + 1]]
t0, t1 = int(sub[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sub)
print(f"step wall {1e-6 * (t1 - t0):.2f} ms, {len(sub)} launches")
byq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
for r in sub:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    e = byq[r["Queue_Id"]][n]
    e[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    e[1] += 1
for q, d in byq.items():
    tot = sum(v[0] for v in d.values())
    print(f"--- queue {q}: {1e-6 * tot:.2f} ms summed")
    for n, (t, c) in sorted(d.items(), key=lambda kv: -kv[1][0])[:28]:
        print(f"  {1e-6 * t:7.3f} ms  {c:4d} x {1e-3 * t / c:8.1f} us  {n[:90]}")
