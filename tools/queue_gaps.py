"""From a rocprofv3 kernel trace csv: busy time per queue and the largest idle gaps of the main queue in one step."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "embed_fwd_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sub = rows[marks[k]:marks[k + 1]]
t0, t1 = int(sub[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sub)
print(f"step wall {1e-6 * (t1 - t0):.2f} ms, {len(sub)} launches")
byq = collections.defaultdict(list)
for r in sub:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")))


def union(iv):
    iv = sorted(iv)
    tot, (cs, ce) = 0, iv[0][:2]
    for s, e, *_ in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


for q, iv in byq.items():
    print(f"queue {q}: {len(iv)} launches, busy {1e-6 * union(iv):.2f} ms")
mainq = max(byq, key=lambda q: len(byq[q]))
iv = sorted(byq[mainq])
gaps = [(iv[i + 1][0] - iv[i][1], iv[i][2][:48], iv[i + 1][2][:48]) for i in range(len(iv) - 1)]
print(f"main queue idle {1e-6 * sum(g[0] for g in gaps if g[0] > 0):.2f} ms; largest gaps:")
for g in sorted(gaps, reverse=True)[:10]:
    print(f"  {g[0] / 1e3:7.1f} us after {g[1]} | before {g[2]}")
