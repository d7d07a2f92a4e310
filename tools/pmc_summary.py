"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, average counter value per dispatch."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            if "gemm" not in name and "attn" not in name:
                continue
            acc[name[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(d.split("/")[-1], "|", k)
            for c, v in cs.items():
                print(f"    {c:28s} avg {sum(v) / len(v):16.1f}  (n={len(v)})")
