"""From a rocprofv3 kernel trace csv: time from the end of one step's last backward kernel (embed_bwd) to the first
text-layer GEMM of the next step, and the kernels in between."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "embed_fwd_kernel" in r["Kernel_Name"]]
gaps = []
for i in marks[2:]:
    j = i
    while j > 0 and "embed_bwd_kernel" not in rows[j]["Kernel_Name"]:
        j -= 1
    k = i
    while "gemm_nt_kernel" not in rows[k]["Kernel_Name"] or rows[k]["Queue_Id"] != rows[i]["Queue_Id"]:
        k += 1
    if j > 0:
        gaps.append(((int(rows[k]["Start_Timestamp"]) - int(rows[j]["End_Timestamp"])) / 1e3, k - j - 1))
print("embed_bwd end -> first text GEMM: us, launches in between:", [(round(g, 1), n) for g, n in gaps])
