"""Merge rocprofv3 --pmc passes of one command into a per-kernel table.

    python tools/pmc_report.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE GRBM_GUI_ACTIVE pass> <dir with SQ pass> [out.json]

HBM-side bytes per launch = 2 x FETCH_SIZE (gfx950 counts 64 B per 128-B request of a wide coalesced read,
MI355X_MICROARCH.md "HBM") + WRITE_SIZE (exact, also for fp32 atomics); both counters are in KB.
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (1024 x kernel cycles), kernel
cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)."""
import csv, glob, hashlib, json, os, sys, collections


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write, sq = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
rows = []
for k in fetch:
    f = fetch[k].get("FETCH_SIZE", [])
    w = write.get(k, {}).get("WRITE_SIZE", [])
    g = write.get(k, {}).get("GRBM_GUI_ACTIVE", [])
    m = sq.get(k, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", [])
    if not f or not w:
        continue
    avg = lambda v: sum(v) / len(v) if v else 0.0
    cyc = avg(g) / 8.0
    rows.append(dict(kernel=k, launches=len(f), fetch_kb_raw=avg(f), read_mb=2 * avg(f) / 1e3, write_kb=avg(w),
                     bytes_per_launch=2e3 * avg(f) + 1e3 * avg(w), total_bytes=(2e3 * avg(f) + 1e3 * avg(w)) * len(f),
                     kernel_cycles=cyc, mfma_util=(avg(m) / (1024.0 * cyc) if cyc and m else None)))
rows.sort(key=lambda r: -r["kernel_cycles"] * r["launches"])
print(f"{'kernel':88s} {'launches':>8s} {'FETCH KB raw':>13s} {'read MB (x2)':>13s} {'WRITE KB':>10s} {'MFMA busy':>10s}")
for r in rows[:28]:
    mu = f"{100 * r['mfma_util']:9.1f}%" if r["mfma_util"] is not None else "       n/a"
    print(f"{r['kernel'][:88]:88s} {r['launches']:8d} {r['fetch_kb_raw']:13.1f} {r['read_mb']:13.1f} {r['write_kb']:10.1f} {mu}")
if len(sys.argv) > 4:
    dom = next((r for r in rows if "gemm_tn_pp_kernel" in r["kernel"]), None) or next(r for r in rows if "gemm_tn_kernel" in r["kernel"])
    json.dump({"kernel": "gemm_tn_pp", "source": sys.argv[4 + 1] if len(sys.argv) > 5 else "",
               "fetch_kb_raw_per_launch": dom["fetch_kb_raw"], "write_kb_per_launch": dom["write_kb"],
               "bytes_per_launch_corrected": dom["bytes_per_launch"], "mfma_busy_fraction": dom["mfma_util"],
               "launches_profiled": dom["launches"],
               "commit": os.environ.get("UNIMM_COMMIT") or None,      # the commit the passes ran at (tools/profile_round.sh <tag> <commit>)
               # bench.py nulls `roofline.traffic` when the kernel source no longer hashes to this
               "gemm_hip_sha256": hashlib.sha256(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "unimm_amd",
                                                                   "csrc", "gemm.hip"), "rb").read()).hexdigest()},
              open(sys.argv[4], "w"), indent=1)
