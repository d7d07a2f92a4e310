# kernel trace + SQ counters of the fp32 matrix attention kernels on one shape: bash tools/exp/x3_attn_pmc.sh <out-tag> [case substring]
tag=$1; only=${2:-text self}
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o run -- python tools/exp/x3_attn_shapes.py --only "$only" --impls 1 --iters 10 > $out/kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc1 -o run -- python tools/exp/x3_attn_shapes.py --only "$only" --impls 1 --iters 4 > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $out/pmc2 -o run -- python tools/exp/x3_attn_shapes.py --only "$only" --impls 1 --iters 4 > $out/pmc2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("kt",):
    for f in glob.glob("$out/%s/**/*kernel_stats.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "x3m" in r["Name"]: print(r["Name"][:70], r["Calls"], r["AverageNs"])
for d in ("pmc1", "pmc2"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "x3m" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, cs in acc.items():
            print(k[:70]); print("   " + "  ".join(f"{c}={v / n[(k, c)]:.4g}" for c, v in cs.items()))
PY
