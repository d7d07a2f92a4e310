# SQ counters of the attention kernels alone: [ATTN_SHAPES=coatt] bash tools/exp/attn_text_pmc.sh <out-tag>   (text self-attention, or the two co-attention directions)
tag=$1
script=tools/exp/attn_${ATTN_SHAPES:-text}_shapes.py
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python $script > $out/time.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc1 -o run -- python3 $script --iters 4 > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $out/pmc2 -o run -- python3 $script --iters 4 > $out/pmc2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out/pmc3 -o run -- python3 $script --iters 4 > $out/pmc3.log 2>&1
python - <<PY > $out/summary.txt 2>&1
import csv, glob, collections
print(open("$out/time.txt").read())
for d in ("pmc1", "pmc2", "pmc3"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "attn_" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, cs in acc.items():
            print(d, k[:60]); print("   " + "  ".join(f"{c}={v / n[(k, c)]:.4g}" for c, v in cs.items()))
PY
cat $out/summary.txt
