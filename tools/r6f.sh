out=gpurun_out/r6f; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/blas -o run -- python tools/hipblaslt_names.py > $out/names.log 2>&1
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/r6f/blas/**/*kernel_stats.csv',recursive=True)
for r in csv.DictReader(open(f[0])):
    print(r['Name'][:230], r['Calls'], r['AverageNs'])
P
