# kernel stats of the dense fine-tune micro-step in fp32x3: bash tools/exp/x3_step_profile.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python bench.py --workload dense --compute fp32x3 --steps 8 --warmup 2 --no-cpu-baseline --no-padded > $out/bench.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/run_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
steps=max([int(r["Calls"]) for r in rows if "neural_ndcg" in r["Name"]] or [1])      # one launch per step
print("%d steps in the trace; summed kernel time (both queues) %.2f ms per step" % (steps, tot/1e6/steps))
print("%-96s %9s %9s %9s" % ("kernel", "launches", "ms/step", "us avg"))
for r in rows[:26]:
    print("%-96s %9.1f %9.2f %9.1f" % (r["Name"][:96], int(r["Calls"])/steps, float(r["TotalDurationNs"])/1e6/steps, float(r["AverageNs"])/1e3))
PY
tail -1 $out/bench.log | cut -c1-200
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$out/run_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last full step: take the last 40 % of the trace window and report per-queue busy time / span
t0=int(rows[0]["Start_Timestamp"]); t1=max(int(r["End_Timestamp"]) for r in rows)
lo=t0+int((t1-t0)*0.5); hi=t0+int((t1-t0)*0.9)
q=collections.defaultdict(lambda:[0,0])
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if s>=lo and e<=hi:
        q[r["Queue_Id"]][0]+=e-s; q[r["Queue_Id"]][1]+=1
for k,(busy,n) in q.items():
    print("queue %s: busy %.1f %% of the window, %d launches (%.1f us avg)" % (k, 100.0*busy/(hi-lo), n, busy/max(n,1)/1e3))
PY
