cd /tmp && export TMPDIR=/tmp
for x in 1 0; do
  export MK_XSEG=$x
  timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ws_x$x -- python3 $GRAFT_REPO_ROOT/tools/parse_probe.py 3 > $GRAFT_REPO_ROOT/gpurun_out/ws_x$x.log 2>&1
  python3 - <<E
import csv,glob,collections
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/ws_x$x/**/*counter_collection.csv", recursive=True)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"]=="WRITE_SIZE": agg[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "scatterq" in k or "count_k" in k: print("xseg=$x", k, len(v), round(sum(v)/len(v)), [round(q) for q in v[:8]])
E
done
