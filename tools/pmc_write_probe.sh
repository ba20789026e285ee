# GPU box: WRITE_SIZE per kernel of a probe binary:  bash tools/pmc_write_probe.sh build/<probe> <tag>
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wp_$2 -- $GRAFT_REPO_ROOT/$1 > $GRAFT_REPO_ROOT/gpurun_out/wp_$2.log 2>&1
python3 - <<E
import csv,glob,collections
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/wp_$2/**/*counter_collection.csv", recursive=True)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"]=="WRITE_SIZE": agg[r["Kernel_Name"][:44]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print("$2", k, len(v), round(sum(v)/len(v)))
E
