#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]   (run on the GPU box from the repo root)
# rocprofv3 kernel-trace of bench.py; prints a trimmed per-kernel table, keeps it in gpurun_out/<tag>.stats.csv
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/bench.py "$@" > $root/gpurun_out/prof_$tag.log 2>&1
echo "[$tag] rc=$? $(grep -o '"value": [0-9.e+]*' $root/gpurun_out/prof_$tag.log) $(grep -o '"ms_per_step": [0-9.]*' $root/gpurun_out/prof_$tag.log)"
python3 - "$out" "$root/gpurun_out/$tag.stats.csv" <<'PY'
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(mk_\w+|radix_sort_onesweep_\w+|__amd_rocclr_\w+)", r["Name"])
        rows.append((m.group(1) if m else r["Name"][:40], int(r["Calls"]), int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
with open(sys.argv[2], "w") as w:
    w.write("name,calls,total_ms,avg_us,pct\n")
    for r in rows:
        w.write("%s,%d,%.3f,%.1f,%s\n" % r)
for r in rows[:9]:
    print("   %-22s calls=%-4d avg_us=%-9.1f pct=%s" % (r[0], r[1], r[3], r[4]))
PY
