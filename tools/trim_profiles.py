"""Turn the rocprofv3 outputs of tools/make_profiles.sh (gpurun_out/) into the small committed
summaries under profiles/."""
import collections, csv, glob, json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
rnd = sys.argv[2] if len(sys.argv) > 2 else "round1"

def short(name):
    m = re.search(r"(mk_\w+|radix_sort_onesweep_\w+|__amd_rocclr_\w+)", name)
    return m.group(1) if m else name[:40]

for ctx in (2, 1):
    files = sorted(glob.glob("gpurun_out/prof_%s_ctx%d/*/*kernel_stats.csv" % (tag, ctx)), key=os.path.getmtime)
    if not files:
        continue
    rows = list(csv.DictReader(open(files[-1])))  # (the newest run)
    with open("profiles/%s_kernel_stats_ctx%d.csv" % (rnd, ctx), "w") as w:
        w.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-file-leg --contexts %d\n" % ctx)
        w.write("# S2 workload (10M x 150bp, k=31, -c 10, 16 chunks); 4 passes incl. warmup; kernel names trimmed\n")
        w.write("name,calls,total_ms,avg_us,pct\n")
        for r in rows:
            w.write("%s,%s,%.3f,%.1f,%s\n" % (short(r["Name"]), r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    line = [l for l in open("gpurun_out/prof_%s_ctx%d.log" % (tag, ctx)) if l.startswith('{"metric"')]
    if line:
        open("profiles/%s_bench_under_rocprof_ctx%d.json" % (rnd, ctx), "w").write(line[-1])

files = sorted(glob.glob("gpurun_out/prof_%s_k63/*/*kernel_stats.csv" % tag), key=os.path.getmtime)
if files:
    rows = list(csv.DictReader(open(files[-1])))
    with open("profiles/%s_kernel_stats_k63.csv" % rnd, "w") as w:
        w.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-file-leg --k 63 --reads 50000000 --genome 50000000 --genome-seed 6 --read-seed 7\n")
        w.write("# S3 workload (50M x 150bp from a 50 Mbp genome, k=63, -c 10, 78 chunks; two-word keys); kernel names trimmed\n")
        w.write("name,calls,total_ms,avg_us,pct\n")
        for r in rows:
            w.write("%s,%s,%.3f,%.1f,%s\n" % (short(r["Name"]), r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    line = [l for l in open("gpurun_out/prof_%s_k63.log" % tag) if l.startswith('{"metric"')]
    if line:
        open("profiles/%s_bench_under_rocprof_k63.json" % rnd, "w").write(line[-1])

def pmc_table(prefix, out_stem, command, workload):
    pmc = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = sorted(glob.glob("gpurun_out/%s_%s/*/*counter_collection.csv" % (prefix, c)), key=os.path.getmtime)
        if not files:
            continue
        agg = collections.defaultdict(lambda: [0, 0.0, 0])
        for r in csv.DictReader(open(files[-1])):
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        pmc[c] = agg
    if "FETCH_SIZE" not in pmc:
        return
    out = {}
    with open("profiles/%s.csv" % out_stem, "w") as w:
        w.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) -- %s\n" % command)
        w.write("# %s\n" % workload)
        w.write("# counters in KB per launch. hbm_bytes_per_launch applies the gfx950 correction of MI355X_MICROARCH.md (HBM section):\n")
        w.write("#   FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> doubled; WRITE_SIZE as read.\n")
        w.write("name,launches,fetch_kb,write_kb,hbm_bytes_per_launch,avg_us\n")
        for n, (calls, v, t) in sorted(pmc["FETCH_SIZE"].items(), key=lambda x: -x[1][2]):
            wv = pmc.get("WRITE_SIZE", {}).get(n, [1, 0.0, 0])
            f_kb, w_kb = v / calls, wv[1] / max(1, wv[0])
            hbm = (2 * f_kb + w_kb) * 1024
            out[n] = {"fetch_kb": f_kb, "write_kb": w_kb, "hbm_bytes_per_launch": hbm, "launches": calls}
            w.write("%s,%d,%.0f,%.0f,%.0f,%.1f\n" % (n, calls, f_kb, w_kb, hbm, t / calls / 1e3))
    json.dump(out, open("profiles/%s.json" % out_stem, "w"), indent=1)


pmc_table("pmc_%s_k63" % tag, "%s_pmc_traffic_k63" % rnd,
          "python3 bench.py --steps 1 --warmup 0 --no-cpu --no-file-leg --k 63 --reads 50000000 --genome 50000000 --genome-seed 6 --read-seed 7",
          "S3 workload (50M x 150bp from a 50 Mbp genome, k=63, -c 10, 78 chunks; two-word keys)")

pmc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = sorted(glob.glob("gpurun_out/pmc_%s_%s/*/*counter_collection.csv" % (tag, c)), key=os.path.getmtime)
    if not files:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0, 0])
    for r in csv.DictReader(open(files[-1])):
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    pmc[c] = agg
if pmc:
    out = {}
    with open("profiles/%s_pmc_traffic.csv" % rnd, "w") as w:
        w.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-file-leg --contexts 1\n")
        w.write("# counters in KB per launch. hbm_bytes_per_launch applies the gfx950 correction of MI355X_MICROARCH.md (HBM section):\n")
        w.write("#   FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> doubled; WRITE_SIZE as read.\n")
        w.write("name,launches,fetch_kb,write_kb,hbm_bytes_per_launch,avg_us\n")
        for n, (calls, v, t) in sorted(pmc["FETCH_SIZE"].items(), key=lambda x: -x[1][2]):
            wv = pmc.get("WRITE_SIZE", {}).get(n, [1, 0.0, 0])
            f_kb, w_kb = v / calls, wv[1] / max(1, wv[0])
            hbm = (2 * f_kb + w_kb) * 1024
            out[n] = {"fetch_kb": f_kb, "write_kb": w_kb, "hbm_bytes_per_launch": hbm, "launches": calls}
            w.write("%s,%d,%.0f,%.0f,%.0f,%.1f\n" % (n, calls, f_kb, w_kb, hbm, t / calls / 1e3))
    json.dump(out, open("profiles/%s_pmc_traffic.json" % rnd, "w"), indent=1)
print("ok")
