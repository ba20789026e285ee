"""gpurun_out/<tag>_{A,B,C}/**/counter_collection.csv (tools/pmc_sq.sh) -> one table: per kernel, mean per launch."""
import collections, csv, glob, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "sq"
out = sys.argv[2] if len(sys.argv) > 2 else None
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for p in "ABC":
    for f in glob.glob("gpurun_out/%s_%s/*/*counter_collection.csv" % (tag, p)):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(mk_\w+)", r["Kernel_Name"])
            if not m:
                continue
            a = agg[m.group(1)][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
names = sorted({c for k in agg.values() for c in k})
lines = ["kernel,launches," + ",".join(names)]
for k, cs in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", [0, 0])[1]):
    n = max(v[0] for v in cs.values())
    lines.append("%s,%d," % (k, n) + ",".join("%.4g" % (cs[c][1] / cs[c][0]) if c in cs else "" for c in names))
text = "\n".join(lines) + "\n"
if out:
    open(out, "w").write("# rocprofv3 --pmc (three passes, tools/pmc_sq.sh) -- python3 tools/parse_probe.py 4: one context, S2-shaped 100 MiB chunks, k=31, -c 10; mean per launch\n" + text)
print(text)
