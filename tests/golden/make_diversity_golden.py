"""Builds tests/golden/diversity/alpha_cases.json from data files the reference committed
(results/2023-11-29/<run>/tsv_<type>/<sample>_counts.tsv and <run>/report/diversity-<type>.tsv, or the
single-sample report/diversity/<type>-<sample>.tsv): for every (run, type, sample) the multiset of
counts of the sample's table (as [count, how many rows] pairs) and the nine metric strings the
reference printed for it (lib/mercat2_diversity.py:13-53, scikit-bio's skbio.diversity.alpha).
Run in the build container only (needs /root/reference); the JSON is the fixture."""
import collections
import glob
import json
import os
import sys

ROOT = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/results/2023-11-29"
TYPES = {"protein": "tsv_protein", "Nucleotide": "tsv_nucleotide", "prod": "tsv_prod", "fgs": "tsv_fgs"}
cases = {}
for run in sorted(os.listdir(ROOT)):
    for kind, folder in TYPES.items():
        table = os.path.join(ROOT, run, "report", "diversity-%s.tsv" % kind)
        if not os.path.exists(table):
            continue
        lines = [l.rstrip("\n").split("\t") for l in open(table)]
        for col, sample in enumerate(lines[0][1:], 1):
            tsv = os.path.join(ROOT, run, folder, "%s_counts.tsv" % sample)
            if not os.path.exists(tsv):
                continue
            with open(tsv) as fh:
                fh.readline()
                counts = collections.Counter(int(l.split()[1]) for l in fh)
            cases["%s/%s/%s" % (run, kind, sample)] = {
                "counts": sorted(counts.items()),
                "expected": {row[0]: row[col] for row in lines[1:]},
            }
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diversity", "alpha_cases.json")
json.dump(cases, open(out, "w"), separators=(",", ":"))
print(len(cases), "cases ->", out, os.path.getsize(out), "bytes")
