#!/usr/bin/env python3
"""Golden vectors of merge_tsv / merge_tsv_T (lib/mercat2_report.py:98-156, 160-194) made by running the
REFERENCE's own functions.  Run only in the build container:  python tests/golden/make_report_golden.py

lib/mercat2_report.py imports dominate and plotly (not installed here); the two functions on the path use only
``os`` and ``resource``, so they are taken out of the module's syntax tree and executed on their own.  Inputs are
count tables already among the fixtures (tests/golden/tsv) plus two written here.  Stored: the merged table as the
reference wrote it, and for the transposed table -- whose column order is the iteration order of a Python set,
different in every process -- its header as a sorted list and every sample's row re-ordered to that list.
"""
import ast
import json
import os
import resource
import sys
import tempfile
from pathlib import Path

sys.dont_write_bytecode = True
REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent


def load_functions():
    tree = ast.parse((REF / "lib/mercat2_report.py").read_text())
    ns = {"os": os, "resource": resource}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("merge_tsv", "merge_tsv_T"):
            exec(compile(ast.Module([node], []), "mercat2_report.%s" % node.name, "exec"), ns)
    return ns["merge_tsv"], ns["merge_tsv_T"]


def main():
    if not REF.is_dir():
        sys.exit("needs /root/reference")
    merge_tsv, merge_tsv_T = load_functions()
    out = HERE / "report"
    out.mkdir(exist_ok=True)
    # two small tables written here (disjoint / overlapping keys, a key only in the last sample) + two fixtures
    (out / "in_b.tsv").write_text("k-mer\tb_Count\nAAAAC\t12\nACGTA\t7\nTTTTT\t30\n")
    (out / "in_A.tsv").write_text("k-mer\tA_Count\nAAAAA\t5\nACGTA\t11\nCCCCC\t2\nTTTTG\t9\n")
    cases = {
        "small": {"b": "report/in_b.tsv", "A": "report/in_A.tsv"},
        "k5": {"RW1_clean": "tsv/ref_RW1_clean_k5_c10.tsv", "Test_R1": "tsv/ref_Test_R1_k5_c10.tsv",
               "Scaffolds": "tsv/Scaffolds_with-NNN_k5_c10.tsv", "b": "report/in_b.tsv"},
    }
    index = {}
    for case, files in cases.items():
        tsv_list = {name: str(HERE / rel) for name, rel in files.items()}
        with tempfile.TemporaryDirectory() as tmp:
            merge_tsv(tsv_list, os.path.join(tmp, "m.tsv"))
            merged = Path(tmp, "m.tsv").read_text()
            merge_tsv_T(tsv_list, os.path.join(tmp, "t.tsv"))
            lines = Path(tmp, "t.tsv").read_text().split("\n")
        (out / ("%s_merged.tsv" % case)).write_text(merged)
        head = lines[0].split("\t")
        assert head[0] == "sample" and lines[-1] == ""
        cols = head[1:]
        order = sorted(range(len(cols)), key=lambda i: cols[i])
        rows = {}
        for ln in lines[1:-1]:
            cells = ln.split("\t")
            rows[cells[0]] = [cells[1 + i] for i in order]
        index[case] = {"inputs": files, "columns_sorted": [cols[i] for i in order], "rows": rows,
                       "sample_order": [ln.split("\t")[0] for ln in lines[1:-1]]}
    (out / "transposed.json").write_text(json.dumps(index, indent=0, sort_keys=True))
    print("report cases:", len(index))


if __name__ == "__main__":
    main()
