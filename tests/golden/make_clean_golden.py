#!/usr/bin/env python3
"""Golden vectors of removeN (lib/mercat2_fasta.py:53-119) made by running the REFERENCE's own function.

Run only in the build container (/root/reference mounted):  python tests/golden/make_clean_golden.py

lib/mercat2_fasta.py cannot be imported as a module here (it imports pyrodigal and pkg_resources data that
are not installed), so the two functions on the path -- ``split_sequenceN`` and ``removeN`` -- are taken out of
its syntax tree and executed with the standard-library modules they use.  Nothing of the reference's source is
stored: only inputs written by this script and the outputs' digests / small outputs in full.

Outputs
  inputs/edge_clean.fa, inputs/edge_clean_odd.fa   synthetic inputs (N at the ends, empty pieces, lower-case n,
                                                   CRLF, blank lines, text before the first header, headers with
                                                   several blanks; the second one also blanks / hyphens / tabs
                                                   INSIDE sequences that get split: textwrap's word breaks)
  clean_cases.json   {case: {input, toupper, out_name, bytes, lines, sha256, gc, gz_size}} for the five genomes,
                     Scaffolds_with-NNN and the synthetic inputs
  clean/*.fna        cleaned text in full for the small cases
The reference's committed results/2023-11-29/fna-5genomes_gz-10/clean/*_clean.fna.gz are checked to hold exactly
what removeN produces today (asserted below).
"""
import ast
import gzip
import hashlib
import json
import os
import re
import shutil
import sys
import tempfile
import textwrap
from pathlib import Path

sys.dont_write_bytecode = True
REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent
INPUTS = HERE / "inputs"


def load_functions():
    tree = ast.parse((REF / "lib/mercat2_fasta.py").read_text())
    ns = {"os": os, "gzip": gzip, "re": re, "textwrap": textwrap, "Path": Path}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("split_sequenceN", "removeN"):
            exec(compile(ast.Module([node], []), "mercat2_fasta.%s" % node.name, "exec"), ns)
    return ns["removeN"]


def edge_inputs():
    plain = (
        "text in front of the first header is dropped\nACGTNNACGT\n"
        ">s1 plain record without the letter\nACGTACGTAC\nGTACGT\n\nAC\n"
        ">s2   several   blanks\tand a tab  \nACGTNNNNACGTTTGACCA\nNNAC\n"
        ">s3 leading and trailing runs\nNNNNACGTACGTNNNN\n"
        ">s4 only N\nNNNNNNNN\n"
        ">s5 lower n is not a cut\nacgtnnnnacgtNacgt\n"
        ">s6 crlf\r\nACGTNNAC\r\nGGTT\r\n"
        ">s7 long\n" + "ACGT" * 61 + "N" + "TTGCA" * 40 + "\n"
        ">s8 exactly 80 and 160\n" + "A" * 80 + "N" + "C" * 160 + "N" + "G" * 81 + "\n"
        ">s12 a piece that starts with the header mark: printed as it stands even with -toupper\nacgtNN>tail\nacgtNNNac>gt\n"
        ">s9\nN\n>s10 empty record\n>s11 no newline at the end\nACGNNNT"
    )
    odd = (
        ">o1 blank inside a split sequence\nACGT ACGTNNAC GT\n"
        ">o2 hyphen\nACGT-ACGT-NNAC-GT--TTA\n"
        ">o3 tab\nAC\tGTNNACGT\n"
        ">o4 long words with blanks\n" + ("ACGTTGCA " * 30) + "NN" + ("GG-CC" * 40) + "\n"
        ">o5 no N: blanks stay\nAC GT AC-GT\n"
    )
    return {"edge_clean.fa": plain.encode(), "edge_clean_odd.fa": odd.encode()}


def main():
    if not REF.is_dir():
        sys.exit("needs /root/reference")
    removeN = load_functions()
    for name, data in edge_inputs().items():
        (INPUTS / name).write_bytes(data)
        os.chmod(INPUTS / name, 0o644)
    full = HERE / "clean"
    shutil.rmtree(full, ignore_errors=True)
    full.mkdir()
    cases = {}
    jobs = [(g + ".fna.gz", False) for g in ("DJ", "GIC31", "RW1", "RW2", "Rleg")]
    jobs += [("Scaffolds_with-NNN.fna.gz", False), ("Scaffolds_with-NNN.fna.gz", True),
             ("edge_clean.fa", False), ("edge_clean.fa", True), ("edge_clean_odd.fa", False), ("edge_clean_odd.fa", True)]
    for fname, toupper in jobs:
        with tempfile.TemporaryDirectory() as tmp:
            out, stats = removeN(INPUTS / fname, tmp, toupper)
            text = gzip.open(out, "rb").read()
            case = "%s|%s" % (fname, "upper" if toupper else "asis")
            cases[case] = {"input": fname, "toupper": toupper, "out_name": Path(out).name, "bytes": len(text),
                           "lines": text.count(b"\n"), "sha256": hashlib.sha256(text).hexdigest(),
                           "gc": stats["GC Content"], "gz_size": os.stat(out).st_size}
            if len(text) < 100_000:
                (full / ("%s_%s.fna" % (Path(out).name[:-len("_clean.fna.gz")], "upper" if toupper else "asis"))).write_bytes(text)
            if fname in ("DJ.fna.gz", "GIC31.fna.gz", "RW1.fna.gz", "RW2.fna.gz", "Rleg.fna.gz"):
                committed = gzip.open(REF / "results/2023-11-29/fna-5genomes_gz-10/clean" / Path(out).name, "rb").read()
                assert committed == text, "the committed clean file of %s differs from today's removeN" % fname
    (HERE / "clean_cases.json").write_text(json.dumps(cases, indent=1, sort_keys=True))
    print("clean cases:", len(cases))


if __name__ == "__main__":
    main()
