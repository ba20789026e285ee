#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run only in the build container, where /root/reference is mounted:

    python tests/golden/make_golden.py

It imports the reference's two stdlib-only hot-path modules *by file path*
(lib/mercat2_kmers.py, lib/mercat2_Chunker.py), runs them over (a) small data files the
reference ships and its own tests use and (b) edge-case inputs written by this script,
and records inputs + expected outputs as data.  Nothing of the reference's source text is
stored.  The GPU box has no /root/reference: tests read only what this script committed.

Outputs
  inputs/*                 input files (copies of reference data files; synthetic edge cases)
  expected.json            {case: {input,k,c,rows,sum,sha256}} -- sha256 of the TSV text
                           ("k-mer\\t{base}_Count\\n" + sorted "kmer\\tcount\\n" rows)
  tsv/*                    full TSV text of a few small cases and the reference's own
                           committed count tables (results/2023-11-29/...)
  chunks.json              reference Chunker cut points (byte offsets) + per-chunk sha256
  expected_big.json        the other four genomes / three proteomes of data/5-genomes-* (BASELINE
                           configs 1 and 4 name all five): per case rows, sum, sha256 of the TSV text
                           and sha256 of the raw key bytes / little-endian u64 counts (cheap to check
                           from arrays at millions of rows)
  clean.json               sha256 / size of the decompressed text of the reference's committed
                           clean/*_clean.fna.gz (its removeN outputs) for the five genomes
"""
import gzip
import hashlib
import importlib.util
import json
import os
import random
import shutil
import sys
import tempfile
from pathlib import Path

sys.dont_write_bytecode = True
REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent
INPUTS = HERE / "inputs"
TSV = HERE / "tsv"


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


kmers = load("ref_kmers", "lib/mercat2_kmers.py")
chunker = load("ref_chunker", "lib/mercat2_Chunker.py")


def tsv_text(base, table):
    rows = ["k-mer\t%s_Count\n" % base]
    for key, n in sorted(table.items()):
        rows.append("%s\t%d\n" % (key, n))
    return "".join(rows)


def digest(base, table):
    text = tsv_text(base, table)
    return {"rows": len(table), "sum": int(sum(table.values())),
            "sha256": hashlib.sha256(text.encode()).hexdigest()}


def basename_of(name):
    for ext in (".fasta.gz", ".fa.gz", ".fna.gz", ".ffn.gz", ".faa.gz",
                ".fasta", ".fa", ".fna", ".ffn", ".faa"):
        if name.endswith(ext):
            return name[: -len(ext)]
    return name


# ----------------------------------------------------------------- synthetic edge inputs
def rnd_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def wrap(seq, width):
    return "\n".join(seq[i:i + width] for i in range(0, len(seq), width))


def edge_inputs():
    rng = random.Random(20261003)
    files = {}
    # whitespace / newline / '*' / header quirks (lib/mercat2_kmers.py:50-63)
    ws = (
        "ACGTTGCAACGT pre-header text counts as a record\n"
        ">r1 plain\nACGTACGTACGTAAACCCGGGTTT\nACGT\n"
        ">r2 crlf\r\nACGTACGT\r\nTTTTACGT\r\n"
        ">r3 lone cr\rGGGGACGTAC\rGTACGTAC\r"
        ">r4 stars\nACG*TAC**GTACGT*\n*\n"
        "  >r5 header with leading blanks\n  ACGTAC  \n\tGTACGT\t\n"
        ">r6 inner blank\nACG TAC\tGTA  CGT\n"
        ">r7 gt inside line\nAC>GTACGT\nACGT>\n"
        ">r8 lower and N\nacgtNNNNacgtACGTNACGTnACGT\n"
        ">r9 short\nAC\n>r10 empty\n>r11 empty too\n\n\n"
        ">r12 odd blanks\n\x0bACGTACGTAC\x0c\n\x1cACGTAC\x1f\n AC\x1dGT \n"
        "*>r13 not a header: first char is a star\nACGTACGT\n"
        ">r14 no trailing newline\nACGTACGTACGTACGTAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA"
    )
    files["edge_ws.fa"] = ws.encode()
    files["edge_empty.fa"] = b""
    files["edge_hdr_only.fa"] = b">a\n>b\n>c"
    files["edge_nohdr.fa"] = (wrap(rnd_seq(rng, 700), 70) + "\n").encode()
    # record lengths around k for k in 21,31,32,33,63,64 ; long single-line records
    recs = []
    for i, n in enumerate([20, 21, 22, 30, 31, 32, 33, 34, 62, 63, 64, 65, 66, 127, 128, 129, 1, 0, 5000, 12345]):
        recs.append(">len%d_%d\n%s\n" % (n, i, rnd_seq(rng, n)))
    # low-complexity records: heavy duplicates, all-A / all-T windows (k=32 / k=64 all-ones key)
    recs.append(">polyA\n" + wrap("A" * 300, 60) + "\n")
    recs.append(">polyT\n" + wrap("T" * 300, 60) + "\n")
    recs.append(">polyTT\n" + "T" * 100 + "\n")
    recs.append(">repeat\n" + wrap("ACGTTGCA" * 80, 61) + "\n")
    recs.append(">withN\n" + wrap(rnd_seq(rng, 400, "ACGTN"), 80) + "\n")
    recs.append(">iupac\n" + wrap(rnd_seq(rng, 300, "ACGTRYKMSWN"), 50) + "\n")
    files["edge_lengths.fa"] = "".join(recs).encode()
    # reads sampled from a small genome, both strands: duplicates and min_count effects
    g = rnd_seq(rng, 3000)
    comp = str.maketrans("ACGT", "TGCA")
    reads = []
    for i in range(1500):
        s = rng.randrange(0, len(g) - 100)
        r = g[s:s + 100]
        if rng.random() < 0.5:
            r = r.translate(comp)[::-1]
        reads.append(">r%d\n%s\n" % (i, r))
    files["edge_reads.fna"] = "".join(reads).encode()
    # protein-like: 20 letters + X B Z J O U, stops, lower case, a '-' gap
    aa = "ACDEFGHIKLMNPQRSTVWY"
    prot = []
    for i in range(120):
        n = rng.randrange(1, 400)
        s = rnd_seq(rng, n, aa)
        if i % 7 == 0:
            s = s[: n // 2] + "X" + s[n // 2:]
        if i % 11 == 0:
            s = s + "BZJOU"
        if i % 13 == 0:
            s = s[: n // 3] + "-" + s[n // 3:].lower()
        prot.append(">p%d # 1 # 2\n%s*\n" % (i, wrap(s, 60)))
    files["edge_protein.faa"] = "".join(prot).encode()
    return files


# ----------------------------------------------------------------------------- main
def main():
    if not REF.is_dir():
        sys.exit("needs /root/reference")
    shutil.rmtree(INPUTS, ignore_errors=True)
    shutil.rmtree(TSV, ignore_errors=True)
    INPUTS.mkdir(parents=True)
    TSV.mkdir(parents=True)

    # (a) data files of the reference (inputs of its own committed result trees)
    copies = {
        "A.fasta": "data/simka_test_data/A.fasta",
        "B.fasta": "data/simka_test_data/B.fasta",
        "C.fasta": "data/simka_test_data/C.fasta",
        "D_paired_1.fasta": "data/simka_test_data/D_paired_1.fasta",
        "D_paired_2.fasta": "data/simka_test_data/D_paired_2.fasta",
        "RW1.fna.gz": "data/5-genomes-fna_gz/RW1.fna.gz",
        "RW1_clean.fna.gz": "results/2023-11-29/fna-5genomes_gz-10/clean/RW1_clean.fna.gz",
        "RW1_pro.faa.gz": "data/5-genomes-faa_gz/RW1_pro.faa.gz",
        "RW1_fgs.faa.gz": "results/2023-11-29/fna-5genomes_gz-10/fgs/RW1.faa.gz",
        "DJ_pro.faa.gz": "data/5-genomes-faa_gz/DJ_pro.faa.gz",
        "Test_R1.fna.gz": "results/2023-11-29/test-qc_gz/clean/Test_R1.fna.gz",
        # the rest of data/5-genomes-fna_gz and data/5-genomes-faa_gz (BASELINE configs 1 and 4)
        "DJ.fna.gz": "data/5-genomes-fna_gz/DJ.fna.gz",
        "GIC31.fna.gz": "data/5-genomes-fna_gz/GIC31.fna.gz",
        "RW2.fna.gz": "data/5-genomes-fna_gz/RW2.fna.gz",
        "Rleg.fna.gz": "data/5-genomes-fna_gz/Rleg.fna.gz",
        "GIC31_pro.faa.gz": "data/5-genomes-faa_gz/GIC31_pro.faa.gz",
        "RW2_pro.faa.gz": "data/5-genomes-faa_gz/RW2_pro.faa.gz",
        "Rleg_pro.faa.gz": "data/5-genomes-faa_gz/Rleg_pro.faa.gz",
    }
    for dst, src in copies.items():
        shutil.copyfile(REF / src, INPUTS / dst)
    with open(REF / "data/Scaffolds_with-NNN.fna", "rb") as f, \
            gzip.GzipFile(INPUTS / "Scaffolds_with-NNN.fna.gz", "wb", mtime=0) as g:
        g.write(f.read())
    # (b) synthetic edge cases
    for name, data in edge_inputs().items():
        (INPUTS / name).write_bytes(data)
    for p in INPUTS.iterdir():
        os.chmod(p, 0o644)

    # the reference's own committed count tables (k=5, c=10; results/run-tests.sh:14-28)
    ref_tables = {
        "ref_RW1_clean_k5_c10.tsv": ("results/2023-11-29/fna-5genomes_gz-10/tsv_nucleotide/RW1_counts.tsv", "RW1_clean.fna.gz", "RW1"),
        "ref_RW1_fgs_k5_c10.tsv": ("results/2023-11-29/fna-5genomes_gz-10/tsv_fgs/RW1_counts.tsv", "RW1_fgs.faa.gz", "RW1"),
        "ref_RW1_pro_k5_c10.tsv": ("results/2023-11-29/faa-5genomes-10/tsv_protein/RW1_pro_counts.tsv", "RW1_pro.faa.gz", "RW1_pro"),
        "ref_Test_R1_k5_c10.tsv": ("results/2023-11-29/test-qc_gz/tsv_nucleotide/Test_R1_counts.tsv", "Test_R1.fna.gz", "Test_R1"),
        "ref_DJ_pro_k5_c10_s10.tsv": ("results/2023-11-29/faa-5genomes-10/tsv_protein/DJ_pro_counts.tsv", "DJ_pro.faa.gz", "DJ_pro"),
        "ref_DJ_pro_k5_c10_s1.tsv": ("results/2023-11-29/faa-5genomes-1/tsv_protein/DJ_pro_counts.tsv", "DJ_pro.faa.gz", "DJ_pro"),
    }
    committed = {}
    for dst, (src, inp, base) in ref_tables.items():
        shutil.copyfile(REF / src, TSV / dst)
        os.chmod(TSV / dst, 0o644)
        committed[dst] = {"input": inp, "basename": base, "k": 5, "c": 10,
                          "chunk_mib": 1 if dst.endswith("_s1.tsv") else 0}

    # matrix of (input, k list, c list)
    ks_nt = [1, 2, 3, 5, 8, 13, 14, 15, 16, 21, 31, 32, 33, 47, 63, 64, 65, 100]
    matrix = []
    for f in ["A.fasta", "B.fasta", "C.fasta", "D_paired_1.fasta", "D_paired_2.fasta"]:
        matrix.append((f, [1, 3, 5, 21, 31, 32, 33, 63, 64], [0, 1, 2, 10]))
    matrix.append(("edge_ws.fa", ks_nt, [1, 2]))
    matrix.append(("edge_lengths.fa", ks_nt, [1, 2, 10]))
    matrix.append(("edge_reads.fna", [3, 7, 12, 21, 31, 32, 33, 63, 64], [1, 2, 10]))
    matrix.append(("edge_empty.fa", [3, 31], [1]))
    matrix.append(("edge_hdr_only.fa", [3, 31], [1]))
    matrix.append(("edge_nohdr.fa", [3, 21, 31, 33], [1, 2]))
    matrix.append(("edge_protein.faa", [1, 2, 3, 4, 5, 6, 12, 13, 25, 26, 40], [1, 2, 10]))
    matrix.append(("RW1.fna.gz", [3, 5, 12, 21, 31, 32, 33, 64], [1, 10]))
    matrix.append(("RW1_clean.fna.gz", [3, 5, 31], [10]))
    matrix.append(("Test_R1.fna.gz", [3, 5, 21, 31], [1, 10]))
    matrix.append(("Scaffolds_with-NNN.fna.gz", [5, 21], [2, 10]))
    matrix.append(("RW1_pro.faa.gz", [3, 5, 6, 12, 13, 25, 26], [1, 10]))
    matrix.append(("RW1_fgs.faa.gz", [3, 5, 12], [1, 10]))
    matrix.append(("DJ_pro.faa.gz", [3, 5], [10]))

    expected = {}
    keep_full = {("A.fasta", 31, 1), ("edge_ws.fa", 3, 1), ("edge_ws.fa", 5, 1), ("edge_ws.fa", 31, 1),
                 ("edge_lengths.fa", 32, 2), ("edge_protein.faa", 3, 2), ("Scaffolds_with-NNN.fna.gz", 5, 10)}
    for fname, ks, cs in matrix:
        base = basename_of(fname)
        for k in ks:
            raw = kmers.find_kmers(INPUTS / fname, k, 0)
            for c in cs:
                table = {key: n for key, n in raw.items() if n >= c}
                # the filter is part of find_kmers; cross-check one c per (file,k) through it
                if c == cs[-1]:
                    assert table == kmers.find_kmers(INPUTS / fname, k, c)
                case = "%s|k%d|c%d" % (fname, k, c)
                expected[case] = dict(input=fname, basename=base, k=k, c=c, **digest(base, table))
                if (fname, k, c) in keep_full:
                    (TSV / ("%s_k%d_c%d.tsv" % (base, k, c))).write_text(tsv_text(base, table))

    # Chunker goldens: run the reference Chunker, record cut offsets in the *decompressed,
    # newline-normalised* text and per-chunk digests; plus the composed chunk->count->sum table.
    chunks = {}
    chunk_cases = [("DJ_pro.faa.gz", "1M"), ("RW1_pro.faa.gz", "0.03M"), ("edge_reads.fna", "0.02M"),
                   ("edge_ws.fa", "0.0001M"), ("edge_lengths.fa", "0.001M"), ("A.fasta", "0.004M"),
                   ("edge_ws.fa", "0M"), ("edge_empty.fa", "1M")]
    for fname, size in chunk_cases:
        with tempfile.TemporaryDirectory() as tmp:
            c = chunker.Chunker(str(INPUTS / fname), tmp, size, ">")
            files = sorted(c.files)
            names = [os.path.basename(f) for f in files]
            blobs = [Path(f).read_bytes() for f in files]
            offs, pos = [], 0
            for b in blobs:
                offs.append(pos)
                pos += len(b)
            base = basename_of(fname)
            entry = {"input": fname, "size": size, "bytes": chunker.human2bytes(size), "names": names,
                     "offsets": offs, "total": pos,
                     "sha256": [hashlib.sha256(b).hexdigest() for b in blobs], "counts": {}}
            for k, cmin in [(3, 10), (5, 10), (5, 2), (31, 1)]:
                total = {}
                for f in files:
                    for key, n in kmers.find_kmers(Path(f), k, cmin).items():
                        total[key] = total.get(key, 0) + n
                entry["counts"]["k%d|c%d" % (k, cmin)] = digest(base, total)
            chunks["%s|%s" % (fname, size)] = entry
    # the reference's committed DJ_pro chunks (results/2023-11-29/faa-5genomes-1/chunks_protein)
    ref_chunks = sorted((REF / "results/2023-11-29/faa-5genomes-1/chunks_protein/DJ_pro").iterdir())
    chunks["DJ_pro.faa.gz|1M"]["committed_sha256"] = [hashlib.sha256(p.read_bytes()).hexdigest() for p in ref_chunks]
    assert chunks["DJ_pro.faa.gz|1M"]["committed_sha256"] == chunks["DJ_pro.faa.gz|1M"]["sha256"]

    # ---- the big files: all five genomes at k=3/c=10 (config 1), k=31/c=1 and c=10, k=5/c=10 on the cleaned
    #      twin where the reference committed the table; proteomes at k=3/c=10 (config 4) and k=5/c=10
    import numpy as np
    big = {}

    def big_case(fname, base, k, c, table):
        keys = sorted(table)
        d = digest(base, table)
        d["keys_sha256"] = hashlib.sha256("".join(keys).encode()).hexdigest()
        d["counts_sha256"] = hashlib.sha256(np.array([table[x] for x in keys], dtype="<u8").tobytes()).hexdigest()
        big["%s|k%d|c%d" % (fname, k, c)] = dict(input=fname, basename=base, k=k, c=c, **d)

    for fname in ["DJ.fna.gz", "GIC31.fna.gz", "RW1.fna.gz", "RW2.fna.gz", "Rleg.fna.gz"]:
        base = basename_of(fname)
        raw = kmers.find_kmers(INPUTS / fname, 3, 0)
        big_case(fname, base, 3, 10, {a: b for a, b in raw.items() if b >= 10})
        raw = kmers.find_kmers(INPUTS / fname, 31, 0)
        for c in (1, 10):
            big_case(fname, base, 31, c, {a: b for a, b in raw.items() if b >= c})
        del raw
    for fname in ["DJ_pro.faa.gz", "GIC31_pro.faa.gz", "RW1_pro.faa.gz", "RW2_pro.faa.gz", "Rleg_pro.faa.gz"]:
        base = basename_of(fname)
        for k in (3, 5):
            big_case(fname, base, k, 10, kmers.find_kmers(INPUTS / fname, k, 10))
        # the reference's committed table of the same run (results/run-tests.sh: k=5, -c 10, -s 10: unchunked)
        want = (REF / "results/2023-11-29/faa-5genomes-10/tsv_protein" / (base + "_counts.tsv")).read_text()
        assert hashlib.sha256(want.encode()).hexdigest() == big["%s|k5|c10" % fname]["sha256"], fname
    # SURVEY.md 8(c) known answers (first 16 hex digits of the TSV sha256)
    known = {"DJ.fna.gz|k3|c10": "c6eadaec2188bb43", "GIC31.fna.gz|k3|c10": "0e68d1e690114f80", "RW1.fna.gz|k3|c10": "3ac4bb1f9ca6baa1",
             "RW2.fna.gz|k3|c10": "0fe5a38b181435c3", "Rleg.fna.gz|k3|c10": "4127b01ca65d4e3b", "DJ.fna.gz|k31|c1": "d968c29ebee74951",
             "GIC31.fna.gz|k31|c1": "59c3181f9a7c39b1", "RW1.fna.gz|k31|c1": "985928da4290698c", "RW2.fna.gz|k31|c1": "1537f3abc37cc482",
             "Rleg.fna.gz|k31|c1": "ae6becf6677bee2c", "DJ.fna.gz|k31|c10": "a91a92379c42782f", "Rleg.fna.gz|k31|c10": "14840589c294b8b7",
             "DJ_pro.faa.gz|k3|c10": "2b68e8c6b8a5e68e", "GIC31_pro.faa.gz|k3|c10": "2ee2bd8d48ca4757", "RW1_pro.faa.gz|k3|c10": "e98c90e9c61c7d40",
             "RW2_pro.faa.gz|k3|c10": "5fe201e30219ce8b", "Rleg_pro.faa.gz|k3|c10": "668ed139decf1723"}
    for case, head in known.items():
        assert big[case]["sha256"].startswith(head), (case, big[case]["sha256"])
    (HERE / "expected_big.json").write_text(json.dumps(big, indent=0, sort_keys=True))

    # ---- the reference's committed removeN outputs (results/2023-11-29/fna-5genomes_gz-10/clean) for the five
    #      genomes: digest of the decompressed text (RW1_clean.fna.gz is also among the inputs in full)
    clean = {}
    for g in ["DJ", "GIC31", "RW1", "RW2", "Rleg"]:
        text = gzip.open(REF / "results/2023-11-29/fna-5genomes_gz-10/clean" / (g + "_clean.fna.gz"), "rb").read()
        clean[g + ".fna.gz"] = {"clean_name": g + "_clean.fna.gz", "bytes": len(text), "sha256": hashlib.sha256(text).hexdigest(),
                                "lines": text.count(b"\n")}
    (HERE / "clean.json").write_text(json.dumps(clean, indent=1, sort_keys=True))

    h2b = {s: chunker.human2bytes(s) for s in ["0 B", "1 K", "1 M", "1 Gi", "1 tera", "0.5kilo", "0.1  byte", "1 k", "100M", "1M", "10M", "0M", "1.5G"]}

    (HERE / "expected.json").write_text(json.dumps(expected, indent=0, sort_keys=True))
    (HERE / "chunks.json").write_text(json.dumps({"chunks": chunks, "human2bytes": h2b, "committed_tables": committed}, indent=1, sort_keys=True))
    print("cases:", len(expected), "big cases:", len(big), "chunk cases:", len(chunks))


if __name__ == "__main__":
    main()
