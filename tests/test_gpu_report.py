"""The combined sample x k-mer table (merge_tsv, lib/mercat2_report.py:98-156) from the engine's tables,
against a plain restatement of the reference's merge (sorted union of keys, 0 where absent) and against
the rows of the combined tables the reference committed (results/2023-11-29/*/combined_*.tsv -- their
first header field is 'kmer', from an older release; the current source writes the TSVs' own 'k-mer')."""
import gzip
import random
from pathlib import Path

import numpy as np
import pytest

from mercat2_amd import native, report
from mercat2_amd.harness import run_sample
from oracle import cpu_ref

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).parent / "golden"


def _expected_text(tables, first="k-mer"):
    names = sorted(tables)
    keys = sorted(set().union(*[set(t) for t in tables.values()]))
    out = [first + "\t" + "\t".join(names)]
    for key in keys:
        out.append(key + "\t" + "\t".join(str(tables[n].get(key, 0)) for n in names))
    return "\n".join(out) + "\n"


def _inputs():
    rng = random.Random(1)
    genome = bytes(rng.choice(b"ACGT") for _ in range(4000))
    samples = {}
    for name, (lo, hi, extra) in {"s_b": (0, 2500, b""), "S_a": (1500, 4000, b">n\nACGTNNACGTAC\n"), "z": (500, 900, b"")}.items():
        reads = []
        for i in range(300):
            a = rng.randrange(lo, hi - 120)
            reads.append(b">r%d\n" % i + genome[a:a + 120] + b"\n")
        samples[name] = b"".join(reads) + extra
    return samples


@pytest.mark.parametrize("k,alphabet", [(4, native.ALPHABET_NT2), (21, native.ALPHABET_NT2), (40, native.ALPHABET_NT2),
                                        (7, native.ALPHABET_RAW)])
def test_merged_table_matches_union_of_sample_tables(tmp_path, k, alphabet):
    samples = _inputs()
    tables = {n: cpu_ref.count_text(d, k, 2) for n, d in samples.items()}
    ctxs = {}
    try:
        for n, d in samples.items():
            ctxs[n] = native.Counter(k, alphabet)
            ctxs[n].count_chunk(d, 2)
            ctxs[n].trim()       # working memory released, the table stays
        out = tmp_path / "combined.tsv"
        rows = report.merge_counters(ctxs, out, as_reference=False)  # the true union
        assert out.read_text() == _expected_text(tables)
        out_ref = tmp_path / "combined_ref.tsv"
        report.merge_counters(ctxs, out_ref)                         # as MerCat2's merge_tsv writes it
        assert out_ref.read_text() == cpu_ref.merge_tsv_text(tables)
        assert rows == len(set().union(*[set(t) for t in tables.values()]))
        names = sorted(ctxs)
        kmers, matrix = native.merged_export([ctxs[n] for n in names])
        keys = [bytes(r).decode() for r in kmers]
        assert keys == sorted(keys) and matrix.shape == (rows, 3)
        for j, n in enumerate(names):
            assert {key: int(v) for key, v in zip(keys, matrix[:, j]) if v} == tables[n]
        # a trimmed context counts again
        ctxs["z"].count_chunk(samples["z"], 2)
        assert ctxs["z"].to_dict() == cpu_ref.merge_counts([tables["z"], tables["z"]])
    finally:
        for c in ctxs.values():
            c.close()


def test_merge_tsv_keeps_the_reference_signature(tmp_path):
    samples = _inputs()
    paths, tables = {}, {}
    for n, d in samples.items():
        src = tmp_path / (n + ".fna")
        src.write_bytes(d)
        tsv = tmp_path / (n + "_counts.tsv")
        run_sample(n, src, tsv, 5, 3, report=lambda line: None)
        paths[n] = tsv
        tables[n] = cpu_ref.count_text(d, 5, 3)
    out = tmp_path / "combined_Nucleotide.tsv"
    report.merge_tsv(paths, out)
    assert out.read_text() == cpu_ref.merge_tsv_text(tables)  # the rows of the reference's streaming loop


def test_rows_of_a_combined_table_committed_by_the_reference(tmp_path):
    """tests/golden/combined_protein_k5_c10_head.tsv: the first 400 rows (+ header) of the reference's
    results/2023-11-29/faa-5genomes-1/combined_protein.tsv; RW1_pro is the one of its five samples whose
    input travels with the goldens (135 KB: below that run's -s 1, so one chunk), so its column is checked (all rows whose RW1_pro count is non-zero
    and, conversely, that every RW1_pro k-mer in the covered key range is there)."""
    gold = (GOLDEN / "combined_protein_k5_c10_head.tsv").read_text().splitlines()
    names = gold[0].split("\t")[1:]
    col = names.index("RW1_pro")
    want = {}
    for l in gold[1:]:   # (that release repeated a key when another sample's chunks each listed it: add the rows up)
        f = l.split("\t")
        want[f[0]] = want.get(f[0], 0) + int(f[1 + col])
    src = tmp_path / "RW1_pro.faa"
    src.write_bytes(gzip.open(GOLDEN / "inputs" / "RW1_pro.faa.gz", "rb").read())
    with native.Counter(5, native.ALPHABET_AA5) as a, native.Counter(5, native.ALPHABET_AA5) as b:
        native.count_file([a], src, 1 << 20, 10)      # that run used -s 1
        b.count_chunk(b">x\n" + gold[1].split("\t")[0].encode() * 3 + b"\n", 1)   # a second, tiny sample
        kmers, matrix = native.merged_export([a, b])
    keys = [bytes(r).decode() for r in kmers]
    last = max(want)
    got = {key: int(v) for key, v in zip(keys, matrix[:, 0]) if key <= last and v}
    assert got == {key: v for key, v in want.items() if v}


def test_combined_table_committed_by_the_reference_row_for_row(tmp_path):
    """The same committed table, now with all five proteomes among the fixtures: the reference's run
    (results/run-tests.sh: -k 5 -c 10 -s 1 on data/5-genomes-faa) chunked four of them at 1 MiB (per-chunk filter)
    and merged the five tables with merge_tsv.  Header + first 400 rows must come out line for line -- including the
    rows its streaming loop writes twice (only the first header field differs: that release wrote 'kmer')."""
    gold = (GOLDEN / "combined_protein_k5_c10_head.tsv").read_text().splitlines()
    names = ["DJ_pro", "GIC31_pro", "RW1_pro", "RW2_pro", "Rleg_pro"]
    assert gold[0].split("\t")[1:] == names
    ctxs = {}
    try:
        for n in names:
            src = tmp_path / (n + ".faa")
            src.write_bytes(gzip.open(GOLDEN / "inputs" / (n + ".faa.gz"), "rb").read())
            ctxs[n] = native.Counter(5, native.ALPHABET_AA5)
            native.count_file([ctxs[n]], src, 1 << 20, 10)
        out = tmp_path / "combined_protein.tsv"
        report.merge_counters(ctxs, out, first_column="kmer")
        got = out.read_text().splitlines()
        assert got[:len(gold)] == gold
    finally:
        for c in ctxs.values():
            c.close()


def test_merge_tsv_and_merge_tsv_T_against_the_reference_functions(tmp_path):
    """merge_tsv / merge_tsv_T (lib/mercat2_report.py:98-156, 160-194) with the reference's signatures against the
    outputs of the reference's own functions (tests/golden/make_report_golden.py): the merged table byte for byte;
    the transposed one as the same header set and the same matrix -- its columns are in sorted order here, in the
    order of a Python set there."""
    import json
    idx = json.loads((GOLDEN / "report" / "transposed.json").read_text())
    for case, g in idx.items():
        tsv_list = {name: str(GOLDEN / rel) for name, rel in g["inputs"].items()}
        m = tmp_path / (case + "_m.tsv")
        report.merge_tsv(tsv_list, m)
        assert m.read_text() == (GOLDEN / "report" / (case + "_merged.tsv")).read_text()
        t = tmp_path / (case + "_t.tsv")
        report.merge_tsv_T(tsv_list, t)
        lines = t.read_text().split("\n")
        assert lines[-1] == "" and lines[0].split("\t") == ["sample"] + g["columns_sorted"]
        assert [ln.split("\t")[0] for ln in lines[1:-1]] == g["sample_order"]
        for ln in lines[1:-1]:
            cells = ln.split("\t")
            assert cells[1:] == g["rows"][cells[0]], (case, cells[0])


def test_transposed_table_from_engine_tables(tmp_path):
    samples = _inputs()
    k = 21
    tables = {n: cpu_ref.count_text(d, k, 2) for n, d in samples.items()}
    ctxs = {}
    try:
        for n, d in samples.items():
            ctxs[n] = native.Counter(k, native.ALPHABET_NT2)
            ctxs[n].count_chunk(d, 2)
        out = tmp_path / "combined_T.tsv"
        cols = report.merge_counters_T(ctxs, out)
        keys = sorted(set().union(*[set(t) for t in tables.values()]))
        assert cols == len(keys)
        want = ["sample\t" + "\t".join(keys)] + [n + "\t" + "\t".join(str(tables[n].get(key, 0)) for key in keys) for n in sorted(tables)]
        assert out.read_text() == "\n".join(want) + "\n"
    finally:
        for c in ctxs.values():
            c.close()
