"""Pin the CPU oracle (oracle/cpu_ref.py) to the golden vectors made by the reference."""
import gzip
import hashlib
import io
import json
from pathlib import Path

import pytest

from conftest import GOLDEN, read_input
from oracle import cpu_ref


def _digest(base, table):
    text = cpu_ref.tsv_text(base, table)
    return {"rows": len(table), "sum": sum(table.values()), "sha256": hashlib.sha256(text.encode()).hexdigest()}


def _cases():
    exp = json.loads((GOLDEN / "expected.json").read_text())
    by_file = {}
    for name, case in exp.items():
        by_file.setdefault((case["input"], case["k"]), []).append(case)
    return sorted(by_file.items())


@pytest.mark.parametrize("key,cases", _cases(), ids=lambda v: "%s-k%d" % v if isinstance(v, tuple) else None)
def test_find_kmers_matches_reference(key, cases):
    fname, k = key
    raw = cpu_ref.find_kmers(GOLDEN / "inputs" / fname, k, 0)
    for case in cases:
        table = cpu_ref.apply_min_count(raw, case["c"])
        got = _digest(case["basename"], table)
        assert got == {k2: case[k2] for k2 in ("rows", "sum", "sha256")}, (fname, k, case["c"])


def test_count_text_equals_file_path(inputs_dir):
    for fname in ["edge_ws.fa", "A.fasta", "edge_protein.faa"]:
        data = (inputs_dir / fname).read_bytes()
        for k in (3, 31):
            assert cpu_ref.count_text(data, k, 1) == cpu_ref.find_kmers(inputs_dir / fname, k, 1)


def test_committed_reference_tables(chunk_golden, inputs_dir, tmp_path):
    """The reference's own committed *_counts.tsv (results/2023-11-29) reproduce byte for byte."""
    for tsv, meta in chunk_golden["committed_tables"].items():
        want = (GOLDEN / "tsv" / tsv).read_text()
        data = read_input(meta["input"])
        table = cpu_ref.count_sample_text(data, meta["k"], meta["c"], meta["chunk_mib"])
        assert cpu_ref.tsv_text(meta["basename"], table) == want, tsv


def test_full_tsv_fixtures(expected, inputs_dir):
    for p in sorted((GOLDEN / "tsv").glob("*_k*_c*.tsv")):
        if p.name.startswith("ref_"):
            continue
        stem = p.name[:-4]
        base, kk, cc = stem.rsplit("_", 2)
        k, c = int(kk[1:]), int(cc[1:])
        (fname,) = [v["input"] for v in expected.values() if v["basename"] == base and v["k"] == k and v["c"] == c]
        table = cpu_ref.find_kmers(inputs_dir / fname, k, c)
        assert cpu_ref.tsv_text(base, table) == p.read_text()


def test_chunker_matches_reference(chunk_golden, inputs_dir, tmp_path):
    for name, g in chunk_golden["chunks"].items():
        dest = tmp_path / name.replace("|", "_")
        files = cpu_ref.chunk_file(inputs_dir / g["input"], dest, g["size"])
        assert [Path(f).name for f in files] == g["names"]
        assert [hashlib.sha256(Path(f).read_bytes()).hexdigest() for f in files] == g["sha256"]
        for kc, want in g["counts"].items():
            k, c = int(kc.split("|")[0][1:]), int(kc.split("|")[1][1:])
            total = cpu_ref.merge_counts(cpu_ref.find_kmers(f, k, c) for f in files)
            # basename in the digest is the input's basename (extension stripped)
            assert _digest(_basename(g["input"]), total) == want, (name, kc)


def _basename(name):
    for ext in (".fasta.gz", ".fa.gz", ".fna.gz", ".ffn.gz", ".faa.gz", ".fasta", ".fa", ".fna", ".ffn", ".faa"):
        if name.endswith(ext):
            return name[: -len(ext)]
    return name


def test_human2bytes(chunk_golden):
    for s, want in chunk_golden["human2bytes"].items():
        assert cpu_ref.human2bytes(s) == want
    with pytest.raises(ValueError):
        cpu_ref.human2bytes("12 foo")


def test_run_mercat2_writes_or_skips(inputs_dir, tmp_path, capsys):
    out = tmp_path / "A_counts.tsv"
    base, path = cpu_ref.run_mercat2("A", [inputs_dir / "A.fasta"], out, 31, 1)
    assert (base, path) == ("A", out)
    assert out.read_text() == (GOLDEN / "tsv" / "A_k31_c1.tsv").read_text()
    assert "Significant k-mers: 3840" in capsys.readouterr().out
    out2 = tmp_path / "none.tsv"
    assert cpu_ref.run_mercat2("A", [inputs_dir / "A.fasta"], out2, 31, 1000) == ("A", None)
    assert not out2.exists()
    assert "No significant k-mers found" in capsys.readouterr().out
