"""GPU parity: the HIP path (through the C ABI) against the golden vectors and the CPU oracle.

Bit-exact is the bar: integer counts, identical key sets, identical sorted TSV text.
Run with ``pytest -m gpu`` on an MI355X box; nothing here reads /root/reference.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, read_input
from mercat2_amd import native
from mercat2_amd.chunker import chunk_offsets
from oracle import cpu_ref

pytestmark = pytest.mark.gpu

EXPECTED = json.loads((GOLDEN / "expected.json").read_text())
CHUNKS = json.loads((GOLDEN / "chunks.json").read_text())


def tsv_of(ctx, base):
    kmers, counts = ctx.export()
    k = ctx.k
    flat = kmers.tobytes().decode("ascii")
    rows = ["k-mer\t%s_Count\n" % base]
    rows += ["%s\t%d\n" % (flat[i * k:(i + 1) * k], int(c)) for i, c in enumerate(counts)]
    return "".join(rows)


def digest_of(ctx, base):
    kmers, counts = ctx.export()
    text = tsv_of(ctx, base)
    return {"rows": int(counts.size), "sum": int(counts.sum()), "sha256": hashlib.sha256(text.encode()).hexdigest()}


def alphabet_for(name):
    return native.ALPHABET_AA5 if ".faa" in name else native.ALPHABET_NT2


def _grouped():
    groups = {}
    for case in EXPECTED.values():
        groups.setdefault(case["input"], []).append(case)
    return sorted(groups.items())


@pytest.mark.parametrize("fname,cases", _grouped(), ids=[g[0] for g in _grouped()])
def test_golden_matrix(fname, cases):
    """Every (input, k, c) golden of the reference, through the natural alphabet."""
    data = read_input(fname)
    for case in sorted(cases, key=lambda c: (c["k"], c["c"])):
        with native.Counter(case["k"], alphabet_for(fname)) as ctx:
            ctx.count_chunk(data, case["c"])
            got = digest_of(ctx, case["basename"])
            mode = ctx.stats()["mode_name"]
        assert got == {k: case[k] for k in ("rows", "sum", "sha256")}, (fname, case["k"], case["c"], mode)


BIG = json.loads((GOLDEN / "expected_big.json").read_text())


def _big_grouped():
    groups = {}
    for case in BIG.values():
        groups.setdefault(case["input"], []).append(case)
    return sorted(groups.items())


@pytest.mark.parametrize("fname,cases", _big_grouped(), ids=[g[0] for g in _big_grouped()])
def test_all_five_genomes_and_proteomes(fname, cases, tmp_path):
    """BASELINE configs 1 and 4 on every file of data/5-genomes-fna_gz / 5-genomes-faa_gz: k=3 -c 10, and for
    the genomes k=31 at -c 1 (up to 7.7 M rows) and -c 10 (0 rows = no TSV for three of them); the TSV the
    native writer produces must be the reference's byte for byte (sha256), the arrays of mk_export too."""
    data = read_input(fname)
    for case in sorted(cases, key=lambda c: (c["k"], c["c"])):
        out = tmp_path / ("%s_k%d_c%d.tsv" % (case["basename"], case["k"], case["c"]))
        with native.Counter(case["k"], alphabet_for(fname)) as ctx:
            ctx.count_chunk(data, case["c"])
            rows = ctx.write_tsv(out, case["basename"])
            kmers, counts = ctx.export()
        assert rows == case["rows"], (fname, case["k"], case["c"])
        if rows:
            h = hashlib.sha256()
            with open(out, "rb") as f:
                for block in iter(lambda: f.read(1 << 24), b""):
                    h.update(block)
            assert h.hexdigest() == case["sha256"], (fname, case["k"], case["c"])
            out.unlink()
        else:
            assert not out.exists()  # bin/mercat2.py:135-137
        assert int(counts.sum()) == case["sum"]
        assert hashlib.sha256(kmers.tobytes()).hexdigest() == case["keys_sha256"]
        assert hashlib.sha256(counts.astype("<u8").tobytes()).hexdigest() == case["counts_sha256"]


@pytest.mark.parametrize("alphabet", [native.ALPHABET_RAW, native.ALPHABET_NT2, native.ALPHABET_AA5],
                         ids=["raw", "nt2", "aa5"])
def test_any_alphabet_gives_the_same_answer(alphabet):
    """The alphabet only selects the packed fast path; windows outside it go by reference."""
    for fname, ks in [("edge_ws.fa", [1, 3, 5, 13, 31, 64]), ("edge_lengths.fa", [2, 5, 13, 21, 32, 33]),
                      ("edge_protein.faa", [1, 3, 5, 12, 13]), ("A.fasta", [5, 31])]:
        data = read_input(fname)
        for k in ks:
            for c in (1, 2):
                case = EXPECTED["%s|k%d|c%d" % (fname, k, c)]
                with native.Counter(k, alphabet) as ctx:
                    ctx.count_chunk(data, c)
                    got = digest_of(ctx, case["basename"])
                assert got == {x: case[x] for x in ("rows", "sum", "sha256")}, (fname, k, c, alphabet)


def test_dict_equals_oracle_dict():
    for fname, k, c in [("edge_ws.fa", 5, 1), ("edge_reads.fna", 21, 2), ("edge_protein.faa", 3, 1),
                        ("Scaffolds_with-NNN.fna.gz", 5, 10), ("edge_lengths.fa", 32, 1), ("edge_lengths.fa", 64, 1)]:
        data = read_input(fname)
        with native.Counter(k, alphabet_for(fname)) as ctx:
            ctx.count_chunk(data, c)
            got = ctx.to_dict()
        assert got == cpu_ref.count_text(data, k, c), (fname, k, c)


def test_committed_reference_tables(tmp_path):
    """results/2023-11-29/*/tsv_*/*_counts.tsv of the reference, byte for byte, via mk_write_tsv,
    including the 3-chunk DJ_pro case (per-chunk filter before the merge)."""
    for tsv, meta in CHUNKS["committed_tables"].items():
        data = read_input(meta["input"])
        size = meta["chunk_mib"] * 1024 * 1024
        offs = chunk_offsets(data, size) if meta["chunk_mib"] and len(data) >= size else [0, len(data)]
        out = tmp_path / tsv
        with native.Counter(meta["k"], alphabet_for(meta["input"])) as ctx:
            for a, b in zip(offs[:-1], offs[1:]):
                ctx.count_chunk(memoryview(data)[a:b], meta["c"])
            rows = ctx.write_tsv(out, meta["basename"])
        want = (GOLDEN / "tsv" / tsv).read_text()
        assert rows == want.count("\n") - 1
        assert out.read_text() == want, tsv


def test_chunked_counts_match_reference_chunker():
    for name, g in CHUNKS["chunks"].items():
        data = read_input(g["input"])
        offs = chunk_offsets(data, g["bytes"])
        assert len(offs) - 1 == len(g["names"]), name
        for kc, want in g["counts"].items():
            k, c = int(kc.split("|")[0][1:]), int(kc.split("|")[1][1:])
            with native.Counter(k, alphabet_for(g["input"])) as ctx:
                for a, b in zip(offs[:-1], offs[1:]):
                    ctx.count_chunk(memoryview(data)[a:b], c)
                base = g["input"]
                for ext in (".faa.gz", ".fna.gz", ".fasta", ".fna", ".faa", ".fa"):
                    if base.endswith(ext):
                        base = base[: -len(ext)]
                        break
                assert digest_of(ctx, base) == want, (name, kc)


def test_no_tsv_when_nothing_survives(tmp_path):
    data = read_input("A.fasta")
    out = tmp_path / "none.tsv"
    with native.Counter(31, native.ALPHABET_NT2) as ctx:
        ctx.count_chunk(data, 1000)
        assert ctx.write_tsv(out, "A") == 0
        assert ctx.rows() == 0
    assert not out.exists()


def test_empty_and_ragged_inputs():
    for data in [b"", b"\n\n", b">only header", b">h\nAC\n", b"ACGT", b">a\n>b\n>c\n", b"\r\r\n\r"]:
        for k in (1, 3, 31, 33):
            for alpha in (native.ALPHABET_NT2, native.ALPHABET_RAW):
                with native.Counter(k, alpha) as ctx:
                    ctx.count_chunk(data, 1)
                    assert ctx.to_dict() == cpu_ref.count_text(data, k, 1), (data, k, alpha)


def test_non_ascii_is_refused():
    with native.Counter(3, native.ALPHABET_NT2) as ctx:
        with pytest.raises(native.MercatHipError) as e:
            ctx.count_chunk(">r\nAC\xc3\xa9GT\n".encode("latin-1"), 1)
        assert e.value.code == -5
        ctx.count_chunk(b">r\nACGT\n", 1)  # the context stays usable
        assert ctx.to_dict() == {"ACG": 1, "CGT": 1}


def test_non_ascii_in_header_lines_is_fine():
    """Header text never enters a k-mer (lib/mercat2_kmers.py:52-53): a FASTA whose '>' lines hold UTF-8 is
    counted like the reference counts it; only non-ASCII SEQUENCE characters are refused.  Both parsers."""
    head = ">contig_1 Caf\u00e9 M\u00fcller \u4e2d\u6587\n".encode("utf-8")
    data = head + b"ACGTACGTTGCA\nACGT\n" + ">r2 \u00e9\n".encode("utf-8") + b"TTGACC\n"
    plain = b">contig_1 x\nACGTACGTTGCA\nACGT\n>r2 y\nTTGACC\n"
    general = data.replace(b"ACGTACGTTGCA\n", b"ACGT ACGTTGCA\n")  # a blank inside a sequence line: the general parser
    for k in (3, 5):
        for alpha in (native.ALPHABET_NT2, native.ALPHABET_RAW):
            with native.Counter(k, alpha) as ctx:
                ctx.count_chunk(data, 1)
                assert ctx.to_dict() == cpu_ref.count_text(plain, k, 1)
                ctx.reset()
                ctx.count_chunk(general, 1)
                assert ctx.to_dict() == cpu_ref.count_text(plain.replace(b"ACGTACGTTGCA\n", b"ACGT ACGTTGCA\n"), k, 1)
                with pytest.raises(native.NonAsciiInput):
                    ctx.count_chunk(data + "AC\u00e9GT\n".encode("utf-8"), 1)


def test_reset_and_reuse():
    a, b = read_input("A.fasta"), read_input("B.fasta")
    with native.Counter(21, native.ALPHABET_NT2) as ctx:
        ctx.count_chunk(a, 1)
        ctx.count_chunk(b, 1)
        both = ctx.to_dict()
        ctx.reset()
        ctx.count_chunk(b, 1)
        only_b = ctx.to_dict()
    assert only_b == cpu_ref.count_text(b, 21, 1)
    assert both == cpu_ref.merge_counts([cpu_ref.count_text(a, 21, 1), cpu_ref.count_text(b, 21, 1)])


def _synth(reads, k, genome=200_000, sub_ppm=0):
    return native.synth_reads(genome, 11, reads, 150, 12, sub_ppm).tobytes()


@pytest.mark.parametrize("k", [21, 31])
def test_synthetic_reads_vs_oracle(k):
    """60k x 150 bp reads from a 200 kbp genome (9 Mbases): full dict equality with the oracle."""
    data = _synth(60_000, k)
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        ctx.count_chunk(data, 10)
        got = ctx.to_dict()
        st = ctx.stats()
    want = cpu_ref.count_text(data, k, 10)
    assert got == want
    assert st["symbols"] == 60_000 * 150 and st["windows"] == 60_000 * (150 - k + 1)


def test_full_size_properties_config2():
    """BASELINE config 2 size (1M x 150 bp, k=21): size-independent properties.
    (a) sum of counts at c=1 == number of windows; (b) counting in 2 pieces at c=1 equals
    counting in one piece (linearity of the merge); (c) rows sorted strictly ascending."""
    k = 21
    data = native.synth_reads(1_000_000, 1, 1_000_000, 150, 2).tobytes()
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        ctx.count_chunk(data, 1)
        kmers, counts = ctx.export()
        st = ctx.stats()
    assert int(counts.sum()) == 1_000_000 * (150 - k + 1) == st["windows"]
    keys = kmers.view("S%d" % k).reshape(-1)
    assert np.all(keys[:-1] < keys[1:])
    cut = chunk_offsets(data, len(data) // 2)
    assert len(cut) == 3
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        ctx.count_chunk(memoryview(data)[cut[0]:cut[1]], 1)
        ctx.count_chunk(memoryview(data)[cut[1]:cut[2]], 1)
        kmers2, counts2 = ctx.export()
    assert np.array_equal(kmers, kmers2) and np.array_equal(counts, counts2)


def _merge_sorted_tables(tables, k):
    """Sum (kmers, counts) tables per key with numpy; result sorted by key bytes."""
    keys = np.concatenate([t[0].reshape(-1, k).view("S%d" % k).reshape(-1) for t in tables])
    cnts = np.concatenate([t[1] for t in tables])
    uniq, inv = np.unique(keys, return_inverse=True)
    out = np.zeros(uniq.size, dtype=np.uint64)
    np.add.at(out, inv, cnts)
    return uniq, out


def test_config2_full_size_bit_exact_vs_c_oracle():
    """BASELINE config 2 (S1): 1M x 150 bp, genome 1 Mbp (seeds 1/2), k=21, -c 10 -- the whole
    table, bit for bit, against the C oracle: unchunked (-s 0) and with the reference's chunking
    at -s 100 (2 chunks, each filtered on its own before the merge)."""
    from oracle import c_oracle
    k, c = 21, 10
    data = native.synth_reads(1_000_000, 1, 1_000_000, 150, 2).tobytes()
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        ctx.count_chunk(data, c)
        kmers, counts = ctx.export()
    okm, ocn = c_oracle.count(data, k, c)
    assert np.array_equal(kmers, okm) and np.array_equal(counts, ocn)
    offs = chunk_offsets(data, 100 * 1024 * 1024)
    assert len(offs) == 3
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        for a, b in zip(offs[:-1], offs[1:]):
            ctx.count_chunk(memoryview(data)[a:b], c)
        kmers2, counts2 = ctx.export()
    want_k, want_c = _merge_sorted_tables([c_oracle.count(data[a:b], k, c) for a, b in zip(offs[:-1], offs[1:])], k)
    assert np.array_equal(kmers2.view("S%d" % k).reshape(-1), want_k) and np.array_equal(counts2, want_c)
    assert counts2.size < counts.size  # the per-chunk filter loses k-mers (README "least significant k-mers")


@pytest.mark.parametrize("k,sub_ppm", [(31, 0), (32, 10000), (18, 0), (25, 0)])
def test_superkmer_path_vs_c_oracle(k, sub_ppm):
    """300k reads (45 Mbases) through the super-k-mer kernels at several k, incl. k=32 (all-ones key)
    and 1 % substitutions (more distinct keys per bucket)."""
    from oracle import c_oracle
    data = native.synth_reads(300_000, 21, 300_000, 150, 22, sub_ppm).tobytes() + b">polyT\n" + b"T" * 5000 + b"\n"
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        ctx.count_chunk(data, 2)
        kmers, counts = ctx.export()
    okm, ocn = c_oracle.count(data, k, 2)
    assert np.array_equal(kmers, okm) and np.array_equal(counts, ocn)


def test_bucket_overflow_splits_sub_ranges():
    """All-distinct keys (uniform random reads, nothing repeats) overflow the LDS tables' target
    load and force the sub-range splitting of the count kernel; the result must not change."""
    from oracle import c_oracle
    rng = np.random.default_rng(7)
    seq = rng.integers(0, 4, size=(40_000, 150), dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    body = lut[seq]
    lines = [b">u%d\n" % i + body[i].tobytes() + b"\n" for i in range(body.shape[0])]
    data = b"".join(lines)
    for k in (31, 21, 12):
        with native.Counter(k, native.ALPHABET_NT2) as ctx:
            ctx.count_chunk(data, 1)
            kmers, counts = ctx.export()
        okm, ocn = c_oracle.count(data, k, 1)
        assert np.array_equal(kmers, okm) and np.array_equal(counts, ocn), k


def test_cli_writes_reference_tsv(tmp_path, capsys):
    """python -m mercat2_amd.cli with the reference's flags: same file name and bytes as the
    reference's committed table for RW1_pro (k=5, -c 10)."""
    from mercat2_amd import cli
    out = tmp_path / "res"
    src = tmp_path / "RW1_pro.faa"
    src.write_bytes(read_input("RW1_pro.faa.gz"))
    assert cli.main(["-i", str(src), "-k", "5", "-c", "10", "-o", str(out)]) == 0
    got = (out / "tsv_protein" / "RW1_pro_counts.tsv").read_text()
    assert got == (GOLDEN / "tsv" / "ref_RW1_pro_k5_c10.tsv").read_text()
    text = capsys.readouterr().out
    assert "Significant k-mers:" in text and "Time to count 5-mers:" in text
    # nothing survives -> no file, sample dropped (bin/mercat2.py:135-137)
    out2 = tmp_path / "res2"
    assert cli.main(["-i", str(src), "-k", "5", "-c", "100000", "-o", str(out2)]) == 0
    assert not (out2 / "tsv_protein" / "RW1_pro_counts.tsv").exists()
    assert "No significant k-mers found" in capsys.readouterr().out


def test_cli_nucleotide_default_runs_removeN_first(tmp_path, capsys):
    """Without -skipclean a nucleotide FASTA goes through removeN before counting (bin/mercat2.py:239-244): the
    reference's committed run on RW1.fna.gz (k=5, -c 10) left clean/RW1_clean.fna.gz and tsv_nucleotide/
    RW1_counts.tsv; both are reproduced.  With -skipclean the raw file is counted (find_kmers on RW1.fna.gz)."""
    import gzip
    import shutil
    from mercat2_amd import cli
    src = tmp_path / "RW1.fna.gz"
    shutil.copyfile(GOLDEN / "inputs" / "RW1.fna.gz", src)
    out = tmp_path / "res"
    assert cli.main(["-i", str(src), "-k", "5", "-c", "10", "-o", str(out), "-lowmem", "-pca"]) == 0
    assert gzip.open(out / "clean" / "RW1_clean.fna.gz", "rb").read() == read_input("RW1_clean.fna.gz")
    assert (out / "tsv_nucleotide" / "RW1_counts.tsv").read_text() == (GOLDEN / "tsv" / "ref_RW1_clean_k5_c10.tsv").read_text()
    out2 = tmp_path / "res2"
    assert cli.main(["-i", str(src), "-k", "5", "-c", "10", "-o", str(out2), "-skipclean"]) == 0
    assert not (out2 / "clean").exists()
    want = EXPECTED["RW1.fna.gz|k5|c10"]
    text = (out2 / "tsv_nucleotide" / "RW1_counts.tsv").read_text()
    assert hashlib.sha256(text.encode()).hexdigest() == want["sha256"]
    # Scaffolds_with-NNN: N runs, lower case; -toupper turns the lower-case bases into countable ones
    src2 = tmp_path / "Scaffolds_with-NNN.fna"
    src2.write_bytes(read_input("Scaffolds_with-NNN.fna.gz"))
    for flags, upper in ((["-toupper"], True), ([], False)):
        out3 = tmp_path / ("res3_%d" % upper)
        assert cli.main(["-i", str(src2), "-k", "5", "-c", "10", "-o", str(out3)] + flags) == 0
        cleaned = gzip.open(out3 / "clean" / "Scaffolds_with-NNN_clean.fna.gz", "rb").read()
        table = cpu_ref.count_text(cleaned, 5, 10)
        assert (out3 / "tsv_nucleotide" / "Scaffolds_with-NNN_counts.tsv").read_text() == cpu_ref.tsv_text("Scaffolds_with-NNN", table)
    capsys.readouterr()
    for bad in (["-prod"], ["-fgs"]):
        with pytest.raises(SystemExit):
            cli.main(["-i", str(src), "-k", "5", "-o", str(tmp_path / "x")] + bad)
    fq = tmp_path / "reads.fastq"
    fq.write_text("@r\nACGT\n+\nIIII\n")
    with pytest.raises(SystemExit):
        cli.main(["-i", str(fq), "-k", "5", "-o", str(tmp_path / "y")])


def test_cli_counts_both_tables_until_the_clean_gz_decides(tmp_path, capsys):
    """A nucleotide FASTA whose CLEANED text reaches -s MiB: whether MerCat2 chunks it depends on the size of the level-9
    clean/<base>_clean.fna.gz (bin/mercat2.py:101, 243).  The CLI counts both candidates from the rewritten text and
    publishes the one the growing .gz selects: (a) a text whose .gz passes the limit -> the chunked table (the Chunker's
    cuts of the CLEANED text, every chunk filtered on its own), (b) one whose .gz stays below -> the unchunked table."""
    import gzip
    import io
    from mercat2_amd import cli
    from oracle import clean_ref
    rng = np.random.default_rng(9)

    def sample(name, bases, genome):
        recs, i = [], 0
        g = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=genome).tobytes()
        total = 0
        while total < bases:
            a = int(rng.integers(0, genome - 400))
            seq = bytearray(g[a:a + 400])
            if i % 7 == 0:
                seq[100:100 + int(rng.integers(1, 30))] = b"N" * len(seq[100:100 + int(rng.integers(1, 30))])
            recs.append(b">s%d d\n" % i + b"\n".join(bytes(seq[j:j + 70]) for j in range(0, len(seq), 70)) + b"\n")
            total += 400
            i += 1
        p = tmp_path / (name + ".fna")
        p.write_bytes(b"".join(recs))
        return p
    # (a): a large genome compresses to ~0.29 of the text -> 5 MB of text gives ~1.4 MiB of .gz; (b): reads of a tiny
    # genome compress far below the limit although the text is above it
    for name, bases, genome, want_chunked in (("big", 5_000_000, 4_000_000, True), ("rep", 2_500_000, 3_000, False)):
        src = sample(name, bases, genome)
        out = tmp_path / ("res_" + name)
        assert cli.main(["-i", str(src), "-k", "21", "-c", "2", "-s", "1", "-o", str(out), "-debug"]) == 0
        printed = capsys.readouterr().out
        gz = out / "clean" / (name + "_clean.fna.gz")
        cleaned = gzip.open(gz, "rb").read()
        assert cleaned == clean_ref.clean_text(src.read_bytes().decode(), False)[0].encode()
        assert len(cleaned) >= 1 << 20, "the case must be one the text's size alone does not decide"
        assert (gz.stat().st_size >= (1 << 20)) == want_chunked
        if want_chunked:
            groups = cpu_ref.split_lines(io.TextIOWrapper(io.BytesIO(cleaned), encoding="utf-8", newline=None), 1 << 20)
            assert len(groups) > 2
            table = cpu_ref.merge_counts(cpu_ref.count_lines(g, 21, 2) for g in groups)
        else:
            table = cpu_ref.count_text(cleaned, 21, 2)
        tsv = out / "tsv_nucleotide" / (name + "_counts.tsv")
        assert tsv.read_text() == cpu_ref.tsv_text(name, table)
        assert not list((out / "tsv_nucleotide").glob("*.whole")) and not list((out / "tsv_nucleotide").glob("*.chunked"))
        assert "decide_s=" in printed and printed.count("Significant k-mers:") == 1


def _fold_filter(table, c):
    return {key: n for key, n in cpu_ref.canonical_fold(table).items() if n >= c}


@pytest.mark.parametrize("k", [3, 7, 12, 17, 21, 31, 32, 33, 40, 47, 63, 64])
def test_canonical_mode_is_the_folded_reference(k):
    """Opt-in canonical counting (north_star / BASELINE config 3; NOT reference behaviour):
    oracle = reference counts with min_count 0, every ACGT-only key folded onto
    min(key, reverse complement), then the per-chunk filter."""
    from oracle import c_oracle
    synth = native.synth_reads(30_000, 41, 20_000, 150, 42).tobytes()
    for data, c in [(read_input("edge_lengths.fa"), 1), (read_input("edge_reads.fna"), 2), (synth, 3)]:
        with native.Counter(k, native.ALPHABET_NT2, canonical=True) as ctx:
            ctx.count_chunk(data, c)
            got = ctx.to_dict()
        assert got == _fold_filter(c_oracle.count_dict(data, k, 0), c), (k, c, len(data))


@pytest.mark.parametrize("k", [31, 63])
def test_canonical_mode_chunked_and_guards(k):
    from oracle import c_oracle
    c = 2
    data = native.synth_reads(30_000, 43, 40_000, 150, 44).tobytes()
    offs = chunk_offsets(data, 2_000_000)
    assert len(offs) > 3
    with native.Counter(k, native.ALPHABET_NT2, canonical=True) as ctx:
        for a, b in zip(offs[:-1], offs[1:]):
            ctx.count_chunk(memoryview(data)[a:b], c)
        got = ctx.to_dict()
        with pytest.raises(native.MercatHipError):
            ctx.set_canonical(False)  # rows counted in the other mode are in the table
        ctx.reset()
        ctx.set_canonical(False)
    want = cpu_ref.merge_counts(_fold_filter(c_oracle.count_dict(data[a:b], k, 0), c) for a, b in zip(offs[:-1], offs[1:]))
    assert got == want
    with pytest.raises(native.MercatHipError):
        native.Counter(3, native.ALPHABET_AA5, canonical=True)
    with pytest.raises(native.MercatHipError):
        native.Counter(70, native.ALPHABET_NT2, canonical=True)  # text keys have no complement


def test_run_sample_streams_and_merge_from(tmp_path):
    """harness.run_sample with 1, 2 and 3 contexts writes the same TSV as the reference
    composition (chunk -> count with its own filter -> sum), incl. by-reference rows."""
    from mercat2_amd import harness
    data = native.synth_reads(40_000, 51, 30_000, 150, 52).tobytes() + read_input("edge_lengths.fa")
    src = tmp_path / "s.fna"
    src.write_bytes(data)
    k, c = 21, 2
    offs = chunk_offsets(data, 1024 * 1024)
    want = cpu_ref.tsv_text("s", cpu_ref.merge_counts(cpu_ref.count_text(data[a:b], k, c) for a, b in zip(offs[:-1], offs[1:])))
    assert len(offs) > 4
    for streams in (1, 2, 3):
        out = tmp_path / ("s_%d.tsv" % streams)
        assert harness.run_sample("s", src, out, k, c, chunk_mib=1, streams=streams) == ("s", out)
        assert out.read_text() == want, streams


@pytest.mark.parametrize("k", [33, 48, 63, 64])
def test_large_k_packed_by_reference_vs_c_oracle(k):
    """33..64-mers (BASELINE config 5 uses k=63): packed by-reference kernel for the clean windows,
    byte-wise by-reference kernel for windows with N / lower case, one shared table."""
    from oracle import c_oracle
    reads = native.synth_reads(200_000, 61, 100_000, 150, 62, 3000).tobytes()
    odd = read_input("edge_lengths.fa") + b">mixed\n" + b"ACGTNNNNacgtACGTTGCA" * 40 + b"\n"
    data = reads + odd
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        ctx.count_chunk(data, 2)
        kmers, counts = ctx.export()
        assert ctx.stats()["mode_name"] == "hash128"  # mode 2: 33..64-mers, two-word packed keys
    okm, ocn = c_oracle.count(data, k, 2)
    assert np.array_equal(kmers, okm) and np.array_equal(counts, ocn)


def test_low_complexity_and_repeats():
    """Skewed input: long homopolymer and dinucleotide runs and a tandem repeat put most windows
    into a few minimizer buckets (and a handful of keys); counts must stay exact."""
    from oracle import c_oracle
    rng = np.random.default_rng(11)
    unit = "".join("ACGT"[i] for i in rng.integers(0, 4, 500))
    recs = []
    for i in range(300):
        recs.append(">a%d\n%s\n" % (i, "A" * 2000))
        recs.append(">at%d\n%s\n" % (i, "AT" * 1000))
        recs.append(">rep%d\n%s\n" % (i, unit * 5))
        recs.append(">t%d\n%s\n" % (i, "T" * 700 + unit[:300]))
    data = "".join(recs).encode()
    for k in (12, 21, 31, 32, 40):
        with native.Counter(k, native.ALPHABET_NT2) as ctx:
            ctx.count_chunk(data, 10)
            kmers, counts = ctx.export()
        okm, ocn = c_oracle.count(data, k, 10)
        assert np.array_equal(kmers, okm) and np.array_equal(counts, ocn), k


def test_count_device_unaligned_offsets():
    """mk_count_device on text resident in HBM at every 16-byte misalignment (the fast parser reads
    from the aligned address below and ignores the bytes in front) == the host-fed result."""
    import torch
    data = native.synth_reads(20_000, 71, 3_000, 150, 72).tobytes() + read_input("edge_lengths.fa")
    odd = read_input("edge_ws.fa")  # blanks in sequence lines: general-parser fallback + aligned copy
    for payload, k in [(data, 21), (odd, 5)]:
        with native.Counter(k, native.ALPHABET_NT2) as ref:
            ref.count_chunk(payload, 1)
            want = ref.to_dict()
        for off in (0, 1, 7, 15):
            buf = torch.zeros(len(payload) + 64, dtype=torch.uint8, device="cuda")
            buf[:16] = ord(">")  # garbage in front of the chunk must be ignored
            buf[off:off + len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).cuda()
            torch.cuda.synchronize()
            with native.Counter(k, native.ALPHABET_NT2) as ctx:
                ctx.count_device(buf.data_ptr() + off, len(payload), 1)
                assert ctx.to_dict() == want, (k, off)


def _random_fasta(rng, alphabet, n_records, max_len, quirks):
    out = []
    for i in range(n_records):
        n = int(rng.integers(0, max_len))
        seq = bytes(alphabet[rng.integers(0, len(alphabet), n)])
        width = int(rng.integers(1, 120))
        lines = [seq[j:j + width] for j in range(0, len(seq), width)] or [b""]
        nl = b"\r\n" if quirks and rng.random() < 0.2 else (b"\r" if quirks and rng.random() < 0.1 else b"\n")
        hdr = b">r%d some text > here" % i if rng.random() < 0.5 else b">r%d" % i
        if quirks and rng.random() < 0.15:
            hdr = b"  " + hdr
        body = []
        for ln in lines:
            if quirks and rng.random() < 0.1:
                ln = b" " + ln + b"\t"
            if quirks and rng.random() < 0.05:
                ln = ln[: len(ln) // 2] + b" " + ln[len(ln) // 2:]
            if quirks and rng.random() < 0.1:
                ln = ln + b"*"
            body.append(ln)
        out.append(hdr + nl + nl.join(body) + nl)
        if quirks and rng.random() < 0.1:
            out.append(nl)
    text = b"".join(out)
    if quirks and rng.random() < 0.3:
        text = text.rstrip(b"\r\n")
    if quirks and rng.random() < 0.2:
        text = b"ACGTTGCA" + b"\n" + text
    return text


def test_fuzz_against_oracle():
    """Random small FASTA texts (wrapped lines, CRLF / CR, blanks, stars, '>' inside headers, empty
    records, lower case, IUPAC, proteins), random k, min_count and chunk size: GPU == oracle."""
    from oracle import c_oracle
    rng = np.random.default_rng(2026)
    nt = np.frombuffer(b"ACGT", dtype=np.uint8)
    nt_odd = np.frombuffer(b"ACGTACGTACGTNnacgtRY", dtype=np.uint8)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYXBZ", dtype=np.uint8)
    for trial in range(120):
        kind = trial % 4
        alphabet = [nt, nt_odd, aa, nt][kind]
        quirks = kind != 0 or trial % 8 == 0
        data = _random_fasta(rng, alphabet, int(rng.integers(1, 60)), int(rng.integers(1, 900)), quirks)
        k = int(rng.choice([1, 2, 3, 5, 7, 8, 11, 12, 13, 17, 18, 21, 25, 31, 32, 33, 40, 63, 64, 65, 90]))
        c = int(rng.choice([1, 1, 2, 3]))
        alpha = native.ALPHABET_AA5 if kind == 2 else (native.ALPHABET_RAW if trial % 17 == 0 else native.ALPHABET_NT2)
        size = int(rng.integers(200, 20000))
        offs = chunk_offsets(data, size) if trial % 3 == 0 else [0, len(data)]
        with native.Counter(k, alpha) as ctx:
            for a, b in zip(offs[:-1], offs[1:]):
                ctx.count_chunk(data[a:b], c)
            got = ctx.to_dict()
        want = cpu_ref.merge_counts(c_oracle.count_dict(data[a:b], k, c) for a, b in zip(offs[:-1], offs[1:]))
        assert got == want, (trial, kind, k, c, len(data), len(offs))


def test_blanks_in_header_lines_keep_the_fast_parser():
    """Read files whose header lines hold blanks (">SRR1.7 7 length=150": most real ones) stay on the fast parser --
    up to round 3 its first pass, which tries both entry states of a wave, took the blanks of a header line it started
    inside for blanks in sequence text and sent the whole chunk to the general parser (three times the work of the rest
    of the pipeline).  A blank inside a SEQUENCE line still takes the general parser, and both are exact."""
    from oracle import c_oracle
    rng = np.random.default_rng(77)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=20000).tobytes()
    recs = []
    for i in range(6000):
        a = int(rng.integers(0, len(genome) - 150))
        recs.append(b">SRR000001.%d %d length=150 some\ttext\n" % (i, i) + genome[a:a + 150] + b"\n")
    data = b"".join(recs)
    for k, c in ((31, 2), (21, 1), (63, 2), (5, 10)):
        with native.Counter(k, native.ALPHABET_NT2) as ctx:
            ctx.count_chunk(data, c)
            assert ctx.to_dict() == c_oracle.count_dict(data, k, c), (k, c)
            assert ctx.stats()["parse_retries"] == 0, (k, c)
    spoiled = data.replace(genome[300:320], genome[300:310] + b" " + genome[310:320], 1)
    assert spoiled != data
    with native.Counter(31, native.ALPHABET_NT2) as ctx:
        ctx.count_chunk(spoiled, 2)
        ctx.count_chunk(spoiled, 2)  # (the second chunk of a sample would be a fused launch: it must stand back too)
        want = c_oracle.count_dict(spoiled, 31, 2)
        assert ctx.to_dict() == {key: 2 * n for key, n in want.items()}
        assert ctx.stats()["parse_retries"] == 2


@pytest.mark.parametrize("canonical", [False, True])
def test_fused_upsert_paths_are_exact(monkeypatch, canonical):
    """The count kernel's own merge into the running table (mk_skcount.hip, FCAP > 0) against the regions + import kernel
    it replaces, on chunks of one sample (the second chunk on is a fused launch), and its two side doors: (a) the spill
    list -- MK_FUSE_MAX_PROBE=0 sends every survivor whose first slot holds another key there, the host imports it
    afterwards; (b) the blocking finish of survivors that do not fit a sweep's list -- a bucket of a small chunk with
    -c 1... -c 2 on a tiny genome keeps far more than 512 keys.  All three equal the oracle; the counters say which ran."""
    from oracle import c_oracle
    data = native.synth_reads(60_000, 21, 120_000, 150, 22).tobytes()
    offs = chunk_offsets(data, 4_000_000)
    assert len(offs) > 4
    spans = list(zip(offs[:-1], offs[1:]))

    def want(c):
        parts = [c_oracle.count_dict(data[a:b], 31, 0 if canonical else c) for a, b in spans]
        if canonical:
            parts = [_fold_filter(p, c) for p in parts]
        return cpu_ref.merge_counts(parts)

    def run(c):
        with native.Counter(31, native.ALPHABET_NT2, canonical=canonical) as ctx:
            for a, b in spans:
                ctx.count_chunk(memoryview(data)[a:b], c)
            return ctx.to_dict(), ctx.stats()
    for c in (2, 3):
        got, st = run(c)
        assert got == want(c) and st["fused_chunks"] == len(spans) - 1 and st["fuse_spilled"] == 0, (c, st)
    monkeypatch.setenv("MK_FUSE_MAX_PROBE", "0")
    got, st = run(2)
    assert got == want(2) and st["fused_chunks"] == len(spans) - 1 and st["fuse_spilled"] > 1000, st
    monkeypatch.delenv("MK_FUSE_MAX_PROBE")
    monkeypatch.setenv("MK_NO_FUSE", "1")
    # (read when a context is created)
    got, st = run(2)
    assert got == want(2) and st["fused_chunks"] == 0
    monkeypatch.delenv("MK_NO_FUSE")
    # (b): small chunks have 256 buckets; every k-mer of the 60 kbp genome survives -c 2 in every chunk: ~470 survivors per
    # bucket forward, twice that many canonical keys per bucket when both strands fold -- beyond the 512-entry list
    with native.Counter(31, native.ALPHABET_NT2, canonical=canonical) as ctx:
        small = chunk_offsets(data, 1_000_000)
        for a, b in zip(small[:-1], small[1:]):
            ctx.count_chunk(memoryview(data)[a:b], 2)
        parts = [c_oracle.count_dict(data[a:b], 31, 0 if canonical else 2) for a, b in zip(small[:-1], small[1:])]
        if canonical:
            parts = [_fold_filter(p, 2) for p in parts]
        assert ctx.to_dict() == cpu_ref.merge_counts(parts)
        assert ctx.stats()["fused_chunks"] == len(small) - 2


@pytest.mark.parametrize("alpha,k", [("nt", 8), ("nt", 9), ("nt", 10), ("nt", 11), ("aa", 4), ("aa", 5)])
def test_short_keys_counted_by_direct_index(alpha, k, monkeypatch):
    """Keys of 16..26 bits (nucleotide 8 <= k <= 11, protein k = 4, 5) take mk_bin.hip: bucket = the key's top bits, one LDS
    add per window into the bucket's bins.  Reads, a homopolymer and a two-letter repeat (every window in one or two
    buckets), characters outside the alphabet, several chunks with their own filters, canonical keys; against the C
    oracle."""
    from oracle import c_oracle
    rng = np.random.default_rng(31 + k)
    if alpha == "nt":
        body = native.synth_reads(40_000, 5, 30_000, 150, 6).tobytes()
        odd = b">poly\n" + b"A" * 70_000 + b"\n>rep\n" + b"AC" * 30_000 + b"\n>n\n" + b"ACGTNNACGTRYACGT" * 500 + b"\n"
        a = native.ALPHABET_NT2
    else:
        letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
        body = b"".join(b">p%d\n" % i + letters[rng.integers(0, 20, int(rng.integers(5, 400)))].tobytes() + b"*\n" for i in range(6000))
        odd = b">poly\n" + b"L" * 50_000 + b"\n>x\n" + b"MKVLAXXBZJOUACDEF" * 300 + b"\n"
        a = native.ALPHABET_AA5
    data = body + odd + body[: len(body) // 3]
    offs = chunk_offsets(data, 900_000 if alpha == "nt" else 400_000)
    assert len(offs) > 3
    for c, canonical in ((1, False), (3, False), (2, True)):
        if canonical and alpha != "nt":
            continue
        parts = [c_oracle.count_dict(data[x:y], k, 0 if canonical else c) for x, y in zip(offs[:-1], offs[1:])]
        if canonical:
            parts = [_fold_filter(p, c) for p in parts]
        want = cpu_ref.merge_counts(parts)
        with native.Counter(k, a, canonical=canonical) as ctx:
            for x, y in zip(offs[:-1], offs[1:]):
                ctx.count_chunk(memoryview(data)[x:y], c)
            assert ctx.to_dict() == want, (alpha, k, c, canonical)
            assert ctx.stats()["mode_name"] == "hash64"
    del monkeypatch


def test_long_lines_and_long_headers():
    """A 3 Mbp record on ONE line, a 200 kB header, and 70 kB of text in front of the first header:
    lines far longer than a parser wave (4 KiB) or workgroup."""
    from oracle import c_oracle
    rng = np.random.default_rng(5)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    big = lut[rng.integers(0, 4, 3_000_000)].tobytes()
    pre = lut[rng.integers(0, 4, 70_000)].tobytes()
    data = pre + b"\n>" + b"h" * 200_000 + b"\n" + big + b"\n>tail\n" + big[:5000] + b"\n"
    for k in (21, 31, 63):
        with native.Counter(k, native.ALPHABET_NT2) as ctx:
            ctx.count_chunk(data, 1)
            kmers, counts = ctx.export()
        okm, ocn = c_oracle.count(data, k, 1)
        assert np.array_equal(kmers, okm) and np.array_equal(counts, ocn), k


def test_full_size_properties_config3():
    """BASELINE config 3 size (S2: 10 M x 150 bp from a 10 Mbp genome, k=31), size-independent
    properties: (a) unchunked at c=1 the counts sum to the number of windows and the rows are
    strictly ascending; (b) the 16 reference chunks at c=1 give the identical table (linearity of
    the per-chunk merge); (c) at c=10 every chunked row also exists at c=1 with a count no larger;
    (d) two contexts + device-side merge == one context."""
    k = 31
    data = native.synth_reads(10_000_000, 3, 10_000_000, 150, 4)
    offs = chunk_offsets(data, 100 * 1024 * 1024)
    assert len(offs) == 17
    view = memoryview(data)
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        ctx.count_chunk(view, 1)  # one 1.6 GB chunk: buckets far larger than the LDS table -> sub-range splitting
        km1, cn1 = ctx.export()
        st = ctx.stats()
    assert int(cn1.sum()) == 10_000_000 * (150 - k + 1) == st["windows"]
    keys1 = km1.view("S%d" % k).reshape(-1)
    assert np.all(keys1[:-1] < keys1[1:])
    with native.Counter(k, native.ALPHABET_NT2) as a, native.Counter(k, native.ALPHABET_NT2) as b:
        for i, (lo, hi) in enumerate(zip(offs[:-1], offs[1:])):
            (a if i % 2 == 0 else b).count_chunk(view[lo:hi], 1)
        a.merge_from(b)
        km2, cn2 = a.export()
    assert np.array_equal(km1, km2) and np.array_equal(cn1, cn2)
    del km2, cn2
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        for lo, hi in zip(offs[:-1], offs[1:]):
            ctx.count_chunk(view[lo:hi], 10)
        km10, cn10 = ctx.export()
    keys10 = km10.view("S%d" % k).reshape(-1)
    pos = np.searchsorted(keys1, keys10)
    assert np.all(pos < keys1.size) and np.array_equal(keys1[pos], keys10)
    assert np.all(cn10 <= cn1[pos]) and np.all(cn10 >= 10) and 0 < keys10.size < keys1.size


def _revcomp_rows(km):
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    return comp[km[:, ::-1]]


def test_full_size_properties_config3_canonical():
    """BASELINE config 3 as it is worded (canonical k-mers; an opt-in extension here) at full size: S2 in
    its 16 reference chunks.  (a) at c=1 the counts sum to the number of windows; (b) every key is the smaller of
    itself and its reverse complement, rows strictly ascending; (c) the table is the forward-strand table of the same
    chunks folded key by key onto min(key, revcomp) -- the definition of the mode -- checked in full; (d) at c=10
    (per chunk) every row exists at c=1 with a count no larger."""
    k = 31
    data = native.synth_reads(10_000_000, 3, 10_000_000, 150, 4)
    offs = chunk_offsets(data, 100 * 1024 * 1024)
    view = memoryview(data)
    spans = list(zip(offs[:-1], offs[1:]))
    with native.Counter(k, native.ALPHABET_NT2, canonical=True) as ctx:
        for lo, hi in spans:
            ctx.count_chunk(view[lo:hi], 1)
        kmc, cnc = ctx.export()
        st = ctx.stats()
    assert int(cnc.sum()) == 10_000_000 * (150 - k + 1) == st["windows"]
    keys = kmc.view("S%d" % k).reshape(-1)
    assert np.all(keys[:-1] < keys[1:])
    rc = np.ascontiguousarray(_revcomp_rows(kmc)).view("S%d" % k).reshape(-1)
    assert np.all(keys <= rc)
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        for lo, hi in spans:
            ctx.count_chunk(view[lo:hi], 1)
        kmf, cnf = ctx.export()
    fkeys = kmf.view("S%d" % k).reshape(-1)
    frc = np.ascontiguousarray(_revcomp_rows(kmf)).view("S%d" % k).reshape(-1)
    folded = np.where(frc < fkeys, frc, fkeys)
    uniq, inv = np.unique(folded, return_inverse=True)
    summed = np.zeros(uniq.size, dtype=np.uint64)
    np.add.at(summed, inv, cnf)
    assert np.array_equal(uniq, keys) and np.array_equal(summed, cnc)
    del kmf, cnf, fkeys, frc, folded, uniq, inv, summed
    with native.Counter(k, native.ALPHABET_NT2, canonical=True) as ctx:
        for lo, hi in spans:
            ctx.count_chunk(view[lo:hi], 10)
        km10, cn10 = ctx.export()
    keys10 = km10.view("S%d" % k).reshape(-1)
    pos = np.searchsorted(keys, keys10)
    assert np.all(pos < keys.size) and np.array_equal(keys[pos], keys10)
    assert np.all(cn10 <= cnc[pos]) and np.all(cn10 >= 10) and 0 < keys10.size < keys.size


def test_full_size_properties_config5(tmp_path):
    """BASELINE config 5 at its full size: 50 M x 150 bp from a 50 Mbp genome (S3, seeds 6/7), k=63, -c 10 -s 100,
    streamed from a file through mk_count_file (reader threads -> pinned blocks -> chunks on the GPU, never the
    whole sample in HBM).  (a) the Chunker rule gives the chunks of the text (78); every window is counted
    (50 M x 88); the rows that survive -c 10 per chunk are in strictly ascending order; (b) device memory in use
    after the 8 GB sample is bounded by the per-chunk working set, far below the sample; (c) linearity at c=1 on the
    sample's first 100 MiB cut into 32 MiB chunks (55 M rows): the file counted as a whole == its chunks counted one by
    one and summed, and the counts sum to the windows; (d) the same chunks at c=2 (per chunk) are a subset with counts
    no larger."""
    import torch
    k, reads = 63, 50_000_000
    data = native.synth_reads(50_000_000, 6, reads, 150, 7)
    path = tmp_path / "S3.fna"
    with open(path, "wb") as f:
        f.write(memoryview(data))
    first = chunk_offsets(memoryview(data)[:220 * 1024 * 1024], 100 * 1024 * 1024)[1]   # end of the first chunk
    head = bytes(memoryview(data)[:first])
    offs3 = chunk_offsets(head, 32 * 1024 * 1024)
    nbytes = data.nbytes
    del data
    free0, _ = torch.cuda.mem_get_info()
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        st = native.count_file([ctx], path, 100 * 1024 * 1024, 10)
        free1, _ = torch.cuda.mem_get_info()
        stats = ctx.stats()
        kmers, counts = ctx.export()
    assert st["chunked"] == 1 and st["chunks"] == 78 and st["text_bytes"] == nbytes
    assert stats["windows"] == reads * (150 - k + 1) and stats["exotic_windows"] == 0 and stats["mode_name"] == "hash128"
    # (round 4) the table itself is pinned: the reference's Chunker + find_kmers(chunk, 63, 10) over all 78 chunks
    # (tests/golden/make_s2_golden.py --only s3full) -> rows, sum, sha256 of keys and counts
    import hashlib
    import json
    from conftest import GOLDEN
    gold = json.loads((GOLDEN / "expected_s2.json").read_text()).get("S3|k63|c10|s100")
    if gold is not None:
        want = gold["forward"]
        assert gold["chunks_total"] == 78 and kmers.shape[0] == want["rows"] and int(counts.sum()) == want["sum"]
        assert hashlib.sha256(np.ascontiguousarray(kmers).tobytes()).hexdigest() == want["keys_sha256"]
        assert hashlib.sha256(counts.astype("<u8").tobytes()).hexdigest() == want["counts_sha256"]
    keys = kmers.view("S%d" % k).reshape(-1)
    assert np.all(keys[:-1] < keys[1:]) and np.all(counts >= 10)
    assert free0 - free1 < 6 * (1 << 30) < nbytes  # working set of a 100 MiB chunk, not of the 8 GB sample
    os.unlink(path)
    # linearity at c = 1 on the head
    hpath = tmp_path / "S3_head.fna"
    hpath.write_bytes(head)
    with native.Counter(k, native.ALPHABET_NT2) as a, native.Counter(k, native.ALPHABET_NT2) as b:
        sth = native.count_file([a], hpath, 32 * 1024 * 1024, 1)
        assert sth["chunks"] == len(offs3) - 1 >= 3
        for lo, hi in zip(offs3[:-1], offs3[1:]):
            b.count_chunk(memoryview(head)[lo:hi], 1)
        ka, ca = a.export()
        kb, cb = b.export()
        wa = a.stats()["windows"]
    assert np.array_equal(ka, kb) and np.array_equal(ca, cb)
    assert int(ca.sum()) == wa == head.count(b">") * (150 - k + 1)
    akeys = ka.view("S%d" % k).reshape(-1)
    assert np.all(akeys[:-1] < akeys[1:])
    with native.Counter(k, native.ALPHABET_NT2) as c10:
        native.count_file([c10], hpath, 32 * 1024 * 1024, 2)
        k10, n10 = c10.export()
    keys10 = k10.view("S%d" % k).reshape(-1)
    pos = np.searchsorted(akeys, keys10)
    assert np.all(pos < max(akeys.size, 1)) and np.array_equal(akeys[pos], keys10) and np.all(n10 <= ca[pos])


def test_many_contexts_and_threads():
    """Contexts are independent: 6 of them counting different inputs from 6 host threads at once,
    then 150 create/count/destroy cycles (no leak, no cross-talk)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle
    jobs = []
    for i, (k, alpha) in enumerate([(31, native.ALPHABET_NT2), (3, native.ALPHABET_NT2), (12, native.ALPHABET_NT2),
                                    (63, native.ALPHABET_NT2), (5, native.ALPHABET_AA5), (21, native.ALPHABET_RAW)]):
        data = native.synth_reads(20_000 + i, 80 + i, 4_000, 150, 90 + i).tobytes()
        if alpha == native.ALPHABET_AA5:
            data = read_input("edge_protein.faa") * 20
        jobs.append((k, alpha, data))

    def run(job):
        k, alpha, data = job
        with native.Counter(k, alpha) as ctx:
            for _ in range(3):
                ctx.reset()
                ctx.count_chunk(data, 2)
            return ctx.to_dict()

    with ThreadPoolExecutor(6) as pool:
        got = list(pool.map(run, jobs))
    for (k, alpha, data), g in zip(jobs, got):
        assert g == c_oracle.count_dict(data, k, 2), (k, alpha)
    small = read_input("A.fasta")
    want = cpu_ref.count_text(small, 21, 1)
    for i in range(150):
        with native.Counter(21, native.ALPHABET_NT2) as ctx:
            ctx.count_chunk(small, 1)
            if i % 50 == 0:
                assert ctx.to_dict() == want


@pytest.mark.parametrize("sigmas,xseg,expect_retry", [("6", "1", False), ("6", "0", False), ("0", "0", True), ("0", "1", None)])
def test_sampled_bucket_sizes_and_exact_second_pass(monkeypatch, sigmas, xseg, expect_retry):
    """Big chunks size their super-k-mer buckets from a 1-in-8 sample of the histogram; a chunk whose
    sample was too small anywhere is partitioned again exactly.  Same tables either way (forced here on
    a small input: MK_SAMPLE_MIN=0; MK_SAMPLE_SIGMAS=0 removes the safety margin so the second pass runs).
    With a region per XCD and bucket (the default, MK_XSEG=1) runs that do not fit their region go to the bucket's
    shared region first: without a margin most of the records travel that way, and the exact pass is not needed."""
    monkeypatch.setenv("MK_SAMPLE_MIN", "0")
    monkeypatch.setenv("MK_SAMPLE_SIGMAS", sigmas)
    monkeypatch.setenv("MK_XSEG", xseg)
    data = native.synth_reads(200_000, 41, 60_000, 150, 42).tobytes()
    low = b">poly\n" + b"A" * 30_000 + b"\n>rep\n" + b"ACGTTGCAAG" * 4_000 + b"\n"
    from oracle import c_oracle
    for k, c in ((21, 2), (31, 1), (32, 3), (40, 2), (63, 1)):      # one-word and two-word keys
        for payload in (data, data + low):
            half = payload[: len(payload) // 2]
            want = cpu_ref.merge_counts([c_oracle.count_dict(payload, k, c), c_oracle.count_dict(half, k, c)])
            with native.Counter(k, native.ALPHABET_NT2) as ctx:
                ctx.count_chunk(payload, c)
                ctx.count_chunk(half, c)      # a second chunk on the same context
                got = ctx.to_dict()
                retries = ctx.stats()["part_retries"]
            assert got == want, (k, c, sigmas)
            if expect_retry is not None:
                assert (retries > 0) == expect_retry, (k, c, sigmas, retries)


@pytest.mark.parametrize("canonical", [False, True])
def test_bucket_regions_per_xcd_give_the_same_tables(monkeypatch, canonical):
    """Sampled bucket sizes come with one region per XCD and bucket (mk_sk_scatterq_k: a workgroup fills the regions of the
    XCD it runs on; the count kernel reads a bucket's eight regions one after the other); MK_XSEG=0 keeps one region per
    bucket.  Same tables, no exact second pass for reads, and chunks that inherit their regions from the chunk before
    still do so when the layout changes in between (they size their own then)."""
    monkeypatch.setenv("MK_SAMPLE_MIN", "0")
    from oracle import c_oracle
    data = native.synth_reads(300_000, 11, 80_000, 150, 12).tobytes()
    tail = data[: len(data) // 3]
    for k, c in ((31, 2), (21, 1), (14, 3), (32, 1)):
        parts = [c_oracle.count_dict(x, k, 0 if canonical else c) for x in (data, data, tail)]
        want = cpu_ref.merge_counts([_fold_filter(p, c) for p in parts] if canonical else parts)
        for xseg in ("1", "0"):
            monkeypatch.setenv("MK_XSEG", xseg)
            with native.Counter(k, native.ALPHABET_NT2, canonical=canonical) as ctx:
                ctx.count_chunk(data, c)
                ctx.count_chunk(data, c)   # inherits the regions of the first
                ctx.count_chunk(tail, c)
                got = ctx.to_dict()
                st = ctx.stats()
            assert got == want, (k, c, xseg)
            assert st["part_retries"] == 0 and st["part_reused"] == 1, (k, c, xseg, st["part_retries"], st["part_reused"])
        # the layout changing between chunks of one context
        with native.Counter(k, native.ALPHABET_NT2, canonical=canonical) as ctx:
            for xseg, chunk in (("1", data), ("0", data), ("1", tail)):
                monkeypatch.setenv("MK_XSEG", xseg)
                ctx.count_chunk(chunk, c)
            assert ctx.to_dict() == want, (k, c, "mixed")
            assert ctx.stats()["part_reused"] == 0


def test_equal_chunks_inherit_bucket_regions(monkeypatch):
    """A chunk as long as the one before it (same k, same min_count, no overflow) inherits that chunk's bucket
    regions instead of sizing its own (mk_stats_t.part_reused).  Same tables as with MK_NO_REUSE; a chunk of the
    same length whose content does not fit the inherited regions (a repeat array after random reads) is
    partitioned again exactly, and a shorter last chunk sizes its own."""
    monkeypatch.setenv("MK_SAMPLE_MIN", "0")
    from oracle import c_oracle
    n = 3_000_000
    reads = [native.synth_reads(n // 152 + 1, 41, 60_000 + 7 * i, 150, 42 + i).tobytes()[:n] for i in range(3)]
    unit = b"ACGTTGCAAGGCTTAACGGATCCATGCAAGTCC"
    skew = (b">rep\n" + unit * (n // len(unit)))[: n - 1] + b"\n"
    assert len(skew) == n
    for k, c in ((31, 2), (21, 1), (40, 2)):
        chunks = [reads[0], reads[1], skew, reads[2], reads[1], reads[0][: n // 3]]
        want = cpu_ref.merge_counts([c_oracle.count_dict(x, k, c) for x in chunks])
        seen = {}
        for env in ("", "1"):
            if env:
                monkeypatch.setenv("MK_NO_REUSE", env)
            else:
                monkeypatch.delenv("MK_NO_REUSE", raising=False)
            with native.Counter(k, native.ALPHABET_NT2) as ctx:
                for x in chunks:
                    ctx.count_chunk(x, c)
                got = ctx.to_dict()
                seen[env] = ctx.stats()
            assert got == want, (k, c, env)
        assert seen["1"]["part_reused"] == 0
        assert seen[""]["part_reused"] >= 2, seen[""]
        if not os.environ.get("MK_NO_SPECULATION"):
            # (the one-read-back lane compares raw lengths; the general lane compares sequence lengths, and the repeat
            # array's single header makes its sequence 6 % longer: it sizes its own regions there)
            assert seen[""]["part_retries"] >= 1, seen[""]     # the repeat array did not fit the reads' regions


@pytest.mark.parametrize("qcap", ["0", "300", None])
@pytest.mark.parametrize("walk", [False, True])
def test_scatter_queue_and_walk_agree(monkeypatch, qcap, walk):
    """The scatter lists a wave's runs in an LDS queue and works them off with every lane busy; a wave whose runs do
    not fit the queue walks them lane by lane instead (and analyses the sub-tile again for the second pass).
    MK_SKQ_CAP lowers the queue's capacity: 0 walks every wave, 300 about half of them (k = 21: ~6 runs per
    thread), unset is the product setting; MK_SCATTER_WALK is the first version of the kernel.  Same tables."""
    from oracle import c_oracle
    if qcap is not None:
        monkeypatch.setenv("MK_SKQ_CAP", qcap)
    if walk:
        if qcap is not None:
            pytest.skip("the walking kernels have no queue")
        monkeypatch.setenv("MK_SCATTER_WALK", "1")
    data = native.synth_reads(300_000, 5, 70_000, 150, 6).tobytes()
    low = b">poly\n" + b"A" * 20_000 + b"\n>n\n" + (b"ACGTTGCAAGGCTTAACGGATCCATGCAAGTCCN" * 1500) + b"\n"
    # (two-word keys, 33 <= k <= 64: the same queue form in mk_sk2_scatterq_k, forward-strand keys)
    for k, c, canon in ((21, 2, False), (31, 1, False), (18, 1, False), (32, 2, False), (25, 1, True), (12, 2, False), (14, 1, True),
                        (63, 1, False), (33, 2, False), (48, 1, False), (64, 1, False), (63, 1, True)):
        payload = data + low
        want = _fold_filter(c_oracle.count_dict(payload, k, 0), c) if canon else c_oracle.count_dict(payload, k, c)
        with native.Counter(k, native.ALPHABET_NT2, canonical=canon) as ctx:
            ctx.count_chunk(payload, c)
            got = ctx.to_dict()
        assert got == want, (k, c, canon, qcap)


@pytest.mark.parametrize("qcap", [None, "250"])
def test_scatter_tile_shapes_agree(monkeypatch, qcap):
    """The queue scatter has two tile shapes -- 2 sub-tiles with 512-item queues, 3 with 376-item queues, picked from
    what the chunk before listed per lane (MK_SKQ_SUBT forces one).  Three sub-tiles on a first chunk, on input whose
    waves overflow the shorter queues (k = 12, 14: many runs per lane; the poly-A and repeat records: few), with the
    queue cut down further, and over several chunks so that the hint switches shapes by itself: same tables."""
    from oracle import c_oracle
    if qcap is not None:
        monkeypatch.setenv("MK_SKQ_CAP", qcap)
    data = native.synth_reads(300_000, 5, 70_000, 150, 6).tobytes()
    low = b">poly\n" + b"A" * 20_000 + b"\n>n\n" + (b"ACGTTGCAAGGCTTAACGGATCCATGCAAGTCCN" * 1500) + b"\n"
    payload = data + low
    for k, c, canon in ((31, 1, False), (21, 2, False), (12, 2, False), (14, 1, True), (32, 2, True), (63, 1, False), (40, 2, False)):
        want = _fold_filter(c_oracle.count_dict(payload, k, 0), c) if canon else c_oracle.count_dict(payload, k, c)
        monkeypatch.setenv("MK_SKQ_SUBT", "3")
        with native.Counter(k, native.ALPHABET_NT2, canonical=canon) as ctx:
            ctx.count_chunk(payload, c)
            got = ctx.to_dict()
        assert got == want, (k, c, canon, qcap, "forced 3")
        monkeypatch.delenv("MK_SKQ_SUBT")
        if c == 1:  # three equal chunks: the second and third take whatever shape the first one's records suggest
            with native.Counter(k, native.ALPHABET_NT2, canonical=canon) as ctx:
                for _ in range(3):
                    ctx.count_chunk(payload, 1)
                got3 = ctx.to_dict()
            assert got3 == {key: 3 * n for key, n in want.items()}, (k, canon, qcap, "by hint")


@pytest.mark.parametrize("nkmax", ["1", "3", "5", "12", "31"])
def test_record_length_limit_is_only_a_layout_choice(monkeypatch, nkmax):
    """Runs of windows that share a minimizer are cut into records of at most 8 windows (MK_NKMAX moves the limit:
    the histogram cuts them while it walks, the scatter with mask arithmetic beforehand -- both must cut alike or the
    bucket sizes would not match what is written).  Any limit gives the same table."""
    from oracle import c_oracle
    monkeypatch.setenv("MK_NKMAX", nkmax)
    monkeypatch.setenv("MK_SAMPLE_MIN", "0")
    data = native.synth_reads(200_000, 9, 50_000, 150, 10).tobytes() + b">t\n" + b"ACGTTGCAAG" * 3000 + b"\n>p\n" + b"C" * 9000 + b"\n"
    for k, c in ((21, 2), (31, 1), (32, 1), (18, 3)):
        with native.Counter(k, native.ALPHABET_NT2) as ctx:
            ctx.count_chunk(data, c)
            got = ctx.to_dict()
        assert got == c_oracle.count_dict(data, k, c), (k, c, nkmax)


def test_skewed_genome_like_input_at_scale_vs_c_oracle():
    """60 MB that look like an assembly rather than reads: megabase single-line records, a satellite
    array (171-bp unit, 1 % mutated copies), poly-A, a dinucleotide repeat and N gaps next to random
    sequence and reads.  The sampled bucket sizes are badly off where the repeats sit (exact second pass),
    some buckets hold hundreds of thousands of windows of a handful of k-mers, others need sub-range
    splitting; the table must still be the C oracle's, bit for bit."""
    from oracle import c_oracle
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rand = lambda n: acgt[rng.integers(0, 4, n)].tobytes()
    unit = rand(171)
    sat = bytearray(unit * 50_000)
    for pos in rng.integers(0, len(sat), len(sat) // 100):
        sat[pos] = acgt[rng.integers(0, 4)]
    genome = rand(20_000_000)
    reads = native.synth_reads(2_000_000, 5, 150_000, 150, 6).tobytes()
    parts = [b">chr1 random\n", genome[:12_000_000], b"\n>sat\n", bytes(sat), b"\n>polyA\n", b"A" * 1_000_000,
             b"\n>ac\n", b"AC" * 1_000_000, b"\n>gaps\n", genome[12_000_000:16_000_000], b"N" * 5000,
             genome[16_000_000:], b"\n", reads, b">chr1_again\n", genome[:6_000_000], b"\n"]
    data = b"".join(parts)
    for k, c in ((31, 2), (21, 3), (45, 2)):
        okm, ocn = c_oracle.count(data, k, c)
        with native.Counter(k, native.ALPHABET_NT2) as ctx:
            ctx.count_chunk(data, c)
            kmers, counts = ctx.export()
            st = ctx.stats()
        assert np.array_equal(kmers, okm) and np.array_equal(counts, ocn), (k, c)
        assert st["windows"] > 50_000_000


def test_homopolymer_megabases_count_in_bounded_time():
    """Millions of windows of ONE k-mer land in one bucket.  The count kernel used to start such a bucket at
    the split depth its size suggested (2^16 passes over the records: 30 s for 2 Mbases of poly-A); it now
    starts at most 8 sub-ranges deep and splits further only where a table overflows."""
    import time
    from oracle import c_oracle
    data = b">a\n" + b"A" * 3_000_000 + b"\n>r\n" + b"ACGT" * 500_000 + b"\n>t\n" + b"T" * 200_000 + b"\n"
    for k in (31, 32, 21, 40, 12):
        with native.Counter(k, native.ALPHABET_NT2) as ctx:
            t0 = time.perf_counter()
            ctx.count_chunk(data, 1)
            got = ctx.to_dict()
            dt = time.perf_counter() - t0
        assert got == c_oracle.count_dict(data, k, 1), k
        assert dt < 5.0, (k, dt)


def test_input_shapes_vs_c_oracle():
    """One megabase line, records of exactly k bases, an N gap, soft-masked (lower-case) stretches, wrapped and
    CRLF-wrapped lines, blanks inside sequence lines: same tables as the C oracle, each within a second."""
    import time
    from oracle import c_oracle
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    n = 2_000_000
    rand = acgt[rng.integers(0, 4, n)].tobytes()
    shapes = {
        "one_long_line": b">x\n" + rand + b"\n",
        "tiny_records": b"".join(b">r%d\n" % i + rand[i * 31:i * 31 + 31] + b"\n" for i in range(n // 80)),
        "n_gap": b">n\n" + rand[:1000] + b"N" * (n // 2) + rand[1000:2000] + b"\n",
        "soft_masked": b">m\n" + b"".join((rand[i:i + 5000].lower() if (i // 5000) % 2 else rand[i:i + 5000]) for i in range(0, n // 2, 5000)) + b"\n",
        "wrapped_60": b">w\n" + b"\n".join(rand[i:i + 60] for i in range(0, n, 60)) + b"\n",
        "crlf_wrapped": b">w\r\n" + b"\r\n".join(rand[i:i + 70] for i in range(0, n // 2, 70)) + b"\r\n",
        "blank_in_lines": b">b\n" + b"\n".join(rand[i:i + 30] + b" " + rand[i + 30:i + 60] for i in range(0, n // 4, 60)) + b"\n",
    }
    for name, data in shapes.items():
        for k, c in ((21, 1), (31, 2), (40, 1)):
            with native.Counter(k, native.ALPHABET_NT2) as ctx:
                t0 = time.perf_counter()
                ctx.count_chunk(data, c)
                kmers, counts = ctx.export()
                dt = time.perf_counter() - t0
            okm, ocn = c_oracle.count(data, k, c)
            assert np.array_equal(kmers, okm) and np.array_equal(counts, ocn), (name, k, c)
            assert dt < 2.0, (name, k, dt)


@pytest.mark.parametrize("k,c,genome,reads", [(63, 10, 400_000, 60_000), (33, 4, 400_000, 60_000), (48, 2, 20_000, 40_000),
                                              (31, 10, 400_000, 60_000), (32, 3, 20_000, 40_000), (21, 2, 5_000, 40_000)])
@pytest.mark.parametrize("canonical", [False, True])
def test_counting_prefilter_kernels_are_exact(monkeypatch, k, c, genome, reads, canonical):
    """The count kernels with a counting pre-filter (a count-min row in LDS; only keys whose counter reaches min_count
    enter the exact table): the two-word one is what a sample's later chunks take when min_count is well above the mean
    count (BASELINE config 5); forced here from the first chunk on, for one- and two-word keys, at coverages from 20x
    (few candidates) to 1200x (nearly every key a candidate: the candidates' table overflows and the hash range is
    split), against the C oracle chunk by chunk."""
    from oracle import c_oracle
    monkeypatch.setenv("MK_FORCE_PREFILTER", "1")
    data = native.synth_reads(genome, 41, reads, 150, 42).tobytes() + b">polyT\n" + b"T" * 400 + b"\n"
    offs = chunk_offsets(data, 3_000_000)
    want = {}
    for a, b in zip(offs[:-1], offs[1:]):
        if canonical:
            raw = cpu_ref.count_text(data[a:b], k, 1)
            part = _fold_filter(raw, c)
        else:
            part = c_oracle.count_dict(data[a:b], k, c)
        for key, n in part.items():
            want[key] = want.get(key, 0) + n
    with native.Counter(k, native.ALPHABET_NT2, canonical=canonical) as ctx:
        for a, b in zip(offs[:-1], offs[1:]):
            ctx.count_chunk(memoryview(data)[a:b], c)
        got = ctx.to_dict()
    assert got == want


@pytest.mark.parametrize("k", [13, 14, 20, 24, 25])
def test_amino_acid_two_word_keys(k):
    """Protein k-mers of 13..25 residues are packed two-word keys (5 k <= 125 bits: mk_count_ref128aa_k), not text:
    same table as the oracle -- rows in byte order, residues outside 'A'..'Z' and lower case kept as text rows --
    also when the chunks are dealt to three contexts and merged by key range (owner bounds over the 5 k - 64 bits of
    the key's first word)."""
    rng = np.random.default_rng(k)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    base = ["".join(aa[x] for x in rng.integers(0, 20, 400)) for _ in range(40)]
    recs = []
    for i in range(1500):
        s = base[int(rng.integers(0, 40))]
        a = int(rng.integers(0, 300))
        piece = s[a:a + int(rng.integers(k - 3, 100))]
        if i % 17 == 0:
            piece = piece[:5] + "X*" + piece[5:] + "BZJOU"
        if i % 29 == 0:
            piece = piece[:9] + piece[9:].lower()
        recs.append(">p%d # 1\n%s*\n" % (i, "\n".join(piece[j:j + 60] for j in range(0, len(piece), 60))))
    data = "".join(recs).encode()
    for c in (1, 3):
        want = cpu_ref.count_text(data, k, c)
        with native.Counter(k, native.ALPHABET_AA5) as ctx:
            ctx.count_chunk(data, c)
            assert ctx.stats()["mode_name"] == "hash128" and ctx.words_per_key() == 2
            kmers, counts = ctx.export()
            assert ctx.to_dict() == want
        keys = [bytes(r) for r in kmers]
        assert keys == sorted(keys)
    offs = chunk_offsets(data, 30_000)
    spans = list(zip(offs[:-1], offs[1:]))
    want = {}
    for a, b in spans:
        for key, n in cpu_ref.count_text(data[a:b], k, 2).items():
            want[key] = want.get(key, 0) + n
    ctxs = [native.Counter(k, native.ALPHABET_AA5) for _ in range(3)]
    try:
        for i, (a, b) in enumerate(spans):
            ctxs[i % 3].count_chunk(data[a:b], 2)
        native.merge_devices(ctxs, native.MERGE_RANGES)
        km, cn = native.export_multi(ctxs)
        flat = km.tobytes().decode()
        assert dict(zip((flat[i:i + k] for i in range(0, len(flat), k)), cn.tolist())) == want
    finally:
        for x in ctxs:
            x.close()
    with native.Counter(26, native.ALPHABET_AA5) as ctx:  # 130 bits: text rows, as before
        ctx.count_chunk(data, 1)
        assert ctx.stats()["mode_name"] == "byref" and ctx.to_dict() == cpu_ref.count_text(data, 26, 1)
