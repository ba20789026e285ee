"""Host-side mirror of the reference interface (mercat2_amd/{kmers,chunker,harness}.py): the
parts that run without a GPU."""
import gzip
import hashlib
import inspect
import os
from pathlib import Path

from conftest import GOLDEN
from mercat2_amd import harness, kmers, native
from oracle import cpu_ref


def test_signatures_mirror_the_reference():
    # lib/mercat2_kmers.py:32 find_kmers(file, kmer, min_count)
    assert list(inspect.signature(kmers.find_kmers).parameters)[:3] == ["file", "kmer", "min_count"]
    # bin/mercat2.py:116 run_mercat2(basename, files, out_file, kmer, min_count, num_cores)
    assert list(inspect.signature(harness.run_mercat2).parameters)[:6] == [
        "basename", "files", "out_file", "kmer", "min_count", "num_cores"]
    # bin/mercat2.py:87 chunk_files(name, filename, chunk_size, outpath)
    assert list(inspect.signature(harness.chunk_files).parameters) == ["name", "filename", "chunk_size", "outpath"]
    assert list(inspect.signature(harness.countKmers).parameters)[:3] == ["file", "kmer", "min_count"]


def test_read_fasta_bytes_gz_rule(tmp_path):
    raw = b">a\nACGT\n"
    (tmp_path / "x.fna").write_bytes(raw)
    with gzip.open(tmp_path / "y.fna.gz", "wb") as f:
        f.write(raw)
    assert kmers.read_fasta_bytes(tmp_path / "x.fna") == raw
    assert kmers.read_fasta_bytes(tmp_path / "y.fna.gz") == raw


def test_guess_alphabet():
    assert kmers.guess_alphabet("a/b/DJ_pro.faa") == native.ALPHABET_AA5
    assert kmers.guess_alphabet("x.faa.gz") == native.ALPHABET_AA5
    assert kmers.guess_alphabet("x.fna", b">r\nACGTNNACGT\n") == native.ALPHABET_NT2
    assert kmers.guess_alphabet("chunk.00001", b">p\nMKVLAAGIVGLLLAQWERTY\n") == native.ALPHABET_AA5


def test_chunk_files_threshold_and_files(tmp_path):
    src = GOLDEN / "inputs" / "edge_reads.fna"  # 160 KB: below 1 MiB -> not chunked
    name, files = harness.chunk_files("s", str(src), 1, str(tmp_path / "c1"))
    assert (name, files) == ("s", [str(src)]) and not (tmp_path / "c1").exists()
    # force chunking with a 0 MiB threshold: every header line opens a chunk
    name, files = harness.chunk_files("s", str(GOLDEN / "inputs" / "edge_hdr_only.fa"), 0, str(tmp_path / "c0"))
    want = cpu_ref.chunk_file(GOLDEN / "inputs" / "edge_hdr_only.fa", tmp_path / "ref0", "0M")
    assert sorted(Path(f).name for f in files) == sorted(Path(f).name for f in want)
    for f in want:
        assert Path(f).read_bytes() == (tmp_path / "c0" / Path(f).name).read_bytes()
