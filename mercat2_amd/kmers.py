"""Drop-in for MerCat2's ``mercat2_kmers`` module (lib/mercat2_kmers.py).

``find_kmers(file, kmer, min_count)`` keeps the reference's signature and result (a dict
``{kmer_string: count}`` holding the k-mers whose count in THIS file is >= min_count,
lib/mercat2_kmers.py:32-78) but counts on the GPU through libmercat_hip.so.
"""
from __future__ import annotations

import gzip
from pathlib import Path
from typing import Dict, Optional, Union

from . import native

PROTEIN_SUFFIXES = (".faa", ".faa.gz")


def read_fasta_bytes(file: Union[str, Path]) -> bytes:
    """Raw (decompressed) bytes of a FASTA; gzip iff the last suffix is '.gz', exactly the
    reference's test (lib/mercat2_kmers.py:47)."""
    p = Path(file)
    if p.suffix == ".gz":
        with gzip.open(p, "rb") as fh:
            return fh.read()
    return p.read_bytes()


def read_head(file: Union[str, Path], n: int = 4096) -> bytes:
    """The first ``n`` (decompressed) bytes of a FASTA, to pick the alphabet from."""
    p = Path(file)
    with (gzip.open(p, "rb") if p.suffix == ".gz" else open(p, "rb")) as fh:
        return fh.read(n)


def map_fasta(file: Union[str, Path]):
    """Buffer over the (decompressed) bytes of a FASTA without an extra copy where possible:
    plain files are memory-mapped (the engine copies straight from the page cache to the GPU),
    '.gz' files are inflated into memory.  Returns an object supporting the buffer protocol."""
    import mmap
    p = Path(file)
    if p.suffix == ".gz":
        return read_fasta_bytes(p)
    size = p.stat().st_size
    if size == 0:
        return b""
    with open(p, "rb") as fh:
        return mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)


def guess_alphabet(file: Union[str, Path], data: Optional[bytes] = None) -> int:
    """Pick the packed fast path. Correctness never depends on it: characters outside the
    chosen alphabet are still counted exactly (by the by-reference kernel)."""
    name = str(file).lower()
    if name.endswith(PROTEIN_SUFFIXES):
        return native.ALPHABET_AA5
    if data is not None and len(data):
        # chunk files written by the Chunker lose the '.gz' but keep '.faa'; anything else:
        # sniff the first sequence lines
        seq = b"".join(l for l in bytes(data[:4096]).splitlines() if l and not l.startswith(b">"))
        if seq:
            acgt = sum(seq.count(c) for c in (b"A", b"C", b"G", b"T", b"N", b"a", b"c", b"g", b"t", b"n"))
            if acgt < 0.9 * len(seq):
                return native.ALPHABET_AA5
    return native.ALPHABET_NT2


def calculateKmerCount(seq: str, kmer: int, device: int = 0) -> Dict[str, int]:
    """Reference helper of the same name (lib/mercat2_kmers.py:10-28): all k-mers of ONE
    sequence string, no filter.  The string is counted as it stands (no stripping, '*' kept),
    so it is fed as a single-line record and must not hold characters the FASTA parser acts on."""
    data = seq.encode("ascii")
    if data != data.strip() or any(c in data for c in b"\r\n*") or data.startswith(b">"):
        raise ValueError("calculateKmerCount: sequence holds FASTA control characters")
    with native.Counter(kmer, guess_alphabet("", b">s\n" + data), device) as ctx:
        ctx.count_chunk(b">s\n" + data + b"\n", 0)
        return ctx.to_dict()


def find_kmers(file: Path, kmer: int, min_count: int, *, device: int = 0, alphabet: Optional[int] = None) -> Dict[str, int]:
    """Calculates the k-mer count in a fasta file (same contract as the reference).

    Parameters:
        file (Path): path to a fasta file (plain or .gz) to scan for k-mers.
        kmer (int): k-mer length.
        min_count (int): minimum count of k-mers found to be considered significant.

    Returns:
        dict: {k-mer string: count} for every k-mer with count >= min_count in this file.
    """
    if alphabet is None:
        alphabet = guess_alphabet(file, read_head(file))
    with native.Counter(kmer, alphabet, device) as ctx:
        native.count_file([ctx], file, 0, min_count)  # the whole file is one chunk
        return ctx.to_dict()
