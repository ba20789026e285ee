"""Drop-in for the part of MerCat2's ``mercat2_fasta`` module that stands in front of the counting path:
``removeN`` (lib/mercat2_fasta.py:53-119) with its helper ``split_sequenceN`` (:21-49).

The reference runs removeN on every nucleotide FASTA before counting (bin/mercat2.py:239-244, 276) unless
``-skipclean`` is given: sequences are cut at runs of 'N' into ``>{name}_{i} {info}`` records re-wrapped at 80
columns, and the result is written as ``<outpath>/<base>_clean.fna.gz``.  Here the rewrite is native host code
behind the C ABI (``mk_remove_n``, csrc/mk_host.cpp); the cleaned text can be handed to the counting engine
straight from memory (``clean_text``) instead of being read back from the ``.gz``.
"""
from __future__ import annotations

import gzip
import os
import re
import textwrap
from pathlib import Path
from typing import Optional, Dict, Tuple

from . import native
from .kmers import read_fasta_bytes


def split_sequenceN(header: str, sequence: str):
    """lib/mercat2_fasta.py:21-49 (same arguments and result): used for the records ``mk_remove_n`` leaves to
    this layer -- a sequence to be split that holds blanks or hyphens, which textwrap treats as word breaks."""
    n_lengths = [len(m.group(1)) for m in re.finditer(r"(N+)", sequence)]
    pieces = re.sub(r"(N+)", "\n", sequence).split("\n")
    words = header.split()
    basename, info = words[0], " ".join(words[1:])
    seqs = []
    for i, seq in enumerate(pieces, 1):
        seqs.append(f">{basename}_{i} {info}")
        seqs += textwrap.wrap(seq, 80)
    return seqs, n_lengths


def _clean_text_py(text: str, toupper: bool) -> Tuple[str, int, int]:
    """The whole rewrite in Python, for the rare file the native code declines (see split_sequenceN)."""
    out = []
    gc = total = 0
    lines = text.splitlines(keepends=True) if False else None  # (text-mode readline semantics: only \n, \r, \r\n end a line)
    lines = re.split(r"\r\n|\r|\n", text)
    if lines and lines[-1] == "":
        lines.pop()
    i = 0
    while i < len(lines):
        line = lines[i].strip()
        if not line.startswith(">"):
            i += 1
            continue
        name = line[1:]
        i += 1
        seq_line = []
        while i < len(lines):
            cur = lines[i].strip()
            if cur.startswith(">"):
                break
            seq_line.append(cur)
            i += 1
        sequence = "".join(seq_line)
        if "N" in sequence:
            pieces, _ = split_sequenceN(name, sequence)
            for s in pieces:
                out.append(s if s.startswith(">") or not toupper else s.upper())
                gc += s.count("G") + s.count("C")
                total += len(s)
        else:
            out.append(">" + name)
            out += [s.upper() for s in seq_line] if toupper else seq_line
            gc += sequence.count("G") + sequence.count("C")
            total += len(sequence)
    return "".join(s + "\n" for s in out), gc, total


def clean_text(raw, toupper: bool = False) -> Tuple[bytes, Dict[str, float]]:
    """removeN on FASTA bytes already in memory: (cleaned text, {'GC Content': percent}).  Raises
    ZeroDivisionError for input without sequence and IndexError for a record to be split whose header is
    empty, as the reference does."""
    cleaned, st = native.remove_n(raw, toupper)
    if st["unsupported_record"] >= 0:
        text, gc, total = _clean_text_py(bytes(raw).decode("utf-8"), toupper)
        cleaned, st = text.encode("utf-8"), dict(st, gc_count=gc, total_length=total)
    return cleaned, {"GC Content": 100.0 * st["gc_count"] / st["total_length"]}


def removeN_text(fasta: Path, outpath: Path, toupper: bool, timings: Optional[dict] = None):
    """removeN that also hands back the cleaned text: ``(path, stats, cleaned bytes)`` -- the counting path takes
    the bytes from memory instead of inflating the file it has just written.  ``timings`` receives read_s / clean_s /
    gzip_s (the CLI's -debug line)."""
    import timeit
    os.makedirs(outpath, exist_ok=True)
    basename = Path(fasta).stem.split(".")[0]
    out_fasta = Path(outpath, f"{basename}_clean.fna.gz")
    t0 = timeit.default_timer()
    raw = read_fasta_bytes(fasta)
    t1 = timeit.default_timer()
    cleaned, stats = clean_text(raw, toupper)
    t2 = timeit.default_timer()
    with gzip.open(out_fasta, "wb") as writer:
        writer.write(cleaned)
        writer.flush()  # the reference's text-mode writer flushes once when it is closed: one sync-flush marker in the stream
    if timings is not None:
        timings.update(read_s=t1 - t0, clean_s=t2 - t1, gzip_s=timeit.default_timer() - t2)
    return out_fasta.absolute(), stats, cleaned


def removeN(fasta: Path, outpath: Path, toupper: bool):
    """Splits sequences in a scaffold fasta file at N repeats (lib/mercat2_fasta.py:53-119: same arguments, same
    ``(path of <base>_clean.fna.gz, {'GC Content': ...})`` result, same file content).  The file is written with
    Python's gzip module exactly as the reference writes it (level 9), so that its size -- which decides whether
    the sample is chunked, bin/mercat2.py:101 -- is the same."""
    path, stats, _ = removeN_text(fasta, outpath, toupper)
    return path, stats
